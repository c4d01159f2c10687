"""bench.py prints one JSON line with the fields the driver reads (a small ensemble, a few windows)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--chains", "4096",
                        "--no-cpu-baseline", "--no-ess", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):   # cpu_baseline: skipped here (12 s of host work)
        assert key in out, key
    assert out["unit"] == "chain-steps/s" and out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and "workload" in out["config"]
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "fp64_issue_frac", "measured_hbm_GBps"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # value = chain-steps of the timed windows / wall time
    assert abs(out["value"] - 4096 * out["config"]["window"] * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
