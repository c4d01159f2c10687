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


@pytest.mark.gpu
def test_bench_config_5_line(gpu):
    """`bench.py --config c5`: the sharded TSimpleHMC config through distributed.run_windows over HmcBackend, one rank."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--steps", "2", "--warmup", "1",
                        "--chains", "1024", "--dim", "80", "--window", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["unit"] == "trajectories/s" and out["config"]["baseline_config"] == 5 and out["n_gpus"] == 1
    assert out["covariance_updates"] >= 1 and 0.0 < out["accept_rate"] <= 1.0
    assert abs(out["value"] - 1024 * 4 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["frac"] > 0.0


# ---- `python bench.py --gpus N` launches its own ranks (no GPU needed for the logic) ----
def test_rank_environments(smcmc):
    envs = smcmc.distributed.rank_environments(4, 29511, {"PATH": "/bin", "WORLD_SIZE": "9"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" and e["PATH"] == "/bin" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)       # dmabuf IPC, or RCCL cannot share buffers
    assert smcmc.distributed.rank_environments(1, 1, {"HSA_ENABLE_IPC_MODE_LEGACY": "1"})[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    with pytest.raises(ValueError):
        smcmc.distributed.rank_environments(0, 1)


def test_local_launcher_starts_one_process_per_rank(smcmc, tmp_path):
    """The launcher bench.py uses: N child processes with the ranks' environments, rank 0's stdout handed back, the
    worst exit code reported; fewer devices than ranks is refused before anything starts."""
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\n"
                      "r, w = os.environ['RANK'], os.environ['WORLD_SIZE']\n"
                      "open(os.path.join(sys.argv[1], 'rank' + r), 'w').write(w + ' ' + os.environ['LOCAL_RANK'] + ' ' + os.environ['MASTER_PORT'])\n"
                      "print('line from rank', r)\n"
                      "sys.exit(3 if r == sys.argv[2] else 0)\n")
    code, out = smcmc.distributed.launch_local_ranks([sys.executable, str(script), str(tmp_path), "none"], 3, 8)
    assert code == 0 and out.strip() == "line from rank 0"
    seen = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank") and not f.endswith(".py"))
    assert seen == ["rank0", "rank1", "rank2"]
    ports = {open(tmp_path / f).read().split()[2] for f in seen}
    assert len(ports) == 1 and all(open(tmp_path / f).read().split()[:2] == ["3", f[4:]] for f in seen)
    code, _ = smcmc.distributed.launch_local_ranks([sys.executable, str(script), str(tmp_path), "2"], 3, 8)
    assert code == 3                                                       # a failing rank fails the launch
    with pytest.raises(RuntimeError):
        smcmc.distributed.launch_local_ranks([sys.executable, str(script), str(tmp_path), "none"], 4, 2)


def test_bench_refuses_more_ranks_than_gpus_before_touching_one():
    """`python bench.py --gpus 64` on any box: a clear message and a non-zero exit code, no HIP initialisation."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "64 ranks asked for" in r.stderr and "GPU(s) visible" in r.stderr
    assert "{" not in r.stdout
    # and a WORLD_SIZE that disagrees with --gpus is a usage error, not a silent single-rank run
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
