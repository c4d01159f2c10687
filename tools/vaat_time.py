"""Throughput of the variable-at-a-time engine (smcmc_vaat_step), HIP events on the engine's stream.
usage: python tools/vaat_time.py [out.json]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smcmc_amd_loader import load_package  # noqa: E402


def tdummy_error(dim):
    cov = np.eye(dim)
    cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
    return np.linalg.inv(cov)


def run(pkg, torch, name, dim, chains, like, prm, exact, steps, reps=3):
    stream = torch.cuda.Stream()
    e = pkg.VaatEngine(dim, chains, likelihood=like, likelihood_params=prm, exact=exact, stream=stream.cuda_stream)
    rng = np.random.default_rng(0)
    assert e.Start(rng.uniform(-1.0, 1.0, size=(dim, chains)))
    e.UpdateProposal()
    e.Step(steps)
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); e.Step(steps); b.record(stream)
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    ms = float(np.median(times))
    out = {"workload": name, "dim": dim, "chains": chains, "steps_per_launch": steps, "ms_per_launch": ms,
           "chain_steps_per_s": chains * steps / ms * 1e3, "arithmetic": "reference-order" if exact else "fused",
           "acceptance": float(e.lane("naccept").sum() / (e.total_steps * chains))}
    e.close()
    return out


def main():
    import torch
    pkg = load_package()
    pkg.load()
    rows = [run(pkg, torch, "README-form TDummy (iso) D=50", 50, 65536, pkg.LIKE_ISO_GAUSS, None, True, 1000),
            run(pkg, torch, "README-form TDummy (iso) D=50", 50, 65536, pkg.LIKE_ISO_GAUSS, None, False, 1000),
            run(pkg, torch, "header-form TDummy (quadratic form) D=50", 50, 65536, pkg.LIKE_QUADFORM, tdummy_error(50), True, 200),
            run(pkg, torch, "Rosenbrock D=6", 6, 1 << 20, pkg.LIKE_ROSENBROCK, [100.0], True, 1000),
            run(pkg, torch, "SimpleVAAT.C: header-form TDummy D=100", 100, 32768, pkg.LIKE_QUADFORM, tdummy_error(100), True, 50),
            run(pkg, torch, "README-form D=500", 500, 32768, pkg.LIKE_ISO_GAUSS, None, True, 200)]
    for r in rows:
        print(f"{r['workload']:48s} D={r['dim']:4d} N={r['chains']:8d} {r['arithmetic']:16s} "
              f"{r['chain_steps_per_s']:.3e} chain-steps/s  {r['ms_per_launch'] / r['steps_per_launch'] * 1e3:9.2f} us/step  "
              f"acc {r['acceptance']:.3f}")
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
