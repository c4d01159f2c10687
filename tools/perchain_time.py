#!/usr/bin/env python3
"""SMCMC_MODE_PER_CHAIN timing: chain-steps/s and the stream rate of the one configuration whose HBM traffic per
chain-step is O(D^2) (every chain reads its decomposition and reads + writes its covariance every step).
  algorithmic bytes per chain-step = 8 * 3 * D (D + 1) / 2  (the three streams)  +  8 * 7 D + 16
    (x read by the centre loop and by the trial step, centre read + written, last point written, proposal written and
     read by the likelihood, x written on accept is not counted)
usage: python tools/perchain_time.py [--dim 50] [--chains 4096 65536] [--steps 64] [--json out.json]
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this command gives the measured traffic of perchain_step_kernel."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def algorithmic_bytes(dim):
    return 8 * 3 * dim * (dim + 1) // 2 + 8 * 7 * dim + 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--chains", type=int, nargs="+", default=[4096, 65536])
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--frozen", action="store_true", help="SetCovarianceFrozen: the covariance stream is skipped")
    ap.add_argument("--header-tdummy", action="store_true",
                    help="the header form of TDummyLogLikelihood (quadratic form, TDummyLogLikelihood.H:24-28) instead of the README form")
    ap.add_argument("--kernel", choices=("auto", "lane", "wave"), default="auto",
                    help="SMCMC_P_PERCHAIN_WAVE: one chain per lane (perchain_step_kernel) or per wavefront (perchain_wave_kernel)")
    ap.add_argument("--library", default=None, help="another build of the library (an experiment)")
    ap.add_argument("--json")
    a = ap.parse_args()
    import torch
    from smcmc_amd_loader import load_package
    pkg = load_package()
    stream = torch.cuda.current_stream()
    rows = []
    for n in a.chains:
        kw = {}
        if a.header_tdummy:
            cov = np.eye(a.dim)
            cov[0, a.dim - 1] = cov[a.dim - 1, 0] = 0.999999                     # TDummyLogLikelihood::Init(), :44-142
            kw = {"likelihood": pkg.LIKE_QUADFORM, "likelihood_params": np.linalg.inv(cov)}
        e = pkg.Engine(a.dim, n, mode=pkg.MODE_PER_CHAIN, stream=stream.cuda_stream, library=a.library, **kw)
        e.set_param("PERCHAIN_WAVE", {"auto": -1, "lane": 0, "wave": 1}[a.kernel])
        if a.frozen:
            e.SetCovarianceFrozen(True)
        assert e.Start(np.zeros(a.dim))
        e.Step(8)
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        for _ in range(a.launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); e.Step(a.steps); e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kms = float(np.mean([x.elapsed_time(y) for x, y in evs]))
        rate = n * a.steps / (kms * 1e-3)
        rows.append({"dim": a.dim, "chains": n, "kernel": "one chain per wavefront" if e.get_param("PERCHAIN_WAVE") else "one chain per lane", "likelihood": "header TDummy" if a.header_tdummy else "README TDummy", "steps_per_launch": a.steps, "kernel_ms_per_launch": kms,
                     "us_per_ensemble_step": kms * 1e3 / a.steps, "chain_steps_per_s": rate,
                     "wall_chain_steps_per_s": n * a.steps * a.launches / dt,
                     "algorithmic_bytes_per_chain_step": algorithmic_bytes(a.dim),
                     "stream_GBps": rate * algorithmic_bytes(a.dim) / 1e9,
                     "hbm_frac": rate * algorithmic_bytes(a.dim) / 1e9 / 8000.0,
                     "state_MB": n * (8 * (a.dim * (a.dim + 1) // 2 + a.dim * a.dim) + 8 * 5 * a.dim) / 1e6,
                     "updates_per_chain": float(e.lane("update_count").mean())})
        print(json.dumps(rows[-1]), flush=True)
        e.close()
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
