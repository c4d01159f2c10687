"""Diagnostic (GPU box): window time of the headline workload (D = 50, 65 536 chains, pooled, 256 steps / launch), pooled
and with the covariance frozen (the same kernel without the moment fold).
usage: python tools/headline_time.py [dim [chains]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcmc_amd_loader import load_package  # noqa: E402
import torch  # noqa: E402

pkg = load_package()
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 50
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for mode, name in ((pkg.MODE_POOLED, "pooled"), (pkg.MODE_FROZEN, "frozen"), (pkg.MODE_POOLED, "pooled")):
    e = pkg.Engine(dim, chains, mode=mode)
    e.Start(np.zeros(dim))
    for _ in range(4):
        e.Step(256)
        if mode == pkg.MODE_POOLED:
            e.sync()
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e.Step(256); b.record(); evs.append((a, b))
        if mode == pkg.MODE_POOLED:
            e.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    k = np.mean([a.elapsed_time(b) for a, b in evs])
    print("%s: window %.3f ms, step launch %.3f ms, model frac %.4f" %
          (name, dt * 1e3, k, chains * 256 * (16 * dim + 16) / (k * 1e-3) / 8e12), flush=True)
    e.close()
