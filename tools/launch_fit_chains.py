import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from smcmc_amd_loader import load_package
pkg = load_package(); pkg.load()
stream = torch.cuda.Stream()
for N in (2048, 4096, 8192, 16384, 32768, 65536):
    e = pkg.Engine(500, N, mode=pkg.MODE_FROZEN, exact=False, stream=stream.cuda_stream)
    e.Start(np.zeros(500)); e.Step(4); torch.cuda.synchronize()
    res = {}
    for n in (1, 8):
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); e.Step(n); b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[n] = float(np.median(ts))
    per_step = (res[8] - res[1]) / 7
    over = res[1] - per_step
    print(f"N={N:6d}: one-step launch {res[1]*1e3:7.0f} us, per step {per_step*1e3:6.0f} us, launch overhead {over*1e3:6.0f} us = {2*N*500*8/over/1e6:6.2f} GB/s of state traffic")
    e.close()
