"""Diagnostic (GPU box): window time of the headline workload (D = 50, 65 536 chains, pooled, 256 steps / launch)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from smcmc_amd_loader import load_package
import torch
pkg = load_package()
ext = int(sys.argv[1]) if len(sys.argv) > 1 else 0
e = pkg.Engine(50, 65536)
e.Start(np.zeros(50))
for _ in range(4):
    e.Step(256); e.sync()
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); e.Step(256); b.record(); evs.append((a, b)); e.sync()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
k = np.mean([a.elapsed_time(b) for a, b in evs])
print("ext %d: window %.3f ms, step launch %.3f ms, model frac %.4f" % (ext, dt * 1e3, k, 65536 * 256 * 816 / (k * 1e-3) / 8e12))
