"""Timing probe of the pooled update at the headline shape: 200 x (one step + sync); run under rocprofv3 --kernel-trace --stats
for the per-kernel split of a window's tail."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from smcmc_amd_loader import load_package
pkg=load_package()
e=pkg.Engine(50,65536,mode=pkg.MODE_POOLED)
assert e.Start(np.zeros(50))
for _ in range(3): e.Step(64); e.sync()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200): e.Step(1); e.sync()
torch.cuda.synchronize()
print("ms per (1 step + sync)", (time.perf_counter()-t0)/200*1e3)
print(e.decomposition[:2,:3])
