"""Diagnostic (GPU box): does a second / fourth wavefront per SIMD buy throughput for the step kernel?  The D = 15 and
D = 31 families leave LDS for several workgroups per CU, so the chain count sets the wavefronts per SIMD."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from smcmc_amd_loader import load_package
import torch
pkg = load_package()
for dim in (15, 31):
    for chains in (65536, 131072, 262144):
        e = pkg.Engine(dim, chains)
        e.Start(np.zeros(dim))
        e.Step(64); e.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            e.Step(256)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("D=%d chains=%d (%.0f wavefronts/SIMD): %.3f ms per 256 steps, %.3e chain-steps/s" % (dim, chains, chains / 65536, dt * 1e3, chains * 256 / dt))
        e.close()
