"""Diagnostic: wall time per step of the self-tuned HMC path at config 5's size (run on the GPU box)."""
import sys, time, gc, numpy as np
sys.path.insert(0, "/root/repo")
from smcmc_amd_loader import load_package
import torch
pkg = load_package()
import bench
err = bench.tdummy_error(500)
for exact in (True, False):
    h = pkg.HmcEngine(500, 8192, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, exact=exact,
                      stream=torch.cuda.current_stream().cuda_stream)
    h.Start(np.ones(500)); h.SetLeapFrog(20)
    ts = []
    for s in range(5 if exact else 14):
        t0 = time.perf_counter(); h.Step(1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("exact" if exact else "fused", " ".join("%.1f" % t for t in ts), "gc", gc.get_count())
    t0 = time.perf_counter(); h.Step(10); torch.cuda.synchronize(); print("  Step(10): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    h.close()
