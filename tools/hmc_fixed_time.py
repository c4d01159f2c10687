"""Config 5, fixed step and tuned, fused and reference order (HIP events around smcmc_hmc_step)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
dim, chains = 500, 8192
cov = np.eye(dim); cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
err = np.linalg.inv(cov)
stream = torch.cuda.Stream()
for exact in (False, True):
    for tuned in (False, True):
        h = pkg.HmcEngine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, exact=exact, stream=stream.cuda_stream)
        h.Start(np.ones(dim))
        if not tuned:
            h.SetMeanEpsilon(-0.001)
        h.SetLeapFrog(20)
        h.Step(3); torch.cuda.synchronize()
        n = 8 if not exact else 4
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); h.Step(n); b.record(stream); torch.cuda.synchronize()
        print(f"{'reference order' if exact else 'fused':16s} {'tuned' if tuned else 'fixed step':10s}: {a.elapsed_time(b) / n:.3f} ms per step")
        h.close()
