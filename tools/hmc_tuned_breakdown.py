"""Config 5 with the chain tuning its step: where a step's time goes (sync interval 1 vs none)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
dim, chains = 500, 8192
cov = np.eye(dim); cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
err = np.linalg.inv(cov)
for exact in (False, True):
    for sync in (1, 1 << 30):
        h = pkg.HmcEngine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, exact=exact)
        h.SetSyncInterval(sync)
        h.Start(np.ones(dim))
        h.SetLeapFrog(20)
        h.Step(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10 if not exact else 4
        h.Step(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{'reference order' if exact else 'fused':16s} sync every {'step' if sync == 1 else 'never':6s}: {dt * 1e3:.3f} ms per step")
        h.close()
