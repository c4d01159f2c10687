"""Reference-order large-dimension kernel: frozen covariance (long launches) and pooled (one step per launch)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


rng = np.random.default_rng(0)
for exact in (True, False):
    tag = "reference order" if exact else "fused"
    e = pkg.Engine(200, 16384, likelihood=pkg.LIKE_ROSENBROCK, likelihood_params=[100.0], mode=pkg.MODE_FROZEN, exact=exact)
    e.Start(rng.uniform(0.5, 1.5, (200, 16384)))
    dt = timed(lambda: e.Step(32), 3)
    print(f"config 3 D=200 x 16384 frozen {tag:16s}: {16384 * 32 / dt:.3e} chain-steps/s  {dt / 32 * 1e3:.4f} ms/step")
    e = pkg.Engine(500, 32768, mode=pkg.MODE_FROZEN, exact=exact)
    e.Start(np.zeros(500))
    dt = timed(lambda: e.Step(16), 3)
    print(f"config 4 D=500 x 32768 frozen {tag:16s}: {32768 * 16 / dt:.3e} chain-steps/s  {dt / 16 * 1e3:.4f} ms/step")
    e = pkg.Engine(100, 65536, mode=pkg.MODE_FROZEN, exact=exact)
    e.Start(np.zeros(100))
    dt = timed(lambda: e.Step(32), 3)
    print(f"         D=100 x 65536 frozen {tag:16s}: {65536 * 32 / dt:.3e} chain-steps/s  {dt / 32 * 1e3:.4f} ms/step")
