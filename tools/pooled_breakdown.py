"""Pooled covariance fed every step at D > 63: one-step launch vs fold, configs 3 and 4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
stream = torch.cuda.Stream()
rng = np.random.default_rng(0)
for name, dim, chains, like, prm, x0 in (("config 3", 200, 16384, pkg.LIKE_ROSENBROCK, [100.0], rng.uniform(0.5, 1.5, (200, 16384))),
                                         ("config 4", 500, 32768, pkg.LIKE_ISO_GAUSS, None, np.zeros(500))):
    for exact in (True, False):
        res = {}
        for mode in (pkg.MODE_FROZEN, pkg.MODE_POOLED):
            e = pkg.Engine(dim, chains, likelihood=like, likelihood_params=prm, mode=mode, exact=exact, stream=stream.cuda_stream)
            e.Start(x0); e.Step(4); torch.cuda.synchronize()
            for n in (1, 32):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(32 // n):
                    e.Step(n)
                b.record(stream)
                torch.cuda.synchronize()
                res[(mode, n)] = a.elapsed_time(b) / 32 * 1e3
            e.close()
        f1, f32, p1, p32 = res[(pkg.MODE_FROZEN, 1)], res[(pkg.MODE_FROZEN, 32)], res[(pkg.MODE_POOLED, 1)], res[(pkg.MODE_POOLED, 32)]
        print(f"{name} {'reference order' if exact else 'fused':16s}: in a long launch {f32:6.0f} us/step, one-step launches {f1:6.0f}, "
              f"pooled (fold every step) {p32:6.0f} -> fold {p32 - f1:6.0f} us, launch overhead {f1 - f32:6.0f} us")
