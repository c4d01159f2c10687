"""Launch cost of the large-dimension kernels as a + b * nsteps (frozen covariance, config 4 share)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
stream = torch.cuda.Stream()
for exact in (False, True):
    e = pkg.Engine(500, 32768, mode=pkg.MODE_FROZEN, exact=exact, stream=stream.cuda_stream)
    e.Start(np.zeros(500))
    e.Step(4)
    torch.cuda.synchronize()
    rows = []
    for n in (1, 1, 2, 4, 8, 16):
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); e.Step(n); b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        rows.append((n, float(np.median(ts))))
    n_, t_ = np.array([r[0] for r in rows[1:]], float), np.array([r[1] for r in rows[1:]])
    b_, a_ = np.polyfit(n_, t_, 1)
    print(("reference order" if exact else "fused"), " ".join(f"n={n}: {t:.3f} ms" for n, t in rows),
          f"| fit: {a_ * 1e3:.0f} us per launch + {b_ * 1e3:.0f} us per step")
    # back-to-back one-step launches (the pooled pattern without the fold)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(32):
        e.Step(1)
    b.record(stream)
    torch.cuda.synchronize()
    print(f"   32 one-step launches back to back: {a.elapsed_time(b) / 32 * 1e3:.0f} us each")
    e.close()
