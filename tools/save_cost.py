"""What saving every step's point costs inside a launch (frozen covariance, config 4 share)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
stream = torch.cuda.Stream()
dim, chains, n = 500, 32768, 8
for exact in (False, True):
    e = pkg.Engine(dim, chains, mode=pkg.MODE_FROZEN, exact=exact, stream=stream.cuda_stream)
    e.Start(np.zeros(dim)); e.Step(4)
    npad = e.nchains_padded
    for stride in (0, 8, 1):
        slots = n // stride if stride else 0
        sx = torch.zeros((max(slots, 1), dim, npad), dtype=torch.float64, device="cuda")
        sl = torch.zeros((max(slots, 1), npad), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            if stride:
                e.StepSave(n, sx.data_ptr(), sl.data_ptr(), stride=stride)
            else:
                e.Step(n)
            b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"{'reference order' if exact else 'fused':16s} 8-step launch, "
              f"{'no save' if not stride else 'save every %d' % stride:14s}: {np.median(ts) / n * 1e3:7.0f} us per step")
    e.close()
