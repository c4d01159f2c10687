"""Config 4 with the header-form TDummy in the fused order: long launches (frozen) and pooled."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
pkg.load()
stream = torch.cuda.Stream()
dim, chains = 500, 32768
cov = np.eye(dim); cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
err = np.linalg.inv(cov)
for mode, name in ((pkg.MODE_FROZEN, "frozen"), (pkg.MODE_POOLED, "pooled")):
    e = pkg.Engine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, mode=mode, exact=False, stream=stream.cuda_stream)
    e.Start(np.zeros(dim)); e.Step(4); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); e.Step(32); b.record(stream); torch.cuda.synchronize()
    print(f"header-form TDummy D=500 fused, {name}: {a.elapsed_time(b) / 32 * 1e3:.0f} us per step")
    e.close()
