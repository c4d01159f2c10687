"""Diagnostic (GPU box): ms per ensemble step of the header-form TDummyLogLikelihood (quadratic form, the reference's own
Error matrix) in the reference's order, frozen covariance, 16 384 chains, D = 64 ... 500 -- with the sparse walk of the
non-zero entries (default) and with the dense D^2-term sum (SMCMC_P_DENSE_QUADFORM = 1).
usage: python tools/header_exact_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
def err(dim):
    cov = np.eye(dim); cov[0, dim-1] = cov[dim-1, 0] = 0.999999
    return np.linalg.inv(cov)
for dim, n in ((64, 16384), (80, 16384), (100, 16384), (128, 16384), (150, 16384), (200, 16384), (256, 16384), (300, 16384), (400, 16384), (500, 16384)):
    for label, like, prm, dense in (("header sparse", pkg.LIKE_QUADFORM, err(dim), 0.0), ("header dense", pkg.LIKE_QUADFORM, err(dim), 1.0)):
        e = pkg.Engine(dim, n, likelihood=like, likelihood_params=prm, mode=pkg.MODE_FROZEN, exact=True)
        if dense is not None: e.set_param("DENSE_QUADFORM", dense)
        assert e.Start(np.zeros(dim))
        e.Step(4); torch.cuda.synchronize()
        t0 = time.perf_counter(); e.Step(8); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(dim, n, label, "%.3f ms/step" % (dt / 8 * 1e3), flush=True)
        e.close()
