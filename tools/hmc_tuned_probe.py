"""Diagnostic (GPU box): acceptance of TSimpleHMC's own tuning (UpdateErrorMatrix, TSimpleHMC.H:703-858) at config 5's
size on quadratic-form targets of different conditioning.  The header-form TDummy (one pair with rho = 0.999999) cannot be
tuned by the reference's rule: it clamps minScale at 0.01 and sets epsilon >= 0.5 minScale = 0.005, five times the
stability limit of that target's stiff direction."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()
dim, chains = 500, 8192
for lo, hi in ((0.25, 4.0), (0.5, 2.0), (0.04, 1.0)):
    var = np.linspace(lo, hi, dim)
    err = np.diag(1.0 / var)
    h = pkg.HmcEngine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, seed=20240607, exact=False)
    h.Start(np.ones(dim))
    h.SetLeapFrog(20)
    h.Step(300)
    a0, t0 = h.lane("naccept").sum(), h.lane("trials").sum()
    h.Step(100)
    a1, t1 = h.lane("naccept").sum(), h.lane("trials").sum()
    print("variances %.2f..%.2f: acceptance %.3f, |epsilon| %.4f, updates %d" %
          (lo, hi, (a1 - a0) / (t1 - t0), np.abs(h.lane("mean_epsilon")).mean(), h.tuning["updates"]), flush=True)
    h.close()
