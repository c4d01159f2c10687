"""Cycles per section of panel_step_kernel (the profiling build of tools/micro/build_panelprof.sh), frozen covariance."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smcmc_amd_loader import load_package
import torch
pkg = load_package()
lib = os.path.join(ROOT, 'root-simple-mcmc_amd', 'build', 'prof', 'libsmcmc_amd_panelprof.so')
rng = np.random.default_rng(0)
for dim, chains, like, prm, x0 in ((200, 16384, pkg.LIKE_ROSENBROCK, [100.0], None), (500, 32768, pkg.LIKE_ISO_GAUSS, None, 0.0)):
    e = pkg.Engine(dim, chains, likelihood=like, likelihood_params=prm, mode=pkg.MODE_FROZEN, exact=True, library=lib)
    e.Start(rng.uniform(0.5, 1.5, (dim, chains)) if x0 is None else np.zeros(dim))
    e.Step(4); torch.cuda.synchronize()
    e.Step(32); torch.cuda.synchronize()
    e.close()
