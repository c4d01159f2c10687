import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smcmc_amd_loader import load_package
pkg = load_package()
lib = os.path.join(ROOT, 'root-simple-mcmc_amd', 'build', 'prof', 'libsmcmc_amd_prof.so')
names = ["loop top", "scalar half + centre", "covariance update", "trigger / last point", "Philox + normals", "proposal columns", "StepRMS sum", "likelihood, accept, record"]
for dim in (50, 5):
    e = pkg.Engine(dim, 1, mode=pkg.MODE_PER_CHAIN, library=lib)
    assert e.Start(np.zeros(dim))
    e.Step(100)
    n = 512
    rec = e.StepRecorded(n)
    t = np.concatenate([rec["accepted"][0], rec["proposed"][0]])[:8]
    print("D =", dim, "cycles per step:", " | ".join("%s %.0f" % (nm, v / n) for nm, v in zip(names, t)), "| total %.0f" % (t.sum() / n))
