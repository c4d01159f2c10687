"""One adaptation window of the headline workload at its full size (D = 50, 65 536 chains, 256 steps, pooled) on the
device and in the CPU oracle, compared bit for bit -- about a minute of oracle time, so not part of the suite."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smcmc_amd_loader import load_package  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = load_package()
pkg.load()
O.build()
dim, chains, window = 50, 65536, 256
e = pkg.Engine(dim, chains, mode=pkg.MODE_POOLED)
o = O.Ensemble(chains, dim, mode=O.MODE_POOLED)
assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
for w in range(2):
    t0 = time.perf_counter()
    e.Step(window); e.sync(); e.lane("logl")   # the read waits for the stream
    t1 = time.perf_counter()
    o.step(window); o.sync()
    t2 = time.perf_counter()
    same = (np.array_equal(e.GetAccepted(), o.x) and np.array_equal(e.lane("logl"), o.lane("logl"))
            and np.array_equal(e.lane("sigma"), o.lane("sigma")) and np.array_equal(e.lane("naccept"), o.lane("naccept"))
            and np.array_equal(e.covariance, o.covariance) and np.array_equal(e.decomposition, o.decomposition)
            and np.array_equal(e.GetEstimatedCenter(), o.center))
    print(f"window {w}: device {t1 - t0:.3f} s, oracle {t2 - t1:.1f} s ({chains * window / (t2 - t1):.3e} chain-steps/s, one core), "
          f"bit-identical: {same}; acceptance {e.lane('naccept').sum() / (chains * window * (w + 1)):.3f}")
    assert same
