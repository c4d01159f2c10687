"""Configs 3 and 4 at their full ensemble size for a few steps, pooled with the covariance fed every step (ring of
per-step points, 25 / 85 moment slices), device against CPU oracle, bit for bit.  Minutes of oracle time: not in the suite."""
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
rng = np.random.default_rng(0)
cases = [("config 3", 200, 16384, 2, [100.0], rng.uniform(0.5, 1.5, (200, 16384)), 5),
         ("config 4 share", 500, 32768, 0, None, np.zeros(500), 3)]
for name, dim, chains, kind, prm, x0, steps in cases:
    for exact in (False, True):
        e = pkg.Engine(dim, chains, likelihood=kind, likelihood_params=prm, mode=pkg.MODE_POOLED, exact=exact)
        o = O.Ensemble(chains, dim, kind=kind, params=prm, mode=O.MODE_POOLED, exact=exact)
        o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), 1)
        assert e.Start(x0) and o.start(x0)
        t0 = time.perf_counter()
        e.Step(steps); e.sync()
        o.step(steps); o.sync()
        dt = time.perf_counter() - t0
        e.Step(1); o.step(1)
        same = (np.array_equal(e.GetAccepted(), o.x) and np.array_equal(e.lane("logl"), o.lane("logl"))
                and np.array_equal(e.lane("sigma"), o.lane("sigma")) and np.array_equal(e.covariance, o.covariance)
                and np.array_equal(e.decomposition, o.decomposition) and np.array_equal(e.GetEstimatedCenter(), o.center))
        print(f"{name}, D={dim}, {chains} chains, {'reference order' if exact else 'fused'}: {steps} steps + sync + 1 step, "
              f"moment group {int(e.get_param('MOMENT_GROUP'))}, {dt:.0f} s: bit-identical {same}", flush=True)
        assert same
        e.close()

# the HMC engine's pooled tuning with many moment groups (more than one reduction chunk of 32)
for dim, chains, steps in ((20, 8192, 6), (500, 8192, 2)):
    prm = None
    kind = 0
    if dim == 500:
        cov = np.eye(dim); cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
        prm, kind = np.linalg.inv(cov), 1
    h = pkg.HmcEngine(dim, chains, likelihood=kind, likelihood_params=prm, seed=5)
    ho = O.HmcEnsemble(chains, dim, kind=kind, params=prm, seed=5, group=h.moment_group, sync_every=1, potential_from_gradient=True)
    h.Start(np.ones(dim)); ho.start(np.ones(dim))
    if dim == 500:
        h.SetLeapFrog(20); ho.set_leapfrog(20)
    t0 = time.perf_counter()
    h.Step(steps); ho.step(steps)
    q, m, logl = h.state()
    oq, om = ho.state()
    same = (np.array_equal(q, oq) and np.array_equal(m, om) and np.array_equal(h.lane("mean_epsilon"), ho.lane("mean_epsilon"))
            and np.array_equal(h.average, ho.average) and np.array_equal(h.covariance, ho.covariance)
            and h.tuning["trace"] == ho.shared["trace"] and h.tuning["cov_trials"] == ho.shared["cov_trials"])
    print(f"HMC D={dim}, {chains} chains, default tuning, {steps} steps, moment group {h.moment_group} "
          f"({-(-chains // h.moment_group)} groups), {time.perf_counter() - t0:.0f} s: bit-identical {same}", flush=True)
    assert same
    h.close()
