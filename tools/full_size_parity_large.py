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

# config 4 with the likelihood it names (header-form TDummy), fused row-wise order, pooled, full size
dim, chains = 500, 32768
prm = O.dummy_error_matrix(dim)[1]
e = pkg.Engine(dim, chains, likelihood=1, likelihood_params=prm, mode=pkg.MODE_POOLED, exact=False)
o = O.Ensemble(chains, dim, kind=1, params=prm, mode=O.MODE_POOLED, exact=False)
o.set_quadform_rowwise(1)
o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), 1)
assert e.Start(np.full(dim, 0.02)) and o.start(np.full(dim, 0.02))
t0 = time.perf_counter()
e.Step(3); o.step(3); e.sync(); o.sync(); e.Step(1); o.step(1)
same = (np.array_equal(e.GetAccepted(), o.x) and np.array_equal(e.lane("logl"), o.lane("logl"))
        and np.array_equal(e.covariance, o.covariance) and np.array_equal(e.decomposition, o.decomposition))
print(f"config 4 header-form TDummy, fused, pooled, {chains} chains: 3 steps + sync + 1 step, {time.perf_counter() - t0:.0f} s: "
      f"bit-identical {same}", flush=True)
assert same
e.close()

# a stress likelihood at the reference's dimension with the headline's chain count (large-dimension kernel, pooled)
dim, chains = 100, 65536
rng = np.random.default_rng(5)
x0 = rng.normal(0.3, 0.2, size=(dim, chains))
e = pkg.Engine(dim, chains, likelihood=4, mode=pkg.MODE_POOLED)
o = O.Ensemble(chains, dim, kind=4, params=np.array([-1.0, 100.0]), mode=O.MODE_POOLED)
o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), 1)
assert e.Start(x0) and o.start(x0)
e.Step(9); o.step(9); e.sync(); o.sync(); e.Step(2); o.step(2)
same = (np.array_equal(e.GetAccepted(), o.x) and np.array_equal(e.lane("logl"), o.lane("logl"))
        and np.array_equal(e.covariance, o.covariance))
print(f"TASymLogLikelihood D={dim}, {chains} chains, pooled: 9 steps + sync + 2 steps, moment group "
      f"{int(e.get_param('MOMENT_GROUP'))}: bit-identical {same}", flush=True)
assert same
e.close()
