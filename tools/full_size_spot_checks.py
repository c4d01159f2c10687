"""Large ensembles, a few chains each compared with the single-chain oracle on the same random stream (chains of these
modes share nothing, so any chain can be checked alone), plus config 2's header-form likelihood pooled at full size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smcmc_amd_loader import load_package  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = load_package()
pkg.load()
O.build()
SEED = 20240607

# (a) variable-at-a-time, D = 50, 65 536 chains
dim, n, steps = 50, 65536, 400
e = pkg.VaatEngine(dim, n, seed=SEED)
x0 = np.full(dim, 0.25)
assert e.Start(x0)
e.UpdateProposal()
e.Step(steps)
x, sg = e.GetAccepted(), e.per_dim("sigma")
for c in (0, 63, 64, 40000, 65535):
    v = O.Vaat(1, dim, seed=SEED, chain_offset=c)
    assert v.start(x0)
    v.update_proposal(); v.step(steps)
    assert np.array_equal(v.x[:, 0], x[:, c]) and np.array_equal(v.per_dim("sigma")[:, 0], sg[:, c]), c
print(f"VAAT D={dim}, {n} chains, {steps} steps: chains 0, 63, 64, 40000, 65535 are their reference chains", flush=True)
e.close()

# (b) adaptive Metropolis, frozen covariance, D = 50, 65 536 chains
steps = 300
e = pkg.Engine(dim, n, seed=SEED, mode=pkg.MODE_FROZEN)
assert e.Start(np.zeros(dim))
e.Step(steps)
x, logl, sg = e.GetAccepted(), e.lane("logl"), e.lane("sigma")
for c in (0, 12345, 65535):
    ch = O.Chain(dim, seed=SEED, chain_id=c)
    ch.set_covariance_frozen(1)
    assert ch.start(np.zeros(dim))
    ch.run_quiet(steps)
    s = ch.scalars
    assert np.array_equal(ch.accepted, x[:, c]) and s["accepted_logl"] == logl[c] and s["sigma"] == sg[c], c
print(f"Metropolis frozen D={dim}, {n} chains, {steps} steps: chains 0, 12345, 65535 are their reference chains", flush=True)
e.close()

# (c) config 2 with the header-form TDummy, pooled, full size, one short window
prm = O.dummy_error_matrix(dim)[1]
e = pkg.Engine(dim, n, likelihood=1, likelihood_params=prm, seed=SEED, mode=pkg.MODE_POOLED)
o = O.Ensemble(n, dim, kind=1, params=prm, seed=SEED, mode=O.MODE_POOLED)
assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
e.Step(12); o.step(12)
e.sync(); o.sync()
e.Step(2); o.step(2)
assert np.array_equal(e.GetAccepted(), o.x) and np.array_equal(e.lane("logl"), o.lane("logl"))
assert np.array_equal(e.covariance, o.covariance) and np.array_equal(e.decomposition, o.decomposition)
print(f"config 2, header-form TDummy, pooled, {n} chains: 12 steps + sync + 2 steps bit-identical", flush=True)
e.close()

# (d) HMC with a fixed step, D = 100, 8 192 chains
dim, n, steps = 100, 8192, 5
h = pkg.HmcEngine(dim, n, seed=SEED)
h.Start(np.ones(dim))
h.SetMeanEpsilon(-0.1); h.SetLeapFrog(8)
h.Step(steps)
q, m, logl = h.state()
for c in (0, 4097, 8191):
    r = O.Hmc(dim, seed=SEED, chain_id=c)
    r.start(np.ones(dim)); r.set_mean_epsilon(-0.1); r.set_leapfrog(8)
    r.run(steps)
    assert np.array_equal(r.accepted, q[:, c]) and np.array_equal(r.momentum, m[:, c]), c
print(f"HMC fixed step D={dim}, {n} chains, {steps} steps: chains 0, 4097, 8191 are their reference chains", flush=True)
