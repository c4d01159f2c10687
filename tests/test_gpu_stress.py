"""The reference's stress likelihoods on the device, through the C ABI against the CPU oracle, bit
for bit:

* TASymLogLikelihood (TAsymLogLikelihood.H:20-31), 100 dimensions in the reference;
* THorrificLogLikelihood (THorrificLogLikelihood.H:26-38), 75 dimensions, -1E+30 outside the unit box;
* example4's TConstrainedLikelihood (TConstrainedLikelihood.H:26-46), 25 dimensions -- plus a known-answer test: its
  posterior is Gaussian with precision diag(1/s_i^2) + 1 1^T / 16^2, which the sampled ensemble must reproduce.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ASYM, HORRIFIC, CONSTRAINED = 4, 5, 6


def _pair(gpu, oracle, dim, nchains, kind, mode, exact, stride=1):
    prm = oracle.like_params(kind, dim)
    prm = prm if prm.size else None
    e = gpu.Engine(dim, nchains, likelihood=kind, likelihood_params=prm, mode=mode, exact=exact)
    o = oracle.Ensemble(nchains, dim, kind=kind, params=prm, mode=mode, exact=exact)
    if dim > 63 and mode == gpu.MODE_POOLED:
        e.set_param("MOMENT_STRIDE", stride)
        o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), stride)
    return e, o


def _same(e, o, tag):
    assert np.array_equal(e.GetAccepted(), o.x), f"{tag}: accepted points differ"
    for name in ("logl", "sigma", "acceptance", "acceptance_trials", "rigidity", "step_rms", "logl_proposed"):
        a, b = e.lane(name), o.lane(name)
        assert np.array_equal(a, b), f"{tag}: lane field {name} differs (max |d| = {np.max(np.abs(a - b))})"
    for name in ("trials", "successes", "next_update", "naccept", "step_rms_trials"):
        assert np.array_equal(e.lane(name), o.lane(name)), f"{tag}: lane field {name} differs"


def _start_point(kind, dim, nchains, rng):
    if kind == ASYM:
        return rng.normal(0.3, 0.2, size=(dim, nchains))           # a few coordinates below zero: both slopes in play
    if kind == HORRIFIC:
        return rng.uniform(-0.9, 0.9, size=(dim, nchains))          # inside the box; proposals leave it at once
    return np.full((dim, nchains), 76.0) + rng.normal(0.0, 1.0, size=(dim, nchains))


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("kind,dim", [(ASYM, 20), (ASYM, 40), (HORRIFIC, 30), (HORRIFIC, 63), (CONSTRAINED, 25),
                                      (CONSTRAINED, 50)])
def test_stress_likelihoods_frozen_small_dimensions(gpu, oracle, kind, dim, exact):
    n = 192
    rng = np.random.default_rng(kind * 100 + dim)
    e, o = _pair(gpu, oracle, dim, n, kind, gpu.MODE_FROZEN, exact)
    x0 = _start_point(kind, dim, n, rng)
    assert e.Start(x0) and o.start(x0)
    _same(e, o, "start")
    for leg in range(3):
        e.Step(40); o.step(40)
        _same(e, o, f"leg {leg}")
    assert e.lane("naccept").sum() > 0
    if kind == HORRIFIC:
        assert (e.lane("logl_proposed") == -1E+30).any(), "no proposal left the unit box: the early return is untested"


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("kind,dim", [(ASYM, 31), (HORRIFIC, 20), (CONSTRAINED, 25)])
def test_stress_likelihoods_pooled_small_dimensions(gpu, oracle, kind, dim, exact):
    n = 256
    rng = np.random.default_rng(kind * 1000 + dim)
    e, o = _pair(gpu, oracle, dim, n, kind, gpu.MODE_POOLED, exact)
    x0 = _start_point(kind, dim, n, rng)
    assert e.Start(x0) and o.start(x0)
    for w in range(3):
        e.Step(25); o.step(25)
        _same(e, o, f"window {w}")
        e.sync(); o.sync()
        assert np.array_equal(e.covariance, o.covariance), f"window {w}: covariance"
        assert np.array_equal(e.decomposition, o.decomposition), f"window {w}: decomposition"
    e.Step(5); o.step(5)
    _same(e, o, "after the last sync")


@pytest.mark.parametrize("mode", ["frozen", "pooled"])
@pytest.mark.parametrize("kind,dim", [(ASYM, 100), (HORRIFIC, 75), (ASYM, 300), (CONSTRAINED, 100), (CONSTRAINED, 260)])
def test_stress_likelihoods_reference_dimensions(gpu, oracle, kind, dim, mode):
    """The dimensions the reference's headers fix (100 and 75): the large-dimension kernel, reference order."""
    n = 128
    m = gpu.MODE_POOLED if mode == "pooled" else gpu.MODE_FROZEN
    rng = np.random.default_rng(dim)
    e, o = _pair(gpu, oracle, dim, n, kind, m, True, stride=2)
    x0 = _start_point(kind, dim, n, rng)
    assert e.Start(x0) and o.start(x0)
    _same(e, o, "start")
    for w in range(2):
        e.Step(12); o.step(12)
        _same(e, o, f"window {w}")
        if m == gpu.MODE_POOLED:
            e.sync(); o.sync()
            assert np.array_equal(e.covariance, o.covariance), f"window {w}: covariance"
    e.Step(3); o.step(3)
    _same(e, o, "end")
    assert e.lane("naccept").sum() > 0


def test_stress_likelihoods_unsupported_corners(gpu, oracle):
    # fused order for dim > 63 is the matrix-pipe kernel, which carries the three smooth likelihoods only
    e = gpu.Engine(100, 64, likelihood=ASYM, exact=False)
    assert e.Start(np.zeros(100))
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1)
    assert err.value.status == 5   # SMCMC_ERR_UNSUPPORTED
    e = gpu.Engine(64, 64, likelihood=CONSTRAINED, likelihood_params=np.ones(2 + 2 * 64), exact=False)
    assert e.Start(np.zeros(64))
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1)
    assert err.value.status == 5   # SMCMC_ERR_UNSUPPORTED
    # wrong parameter count
    with pytest.raises(gpu.SmcmcError):
        gpu.Engine(25, 64, likelihood=CONSTRAINED, likelihood_params=np.ones(10)).Start(np.zeros(25))
    # no gradient of their own: HMC targets only through the gradient types that do not ask the functor for one
    # (TSimpleHMC.H:417-454, 524-528; tests/test_gpu_hmc.py::test_hmc_targets_without_a_gradient) -- the default type
    # calls the functor's gradient, which the reference's stress functors answer with an exception
    for kind in (ASYM, HORRIFIC, CONSTRAINED):
        prm = oracle.like_params(kind, 25)
        h = gpu.HmcEngine(25, 64, likelihood=kind, likelihood_params=prm if prm.size else None)
        h.Start(np.full(25, 0.01))
        with pytest.raises(gpu.SmcmcError) as err:
            h.Step(1, gradient_type=0)
        assert err.value.status == 3   # SMCMC_ERR_RUNTIME
        h.close()


def test_constrained_posterior_known_answer(gpu, oracle):
    """example4's posterior in closed form: Gaussian with precision A = diag(1/s_i^2) + 1 1^T / c^2 and mean
    A^-1 (mu_i / s_i^2 + S / c^2).  A pooled ensemble must land on it (means within 5 standard errors plus the
    thinning allowance, standard deviations within 5 %)."""
    dim, n = 25, 4096
    prm = oracle.constrained_params(dim)
    S, c, mu, s = prm[0], prm[1], prm[2:2 + dim], prm[2 + dim:]
    A = np.diag(1.0 / s ** 2) + np.ones((dim, dim)) / c ** 2
    cov = np.linalg.inv(A)
    mean = cov @ (mu / s ** 2 + S / c ** 2)

    e = gpu.Engine(dim, n, likelihood=CONSTRAINED, likelihood_params=prm, mode=gpu.MODE_POOLED, exact=False)
    assert e.Start(mu.copy())
    for _ in range(40):                                             # burn-in with pooled covariance updates
        e.Step(50)
        e.sync()
    draws = []
    for _ in range(20):
        e.Step(50)
        e.sync()
        draws.append(e.GetAccepted().copy())
    x = np.concatenate(draws, axis=1)                               # [dim][20 n]
    got_mean, got_sd = x.mean(axis=1), x.std(axis=1)
    sd = np.sqrt(np.diag(cov))
    # 20 draws per chain 50 steps apart are correlated: count the chains only
    assert np.all(np.abs(got_mean - mean) < 5.0 * sd / np.sqrt(n)), np.max(np.abs(got_mean - mean) / sd)
    assert np.all(np.abs(got_sd / sd - 1.0) < 0.05), np.max(np.abs(got_sd / sd - 1.0))
    # the constraint on the sum: its posterior spread is 1 / sqrt(1/c^2 + 1/sum s_i^2)-like; compare with the closed form
    tot = x.sum(axis=0)
    tot_sd = np.sqrt(np.ones(dim) @ cov @ np.ones(dim))
    assert abs(tot.mean() - np.ones(dim) @ mean) < 5.0 * tot_sd / np.sqrt(n)
    assert abs(tot.std() / tot_sd - 1.0) < 0.05
