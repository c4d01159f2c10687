"""The many-chain semantics (oracle/ensemble_oracle.c) against the single reference
chain (oracle/oracle_core.h), both on the CPU."""
import numpy as np
import pytest


@pytest.mark.parametrize("dim,kind", [(5, 0), (6, 2), (5, 1), (20, 0)])
def test_frozen_ensemble_lanes_are_reference_chains(oracle, dim, kind):
    n, steps = 70, 2500
    rng = np.random.default_rng(dim)
    x0 = rng.uniform(0.5, 1.5, (dim, n)) if kind == 2 else np.zeros((dim, n))
    e = oracle.Ensemble(n, dim, kind=kind, mode=oracle.MODE_FROZEN)
    e.set_acceptance_window(60.0)                    # short window: many per-chain UpdateProposal calls
    assert e.start(x0)
    e.step(steps)
    x = e.x
    for ch in (0, 1, 37, 64, 69):
        c = oracle.Chain(dim, kind=kind, chain_id=ch)
        c.set_covariance_frozen(1)
        c.set_acceptance_window(60.0)
        # the shared proposal is initialised from chain 0's start
        c.start(x0[:, ch])
        if kind == 2 and ch != 0:
            continue                                  # centre := chain 0's start differs; trajectory does not depend on it
        c.run_quiet(steps)
        s = c.scalars
        assert np.array_equal(c.accepted, x[:, ch])
        for name in ("sigma", "acceptance", "acceptance_trials", "rigidity", "step_rms"):
            assert s[name] == e.lane(name)[ch], name
        for name in ("trials", "successes", "next_update"):
            assert s[name] == e.lane(name)[ch], name
        assert s["accepted_logl"] == e.lane("logl")[ch]
        assert s["update_count"] > 2


def test_frozen_rosenbrock_other_chains_match_too(oracle):
    """With a frozen covariance the centre is never used, so every lane equals its own
    reference chain even when the chains start from different points."""
    dim, n, steps = 6, 8, 1500
    rng = np.random.default_rng(1)
    x0 = rng.uniform(0.5, 1.5, (dim, n))
    e = oracle.Ensemble(n, dim, kind=2, mode=oracle.MODE_FROZEN)
    e.start(x0); e.step(steps)
    for ch in range(n):
        c = oracle.Chain(dim, kind=2, chain_id=ch)
        c.set_covariance_frozen(1)
        c.start(x0[:, ch]); c.run_quiet(steps)
        assert np.array_equal(c.accepted, e.x[:, ch])
        assert c.scalars["sigma"] == e.lane("sigma")[ch]


def test_pooled_moments_are_the_running_sums(oracle):
    dim, n, w = 4, 100, 7
    e = oracle.Ensemble(n, dim, mode=oracle.MODE_POOLED)
    e.start(np.zeros(dim))
    xs = []
    for _ in range(w):
        xs.append(e.x.copy())                        # the point UpdateState sees at the start of each step
        e.step(1)
    m = e.reduce_moments()
    y = np.concatenate(xs, axis=1)                   # c0 = 0
    assert m[-1] == n * w
    s1 = m[dim * (dim + 1) // 2: dim * (dim + 1) // 2 + dim]
    assert np.allclose(s1, y.sum(axis=1), rtol=1e-12, atol=1e-12)
    s2 = np.zeros((dim, dim))
    for i in range(dim):
        for j in range(i + 1):
            s2[i, j] = s2[j, i] = m[i * (i + 1) // 2 + j]
    assert np.allclose(s2, y @ y.T, rtol=1e-12, atol=1e-12)


def test_pooled_update_is_the_batch_running_average(oracle):
    dim, n, w = 3, 200, 10
    e = oracle.Ensemble(n, dim, mode=oracle.MODE_POOLED)
    e.start(np.zeros(dim))
    xs = []
    for _ in range(w):
        xs.append(e.x.copy()); e.step(1)
    pts = np.concatenate(xs, axis=1)
    tc, tv = e.shared["central_trials"], e.shared["cov_trials"]
    c_old, v_old = e.center, e.covariance
    e.sync()
    npts = pts.shape[1]
    c_new = (c_old * tc + pts.sum(axis=1)) / (tc + npts)       # TSimpleMCMC.H:1780-1786 fed with a batch
    d = pts - c_new[:, None]
    v_new = (v_old * tv + d @ d.T) / (tv + npts)               # :1795-1816
    assert np.allclose(e.center, c_new, rtol=1e-12, atol=1e-14)
    assert np.allclose(e.covariance, v_new, rtol=1e-11, atol=1e-14)
    u = e.decomposition
    assert np.allclose(u.T @ u, e.covariance, rtol=1e-12)
    # UpdateProposal de-weighted the trial counts (:1056-1067) after min(window, T+n)
    w_cov = e.shared["cov_window"]
    assert e.shared["cov_trials"] == min(max(1.0, 0.5 * min(w_cov, tv + npts)), 0.5 * w_cov)


def test_pooled_adaptation_converges(oracle):
    dim, n = 5, 512
    e = oracle.Ensemble(n, dim, mode=oracle.MODE_POOLED)
    e.start(np.zeros(dim))
    for _ in range(40):
        e.step(64); e.sync()
    nacc0 = e.lane("naccept").sum()
    for _ in range(10):
        e.step(64); e.sync()
    acc = (e.lane("naccept").sum() - nacc0) / (n * 640)
    assert 0.17 < acc < 0.30                         # sigma adapts every chain to the 0.234 target
    assert np.max(np.abs(e.covariance - np.eye(dim))) < 0.15
    x = e.x
    assert np.max(np.abs(x.mean(axis=1))) < 0.2 and np.max(np.abs(np.cov(x) - np.eye(dim))) < 0.35


def test_sharded_oracle_equals_whole_in_frozen_mode(oracle):
    dim, n = 4, 128
    whole = oracle.Ensemble(n, dim, mode=oracle.MODE_FROZEN); whole.start(np.zeros(dim)); whole.step(200)
    hi = oracle.Ensemble(n // 2, dim, chain_offset=n // 2, mode=oracle.MODE_FROZEN); hi.start(np.zeros(dim)); hi.step(200)
    assert np.array_equal(whole.x[:, n // 2:], hi.x)


def test_posterior_moments_accumulator_algebra(oracle, smcmc):
    """PosteriorMoments (the posterior mean/covariance reducer over the pooled moment sums) re-centres every
    window's sums correctly: checked against the plain sample moments of every visited point, with the CPU
    ensemble standing in for the engine."""
    dim, n, window = 4, 64, 25
    e = oracle.Ensemble(n, dim, seed=3)
    assert e.start(np.full(dim, 0.2))

    class Adapter:                      # the two calls PosteriorMoments makes on an engine
        def __init__(self): self.m = None
        def read_moments(self): return self.m
        def GetEstimatedCenter(self): return e.center

    ad = Adapter()
    acc = smcmc.PosteriorMoments(dim)
    xs = []
    for w in range(6):
        for _ in range(window):
            xs.append(e.x.copy())       # the point every chain holds at the start of the step is what gets folded
            e.step(1)
        ad.m = e.reduce_moments()
        acc.add(ad)                     # before apply_moments: the centre is still the window's c0
        e.apply_moments(ad.m)
    pts = np.concatenate(xs, axis=1)    # [dim][steps * chains]
    assert acc.n == pts.shape[1]
    assert np.allclose(acc.mean, pts.mean(axis=1), rtol=1e-12, atol=1e-14)
    assert np.allclose(acc.covariance, np.cov(pts, bias=True), rtol=1e-10, atol=1e-13)
