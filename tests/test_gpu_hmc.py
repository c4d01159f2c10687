"""HMC on the GPU (smcmc_hmc_*, BASELINE config 5 shape) against the CPU restatement of
sMCMC::TSimpleHMC (oracle/hmc_oracle.c): every chain is an independent reference chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(dim, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((dim, dim)) / np.sqrt(dim)
    return a @ a.T + np.eye(dim)


def _check(gpu, oracle, dim, nchains, kind, params, x0, eps, leap, steps, alpha=0.0, chains=(0, 1, 63, 64),
           exact=True):
    e = gpu.HmcEngine(dim, nchains, likelihood=kind, likelihood_params=params, seed=99, exact=exact)
    e.SetAlpha(alpha)
    e.Start(x0)
    e.SetMeanEpsilon(-eps)         # fixed step length: fMeanEpsilon < 0 (TSimpleHMC.H:297, 304-343)
    e.SetLeapFrog(leap)
    e.Step(1)
    e.Step(steps - 1)
    q, m, logl = e.state()
    for ch in chains:
        if ch >= nchains:
            continue
        h = oracle.Hmc(dim, kind=kind, params=params, seed=99, chain_id=ch, potential_from_gradient=True,
                       fused_gradient=not exact)
        h.set_alpha(alpha)
        h.start(x0 if x0.ndim == 1 else x0[:, ch])
        h.set_mean_epsilon(-eps)
        h.set_leapfrog(leap)
        h.run(steps)
        s = h.scalars
        assert np.array_equal(h.accepted, q[:, ch]), f"chain {ch}: positions differ"
        assert np.array_equal(h.momentum, m[:, ch]), f"chain {ch}: momenta differ"
        assert -s["accepted_potential"] == logl[ch]
        assert s["current_acceptance"] == e.lane("acceptance")[ch]
        assert s["last_accept"] == e.lane("last_accept")[ch]
        assert s["step_count"] == e.lane("trials")[ch] == steps
    return e


@pytest.mark.parametrize("dim,nchains", [(5, 70), (64, 65), (300, 64)])
def test_hmc_iso_matches_reference_chain(gpu, oracle, dim, nchains):
    e = _check(gpu, oracle, dim, nchains, 0, None, np.ones(dim), 0.1, 10, 12)
    assert e.lane("naccept").sum() > 0


@pytest.mark.parametrize("dim,nchains,leap", [(6, 70, 20), (200, 64, 8)])
def test_hmc_rosenbrock_matches_reference_chain(gpu, oracle, dim, nchains, leap):
    rng = np.random.default_rng(dim)
    x0 = rng.uniform(0.9, 1.1, (dim, nchains))
    _check(gpu, oracle, dim, nchains, 2, [100.0], x0, 0.004, leap, 8)


@pytest.mark.parametrize("dim,nchains,leap", [(5, 70, 20), (100, 64, 20), (300, 64, 6), (63, 64, 4), (64, 40, 4),
                                              (128, 64, 3), (129, 33, 3), (256, 64, 2), (257, 64, 2), (512, 32, 2)])
def test_hmc_quadratic_form_matches_reference_chain(gpu, oracle, dim, nchains, leap):
    """The D x D gradient contraction of TDummyLogLikelihood.H:34-42 (config 5's hot loop)."""
    err = _spd(dim, dim)
    _check(gpu, oracle, dim, nchains, 1, err, np.ones(dim), 0.05, leap, 6, alpha=0.3)


@pytest.mark.parametrize("dim,nchains,leap", [(5, 70, 20), (16, 33, 3), (100, 64, 20), (300, 96, 6), (500, 64, 4),
                                              (512, 32, 2), (63, 64, 4), (64, 40, 4), (128, 64, 3), (129, 33, 3),
                                              (256, 64, 2), (257, 64, 2)])
def test_hmc_quadratic_form_on_the_matrix_pipe(gpu, oracle, dim, nchains, leap):
    """Fused order: the gradient contraction as a chain of v_mfma_f64_16x16x4_f64 (32 chains per
    workgroup, positions / momenta / gradients in the matrix layout), bit for bit the oracle's
    fused-gradient chain."""
    err = _spd(dim, dim + 1)
    rng = np.random.default_rng(dim)
    x0 = np.ones(dim) if dim % 2 else rng.uniform(0.5, 1.5, (dim, nchains))
    _check(gpu, oracle, dim, nchains, 1, err, x0, 0.05, leap, 5, alpha=0.3, exact=False,
           chains=(0, 1, 15, 16, 31, 32, 63, 64, 95))


def test_hmc_fused_order_stays_close_to_reference_order(gpu):
    dim, n = 200, 64
    err = _spd(dim, 3)
    out = []
    for exact in (True, False):
        e = gpu.HmcEngine(dim, n, likelihood=1, likelihood_params=err, seed=4, exact=exact)
        e.Start(np.ones(dim)); e.SetMeanEpsilon(-0.02); e.SetLeapFrog(10); e.Step(3)
        out.append(e.state())
    assert np.allclose(out[0][0], out[1][0], rtol=1e-9, atol=1e-11)
    assert np.allclose(out[0][2], out[1][2], rtol=1e-9)


def test_hmc_quadform_potential_close_to_reference_order(gpu, oracle):
    """The engine folds -log L = 1/2 q^T (Error q) per row instead of the reference's single
    D^2-term running sum: same value to rounding."""
    dim = 40
    err = _spd(dim, 1)
    h1 = oracle.Hmc(dim, kind=1, params=err, seed=5)
    h2 = oracle.Hmc(dim, kind=1, params=err, seed=5, potential_from_gradient=True)
    for h in (h1, h2):
        h.start(np.ones(dim)); h.set_mean_epsilon(-0.05); h.set_leapfrog(10); h.run(50)
    assert np.array_equal(h1.accepted, h2.accepted)
    assert abs(h1.scalars["accepted_potential"] - h2.scalars["accepted_potential"]) < 1e-11 * abs(h1.scalars["accepted_potential"])


def test_hmc_default_configuration_runs(gpu):
    """After Start the step length is 0.05 > 0 and the leapfrog count 10 > 0 (TSimpleHMC.H:133, 229): the chains
    retune themselves (nothing is refused; the first rounds of this engine answered SMCMC_ERR_UNSUPPORTED here)."""
    e = gpu.HmcEngine(5, 10)
    e.Start(np.zeros(5))
    e.Step(3)
    assert np.all(e.lane("trials") == 3) and np.all(e.lane("leapfrog") > 0)


def test_hmc_posterior_iso(gpu):
    dim, n = 8, 4096
    e = gpu.HmcEngine(dim, n, seed=3)
    e.Start(np.zeros(dim)); e.SetMeanEpsilon(-0.25); e.SetLeapFrog(8)
    e.Step(300)
    q, _, _ = e.state()
    assert np.max(np.abs(q.mean(axis=1))) < 0.08
    assert np.max(np.abs(np.cov(q) - np.eye(dim))) < 0.12
    assert e.lane("naccept").mean() / 300 > 0.8


# ---------------------------------------------------------------- the chains retune themselves (TSimpleHMC.H:302-345, 665-858)
def _adaptive_pair(gpu, oracle, dim, nchains, kind, params, exact, sync, fix_leapfrog=None, fix_epsilon=None):
    e = gpu.HmcEngine(dim, nchains, likelihood=kind, likelihood_params=params, seed=5, exact=exact)
    e.SetSyncInterval(sync)
    o = oracle.HmcEnsemble(nchains, dim, kind=kind, params=params, seed=5, group=e.moment_group, sync_every=sync,
                           potential_from_gradient=True, fused_gradient=not exact)
    return e, o


def _same_hmc(e, o, tag):
    q, m, logl = e.state()
    oq, om = o.state()
    assert np.array_equal(q, oq), f"{tag}: positions"
    assert np.array_equal(m, om), f"{tag}: momenta"
    assert np.array_equal(logl, -o.lane("accepted_potential")), f"{tag}: potentials"
    assert np.array_equal(e.lane("mean_epsilon"), o.lane("mean_epsilon")), f"{tag}: fMeanEpsilon"
    assert np.array_equal(e.lane("leapfrog"), o.lane("leapfrog_steps").astype(np.int32)), f"{tag}: fLeapFrogSteps"
    assert np.array_equal(e.lane("reversal_len"), o.lane("reversal_len")), f"{tag}: fReversalLen"
    assert np.array_equal(e.lane("acceptance"), o.lane("current_acceptance")), f"{tag}: fCurrentAcceptance"
    t, s = e.tuning, o.shared
    for name in ("trace", "orbit", "updates", "cov_trials", "average_trials", "steps_remaining", "steps_since_update",
                 "est_trace"):
        assert t[name] == s[name], f"{tag}: shared {name}: {t[name]} != {s[name]}"
    assert np.array_equal(e.average, o.average), f"{tag}: fAveragePoint"
    assert np.array_equal(e.covariance, o.covariance), f"{tag}: fEstimatedCovariance"


@pytest.mark.parametrize("kind,dim,nchains,sync", [(0, 5, 1, 1), (0, 5, 70, 1), (2, 6, 64, 1), (0, 20, 130, 4),
                                                   (0, 100, 64, 1), (2, 200, 64, 3)])
def test_hmc_default_tuning_matches_the_ensemble_oracle(gpu, oracle, kind, dim, nchains, sync):
    """SimpleHMC.C:46-72's call sequence -- Start, then Step with no SetMeanEpsilon and no SetLeapFrog: every chain
    retunes its step length and leapfrog count by the reversal test (:302-323) and the pooled covariance (:665-858)."""
    prm = [100.0] if kind == 2 else None
    e, o = _adaptive_pair(gpu, oracle, dim, nchains, kind, prm, True, sync)
    x0 = np.ones(dim)
    e.Start(x0); o.start(x0)
    assert e.GetMeanEpsilon() == 0.05 and e.GetLeapFrog() == 10
    nsteps = 40 if dim <= 20 else 12
    done = 0
    for chunk in (1, 2, nsteps - 3):
        e.Step(chunk); o.step(chunk)
        done += chunk
        _same_hmc(e, o, f"after {done} steps")
    assert e.tuning["updates"] >= 1                          # the pooled UpdateErrorMatrix went through
    assert len(np.unique(e.lane("leapfrog"))) >= 1
    if nchains == 1:
        h = oracle.Hmc(dim, kind=kind, params=prm, seed=5, potential_from_gradient=True)
        h.start(x0)
        h.run(nsteps)
        q, _, _ = e.state()
        assert np.array_equal(h.accepted, q[:, 0])           # one chain, a sync per step: the reference chain
        assert h.scalars["mean_epsilon"] == e.GetMeanEpsilon() and h.scalars["leapfrog_steps"] == e.GetLeapFrog()


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("dim,nchains,sync", [(20, 70, 1), (100, 64, 2), (300, 64, 1), (500, 96, 1)])
def test_hmc_quadratic_form_adaptive(gpu, oracle, dim, nchains, sync, exact):
    """Config 5's likelihood family (quadratic form, analytic gradient) with SetLeapFrog(n) and the step length left to
    the tuning, in both arithmetic orders (matrix layout kernels)."""
    err = np.linalg.inv(_spd(dim, 3))
    e, o = _adaptive_pair(gpu, oracle, dim, nchains, 1, err, exact, sync)
    x0 = np.full(dim, 0.5)
    e.Start(x0); o.start(x0)
    e.SetLeapFrog(6); o.set_leapfrog(6)
    for k in range(3 if dim < 500 else 2):
        e.Step(5); o.step(5)
        _same_hmc(e, o, f"block {k}")
    assert np.all(e.lane("leapfrog") == -6)
    assert e.lane("naccept").sum() > 0


def test_hmc_fixed_step_can_track_the_covariance(gpu, oracle):
    dim, n = 8, 64
    e, o = _adaptive_pair(gpu, oracle, dim, n, 0, None, True, 1)
    e.TrackCovariance()
    x0 = np.zeros(dim)
    e.Start(x0); o.start(x0)
    e.SetMeanEpsilon(-0.2); o.set_mean_epsilon(-0.2)
    e.SetLeapFrog(5); o.set_leapfrog(5)
    e.Step(30); o.step(30)
    _same_hmc(e, o, "fixed step, tracked covariance")
    assert np.all(e.lane("mean_epsilon") == -0.2) and e.tuning["trace"] > 0


def test_hmc_config5_default_tuning_d500(gpu, oracle):
    """BASELINE config 5's shape (quadratic form, D = 500) with nothing fixed: automatic leapfrog count (the reversal
    test runs inside the matrix-layout kernel) and step length, both orders."""
    dim, n = 500, 64
    err = np.linalg.inv(_spd(dim, 7))
    for exact in (True, False):
        e, o = _adaptive_pair(gpu, oracle, dim, n, 1, err, exact, 1)
        x0 = np.full(dim, 0.3)
        e.Start(x0); o.start(x0)
        e.Step(4); o.step(4)
        _same_hmc(e, o, f"exact={exact}")
        assert np.all(e.lane("leapfrog") > 0)


# ---------------------------------------------------------------- Step(save, gradientType), TSimpleHMC.H:467-532
@pytest.mark.parametrize("kind,dim,nchains", [(0, 5, 70), (2, 6, 64), (1, 12, 64), (0, 100, 64), (1, 100, 64), (2, 300, 64)])
def test_hmc_covariant_gradient(gpu, oracle, kind, dim, nchains):
    """Type 2: the gradient from the pooled running covariance, grad = fEstimatedError (q - fAveragePoint) (:447-454);
    the potential stays the likelihood's."""
    prm = [100.0] if kind == 2 else (np.linalg.inv(_spd(dim, 11)) if kind == 1 else None)
    e, o = _adaptive_pair(gpu, oracle, dim, nchains, kind, prm, True, 2)
    x0 = np.full(dim, 0.8)
    e.Start(x0); o.start(x0)
    e.SetGradientType(2); o.set_gradient_type(2)
    assert e.GetGradientType() == 2
    for k in range(3):
        e.Step(5); o.step(5)
        _same_hmc(e, o, f"block {k}")
    if kind != 2:   # the identity-covariance gradient is no guide on the Rosenbrock ridge: nothing is accepted this early
        assert e.lane("naccept").sum() > 0


def test_hmc_covariant_gradient_with_a_fixed_step(gpu, oracle):
    """The covariant gradient needs the running covariance: it is kept although nothing else asks for it."""
    dim, n = 10, 64
    e, o = _adaptive_pair(gpu, oracle, dim, n, 0, None, True, 1)
    x0 = np.full(dim, 0.3)
    e.Start(x0); o.start(x0)
    e.SetMeanEpsilon(-0.1); o.set_mean_epsilon(-0.1)
    e.SetLeapFrog(6); o.set_leapfrog(6)
    e.Step(10, gradient_type=2); o.set_gradient_type(2); o.step(10)
    _same_hmc(e, o, "fixed step, covariant gradient")
    assert e.tuning["cov_trials"] > 0


@pytest.mark.parametrize("kind,dim,nchains", [(0, 5, 70), (2, 6, 64), (1, 9, 64), (0, 70, 64)])
def test_hmc_finite_difference_gradient(gpu, oracle, kind, dim, nchains):
    """Type 3: two potentials per dimension, du = 0.01 (:417-444)."""
    prm = [100.0] if kind == 2 else (np.linalg.inv(_spd(dim, 12)) if kind == 1 else None)
    e, o = _adaptive_pair(gpu, oracle, dim, nchains, kind, prm, True, 1)
    x0 = np.full(dim, 0.9)
    e.Start(x0); o.start(x0)
    e.SetGradientType(3); o.set_gradient_type(3)
    for k in range(2):
        e.Step(3); o.step(3)
        _same_hmc(e, o, f"block {k}")


@pytest.mark.parametrize("kind,dim", [(0, 5), (1, 20), (2, 100)])
def test_hmc_zero_gradient_and_switching_types(gpu, oracle, kind, dim):
    """Type 5 (free flight: the momentum never changes inside LeapFrog), then back to the likelihood's gradient (types
    1 and 4 are type 0 here: every device likelihood has a gradient), the way Step(save, gradientType) allows."""
    n = 64
    prm = [100.0] if kind == 2 else (np.linalg.inv(_spd(dim, 13)) if kind == 1 else None)
    e, o = _adaptive_pair(gpu, oracle, dim, n, kind, prm, True, 1)
    x0 = np.full(dim, 0.7)
    e.Start(x0); o.start(x0)
    for t in (5, 0, 2, 4, 5, 1):
        e.Step(3, gradient_type=t); o.set_gradient_type(t); o.step(3)
        _same_hmc(e, o, f"after type {t}")


def test_hmc_gradient_types_need_reference_order(gpu):
    e = gpu.HmcEngine(8, 64, likelihood=0, exact=False)
    with pytest.raises(gpu.SmcmcError) as err:
        e.SetGradientType(2)
    assert err.value.status == 5   # SMCMC_ERR_UNSUPPORTED
    with pytest.raises(gpu.SmcmcError):
        e.SetGradientType(6)


@pytest.mark.parametrize("kind,dim", [(0, 6), (1, 20), (1, 100)])
def test_hmc_approximate_gradient_schedule(gpu, oracle, kind, dim):
    """SimpleAHMC.C:45-92's three phases: a random walk with the momentum kept (alpha 0.8, SetLeapFrog(0), gradient type 5)
    that only feeds the running covariance, then the covariant gradient (type 2) from that covariance."""
    n = 64
    prm = np.linalg.inv(_spd(dim, 21)) if kind == 1 else None
    e, o = _adaptive_pair(gpu, oracle, dim, n, kind, prm, True, 1)
    e.TrackCovariance()
    rng = np.random.default_rng(dim)
    x0 = rng.uniform(-1.0, 1.0, size=dim)                            # SimpleAHMC.C:43
    e.Start(x0); o.start(x0)
    e.SetAlpha(0.8); o.set_alpha(0.8)                                # :52-54
    e.SetMeanEpsilon(-0.1); o.set_mean_epsilon(-0.1)
    e.SetLeapFrog(0); o.set_leapfrog(0)
    e.Step(40, gradient_type=5); o.set_gradient_type(5); o.step(40)
    _same_hmc(e, o, "first burn-in")
    assert e.tuning["updates"] == 0 and e.tuning["cov_trials"] > 0   # SetLeapFrog(0): UpdateErrorMatrix returns at once (:704)
    e.SetAlpha(0.0); o.set_alpha(0.0)                                # :68-70
    e.SetMeanEpsilon(-0.05); o.set_mean_epsilon(-0.05)
    e.SetLeapFrog(5); o.set_leapfrog(5)
    e.Step(3 * dim + 10, gradient_type=2); o.set_gradient_type(2); o.step(3 * dim + 10)
    _same_hmc(e, o, "second burn-in")
    e.SetAlpha(0.75); o.set_alpha(0.75)                              # :86-88
    e.Step(20, gradient_type=2); o.step(20)
    _same_hmc(e, o, "run")
    assert e.lane("naccept").sum() > 0


def test_hmc_more_moment_groups_than_one_reduction_chunk(gpu, oracle):
    """40 moment groups of 64 chains: the pooled UpdateCovariance adds the group sums in chunks of 32, like every moment
    reduction of the engine."""
    dim, n = 20, 64 * 40
    e, o = _adaptive_pair(gpu, oracle, dim, n, 0, None, True, 1)
    assert e.moment_group == 64
    e.Start(np.ones(dim)); o.start(np.ones(dim))
    e.Step(6); o.step(6)
    _same_hmc(e, o, "six steps")


# ---------------------------------------------------------------- the pooled covariance across engines / ranks
@pytest.mark.parametrize("kind,dim", [(0, 20), (1, 100)])
def test_hmc_sharded_exchange_equals_one_engine(gpu, oracle, kind, dim):
    """Two engines of 2048 chains with chain offsets -- moments reduced, exported, added on the device, imported and
    applied after every step -- are one engine of 4096 chains, bit for bit: the pooled UpdateCovariance /
    UpdateErrorMatrix of a sharded ensemble (what ranks do with one RCCL all-reduce between export and import)."""
    import torch
    n = 4096
    prm = np.linalg.inv(_spd(dim, 3)) if kind == 1 else None
    whole = gpu.HmcEngine(dim, n, likelihood=kind, likelihood_params=prm, seed=5)
    halves = [gpu.HmcEngine(dim, n // 2, likelihood=kind, likelihood_params=prm, seed=5, chain_offset=k * (n // 2)) for k in range(2)]
    x0 = np.full(dim, 0.5)
    whole.Start(x0)
    for h in halves:
        h.SetSyncInterval(10 ** 9)                           # the caller runs the update
        h.Start(x0)
    if kind == 1:
        for h in [whole] + halves:
            h.SetLeapFrog(6)
    for step in range(12 if dim <= 20 else 6):
        whole.Step(1)
        bufs = []
        for h in halves:
            h.Step(1)
            h.reduce_moments()
            b = torch.zeros(h.moments_size, dtype=torch.float64, device="cuda")
            h.export_moments(b.data_ptr())
            bufs.append(b)
        total = bufs[0] + bufs[1]
        for h in halves:
            h.import_moments(total.data_ptr())
            h.apply_moments()
        torch.cuda.synchronize()
        q, m, logl = whole.state()
        for k, h in enumerate(halves):
            sl = slice(k * (n // 2), (k + 1) * (n // 2))
            hq, hm, hl = h.state()
            assert np.array_equal(hq, q[:, sl]) and np.array_equal(hm, m[:, sl]) and np.array_equal(hl, logl[sl]), f"step {step}"
            assert np.array_equal(h.lane("mean_epsilon"), whole.lane("mean_epsilon")[sl]), f"step {step}: epsilon"
            assert np.array_equal(h.lane("leapfrog"), whole.lane("leapfrog")[sl]), f"step {step}: leapfrog"
            assert np.array_equal(h.covariance, whole.covariance) and np.array_equal(h.average, whole.average)
            assert h.tuning == whole.tuning, f"step {step}: {h.tuning} != {whole.tuning}"
    assert whole.tuning["updates"] >= 1 and whole.lane("naccept").sum() > 0


# ---------------------------------------------------------------- likelihoods without a gradient as HMC targets
@pytest.mark.parametrize("kind,dim,gtype", [(4, 8, 3), (4, 70, 3), (6, 25, 3), (5, 12, 3), (4, 20, 5), (6, 25, 2)])
def test_hmc_targets_without_a_gradient(gpu, oracle, kind, dim, gtype):
    """The reference runs any operator()(Vector) under HMC through FiniteDifferenceGradient (TSimpleHMC.H:417-444), the
    covariant gradient (:447-454) or none (:524-528); TAsymLogLikelihood, THorrificLogLikelihood and example4's
    TConstrainedLikelihood have no gradient of their own (their functors throw / return false)."""
    n = 64
    prm = oracle.like_params(kind, dim)
    e, o = _adaptive_pair(gpu, oracle, dim, n, kind, prm, True, 1)
    x0 = {4: np.full(dim, 0.5), 5: np.full(dim, 0.01), 6: np.full(dim, 76.0)}[kind]
    e.Start(x0); o.start(x0)
    e.SetGradientType(gtype); o.set_gradient_type(gtype)
    for k in range(2):
        e.Step(3); o.step(3)
        _same_hmc(e, o, f"block {k}")
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1, gradient_type=0)                               # the functor has no gradient: the reference throws
    assert err.value.status == 3


def test_hmc_backend_exchange_and_the_pending_guard(gpu):
    """distributed.HmcBackend (what run_windows drives for a sharded config 5): two engines with chain offsets exchanging
    their window moments through the backend are one engine syncing at the same interval; and a second reduction before
    the apply of the first is refused (SMCMC_ERR_LOGIC) instead of silently dropping a window's moments."""
    import torch
    from root_simple_mcmc_amd import distributed as D
    dim, n, window = 24, 4096, 3                    # halves of 2 048 chains: whole chunks of 32 moment groups, so the shards'
    whole = gpu.HmcEngine(dim, n, seed=9)           # sums add up in the single engine's order
    whole.SetSyncInterval(window)
    halves = [gpu.HmcEngine(dim, n // 2, seed=9, chain_offset=k * (n // 2)) for k in range(2)]
    x0 = np.full(dim, 0.3)
    whole.Start(x0)
    backs = []
    for h in halves:
        backs.append(D.HmcBackend(h))
        h.Start(x0)
    for _ in range(5):
        whole.Step(window)
        outs = []
        for b in backs:
            b.step(window)
            outs.append(b.moments_out().clone())
        total = outs[0] + outs[1]                       # the all-reduce
        for b in backs:
            b.moments_in(total)
        torch.cuda.synchronize()
        q, m, _ = whole.state()
        for k, h in enumerate(halves):
            sl = slice(k * (n // 2), (k + 1) * (n // 2))
            hq, hm, _ = h.state()
            assert np.array_equal(hq, q[:, sl]) and np.array_equal(hm, m[:, sl])
            assert np.array_equal(h.covariance, whole.covariance) and h.tuning == whole.tuning
    h = halves[0]
    h.Step(1)
    h.reduce_moments()
    h.Step(1)
    with pytest.raises(gpu.SmcmcError) as err:
        h.reduce_moments()
    assert err.value.status == 2                        # SMCMC_ERR_LOGIC
    h.apply_moments()
    h.reduce_moments(); h.apply_moments()
