"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Bars: bit-exact for every integer field and -- because oracle and kernels share
the deterministic math of include/smcmc_detmath.h and the same operation order --
bit-exact for every double as well (the north-star tolerance is "identical
accept/reject sequence, log-likelihood within 1 ulp"; we hold 0 ulp).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIKES = {"iso": 0, "quadform": 1, "rosenbrock": 2}


def _like_params(oracle, kind, dim):
    prm = oracle.like_params(kind, dim)
    return prm if prm.size else None


def _start(kind, dim, nchains, rng):
    if kind == 2:   # SimpleMCMC.C:147: start near the Rosenbrock mode
        return rng.uniform(0.5, 1.5, size=(dim, nchains))
    return np.zeros(dim)


def _pair(gpu, oracle, dim, nchains, kind, mode, exact, seed=20240607, offset=0, setup=None):
    prm = _like_params(oracle, kind, dim)
    e = gpu.Engine(dim, nchains, likelihood=kind, likelihood_params=prm, seed=seed, chain_offset=offset,
                   mode=mode, exact=exact)
    o = oracle.Ensemble(nchains, dim, kind=kind, params=prm, seed=seed, chain_offset=offset, mode=mode,
                        exact=exact)
    if setup:
        setup(e, o)
    return e, o


def _assert_same_state(e, o, tag=""):
    assert np.array_equal(e.GetAccepted(), o.x), f"{tag}: accepted points differ"
    for name in ("logl", "sigma", "acceptance", "acceptance_trials", "rigidity", "step_rms", "logl_proposed"):
        a, b = e.lane(name), o.lane(name)
        assert np.array_equal(a, b), f"{tag}: lane field {name} differs (max |d| = {np.max(np.abs(a - b))})"
    for name in ("trials", "successes", "next_update", "naccept", "step_rms_trials"):
        assert np.array_equal(e.lane(name), o.lane(name)), f"{tag}: lane field {name} differs"
    assert np.array_equal(e.lane("last_accept").astype(np.uint8), o.lane("last_accept")), f"{tag}: accept bits"


# ---------------------------------------------------------------- hardware facts
@pytest.mark.parametrize("kind,name", [(0, "log"), (1, "exp"), (2, "sin2pi"), (3, "cos2pi"), (5, "sqrt")])
def test_detmath_device_equals_host_bitwise(gpu, oracle, kind, name):
    rng = np.random.default_rng(kind)
    if kind == 0:
        x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)),
                            (rng.integers(0, 2 ** 32, 200000) + 0.5) * 2.0 ** -32, rng.uniform(0.5, 2, 100000)])
        ref = oracle.det_log(x)
    elif kind == 1:
        x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-0.02, 0.02, 200000)])
        ref = oracle.det_exp(x)
    elif kind in (2, 3):
        x = (rng.integers(0, 2 ** 32, 400000) + 0.5) * 2.0 ** -32
        ref = oracle.det_sincos2pi(x)[kind - 2]
    else:
        x = np.exp(rng.uniform(-600, 600, 400000))
        ref = np.sqrt(x)
    out = gpu.selftest_detmath(kind, x)
    assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), name


def test_detmath_pow_and_div_bitwise(gpu, oracle):
    rng = np.random.default_rng(9)
    x = rng.uniform(1e-3, 4.3, 300000)
    y = rng.uniform(1e-7, 2e-3, 300000)
    assert np.array_equal(gpu.selftest_detmath(4, x, y), oracle.det_pow_small(x, y))
    a = np.exp(rng.uniform(-300, 300, 300000)) * rng.choice([-1.0, 1.0], 300000)
    b = np.exp(rng.uniform(-300, 300, 300000))
    assert np.array_equal(gpu.selftest_detmath(6, a, b), a / b)


def test_normal_pair_device_equals_host_bitwise(gpu, oracle):
    """Box-Muller from two 32-bit words, including the device-only rsq + Newton square root."""
    rng = np.random.default_rng(17)
    w0 = rng.integers(0, 2 ** 32, size=400000, dtype=np.uint64).astype(np.float64)
    w1 = rng.integers(0, 2 ** 32, size=400000, dtype=np.uint64).astype(np.float64)
    w0[:4] = [0, 1, 2 ** 32 - 1, 2 ** 31]
    w1[:4] = [2 ** 32 - 1, 2 ** 29, 3 * 2 ** 29 - 1, 0]
    n0, n1 = oracle.det_normal_pair(w0, w1)
    assert np.array_equal(gpu.selftest_detmath(7, w0, w1), n0)
    assert np.array_equal(gpu.selftest_detmath(8, w0, w1), n1)
    x = np.concatenate([rng.uniform(2e-10, 45.0, 300000), 10.0 ** rng.uniform(-9.6, 1.6, 300000)])
    assert np.array_equal(gpu.selftest_detmath(9, x), np.sqrt(x))


def test_mfma_f64_is_an_ascending_k_fma_chain(gpu):
    """v_mfma_f64_16x16x4_f64 chained over K folds k = 0..K-1 in order, one fused
    multiply-add per product -- the order oracle/ensemble_oracle.c defines the
    pooled moments in.  Asymmetric operands catch a transposed C layout."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.fma.restype = ctypes.c_double
    libm.fma.argtypes = [ctypes.c_double] * 3
    rng = np.random.default_rng(3)
    K = 64
    a = rng.standard_normal((16, K)) * np.exp(rng.uniform(-8, 8, (16, K)))
    b = rng.standard_normal((K, 16)) * np.exp(rng.uniform(-8, 8, (K, 16)))
    c = gpu.selftest_mfma(a, b)
    ref = np.zeros((16, 16))
    for i in range(16):
        for j in range(16):
            s = 0.0
            for k in range(K):
                s = libm.fma(a[i, k], b[k, j], s)
            ref[i, j] = s
    assert np.array_equal(c, ref)


def test_mfma_f64_4x4x4_is_an_ascending_k_fma_chain_too(gpu):
    """The 4x4x4 four-block form that folds the last rows of the D = 50 moments: same order."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.fma.restype = ctypes.c_double
    libm.fma.argtypes = [ctypes.c_double] * 3
    rng = np.random.default_rng(4)
    K = 64
    a = rng.standard_normal((4, K)) * np.exp(rng.uniform(-8, 8, (4, K)))
    b = rng.standard_normal((K, 16)) * np.exp(rng.uniform(-8, 8, (K, 16)))
    c = gpu.selftest_mfma_strip(a, b)
    ref = np.zeros((4, 16))
    for i in range(4):
        for j in range(16):
            s = 0.0
            for k in range(K):
                s = libm.fma(a[i, k], b[k, j], s)
            ref[i, j] = s
    assert np.array_equal(c, ref)


# ---------------------------------------------------------------- frozen mode
@pytest.mark.parametrize("dim,nchains,steps", [(2, 1, 300), (5, 70, 400), (7, 64, 200), (8, 130, 150),
                                                (20, 65, 120), (50, 128, 60), (63, 64, 30)])
@pytest.mark.parametrize("exact", [True, False])
def test_frozen_iso_matches_oracle(gpu, oracle, dim, nchains, steps, exact):
    e, o = _pair(gpu, oracle, dim, nchains, 0, gpu.MODE_FROZEN, exact)
    assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
    _assert_same_state(e, o, "after start")
    for chunk in (1, 1, steps // 3, steps - steps // 3 - 2):
        e.Step(chunk)
        o.step(chunk)
        _assert_same_state(e, o, f"after +{chunk} steps")


def test_frozen_chain_is_a_reference_chain(gpu, oracle):
    """Every lane of the FROZEN engine is bit for bit the single-chain restatement of
    TSimpleMCMC<L,TProposeAdaptiveStep> with SetCovarianceFrozen(true), including the
    per-chain UpdateProposal schedule (short acceptance window forces many updates)."""
    dim, nchains, steps = 5, 96, 3000
    e = gpu.Engine(dim, nchains, mode=gpu.MODE_FROZEN)
    e.SetAcceptanceWindow(50)
    assert e.Start(np.zeros(dim))
    e.Step(steps)
    x = e.GetAccepted()
    for ch in (0, 1, 63, 64, 95):
        c = oracle.Chain(dim, chain_id=ch)
        c.set_covariance_frozen(1)
        c.set_acceptance_window(50.0)
        assert c.start(np.zeros(dim))
        c.run_quiet(steps)
        sc = c.scalars
        assert np.array_equal(c.accepted, x[:, ch])
        assert sc["sigma"] == e.lane("sigma")[ch]
        assert sc["acceptance"] == e.lane("acceptance")[ch]
        assert sc["acceptance_trials"] == e.lane("acceptance_trials")[ch]
        assert sc["next_update"] == e.lane("next_update")[ch]
        assert sc["trials"] == e.lane("trials")[ch] and sc["successes"] == e.lane("successes")[ch]
        assert sc["step_rms"] == e.lane("step_rms")[ch]
        assert sc["update_count"] > 3


@pytest.mark.parametrize("kind,dim,nchains,steps", [(1, 5, 70, 300), (1, 20, 64, 80), (1, 50, 64, 25),
                                                     (2, 6, 70, 300), (2, 31, 65, 100), (2, 50, 64, 40)])
@pytest.mark.parametrize("exact", [True, False])
def test_frozen_other_likelihoods(gpu, oracle, kind, dim, nchains, steps, exact):
    rng = np.random.default_rng(dim)
    e, o = _pair(gpu, oracle, dim, nchains, kind, gpu.MODE_FROZEN, exact)
    x0 = _start(kind, dim, nchains, rng)
    assert e.Start(x0) and o.start(x0)
    e.Step(steps)
    o.step(steps)
    _assert_same_state(e, o, f"kind {kind}")


def test_metropolis_modes_and_offsets(gpu, oracle):
    for metropolis in (1, 2):
        e, o = _pair(gpu, oracle, 5, 66, 0, gpu.MODE_FROZEN, True, seed=7, offset=640)
        e.Start(np.full(5, 0.3)); o.start(np.full(5, 0.3))
        e.Step(50, metropolis); o.step(50, metropolis)
        _assert_same_state(e, o, f"metropolis {metropolis}")


def test_sharding_is_invisible(gpu):
    """Chains are keyed on their global id: two half-ensembles with chain_offset give
    the chains of the whole one (FROZEN mode, no exchange step)."""
    dim, n, steps = 10, 256, 100
    whole = gpu.Engine(dim, n, mode=gpu.MODE_FROZEN); whole.Start(np.zeros(dim)); whole.Step(steps)
    lo = gpu.Engine(dim, n // 2, mode=gpu.MODE_FROZEN); lo.Start(np.zeros(dim)); lo.Step(steps)
    hi = gpu.Engine(dim, n // 2, chain_offset=n // 2, mode=gpu.MODE_FROZEN); hi.Start(np.zeros(dim)); hi.Step(steps)
    x = whole.GetAccepted()
    assert np.array_equal(x[:, : n // 2], lo.GetAccepted())
    assert np.array_equal(x[:, n // 2:], hi.GetAccepted())


# ---------------------------------------------------------------- pooled mode
@pytest.mark.parametrize("dim,nchains,window,nwin", [(5, 70, 16, 6), (7, 64, 8, 4), (20, 130, 10, 4),
                                                      (50, 192, 6, 3), (63, 64, 4, 2)])
@pytest.mark.parametrize("exact", [True, False])
def test_pooled_iso_matches_oracle(gpu, oracle, dim, nchains, window, nwin, exact):
    e, o = _pair(gpu, oracle, dim, nchains, 0, gpu.MODE_POOLED, exact)
    assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
    for w in range(nwin):
        e.Step(window)
        o.step(window)
        _assert_same_state(e, o, f"window {w}")
        e.reduce_moments()
        m_gpu = e.read_moments()
        m_cpu = o.reduce_moments()
        assert np.array_equal(m_gpu, m_cpu), f"window {w}: moments differ, max rel {np.max(np.abs(m_gpu - m_cpu))}"
        assert m_gpu[-1] == nchains * window
        e.apply_moments()
        o.apply_moments(m_cpu)
        assert np.array_equal(e.covariance, o.covariance)
        assert np.array_equal(e.GetEstimatedCenter(), o.center)
        assert np.array_equal(e.decomposition, o.decomposition)
        # UpdateProposal's sigma rescale and acceptance de-weighting (TSimpleMCMC.H:1042-1043, 1081-1086) are in the
        # chains' columns as soon as the update returns: what SaveStep would write is what the reference holds
        _assert_same_state(e, o, f"window {w}, after the update")
        assert e.saved_state(chain=1)["sigma"] == o.lane("sigma")[1] == e.GetSigma(chain=1)
    e.Step(3); o.step(3)
    _assert_same_state(e, o, "after the last sync")


def test_set_sigma_after_an_update_is_not_rescaled(gpu):
    """SetSigma after UpdateProposal replaces the rescaled value (TSimpleMCMC.H:775); two updates in a row rescale twice."""
    e = gpu.Engine(6, 70)
    assert e.Start(np.zeros(6))
    e.Step(40); e.sync()
    e.SetSigma(0.125)
    assert np.all(e.lane("sigma") == 0.125)
    e.Step(1)
    assert np.all(np.abs(e.lane("sigma") / 0.125 - 1.0) < 0.01)        # one step of the 1/500 power law, no stale factor
    e.Step(30); e.reduce_moments(); e.apply_moments()
    before = e.lane("sigma").copy()
    cov = e.covariance
    e.SetCovariance(cov * 4.0)
    e.UpdateProposal()                                                   # trace x 4: sigma halves
    assert np.array_equal(e.lane("sigma"), before * 0.5)
    e.SetCovariance(cov)
    e.UpdateProposal()
    assert np.array_equal(e.lane("sigma"), before * 0.5 * 2.0)


def test_pooled_split_launches_equal_one_launch(gpu):
    """The moment accumulators persist across launches: 12 x Step(1) == Step(12)."""
    a = gpu.Engine(9, 100); a.Start(np.zeros(9)); a.Step(12); a.reduce_moments()
    b = gpu.Engine(9, 100); b.Start(np.zeros(9))
    for _ in range(12):
        b.Step(1)
    b.reduce_moments()
    assert np.array_equal(a.read_moments(), b.read_moments())
    assert np.array_equal(a.GetAccepted(), b.GetAccepted())


def test_pooled_rosenbrock_and_quadform(gpu, oracle):
    for kind, dim in ((2, 6), (1, 5)):
        rng = np.random.default_rng(kind)
        e, o = _pair(gpu, oracle, dim, 128, kind, gpu.MODE_POOLED, True)
        x0 = _start(kind, dim, 128, rng)
        assert e.Start(x0) and o.start(x0)
        for w in range(4):
            e.Step(20); o.step(20)
            e.sync(); o.sync()
        e.Step(5); o.step(5)
        _assert_same_state(e, o, f"kind {kind}")
        assert np.array_equal(e.decomposition, o.decomposition)


# ---------------------------------------------------------------- boundary behaviour
def test_bad_start_is_reported(gpu):
    e = gpu.Engine(4, 10, likelihood=2, likelihood_params=[100.0])
    assert e.Start(np.full(4, 1e6)) is False       # logL < -0.999999E+10 (TSimpleMCMC.H:265-268)
    assert e.Start(np.ones(4)) is True


def test_step_before_start_raises(gpu):
    e = gpu.Engine(4, 10)
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1)
    assert "Uninitialized starting point" in str(err.value)   # TSimpleMCMC.H:371-374


def test_force_step(gpu):
    e = gpu.Engine(5, 70, mode=gpu.MODE_FROZEN)
    e.Start(np.zeros(5))
    e.Step(10)
    trials = e.lane("trials").copy()
    target = np.full(5, 0.25)
    e.ForceStep(target)
    e.Step(1, 2)                                   # Step(false, 2): accept everything
    assert np.array_equal(e.GetAccepted(), np.repeat(target[:, None], 70, axis=1))
    assert np.array_equal(e.lane("trials"), trials)            # proposal state untouched (:671-678)
    assert np.allclose(e.GetAcceptedLogLikelihood(), -0.5 * 5 * 0.25 ** 2)


@pytest.mark.parametrize("exact", [True, False])
def test_force_step_at_large_dim(gpu, oracle, exact):
    """ForceStep for D > 63 (both large-dimension kernels): the forced point is proposed, tested and, with
    Step(false, 2), taken; the proposal state does not move; the chain then continues like the reference chain
    that was forced the same way."""
    dim, n = 100, 70
    e = gpu.Engine(dim, n, mode=gpu.MODE_FROZEN, exact=exact)
    assert e.Start(np.zeros(dim))
    e.Step(5)
    trials = e.lane("trials").copy()
    target = np.linspace(-0.2, 0.3, dim)
    e.ForceStep(target)
    e.Step(1, 2)
    assert np.array_equal(e.GetAccepted(), np.repeat(target[:, None], n, axis=1))
    assert np.array_equal(e.lane("trials"), trials)
    e.ForceStep(np.full(dim, 10.0))            # far out: tested with the Metropolis rule, rejected everywhere
    e.Step(1)
    assert np.array_equal(e.GetAccepted(), np.repeat(target[:, None], n, axis=1))
    e.Step(6)
    if exact:
        for ch in (0, 69):
            c = oracle.Chain(dim, chain_id=ch)
            c.set_covariance_frozen(1)
            assert c.start(np.zeros(dim))
            c.run_quiet(5)
            c.force_step(target); c.step(False, 2)
            c.force_step(np.full(dim, 10.0)); c.step(False, 0)
            c.run_quiet(6)
            assert np.array_equal(c.accepted, e.GetAccepted()[:, ch])
            assert c.scalars["sigma"] == e.lane("sigma")[ch] and c.scalars["trials"] == e.lane("trials")[ch]


def test_save_buffer(gpu):
    import torch
    dim, n, steps, stride = 5, 70, 20, 4
    e = gpu.Engine(dim, n, mode=gpu.MODE_FROZEN)
    e.Start(np.zeros(dim))
    npad, dpad = e.nchains_padded, e.dim_padded
    sx = torch.zeros((steps // stride, dpad, npad), dtype=torch.float64, device="cuda")
    sl = torch.zeros((steps // stride, npad), dtype=torch.float64, device="cuda")
    ref = gpu.Engine(dim, n, mode=gpu.MODE_FROZEN)
    ref.Start(np.zeros(dim))
    e.StepSave(steps, sx.data_ptr(), sl.data_ptr(), stride=stride)
    torch.cuda.synchronize()
    for slot in range(steps // stride):
        ref.Step(stride)
        assert np.array_equal(sx[slot, :dim, :n].cpu().numpy(), ref.GetAccepted())
        assert np.array_equal(sl[slot, :n].cpu().numpy(), ref.GetAcceptedLogLikelihood())


# ---------------------------------------------------------------- full-size properties
def test_full_size_d50_properties(gpu):
    """BASELINE config 2 shape (D=50, 65 536 chains): size-independent properties."""
    dim, n = 50, 65536
    a = gpu.Engine(dim, n, seed=11); a.Start(np.zeros(dim))
    b = gpu.Engine(dim, n, seed=11); b.Start(np.zeros(dim))
    for _ in range(3):
        a.Step(64); b.Step(64)
        a.reduce_moments(); b.reduce_moments()
        ma, mb = a.read_moments(), b.read_moments()
        assert np.array_equal(ma, mb)                              # run-to-run deterministic
        assert ma[-1] == n * 64                                    # every chain-step was folded in
        a.apply_moments(); b.apply_moments()
    assert np.array_equal(a.GetAccepted(), b.GetAccepted())
    x = a.GetAccepted()
    logl = a.GetAcceptedLogLikelihood()
    assert np.allclose(logl, -0.5 * np.sum(x * x, axis=0), rtol=1e-12)   # state and logL agree
    acc = a.lane("naccept").sum() / (n * 192)
    assert 0.1 < acc < 0.8          # early in the adaptation: sigma still rising toward the 0.234 target
    # the pooled covariance moves toward the target's (identity): trace ~ D
    assert abs(np.trace(a.covariance) / dim - 1.0) < 0.25


# ---------------------------------------------------------------- large dimensions (panel kernel)
@pytest.mark.parametrize("kind,dim,nchains,steps", [(0, 64, 70, 40), (0, 200, 130, 25), (2, 200, 64, 30),
                                                     (0, 257, 64, 12), (0, 500, 96, 10), (2, 500, 64, 10),
                                                     # the edges of the kernels' tilings: 128 | 129, 256 | 257, 512 = max
                                                     (0, 128, 64, 8), (2, 129, 64, 8), (2, 256, 70, 8), (0, 512, 64, 6)])
@pytest.mark.parametrize("exact", [True, False])
def test_frozen_large_dim_matches_oracle(gpu, oracle, kind, dim, nchains, steps, exact):
    """BASELINE configs 3/4 shapes (D=200 Rosenbrock, D=500): a workgroup of 4 or 8
    wavefronts per 64-chain group, still bit for bit the oracle."""
    rng = np.random.default_rng(dim + kind)
    e, o = _pair(gpu, oracle, dim, nchains, kind, gpu.MODE_FROZEN, exact)
    x0 = _start(kind, dim, nchains, rng)
    assert e.Start(x0) and o.start(x0)
    _assert_same_state(e, o, "after start")
    e.Step(1); o.step(1)
    _assert_same_state(e, o, "after 1 step")
    e.Step(steps - 1); o.step(steps - 1)
    _assert_same_state(e, o, f"after {steps} steps")


@pytest.mark.parametrize("dim,nchains,stride,window,nwin", [(100, 256, 1, 6, 3), (200, 192, 4, 8, 2), (300, 640, 3, 6, 2)])
@pytest.mark.parametrize("exact", [True, False])
def test_pooled_large_dim_matches_oracle(gpu, oracle, dim, nchains, stride, window, nwin, exact):
    """Pooled covariance for D > 63: the moment fold runs as its own kernel (one wavefront
    per 16x16 tile and chain slice) every `stride`-th step; still bit for bit the oracle.  In the
    fused order the proposal runs on the matrix pipe (smcmc_panel_mfma_kernel.hip.h)."""
    e, o = _pair(gpu, oracle, dim, nchains, 0, gpu.MODE_POOLED, exact)
    e.set_param("MOMENT_STRIDE", stride)
    o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), stride)
    assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
    for w in range(nwin):
        e.Step(window); o.step(window)
        _assert_same_state(e, o, f"window {w}")
        e.reduce_moments()
        m_gpu, m_cpu = e.read_moments(), o.reduce_moments()
        assert np.array_equal(m_gpu, m_cpu), f"window {w}: moments differ"
        e.apply_moments(); o.apply_moments(m_cpu)
        assert np.array_equal(e.decomposition, o.decomposition)
    e.Step(2); o.step(2)
    _assert_same_state(e, o, "after the last sync")


@pytest.mark.parametrize("mode", ["frozen", "pooled"])
@pytest.mark.parametrize("dim,nchains", [(64, 70), (100, 96), (300, 64)])
def test_quadratic_form_at_large_dim_in_the_fused_order(gpu, oracle, mode, dim, nchains):
    """TDummyLogLikelihood (header form, TDummyLogLikelihood.H:21-31) for D > 63: the row sums of Error p on
    the matrix pipe, the outer sum in dimension order (the oracle's quadform_rowwise association)."""
    m = gpu.MODE_FROZEN if mode == "frozen" else gpu.MODE_POOLED
    e, o = _pair(gpu, oracle, dim, nchains, 1, m, False)
    o.set_quadform_rowwise(1)
    if mode == "pooled":
        e.set_param("MOMENT_STRIDE", 2)
        o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), 2)
    x0 = np.full(dim, 0.05)
    assert e.Start(x0) and o.start(x0)
    _assert_same_state(e, o, "after start")
    for w in range(2):
        e.Step(6); o.step(6)
        _assert_same_state(e, o, f"{mode} window {w}")
        if mode == "pooled":
            e.sync(); o.sync()
            assert np.array_equal(e.decomposition, o.decomposition)
    assert e.lane("naccept").sum() > 0


def test_dimension_limit_is_reported(gpu):
    assert gpu.load().smcmc_max_dim() == 512
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.Engine(513, 64)
    assert err.value.status == 5   # SMCMC_ERR_UNSUPPORTED, no host fallback


@pytest.mark.parametrize("mode", ["frozen", "pooled"])
@pytest.mark.parametrize("dim,nchains,steps", [(64, 70, 10), (100, 96, 8), (300, 64, 5), (500, 64, 4)])
def test_quadratic_form_at_large_dim_in_the_reference_order(gpu, oracle, mode, dim, nchains, steps):
    """TDummyLogLikelihood.H:21-31 for D > 63 exactly as the reference sums it -- logL -= 0.5*p[i]*Error(j,i)*p[j], i
    outer, j inner, one running sum of D^2 terms per chain (a lane walks it) -- with Error from Init() (the pair
    (0, D-1) correlated by 0.999999: cancellations at the 5E+5 scale make the order visible in the last bits)."""
    m = gpu.MODE_FROZEN if mode == "frozen" else gpu.MODE_POOLED
    e, o = _pair(gpu, oracle, dim, nchains, 1, m, True)
    if mode == "pooled":
        e.set_param("MOMENT_STRIDE", 2)
        o.set_moment_grouping(int(e.get_param("MOMENT_GROUP")), 2)
    x0 = np.full(dim, 0.05)
    assert e.Start(x0) and o.start(x0)
    _assert_same_state(e, o, "after start")
    for w in range(2):
        e.Step(steps); o.step(steps)
        _assert_same_state(e, o, f"{mode} window {w}")
        if mode == "pooled":
            e.sync(); o.sync()
            assert np.array_equal(e.decomposition, o.decomposition)
    assert e.lane("naccept").sum() > 0


def test_quadratic_form_orders_agree_to_rounding(gpu, oracle):
    """Reference order (serial sum) and fused order (matrix pipe, row-wise association) of the same likelihood."""
    dim, n = 200, 64
    prm = oracle.like_params(1, dim)
    out = []
    for exact in (True, False):
        e = gpu.Engine(dim, n, likelihood=1, likelihood_params=prm, mode=gpu.MODE_FROZEN, exact=exact)
        assert e.Start(np.full(dim, 0.05))
        out.append(e.GetAcceptedLogLikelihood())
    assert np.allclose(out[0], out[1], rtol=1e-9) and np.all(out[0] < 0)


# ---------------------------------------------------------------- uniform dimensions and scan
@pytest.mark.parametrize("mode", ["frozen", "pooled"])
@pytest.mark.parametrize("kind,dim,nchains", [(0, 5, 70), (2, 6, 64), (0, 50, 128), (0, 70, 130), (2, 300, 64)])
def test_uniform_dimensions_match_oracle(gpu, oracle, mode, kind, dim, nchains):
    """SetUniform (TSimpleMCMC.H:833-848, 711-716, 721): the uniform dimensions are redrawn
    from their range every step and take no part in the Gaussian move."""
    m = gpu.MODE_FROZEN if mode == "frozen" else gpu.MODE_POOLED

    def setup(e, o):
        for d, lo, hi in ((1, -0.5, 0.75), (dim - 1, 0.25, 1.5)):
            e.SetUniform(d, lo, hi)
            o.set_uniform(d, lo, hi)
        e.SetGaussian(0, 0.5)
        o.set_gaussian(0, 0.5)

    rng = np.random.default_rng(3)
    e, o = _pair(gpu, oracle, dim, nchains, kind, m, True, setup=setup)
    x0 = _start(kind, dim, nchains, rng)
    assert e.Start(x0) and o.start(x0)
    for w in range(3):
        e.Step(25); o.step(25)
        _assert_same_state(e, o, f"{mode} window {w}")
        if mode == "pooled":
            e.sync(); o.sync()
            assert np.array_equal(e.decomposition, o.decomposition)
    x = e.GetAccepted()[:, e.lane("naccept") > 0]          # chains that have moved off their start
    assert x.shape[1] > nchains // 2
    assert np.all((x[1] >= -0.5) & (x[1] <= 0.75)) and np.all((x[dim - 1] >= 0.25) & (x[dim - 1] <= 1.5))
    assert len(np.unique(x[1])) > x.shape[1] // 2


@pytest.mark.parametrize("dim", [5, 100])
def test_uniform_needs_reference_order(gpu, dim):
    e = gpu.Engine(dim, 10, exact=False)
    e.SetUniform(2, -1.0, 1.0)
    e.Start(np.zeros(dim))
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1)
    assert err.value.status == 5   # SMCMC_ERR_UNSUPPORTED


@pytest.mark.parametrize("dim,sd", [(6, 3), (90, 77)])
@pytest.mark.parametrize("uniform", [False, True])
def test_scan_dimension_matches_reference_chain(gpu, oracle, uniform, dim, sd):
    """SetScanDimension (TSimpleMCMC.H:685-704, 820-830): only that dimension is redrawn (about
    the estimated centre, or uniformly) and the proposal state stays as it was."""
    nchains, steps = 70, 40
    e = gpu.Engine(dim, nchains, mode=gpu.MODE_FROZEN)
    if uniform:
        e.SetUniform(sd, -2.0, 1.0)
    else:
        e.SetGaussian(sd, 0.7)
    start = np.linspace(-0.3, 0.4, dim)
    assert e.Start(start)
    trials = e.lane("trials").copy()
    e.SetScanDimension(sd)
    e.Step(steps)
    x, logl = e.GetAccepted(), e.GetAcceptedLogLikelihood()
    assert np.array_equal(e.lane("trials"), trials)
    other = [d for d in range(dim) if d != sd]
    assert np.array_equal(x[other], np.repeat(start[other, None], nchains, axis=1))
    for ch in (0, 1, 33, 69):
        c = oracle.Chain(dim, chain_id=ch)
        if uniform:
            c.set_uniform(sd, -2.0, 1.0)
        else:
            c.set_gaussian(sd, 0.7)
        assert c.start(start)
        c.set_scan_dimension(sd)
        c.run_quiet(steps)
        assert np.array_equal(c.accepted, x[:, ch])
        assert c.scalars["accepted_logl"] == logl[ch]
    e.SetScanDimension(-1)
    e.Step(5)
    assert np.array_equal(e.lane("trials"), trials + 5)


# ---------------------------------------------------------------- Restore
@pytest.mark.parametrize("kind,dim", [(0, 5), (2, 6), (0, 50)])
def test_restore_continues_like_the_reference_chain(gpu, oracle, kind, dim):
    """Restore (TSimpleMCMC.H:282-352, 1501-1612): a chain saved by the CPU restatement is picked
    up by the engine; every lane then equals the reference chain restored from the same entry."""
    prm = _like_params(oracle, kind, dim)
    src = oracle.Chain(dim, kind=kind, params=prm, chain_id=2)
    x0 = np.full(dim, 0.9) if kind == 2 else np.zeros(dim)
    assert src.start(x0)
    src.run_quiet(900)
    st = src.saved_state()

    nchains, steps = 70, 300
    e = gpu.Engine(dim, nchains, likelihood=kind, likelihood_params=prm, mode=gpu.MODE_FROZEN)
    assert e.Start(x0)
    e.Restore(st)
    assert e.get_param("TOTAL_STEPS") == 900
    assert np.array_equal(e.GetAccepted(), np.repeat(st["accepted"][:, None], nchains, axis=1))
    e.Step(steps)
    x = e.GetAccepted()
    for ch in (0, 2, 64, 69):
        c = oracle.Chain(dim, kind=kind, params=prm, chain_id=ch)
        c.set_covariance_frozen(1)
        assert c.start(x0)
        c.restore(st)
        c.run_quiet(steps)
        sc = c.scalars
        assert np.array_equal(c.accepted, x[:, ch])
        assert sc["accepted_logl"] == e.GetAcceptedLogLikelihood()[ch]
        for name in ("sigma", "acceptance", "acceptance_trials", "step_rms"):
            assert sc[name] == e.lane(name)[ch], name
        for name in ("trials", "successes", "next_update"):
            assert sc[name] == e.lane(name)[ch], name
    assert np.array_equal(e.decomposition, c.decomposition)


def test_restore_round_trip_of_the_pooled_engine(gpu):
    """saved_state() -> Restore() carries the ensemble's shared centre / covariance and every
    chain's point into a fresh engine."""
    dim, n = 9, 200
    a = gpu.Engine(dim, n)
    a.Start(np.zeros(dim))
    for _ in range(4):
        a.Step(50); a.sync()
    st = a.saved_state(chain=5)
    b = gpu.Engine(dim, n)
    b.Start(np.zeros(dim))
    b.Restore(st, accepted=a.GetAccepted())
    assert np.array_equal(b.GetAccepted(), a.GetAccepted())
    assert np.array_equal(b.GetAcceptedLogLikelihood(), a.GetAcceptedLogLikelihood())
    assert np.array_equal(b.covariance, a.covariance) and np.array_equal(b.GetEstimatedCenter(), a.GetEstimatedCenter())
    assert b.get_param("TOTAL_STEPS") == 200 and np.all(b.lane("trials") == st["trials"])
    # what is saved is what the reference would hold after the update: sigma already rescaled by sqrt(trace0 / trace)
    # (TSimpleMCMC.H:1042-1043), next to the trace it was rescaled to -- and that is what a restored chain starts from
    assert st["sigma"] == a.lane("sigma")[5] and st["acceptance_trials"] == a.lane("acceptance_trials")[5]
    assert a.get_param("SIGMA_TRACE") == a.get_param("COVARIANCE_TRACE")
    assert np.all(b.lane("sigma") == st["sigma"]) and b.get_param("SIGMA_TRACE") == a.get_param("SIGMA_TRACE")
    b.Step(20); b.sync()
    assert np.isfinite(b.GetAcceptedLogLikelihood()).all()


def test_restore_needs_start(gpu):
    e = gpu.Engine(4, 8)
    st = dict(accepted=np.zeros(4), log_likelihood=0.0, total_steps=10, step_rms=0.1, trials=10, successes=3,
              next_update=5, acceptance=0.3, acceptance_trials=10.0, sigma=0.5, central_point=np.zeros(4),
              central_point_trials=10.0, covariance=np.eye(4)[np.tril_indices(4)], covariance_trials=10.0)
    with pytest.raises(gpu.SmcmcError):
        e.Restore(st)


# ---------------------------------------------------------------- posterior known answers
def test_posterior_moments_of_known_targets(gpu):
    """SURVEY.md section 8(c) known answers on the device, through the posterior reducers (PosteriorMoments over
    the pooled moment sums): iso-Gaussian -> mean 0, covariance I; quadratic form -> covariance Error^-1."""
    rng = np.random.default_rng(8)
    a = rng.standard_normal((6, 6))
    cov_true = a @ a.T / 6.0 + np.eye(6)
    for kind, dim, prm, target in ((0, 8, None, np.eye(8)), (1, 6, np.linalg.inv(cov_true), cov_true)):
        e = gpu.Engine(dim, 4096, likelihood=kind, likelihood_params=prm, seed=31)
        assert e.Start(np.zeros(dim))
        for _ in range(12):                                   # adaptation
            e.Step(100); e.sync()
        acc = gpu.PosteriorMoments(dim)
        for _ in range(25):
            e.Step(100); e.reduce_moments(); acc.add(e); e.apply_moments()
        assert acc.n == 4096 * 100 * 25
        scale = np.sqrt(np.diag(target))
        assert np.max(np.abs(acc.mean) / scale) < 0.02
        assert np.max(np.abs(acc.covariance - target) / np.outer(scale, scale)) < 0.03
        assert 0.15 < e.lane("acceptance").mean() < 0.40      # sigma still settling toward the 0.234 target


def test_frozen_lanes_after_a_million_steps(gpu, oracle):
    """D=5, 10^6 Step() calls with the covariance frozen (SetCovarianceFrozen(true), TSimpleMCMC.H:937): after a million
    steps every lane is still bit for bit the reference chain with its chain id, and every chain has settled on the
    target acceptance.  (BASELINE config 1 itself -- the covariance adapting, as SimpleMCMC.C runs it -- is
    tests/test_gpu_perchain.py::test_config1_a_million_adaptive_steps.)"""
    dim, steps = 5, 1000000
    e = gpu.Engine(dim, 64, mode=gpu.MODE_FROZEN)
    assert e.Start(np.zeros(dim))
    e.Step(steps)
    x = e.GetAccepted()
    for ch in (0, 63):
        c = oracle.Chain(dim, chain_id=ch)
        c.set_covariance_frozen(1)
        assert c.start(np.zeros(dim))
        c.run_quiet(steps)
        sc = c.scalars
        assert np.array_equal(c.accepted, x[:, ch])
        assert sc["accepted_logl"] == e.GetAcceptedLogLikelihood()[ch]
        assert sc["sigma"] == e.lane("sigma")[ch] and sc["acceptance"] == e.lane("acceptance")[ch]
        assert sc["trials"] == e.lane("trials")[ch] == steps and sc["successes"] == e.lane("successes")[ch]
        assert sc["next_update"] == e.lane("next_update")[ch] and sc["step_rms"] == e.lane("step_rms")[ch]
    acc = e.lane("naccept") / steps
    assert np.all(np.abs(acc - 0.234) < 0.02)                 # every chain settled on the target acceptance
