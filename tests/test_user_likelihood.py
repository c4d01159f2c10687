"""A user likelihood compiled into the library (SMCMC_LIKE_USER; build.py --user-likelihood): the
reference's TASymLogLikelihood (TAsymLogLikelihood.H:20-31) as examples/user_likelihood_asym.hip.h."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USER_LIB = os.path.join(ROOT, "root-simple-mcmc_amd", "lib", "libsmcmc_amd_user.so")
HEADER = os.path.join(ROOT, "examples", "user_likelihood_asym.hip.h")


def _user_lib(smcmc):
    """The example library: built by __graft_entry__.build(); rebuilt here only if it is missing."""
    if not os.path.exists(USER_LIB):
        smcmc._build_mod.build(user_likelihood=HEADER)
    return USER_LIB


def test_user_library_exports_the_whole_c_abi(smcmc):
    lib = smcmc.load(_user_lib(smcmc))
    for name in smcmc.SIGNATURES:
        assert hasattr(lib, name)


def test_plain_build_refuses_a_user_likelihood(smcmc):
    import ctypes as C
    h = C.c_void_p()
    st = smcmc.load().smcmc_create(5, 8, smcmc.LIKE_USER, 1, 0, 0, C.byref(h))
    assert st == 5               # SMCMC_ERR_UNSUPPORTED, before any device is looked for


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frozen", "pooled"])
def test_asymmetric_user_likelihood_runs_in_the_step_kernel(gpu, mode):
    dim, n = 10, 2048
    slopes = np.array([-1.0, 100.0])
    e = gpu.Engine(dim, n, likelihood=gpu.LIKE_USER, likelihood_params=slopes, library=_user_lib(gpu),
                   mode=gpu.MODE_FROZEN if mode == "frozen" else gpu.MODE_POOLED, seed=5)
    assert e.Start(np.full(dim, 0.5))
    for _ in range(15):
        e.Step(200)
        if mode == "pooled":
            e.sync()
    x, logl = e.GetAccepted(), e.GetAcceptedLogLikelihood()
    # the stored likelihood is the function of the stored point, bit for bit (same operation order)
    ref = np.zeros(n)
    for i in range(dim):
        a = x[i].copy()
        neg = a < 0.0
        a[neg] *= slopes[1]
        a[~neg] *= slopes[0]
        ref += a
    assert np.array_equal(logl, ref)
    # p(x) ~ exp(-x) above zero, exp(100 x) below: mean 1 - 0.01 + O(1e-4) per dimension, almost no mass below zero
    assert abs(x.mean() - 0.99) < 0.06
    assert (x < -0.1).mean() < 1e-3
    assert 0.1 < e.lane("naccept").sum() / (n * 3000) < 0.6


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n", [(10, 6), (30, 3), (50, 2), (63, 2)])
def test_user_likelihood_with_every_chain_adapting_alone(gpu, oracle, dim, n):
    """SMCMC_MODE_PER_CHAIN with a user likelihood (the reference's own mode for a user's functor): the one-chain-per-
    wavefront kernel evaluates smcmc_user_loglike on the whole point in every lane; the example is TASymLogLikelihood, so
    every chain is oracle.Chain with that likelihood, bit for bit, through its own UpdateProposal events; and the
    per-step record (what TSimpleMCMC_amd.H::Step() runs ahead on) is the chain step by step."""
    slopes = np.array([-1.0, 100.0])
    e = gpu.Engine(dim, n, likelihood=gpu.LIKE_USER, likelihood_params=slopes, library=_user_lib(gpu),
                   mode=gpu.MODE_PER_CHAIN, seed=11)
    assert e.get_param("PERCHAIN_WAVE") == 1.0
    chains = [oracle.Chain(dim, kind=oracle.LIKE_ASYM, seed=11, chain_id=c) for c in range(n)]
    x0 = np.full(dim, 0.5)
    assert e.Start(x0)
    for c in chains:
        assert c.start(x0)
    for steps in (700, 1500):
        e.Step(steps)
        for c in chains:
            c.run_quiet(steps)
        x, logl, sigma = e.GetAccepted(), e.GetAcceptedLogLikelihood(), e.lane("sigma")
        for k, c in enumerate(chains):
            assert np.array_equal(x[:, k], c.accepted) and logl[k] == c.scalars["accepted_logl"] and sigma[k] == c.scalars["sigma"]
    rec = e.StepRecorded(40)
    for k in range(40):
        chains[0].step(False, 0)
        assert np.array_equal(rec["accepted"][k], chains[0].accepted)


@pytest.mark.gpu
def test_user_likelihood_limits(gpu):
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.Engine(600, 64, likelihood=gpu.LIKE_USER, library=_user_lib(gpu))      # dim <= 512
    assert err.value.status == 5
    e = gpu.Engine(100, 64, likelihood=gpu.LIKE_USER, likelihood_params=np.array([-1.0, 100.0]), library=_user_lib(gpu),
                   exact=False)
    assert e.Start(np.full(100, 0.5))
    with pytest.raises(gpu.SmcmcError) as err:
        e.Step(1)                                                                   # dim > 63: reference order only
    assert err.value.status == 5
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.Engine(5, 64, likelihood=gpu.LIKE_USER)                                 # the plain library has none
    assert err.value.status == 5


def _lanes_equal(a, b, names):
    for name in names:
        assert np.array_equal(a.lane(name), b.lane(name)), name


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frozen", "pooled"])
@pytest.mark.parametrize("dim", [100, 300])
def test_user_likelihood_above_63_dimensions(gpu, dim, mode):
    """smcmc_user_loglike_at (the point read from its [dim][chain] image) in the large-dimension kernel: the example is
    the library's own ASYM likelihood written as user code, so the two engines must agree bit for bit."""
    n = 192
    slopes = np.array([-1.0, 100.0])
    m = gpu.MODE_FROZEN if mode == "frozen" else gpu.MODE_POOLED
    u = gpu.Engine(dim, n, likelihood=gpu.LIKE_USER, likelihood_params=slopes, library=_user_lib(gpu), mode=m, seed=9)
    b = gpu.Engine(dim, n, likelihood=gpu.LIKE_ASYM, likelihood_params=slopes, mode=m, seed=9)
    x0 = np.random.default_rng(dim).normal(0.3, 0.2, size=(dim, n))
    assert u.Start(x0) and b.Start(x0)
    names = ("logl", "sigma", "acceptance", "step_rms", "logl_proposed", "trials", "successes", "naccept")
    _lanes_equal(u, b, names)
    for w in range(3):
        u.Step(10); b.Step(10)
        if mode == "pooled":
            u.sync(); b.sync()
        assert np.array_equal(u.GetAccepted(), b.GetAccepted()), f"window {w}"
        _lanes_equal(u, b, names)
    assert u.lane("naccept").sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("dim", [10, 40, 100])
def test_user_likelihood_in_the_variable_at_a_time_chains(gpu, dim, exact):
    n = 130
    slopes = np.array([-1.0, 100.0])
    u = gpu.VaatEngine(dim, n, likelihood=gpu.LIKE_USER, likelihood_params=slopes, library=_user_lib(gpu), seed=4, exact=exact)
    b = gpu.VaatEngine(dim, n, likelihood=gpu.LIKE_ASYM, likelihood_params=slopes, seed=4, exact=exact)
    x0 = np.random.default_rng(dim).uniform(-1.0, 1.0, size=(dim, n))
    assert u.Start(x0) and b.Start(x0)
    u.UpdateProposal(); b.UpdateProposal()
    for chunk in (1, dim, dim + 3):
        u.Step(chunk); b.Step(chunk)
        assert np.array_equal(u.GetAccepted(), b.GetAccepted())
        _lanes_equal(u, b, ("logl", "logl_proposed", "step_rms", "trials", "successes", "naccept", "last_index"))
        for name in ("sigma", "acceptance", "acceptance_trials"):
            assert np.array_equal(u.per_dim(name), b.per_dim(name)), name
    assert u.lane("naccept").sum() > 0


@pytest.mark.gpu
def test_cpp_host_with_a_user_likelihood(gpu, tmp_path):
    """examples/UserLikelihood_amd.C: TSimpleMCMC<TASymLogLikelihood> linked against the user library; the device's
    log-likelihood of the accepted point equals the host functor's, bit for bit."""
    import subprocess
    _user_lib(gpu)
    libdir = os.path.dirname(USER_LIB)
    exe = str(tmp_path / "user_amd.exe")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", f"-I{os.path.join(ROOT, 'include')}",
                        os.path.join(ROOT, "examples", "UserLikelihood_amd.C"), f"-L{libdir}", "-lsmcmc_amd_user",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "512"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "(identical)" in r.stdout
    # ... and as the reference is used: ONE chain adapting alone, Step() per call (SMCMC_MODE_PER_CHAIN with the user's
    # function in the one-chain-per-wavefront kernel, Step() running ahead on its record)
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "(identical)" in r.stdout and "Step() runs ahead: 1" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [8, 100])
def test_user_likelihood_as_an_hmc_target(gpu, dim):
    """A compiled-in user likelihood under TSimpleHMC through the finite-difference gradient (TSimpleHMC.H:417-444):
    examples/user_likelihood_asym.hip.h is the built-in TAsymLogLikelihood, chain for chain, bit for bit."""
    n = 64
    slopes = np.array([-1.0, 100.0])
    user = gpu.HmcEngine(dim, n, likelihood=gpu.LIKE_USER, likelihood_params=slopes, seed=5, library=_user_lib(gpu))
    ref = gpu.HmcEngine(dim, n, likelihood=gpu.LIKE_ASYM, likelihood_params=slopes, seed=5)
    x0 = np.full(dim, 0.5)
    for h in (user, ref):
        h.Start(x0)
        h.SetGradientType(3)
    for k in range(2):
        user.Step(3); ref.Step(3)
        for a, b in zip(user.state(), ref.state()):
            assert np.array_equal(a, b), f"block {k}"
        for name in ("mean_epsilon", "leapfrog", "acceptance", "naccept"):
            assert np.array_equal(user.lane(name), ref.lane(name)), name
    assert user.lane("trials").max() == 6
    with pytest.raises(gpu.SmcmcError):
        user.Step(1, gradient_type=0)                            # no gradient of its own
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.HmcEngine(dim, n, likelihood=gpu.LIKE_USER)          # the plain library carries no user likelihood
    assert err.value.status == 5
