"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path
reproduces them with no oracle in the loop."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FROZEN = ["frozen_iso_d5.npz", "frozen_rosenbrock_d6.npz", "frozen_iso_d50.npz"]
POOLED = ["pooled_iso_d5.npz", "pooled_iso_d20.npz"]
LARGE = ["frozen_iso_d200_reference.npz", "frozen_iso_d200_fused.npz", "frozen_rosenbrock_d100_reference.npz",
         "frozen_rosenbrock_d100_fused.npz"]
HMC = ["hmc_quadform_d60_reference.npz", "hmc_quadform_d60_fused.npz"]


def _load(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


@pytest.mark.parametrize("name", FROZEN)
def test_oracle_reproduces_frozen_golden(oracle, name):
    g = _load(name)
    dim, kind, steps = int(g["dim"]), int(g["kind"]), int(g["steps"])
    for ch in range(g["accepted"].shape[0]):
        c = oracle.Chain(dim, kind=kind, seed=int(g["seed"]), chain_id=ch)
        c.set_covariance_frozen(1)
        x0 = g["x0"][:, ch] if g["x0"].ndim == 2 else g["x0"]
        assert c.start(x0)
        r = c.run(steps)
        assert np.array_equal(r["accepted"], g["accepted"][ch])
        assert np.array_equal(r["logl_accepted"], g["logl"][ch])
        assert np.array_equal(r["sigma"], g["sigma"][ch])
        assert np.array_equal(c.accepted, g["x"][:, ch])


@pytest.mark.parametrize("name", POOLED)
def test_oracle_reproduces_pooled_golden(oracle, name):
    g = _load(name)
    e = oracle.Ensemble(int(g["nchains"]), int(g["dim"]), kind=int(g["kind"]), seed=int(g["seed"]),
                        mode=oracle.MODE_POOLED)
    assert e.start(np.zeros(int(g["dim"])))
    m = None
    for _ in range(int(g["nwin"])):
        e.step(int(g["window"]))
        m = e.reduce_moments()
        e.apply_moments(m)
    e.step(4)
    assert np.array_equal(e.x, g["x"]) and np.array_equal(m, g["last_moments"])
    assert np.array_equal(e.decomposition, g["decomposition"]) and np.array_equal(e.covariance, g["covariance"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", FROZEN)
def test_hip_reproduces_frozen_golden(gpu, name):
    g = _load(name)
    dim, kind, steps = int(g["dim"]), int(g["kind"]), int(g["steps"])
    nch = g["accepted"].shape[0]
    prm = [100.0] if kind == 2 else None
    e = gpu.Engine(dim, nch, likelihood=kind, likelihood_params=prm, seed=int(g["seed"]), mode=gpu.MODE_FROZEN)
    assert e.Start(g["x0"])
    acc, logl, sigma = [], [], []
    for _ in range(steps):
        e.Step(1)
        acc.append(e.lane("last_accept").copy()); logl.append(e.lane("logl").copy()); sigma.append(e.lane("sigma").copy())
    assert np.array_equal(np.array(acc).T.astype(np.uint8), g["accepted"])   # identical accept/reject sequence
    assert np.array_equal(np.array(logl).T, g["logl"])                        # log-likelihoods: 0 ulp
    assert np.array_equal(np.array(sigma).T, g["sigma"])
    assert np.array_equal(e.GetAccepted(), g["x"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", POOLED)
def test_hip_reproduces_pooled_golden(gpu, name):
    g = _load(name)
    dim = int(g["dim"])
    e = gpu.Engine(dim, int(g["nchains"]), likelihood=int(g["kind"]), seed=int(g["seed"]), mode=gpu.MODE_POOLED)
    assert e.Start(np.zeros(dim))
    m = None
    for _ in range(int(g["nwin"])):
        e.Step(int(g["window"]))
        e.reduce_moments()
        m = e.read_moments()
        e.apply_moments()
    e.Step(4)
    assert np.array_equal(m, g["last_moments"])
    assert np.array_equal(e.GetAccepted(), g["x"])
    assert np.array_equal(e.lane("logl"), g["logl"]) and np.array_equal(e.lane("sigma"), g["sigma"])
    assert np.array_equal(e.lane("naccept"), g["naccept"])
    assert np.array_equal(e.decomposition, g["decomposition"]) and np.array_equal(e.covariance, g["covariance"])
    assert np.array_equal(e.GetEstimatedCenter(), g["center"])


# ---------------------------------------------------------------- large dimensions and HMC
@pytest.mark.parametrize("name", LARGE)
def test_oracle_reproduces_large_dim_golden(oracle, name):
    g = _load(name)
    dim, kind, n = int(g["dim"]), int(g["kind"]), int(g["nchains"])
    prm = [100.0] if kind == 2 else None
    e = oracle.Ensemble(n, dim, kind=kind, params=prm, seed=int(g["seed"]), mode=oracle.MODE_FROZEN,
                        exact=bool(g["exact"]))
    assert e.start(g["x0"])
    e.step(int(g["steps"]))
    assert np.array_equal(e.x, g["x"]) and np.array_equal(e.lane("logl"), g["logl"])
    assert np.array_equal(e.lane("sigma"), g["sigma"]) and np.array_equal(e.lane("naccept"), g["naccept"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", LARGE)
def test_hip_reproduces_large_dim_golden(gpu, name):
    """D > 63: the panel kernel (reference order) and the matrix-pipe kernel (fused order)."""
    g = _load(name)
    dim, kind, n = int(g["dim"]), int(g["kind"]), int(g["nchains"])
    prm = [100.0] if kind == 2 else None
    e = gpu.Engine(dim, n, likelihood=kind, likelihood_params=prm, seed=int(g["seed"]), mode=gpu.MODE_FROZEN,
                   exact=bool(g["exact"]))
    assert e.Start(g["x0"])
    e.Step(int(g["steps"]))
    assert np.array_equal(e.GetAccepted(), g["x"]) and np.array_equal(e.lane("logl"), g["logl"])
    assert np.array_equal(e.lane("sigma"), g["sigma"]) and np.array_equal(e.lane("naccept"), g["naccept"])
    assert np.array_equal(e.lane("step_rms"), g["step_rms"])


def _hmc_oracle_state(oracle, g):
    dim, n = int(g["dim"]), int(g["nchains"])
    q, m, pot = [], [], []
    for ch in range(n):
        h = oracle.Hmc(dim, kind=1, params=g["error"], seed=int(g["seed"]), chain_id=ch, potential_from_gradient=True,
                       fused_gradient=bool(g["fused"]))
        h.set_alpha(float(g["alpha"])); h.start(np.ones(dim))
        h.set_mean_epsilon(-float(g["epsilon"])); h.set_leapfrog(int(g["leapfrog"]))
        h.run(int(g["steps"]))
        q.append(h.accepted); m.append(h.momentum); pot.append(h.scalars["accepted_potential"])
    return np.array(q).T, np.array(m).T, np.array(pot)


@pytest.mark.parametrize("name", HMC)
def test_oracle_reproduces_hmc_golden(oracle, name):
    g = _load(name)
    q, m, pot = _hmc_oracle_state(oracle, g)
    assert np.array_equal(q, g["q"]) and np.array_equal(m, g["momentum"]) and np.array_equal(pot, g["potential"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", HMC)
def test_hip_reproduces_hmc_golden(gpu, name):
    g = _load(name)
    dim, n = int(g["dim"]), int(g["nchains"])
    e = gpu.HmcEngine(dim, n, likelihood=1, likelihood_params=g["error"], seed=int(g["seed"]),
                      exact=not bool(g["fused"]))
    e.SetAlpha(float(g["alpha"]))
    e.Start(np.ones(dim))
    e.SetMeanEpsilon(-float(g["epsilon"]))
    e.SetLeapFrog(int(g["leapfrog"]))
    e.Step(int(g["steps"]))
    q, m, logl = e.state()
    assert np.array_equal(q, g["q"]) and np.array_equal(m, g["momentum"]) and np.array_equal(-logl, g["potential"])
    assert np.array_equal(e.lane("acceptance"), g["acceptance"])
