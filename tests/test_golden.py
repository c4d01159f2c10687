"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path
reproduces them with no oracle in the loop."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FROZEN = ["frozen_iso_d5.npz", "frozen_rosenbrock_d6.npz", "frozen_iso_d50.npz"]
POOLED = ["pooled_iso_d5.npz", "pooled_iso_d20.npz"]


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
