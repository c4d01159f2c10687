"""The device autocorrelation reducer (smcmc_autocorrelation_sums) against the restatement of
MakeAutocorrelation.C:108-148 in oracle/oracle.py."""
import numpy as np
import pytest


def _rho(total, lagged, nslots, nchains):
    n = (nslots - np.arange(lagged.shape[0]))[:, None] * float(nchains)
    mean = total / n[0]
    var = lagged[0] / n[0] - mean * mean
    return (lagged / n - mean * mean) / var


def test_pooled_sums_reproduce_the_macro_on_one_chain(oracle):
    """One chain, one dimension: the pooled-sum form equals the macro's ring-buffer loop."""
    rng = np.random.default_rng(1)
    n, phi = 4000, 0.8
    s = np.zeros(n)
    for t in range(1, n):
        s[t] = phi * s[t - 1] + rng.standard_normal()
    s += 3.0
    total, lagged = oracle.autocorrelation_sums(s[:, None, None], nlags=16)
    rho = _rho(total, lagged, n, 1)[:, 0]
    ref = oracle.autocorrelation_reference(s, 16)
    assert np.allclose(rho[1:], ref[1:], rtol=0, atol=1e-12)
    assert np.allclose(rho[1:6], phi ** np.arange(1, 6), atol=0.06)        # AR(1): a(k) = phi^k
    shifted, lagged2 = oracle.autocorrelation_sums(s[:, None, None], centre=[2.5], nlags=16)
    # a reference point only enters through the edges of the lagged sums: O(lag / n)
    assert np.allclose(_rho(shifted, lagged2, n, 1)[:, 0], rho, atol=16 / n * 3)


def test_python_autocorrelation_class(smcmc, oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((300, 3, 40)).cumsum(axis=0) * 0.01 + rng.standard_normal((300, 3, 40))
    total, lagged = oracle.autocorrelation_sums(x)
    a = smcmc.Autocorrelation(total, lagged, 300, 40)
    assert np.allclose(a.rho(), _rho(total, lagged, 300, 40))
    half = [oracle.autocorrelation_sums(x[:, :, :20]), oracle.autocorrelation_sums(x[:, :, 20:])]
    b = smcmc.Autocorrelation(*half[0], 300, 20) + smcmc.Autocorrelation(*half[1], 300, 20)
    assert np.allclose(b.rho(), a.rho(), rtol=1e-12, atol=1e-14)              # ranks add their sums
    assert np.all(a.tau() >= 1.0 - 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,nchains,steps,stride", [(5, 70, 640, 4), (50, 256, 512, 8), (100, 64, 80, 1)])
def test_device_sums_match_oracle(gpu, oracle, dim, nchains, steps, stride):
    import torch
    # D > 63 saves a trace with a frozen covariance only
    e = gpu.Engine(dim, nchains, seed=9, mode=gpu.MODE_FROZEN if dim > 63 else gpu.MODE_POOLED)
    assert e.Start(np.zeros(dim))
    e.Step(300)
    slots = steps // stride
    sx = torch.full((slots, e.dim_padded, e.nchains_padded), float("nan"), dtype=torch.float64, device="cuda")
    sl = torch.empty((slots, e.nchains_padded), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    e.StepSave(steps, sx.data_ptr(), sl.data_ptr(), stride=stride)
    torch.cuda.synchronize()
    centre = e.GetEstimatedCenter()
    a = e.AutocorrelationSums(sx.data_ptr(), slots, centre=centre)
    x = sx[:, :dim, :nchains].cpu().numpy()
    total, lagged = oracle.autocorrelation_sums(x, centre, nlags=64)
    assert np.allclose(a.sum, total, rtol=1e-11, atol=1e-9)
    assert np.allclose(a.lagged, lagged, rtol=1e-11, atol=1e-9)
    again = e.AutocorrelationSums(sx.data_ptr(), slots, centre=centre)
    plain = e.AutocorrelationSums(sx.data_ptr(), slots)                         # the macro's own origin
    assert np.allclose(plain.lagged, oracle.autocorrelation_sums(x, None, nlags=64)[1], rtol=1e-11, atol=1e-9)
    assert np.array_equal(a.lagged, again.lagged) and np.array_equal(a.sum, again.sum)   # fixed summation order
    rho = a.rho()
    assert np.allclose(rho[0], 1.0) and np.all(np.abs(rho[:min(slots, 64)]) < 1.5)
    assert np.all(a.tau() > 0.5)


@pytest.mark.gpu
def test_device_sums_reject_bad_arguments(gpu):
    import ctypes as C
    lib = gpu.load()
    out = (C.c_double * 64)()
    assert lib.smcmc_autocorrelation_sums(None, 4, 2, 2, 64, 64, None, out, out, None) == 1       # SMCMC_ERR_INVALID
