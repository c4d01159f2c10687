"""SMCMC_LIKE_QUADFORM with a sparse Error matrix (the reference's own TDummy matrix: identity plus one correlated pair):
the serial sums walk the non-zero entries only (quadform_csr) -- bit for bit the dense D^2-term sum of
TDummyLogLikelihood.H:24-28, which SMCMC_P_DENSE_QUADFORM = 1 brings back.  (The parity tests against the oracle use
this matrix throughout, so they run the sparse walk; this file pins the two walks to each other and the fallbacks.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tdummy_error(dim):
    cov = np.eye(dim)
    cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
    return np.linalg.inv(cov)


def _state(e):
    out = {"x": e.GetAccepted()}
    for name in ("logl", "sigma", "acceptance", "step_rms", "logl_proposed"):
        out[name] = e.lane(name)
    out["naccept"] = e.lane("naccept")
    return out


def _equal(a, b, tag):
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), f"{tag}: {k} differs"


@pytest.mark.parametrize("mode,dim,exact", [("pooled", 50, True), ("pooled", 50, False), ("frozen", 31, True),
                                            ("pooled", 100, True), ("frozen", 200, True), ("per_chain", 20, True), ("per_chain", 50, True)])
def test_sparse_walk_is_the_dense_sum(gpu, mode, dim, exact):
    n = 192
    m = {"pooled": gpu.MODE_POOLED, "frozen": gpu.MODE_FROZEN, "per_chain": gpu.MODE_PER_CHAIN}[mode]
    engines = []
    for dense in (0.0, 1.0):
        e = gpu.Engine(dim, n, likelihood=gpu.LIKE_QUADFORM, likelihood_params=_tdummy_error(dim), mode=m, exact=exact)
        e.set_param("DENSE_QUADFORM", dense)
        assert e.Start(np.full(dim, 0.05))
        assert e.get_param("DENSE_QUADFORM") == dense            # 52 (2 dim + 2) of dim^2 entries: the sparse walk is on
        engines.append(e)
    for block in range(3):
        for e in engines:
            e.Step(40)
            if mode == "pooled":
                e.sync()
        _equal(_state(engines[0]), _state(engines[1]), f"{mode} D={dim} block {block}")
    assert 0 < engines[0].lane("naccept").sum() < 120 * n
    for e in engines:
        e.close()


def test_dense_matrices_keep_the_dense_sum(gpu):
    dim = 12
    rng = np.random.default_rng(3)
    a = rng.normal(size=(dim, dim))
    err = a @ a.T + dim * np.eye(dim)
    e = gpu.Engine(dim, 64, likelihood=gpu.LIKE_QUADFORM, likelihood_params=err)
    assert e.Start(np.zeros(dim))
    assert e.get_param("DENSE_QUADFORM") == 1.0                  # reads back what runs
    e.close()
    # a zero on the diagonal: a non-finite coordinate could hide from the sparse sum, so the dense one stays
    err = np.eye(dim)
    err[3, 3] = 0.0
    e = gpu.Engine(dim, 64, likelihood=gpu.LIKE_QUADFORM, likelihood_params=err)
    assert e.Start(np.zeros(dim))
    assert e.get_param("DENSE_QUADFORM") == 1.0
    e.close()


@pytest.mark.parametrize("mode,dim", [("frozen", 10), ("per_chain", 10), ("frozen", 100)])
def test_a_non_finite_proposal_takes_the_dense_sum(gpu, mode, dim):
    """inf * 0 = NaN in the dense sum (the reference's): a forced proposal with an infinite coordinate must give the
    same proposed log likelihood with and without the compressed form."""
    n = 64
    m = {"frozen": gpu.MODE_FROZEN, "per_chain": gpu.MODE_PER_CHAIN}[mode]
    got = []
    for dense in (0.0, 1.0):
        e = gpu.Engine(dim, n, likelihood=gpu.LIKE_QUADFORM, likelihood_params=_tdummy_error(dim), mode=m)
        e.set_param("DENSE_QUADFORM", dense)
        assert e.Start(np.full(dim, 0.1))
        e.Step(5)
        forced = np.full((dim, n), 0.2)
        forced[4, ::2] = np.inf
        e.ForceStep(forced)
        e.Step(1)
        first = _state(e)
        e.Step(3)
        got.append((first, _state(e)))
        e.close()
    _equal(got[0][0], got[1][0], mode + ", the forced step")
    _equal(got[0][1], got[1][1], mode + ", three steps on")
    assert np.all(np.isnan(got[0][0]["logl_proposed"][::2]))     # inf * 0: NaN, as the dense sum has it
    assert np.all(np.isfinite(got[0][0]["logl_proposed"][1::2]))
    assert np.all(np.isfinite(got[0][1]["x"]))                   # the infinite proposals were turned away
