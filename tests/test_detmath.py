"""include/smcmc_detmath.h on the host: Philox against the published Random123
known-answer vectors, the deterministic log/exp/sincos/pow against libm."""
import numpy as np


def _ulp_err(a, ref_ld):
    ref = ref_ld.astype(np.float64)
    sp = np.spacing(np.abs(ref)).astype(np.longdouble)
    return float(np.max(np.abs(a.astype(np.longdouble) - ref_ld) / sp))


def test_philox4x32_10_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_log_within_one_ulp(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2, 200000),
                        (rng.integers(0, 2 ** 32, 200000) + 0.5) * 2.0 ** -32, [5e-324, 1e-310, 2.2250738585072014e-308]])
    assert _ulp_err(oracle.det_log(x), np.log(x.astype(np.longdouble))) < 1.0
    with np.errstate(all="ignore"):
        special = oracle.det_log(np.array([0.0, np.inf, -1.0, np.nan]))
    assert special[0] == -np.inf and special[1] == np.inf and np.isnan(special[2]) and np.isnan(special[3])


def test_exp_within_one_ulp(oracle):
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(-5, 5, 200000), rng.uniform(-0.02, 0.02, 200000), rng.uniform(-700, 700, 20000)])
    assert _ulp_err(oracle.det_exp(t), np.exp(t.astype(np.longdouble))) < 1.0


def test_sincos2pi(oracle):
    rng = np.random.default_rng(3)
    u = (rng.integers(0, 2 ** 32, 400000) + 0.5) * 2.0 ** -32
    s, c = oracle.det_sincos2pi(u)
    pi = np.longdouble(4) * np.arctan(np.longdouble(1))
    ang = 2 * pi * u.astype(np.longdouble)
    assert np.max(np.abs(s - np.sin(ang))) < 3e-16
    assert np.max(np.abs(c - np.cos(ang))) < 3e-16
    assert np.max(np.abs(s * s + c * c - 1.0)) < 5e-16


def test_pow_small_within_one_ulp(oracle):
    rng = np.random.default_rng(4)
    x = rng.uniform(1e-3, 4.3, 200000)
    y = rng.uniform(1e-7, 2e-3, 200000)
    assert _ulp_err(oracle.det_pow_small(x, y), np.power(x.astype(np.longdouble), y.astype(np.longdouble))) < 1.0


def test_normals_are_standard(oracle):
    v = np.concatenate([oracle.step_draws(7, c, 1, 50)[0] for c in range(8000)])
    n = v.size
    assert abs(v.mean()) < 4 / np.sqrt(n)
    assert abs(v.var() - 1.0) < 4 * np.sqrt(2.0 / n)
    kurt = ((v - v.mean()) ** 4).mean() / v.var() ** 2
    assert abs(kurt - 3.0) < 0.1
    # pairs share nothing: cos/sin members are uncorrelated
    assert abs(np.corrcoef(v[0::2], v[1::2])[0, 1]) < 4 / np.sqrt(n / 2)


def test_draw_slots(oracle):
    """The draw-slot convention: chains, steps and seeds give distinct streams; the
    Metropolis uniform of a D-dim step sits in word 2*ceil(D/2)."""
    n1, u1 = oracle.step_draws(1, 0, 1, 5)
    n2, _ = oracle.step_draws(1, 1, 1, 5)
    n3, _ = oracle.step_draws(1, 0, 2, 5)
    n4, _ = oracle.step_draws(2, 0, 1, 5)
    assert not np.array_equal(n1, n2) and not np.array_equal(n1, n3) and not np.array_equal(n1, n4)
    n50, _ = oracle.step_draws(1, 0, 1, 50)
    assert np.array_equal(n50[:5], n1)            # normals depend on (seed, chain, step, dim index) only
    w = oracle.philox([1, 0, 1, 0], [1, 0])       # block 1 of chain 0, step 1, seed 1
    assert u1 == (w[2] + 0.5) * 2.0 ** -32        # D=5: word 6 = block 1, lane 2


def test_word_forms_equal_the_double_forms(oracle):
    """The integer-path argument reduction and the fused u01 are bit for bit the
    floating-point forms they replace (so the draw streams did not change)."""
    rng = np.random.default_rng(5)
    w = np.concatenate([rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64),
                        np.array([0, 1, 2 ** 29 - 1, 2 ** 29, 2 ** 29 + 1, 2 ** 30 - 1, 2 ** 30, 2 ** 31 - 1, 2 ** 31,
                                  3 * 2 ** 29, 3 * 2 ** 30 - 1, 3 * 2 ** 30, 7 * 2 ** 29 - 1, 7 * 2 ** 29,
                                  2 ** 32 - 2, 2 ** 32 - 1], dtype=np.uint64)]).astype(np.float64)
    u = (w + 0.5) * 2.0 ** -32
    assert np.array_equal(oracle.det_u01(w), u)
    s0, c0 = oracle.det_sincos2pi(u)
    s1, c1 = oracle.det_sincos2pi_u32(w)
    assert np.array_equal(s0, s1) and np.array_equal(c0, c1)
    # and the pair itself: r = sqrt(-2 log u1), (cos, sin)(2 pi u2)
    w0, w1 = w, w[::-1].copy()
    n0, n1 = oracle.det_normal_pair(w0, w1)
    r = np.sqrt(-2.0 * oracle.det_log((w0 + 0.5) * 2.0 ** -32))
    s, c = oracle.det_sincos2pi((w1 + 0.5) * 2.0 ** -32)
    assert np.array_equal(n0, r * c) and np.array_equal(n1, r * s)
