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


def test_draws_use_seven_rounds_of_the_same_generator(oracle):
    """smcmc_draw_block is Philox4x32-7: the same round function and key schedule as the 10-round generator the
    known answers pin -- three more rounds applied to a 7-round block give the published 10-round vectors."""
    assert oracle.philox_draw_rounds() == 7
    for ctr, key, kat in (([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
                          ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
                          ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
                           [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])):
        assert oracle.philox_rounds(ctr, key, 0, 10) == kat
        seven = oracle.philox_rounds(ctr, key, 0, 7)
        assert seven != kat
        assert oracle.philox_rounds(seven, key, 7, 3) == kat
    # the block of (seed, chain, step, block, stream): counter (block, chain, step lo, step hi | stream << 28), key = seed
    seed = (0x299f31d0 << 32) | 0xa4093822
    assert oracle.draw_block(seed, 5, (3 << 32) | 9, 2, 1) == oracle.philox_rounds([2, 5, 9, 3 | (1 << 28)],
                                                                                  [0xa4093822, 0x299f31d0], 0, 7)


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
    w = oracle.philox_rounds([1, 0, 1, 0], [1, 0], 0, 7)   # block 1 of chain 0, step 1, seed 1
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


def _exact_pair(w0, w1):
    """Box-Muller in extended precision: r = sqrt(-2 ln u1), (cos, sin)(2 pi u2), u = (w + 1/2) 2^-32."""
    ld = np.longdouble
    u1 = (w0.astype(ld) + ld(0.5)) * ld(2.0) ** -32
    u2 = (w1.astype(ld) + ld(0.5)) * ld(2.0) ** -32
    r = np.sqrt(ld(-2.0) * np.log(u1))
    pi = ld(4) * np.arctan(ld(1))
    # reduce the angle exactly before the library call: theta = 2 pi u2, u2 in (0, 1)
    return r * np.cos(2 * pi * u2), r * np.sin(2 * pi * u2)


def test_normal_pair_against_the_exact_transform(oracle):
    """The table-driven pair (smcmc_normal_pair) against Box-Muller in extended precision: absolute error far below
    the 2^-32 = 2.3e-10 resolution of its input words -- a few ulp except next to u1 = 1 -- over random words, every
    table boundary of radius and angle, both ends of the radius (6.8 sigma and 1.5e-5) and every quadrant edge."""
    rng = np.random.default_rng(5)
    edge = []
    for k in range(64):                                     # words whose u1 mantissa sits on a table boundary
        for e in (0, 3, 17, 31):
            base = (64 + k) << 25 >> e
            edge += [max(base - 1, 0), base, base + 1]
    edge += [0, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1]
    ang = []
    for q in range(4):
        for k in (0, 1, 31, 32, 63):
            b = (q << 30) + (k << 24)
            ang += [b, b + 1, b + (1 << 23) - 1, b + (1 << 23), b + (1 << 24) - 1]
    edge = np.array(edge, dtype=np.uint64) % (2 ** 32)
    ang = np.array(ang, dtype=np.uint64)
    w0 = np.concatenate([rng.integers(0, 2 ** 32, 400000, dtype=np.uint64), np.repeat(edge, ang.size)])
    w1 = np.concatenate([rng.integers(0, 2 ** 32, 400000, dtype=np.uint64), np.tile(ang, edge.size)])
    n0, n1 = oracle.det_normal_pair(w0.astype(np.float64), w1.astype(np.float64))
    e0, e1 = _exact_pair(w0, w1)
    err = np.maximum(np.abs(n0 - e0), np.abs(n1 - e1)).astype(np.float64)
    # error model: a few ulp of the normal itself, plus the 2e-16 absolute error of -2 ln u1 (the rounding of its two
    # constants) seen through the square root, which matters only next to u1 = 1 (r down to 1.5e-5, where it is 1e-11)
    r = np.hypot(e0, e1).astype(np.float64)
    assert np.all(err < 4e-15 + 4e-16 / r), float(np.max(err / (4e-15 + 4e-16 / r)))
    assert float(err.max()) < 2.3e-10 / 20                             # the input's resolution is 2^-32 = 2.3e-10
    assert np.max(np.abs(n0)) > 6.5 and np.max(np.abs(n0)) < 6.8        # w0 = 0: r = sqrt(66 ln 2) = 6.76
    # exact symmetries of the construction: the angle word's top two bits are the quadrant
    a0, a1 = oracle.det_normal_pair(w0.astype(np.float64), ((w1 + 2 ** 31) % 2 ** 32).astype(np.float64))
    assert np.array_equal(a0, -n0) and np.array_equal(a1, -n1)           # theta + pi: the same pair negated
    b0, b1 = oracle.det_normal_pair(w0.astype(np.float64), ((w1 + 2 ** 30) % 2 ** 32).astype(np.float64))
    assert np.array_equal(b0, -n1) and np.array_equal(b1, n0)            # theta + pi / 2


def test_normal_tails_and_moments(oracle):
    """The first four moments over a stratified sample of the 2^32 radius words (every 2^10-th word, each with a
    random angle word) and the tail: P(|z| > t) against the normal law out to 6 sigma through the radius words that
    reach it."""
    from math import erfc, sqrt
    rng = np.random.default_rng(6)
    w0 = (np.arange(2 ** 22, dtype=np.uint64) << 10) + rng.integers(0, 2 ** 10, 2 ** 22, dtype=np.uint64)
    w1 = rng.integers(0, 2 ** 32, w0.size, dtype=np.uint64)
    n0, n1 = oracle.det_normal_pair(w0.astype(np.float64), w1.astype(np.float64))
    z = np.concatenate([n0, n1])
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs((z ** 2).mean() - 1.0) < 4 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 4 * np.sqrt(96.0 / n)
    for t in (1.0, 2.0, 3.0, 4.0):
        p = erfc(t / sqrt(2.0))
        got = float((np.abs(z) > t).mean())
        assert abs(got - p) < 5 * np.sqrt(p / n) + 1e-7, (t, got, p)
    # the radius is monotone in its word and reaches the tails: r > t exactly for the words below 2^32 exp(-t^2 / 2)
    for t in (4.0, 5.0, 6.0, 6.5):
        edge = int(np.floor(2.0 ** 32 * np.exp(-t * t / 2.0) - 0.5))
        ws = np.arange(max(edge - 3, 0), edge + 4, dtype=np.float64)
        r = np.hypot(*oracle.det_normal_pair(ws, np.full(ws.size, 12345.0)))
        exact = np.sqrt(-2.0 * np.log((ws + 0.5) * 2.0 ** -32))
        assert np.max(np.abs(r - exact)) < 1e-14
        assert np.all((r > t) == (exact > t))
    # tail mass out to 6 sigma: the 2^32 exp(-18) = 65 radius words beyond r = 6, all angles equally likely
    ws = np.arange(0, 4096, dtype=np.float64)
    r = np.hypot(*oracle.det_normal_pair(ws, np.full(ws.size, 999.0)))
    assert np.all(np.diff(r) < 0)
    assert (r > 6.0).sum() == int(np.floor(2.0 ** 32 * np.exp(-18.0) - 0.5)) + 1


def test_halfcircle_form_of_the_pair_is_the_same_function(oracle):
    """The step kernels run SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE (a 128-entry angle table over the half circle, the other
    half as the sign of the radius) instead of the quadrant logic of smcmc_normal_pair: the same bits for every word --
    all 256 table cells with the extreme and random remainders, and random words."""
    rng = np.random.default_rng(11)
    cells = np.arange(256, dtype=np.uint64) << np.uint64(24)
    rests = np.concatenate([[0, 1, 0x7fffff, 0x800000, 0x800001, 0xffffff], rng.integers(0, 1 << 24, 58)]).astype(np.uint64)
    w1 = (cells[:, None] | rests[None, :]).ravel()
    w0 = rng.integers(0, 1 << 32, w1.size, dtype=np.uint64)
    w0[:8] = [0, 1, 2, 0xffffffff, 0xfffffffe, 0x80000000, 0x7fffffff, 0x00010000]
    more = rng.integers(0, 1 << 32, (2, 200000), dtype=np.uint64)
    w0 = np.concatenate([w0, more[0]]).astype(np.float64)
    w1 = np.concatenate([w1, more[1]]).astype(np.float64)
    a0, a1 = oracle.det_normal_pair(w0, w1)
    b0, b1 = oracle.det_normal_pair_halfcircle(w0, w1)
    assert np.array_equal(a0.view(np.uint64), b0.view(np.uint64))
    assert np.array_equal(a1.view(np.uint64), b1.view(np.uint64))
