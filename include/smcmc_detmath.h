/* smcmc_detmath.h -- deterministic scalar math + Philox4x32 shared by the
 * HIP kernels (device), the C++ host layer and the CPU oracle.
 *
 * Why this exists: the hot path needs log (Metropolis test, reference
 * TSimpleMCMC.H:455), pow (sigma adaptation, TSimpleMCMC.H:1772-1775), a
 * N(0,1) draw (TSimpleMCMC.H:719, ROOT TRandom::Gaus) and a U(0,1) draw
 * (TSimpleMCMC.H:455, ROOT TRandom3::Rndm).  glibc and ocml differ in the last
 * bit for log/pow/sincos, and ROOT's TRandom3 is one sequential MT19937 behind
 * a process global, so "same seed => same chain" cannot be defined through
 * them.  Everything here is built only from IEEE-754 binary64 + - * / sqrt and
 * explicit fma() plus 32/64-bit integer ops, so that gcc on the host and hipcc
 * on gfx950 produce bit-identical results when both are compiled with
 * -ffp-contract=off (tests/test_detmath.py checks the host side against libm to
 * <= 2 ulp, tests/test_gpu_detmath.py checks device == host bit for bit).
 *
 * Draw-slot convention (one chain-step of dimension D):
 *   Philox key     = (seed_lo, seed_hi)                      [wave-uniform]
 *   Philox counter = (block, chain_id, step_lo, step_hi | stream<<28)
 *   word w = 4*block + lane-in-block, w = 0 .. :
 *     normal for dimension i : Box-Muller pair p = i/2 built from words
 *                              (2p, 2p+1); even i takes the cosine, odd i the sine
 *     uniform-proposal dim i : word i
 *     Metropolis uniform     : word 2*ceil(D/2)   (first word after the pairs)
 *   a 32-bit word w maps to u = (w + 0.5) * 2^-32 in (0,1), the resolution of
 *   TRandom3::Rndm.
 */
#ifndef SMCMC_DETMATH_H_SEEN
#define SMCMC_DETMATH_H_SEEN

#include <stdint.h>

#if defined(__HIPCC__)
#define SMCMC_HD __host__ __device__ __forceinline__
#elif defined(__cplusplus)
#define SMCMC_HD inline
#else
#define SMCMC_HD static inline
#endif

/* Contraction must be off in every TU that includes this header (build flags
 * carry -ffp-contract=off); the pragma is a second line of defence for clang. */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define SMCMC_FMA(a, b, c) __builtin_fma((a), (b), (c))

#if defined(__clang__)
#define SMCMC_UNROLL _Pragma("unroll")
#else
#define SMCMC_UNROLL
#endif

/* ---- bit casts -------------------------------------------------------- */
SMCMC_HD uint64_t smcmc_d2u(double x) {
    union { double d; uint64_t u; } v; v.d = x; return v.u;
}
SMCMC_HD double smcmc_u2d(uint64_t x) {
    union { double d; uint64_t u; } v; v.u = x; return v.d;
}

/* ---- Philox4x32 (Salmon et al., SC'11): the round function and key schedule, for any number of rounds.  The engine
 * draws with SMCMC_PHILOX_ROUNDS = 7 of them (smcmc_draw_block below); the 10-round generator of the paper is here for
 * the Random123 reference vectors of tests/test_detmath.py, which pin both. ------------------------------------- */
#define SMCMC_PHILOX_M0 0xD2511F53u
#define SMCMC_PHILOX_M1 0xCD9E8D57u
#define SMCMC_PHILOX_W0 0x9E3779B9u
#define SMCMC_PHILOX_W1 0xBB67AE85u

typedef struct { uint32_t v[4]; } smcmc_u32x4;

/* `rounds` rounds of Philox4x32 starting at round `first` of the key schedule (first = 0 for a whole generator; the
 * tests continue a 7-round block by three more rounds, first = 7, and compare with the 10-round known answers). */
SMCMC_HD smcmc_u32x4 smcmc_philox4x32_rounds(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1, int first, int rounds) {
    k0 += (uint32_t)first * SMCMC_PHILOX_W0;
    k1 += (uint32_t)first * SMCMC_PHILOX_W1;
    SMCMC_UNROLL
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)SMCMC_PHILOX_M0 * (uint64_t)c0;
        uint64_t p1 = (uint64_t)SMCMC_PHILOX_M1 * (uint64_t)c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += SMCMC_PHILOX_W0; k1 += SMCMC_PHILOX_W1;
    }
    smcmc_u32x4 out; out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

SMCMC_HD smcmc_u32x4 smcmc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                         uint32_t c3, uint32_t k0, uint32_t k1) {
    SMCMC_UNROLL
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)SMCMC_PHILOX_M0 * (uint64_t)c0;
        uint64_t p1 = (uint64_t)SMCMC_PHILOX_M1 * (uint64_t)c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += SMCMC_PHILOX_W0; k1 += SMCMC_PHILOX_W1;
    }
    smcmc_u32x4 out; out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

/* Counter layout of the draw-slot convention above. */
#define SMCMC_STREAM_STEP  0u   /* proposal + Metropolis draws of Step()      */
#define SMCMC_STREAM_START 1u   /* randomised start points (SimpleMCMC.C:147) */
#define SMCMC_STREAM_HMC   2u   /* momentum / epsilon / accept draws of HMC   */
#define SMCMC_STREAM_VAAT  3u   /* TProposeVAATStep chains: words 0,1 the normal pair of the step's Gaus (first normal
                                 * used), word 2 its Uniform(a,b), word 3 the Metropolis uniform, words 4+i the i-th
                                 * Uniform() of a queue shuffle made during that step (TProposeVAATStep.H:190-193) */

/* Every draw of the engine is a block of Philox4x32-7: seven rounds is the count Salmon et al. (SC'11, table 2) report
 * as the fewest that pass BigCrush ("Crush-resistant"), ten their default with a safety margin.  The headline kernel
 * spends 13 blocks per chain-step, so the three rounds are 3 % of its instructions; the round function and the key
 * schedule are pinned by the Random123 known answers of the 10-round generator (tests/test_detmath.py). */
/* Build switches of the FROZEN-DEFINITION build (tests/golden/frozen_definition_*.npz, oracle/Makefile `frozen`,
 * root-simple-mcmc_amd/build.py build_frozen_definition): -DSMCMC_PHILOX_ROUNDS=10 -DSMCMC_NORMAL_TEXTBOOK give the draws
 * their textbook definition -- the paper's ten rounds, r = sqrt(-2 ln u1), (cos, sin)(2 pi u2) through the <= 1 ulp
 * functions of this header -- which no tuning of the production transform touches: one golden set that stays what it is. */
#ifndef SMCMC_PHILOX_ROUNDS
#define SMCMC_PHILOX_ROUNDS 7
#endif
SMCMC_HD smcmc_u32x4 smcmc_draw_block(uint64_t seed, uint32_t chain, uint64_t step,
                                      uint32_t block, uint32_t stream) {
    return smcmc_philox4x32_rounds(block, chain, (uint32_t)step,
                                   ((uint32_t)(step >> 32) & 0x0FFFFFFFu) | (stream << 28),
                                   (uint32_t)seed, (uint32_t)(seed >> 32), 0, SMCMC_PHILOX_ROUNDS);
}

/* 32-bit word -> u in (0,1): (w + 0.5) * 2^-32, exact in binary64. */
SMCMC_HD double smcmc_u01(uint32_t w) {
    /* w 2^-32 + 2^-33 = (2w + 1) 2^-33: one exact fused operation */
    return SMCMC_FMA((double)w, 2.3283064365386962890625e-10, 1.16415321826934814453125e-10);
}

/* ---- log(x), x > 0 finite normal or subnormal ---------------------------
 * fdlibm e_log.c scheme: x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2),
 * s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2)).  < 1 ulp. */
/* Core for finite normal x > 0 (no special cases, no branches). */
SMCMC_HD double smcmc_log_pos(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix = smcmc_d2u(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    /* move the split point to sqrt(2)/2: add 0x3ff00000 - 0x3fe6a09e */
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffffull);
    double f = smcmc_u2d(ix) - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * SMCMC_FMA(w, SMCMC_FMA(w, Lg6, Lg4), Lg2);
    double t2 = z * SMCMC_FMA(w, SMCMC_FMA(w, SMCMC_FMA(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

SMCMC_HD double smcmc_log(double x) {
    uint64_t ix = smcmc_d2u(x);
    if (x == 0.0) return -__builtin_inf();
    if ((int64_t)ix < 0) return __builtin_nan("");
    if ((ix >> 52) == 0x7FFu) return x;           /* +inf, nan */
    if ((ix >> 52) == 0) {                        /* subnormal: scale by 2^54 */
        return smcmc_log_pos(x * 18014398509481984.0) - 54.0 * 6.93147180559945286227e-01;
    }
    return smcmc_log_pos(x);
}

/* ---- exp(x), |x| < 700 (fdlibm e_exp.c scheme) --------------------------- */
SMCMC_HD double smcmc_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.0) return __builtin_inf();
    if (x < -745.0) return 0.0;
    double kf = invln2 * x;
    /* round to nearest integer without libm: valid for |kf| < 2^51 */
    kf = (kf + 6755399441055744.0) - 6755399441055744.0;
    int k = (int)kf;
    double hi = x - kf * ln2HI;
    double lo = kf * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * SMCMC_FMA(t, SMCMC_FMA(t, SMCMC_FMA(t, SMCMC_FMA(t, P5, P4), P3), P2), P1);
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    /* scale by 2^k through the exponent field (two steps keep subnormals right) */
    if (k > -1021 && k < 1023) {
        return y * smcmc_u2d((uint64_t)(0x3ff + k) << 52);
    }
    if (k >= 1023) {
        return (y * smcmc_u2d((uint64_t)(0x3ff + (k - 1000)) << 52)) * 1.07150860718626732095e+301; /* 2^1000 */
    }
    return (y * smcmc_u2d((uint64_t)(0x3ff + (k + 1000)) << 52)) * 9.33263618503218878990e-302;     /* 2^-1000 */
}

/* exp(x) for |x| <= 700 without the overflow/underflow branches. */
SMCMC_HD double smcmc_exp_mid(double x) {
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    double kf = invln2 * x;
    kf = (kf + 6755399441055744.0) - 6755399441055744.0;
    int k = (int)kf;
    double hi = x - kf * ln2HI;
    double lo = kf * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * SMCMC_FMA(t, SMCMC_FMA(t, SMCMC_FMA(t, SMCMC_FMA(t, P5, P4), P3), P2), P1);
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return y * smcmc_u2d((uint64_t)(0x3ff + k) << 52);
}

/* pow(x, y) for finite normal x > 0 and small |y| (the sigma update uses
 * y <= 1/500, TSimpleMCMC.H:1772-1775).  exp(y*log x): with |y log x| << 1 the
 * result is within ~1 ulp of the true power. */
SMCMC_HD double smcmc_pow_small(double x, double y) {
    return smcmc_exp_mid(y * smcmc_log_pos(x));
}

/* ---- sin/cos(2*pi*u), u in [0,1) ----------------------------------------
 * j = round(4u), g = 4u - j in [-1/2,1/2] (exact for the <= 33-bit u used
 * here), phi = g*pi/2, fdlibm k_sin/k_cos polynomials on [-pi/4,pi/4], then
 * the quadrant rotation. */
SMCMC_HD void smcmc_sincos2pi(double u, double* sn, double* cs) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double pio2 = 1.57079632679489655800e+00;
    double t = 4.0 * u;
    double jf = (t + 6755399441055744.0) - 6755399441055744.0;   /* rint */
    int j = (int)jf;
    double g = t - jf;
    double x = g * pio2;
    double z = x * x;
    double ps = SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, S6, S5), S4), S3), S2), S1);
    double s = SMCMC_FMA(z * x, ps, x);
    double pc = SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, C6, C5), C4), C3), C2), C1);
    double hz = 0.5 * z;
    double c = (1.0 - hz) + (z * z) * pc;
    /* quadrant rotation: q=0 (s,c)  q=1 (c,-s)  q=2 (-s,-c)  q=3 (-c,s) */
    const int swap = j & 1;
    double rs = swap ? c : s;
    double rc = swap ? s : c;
    *sn = (j & 2) ? -rs : rs;
    *cs = ((j + 1) & 2) ? -rc : rc;
}

/* sin/cos(2*pi*u) for u = smcmc_u01(w), bit for bit smcmc_sincos2pi(smcmc_u01(w), ...)
 * with the quadrant and the reduced argument taken from the bits of w instead of
 * through floating-point rounding tricks: 4u = (w + 1/2) 2^-30 is never half-way
 * between integers, so j = rint(4u) = (w + 2^29) >> 30 and g = 4u - j =
 * (d + 1/2) 2^-30 with d = (int32)(w - (j << 30)); x = g pi/2 is one fused operation. */
SMCMC_HD void smcmc_sincos2pi_u32(uint32_t w, double* sn, double* cs) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double pio2_30 = 1.57079632679489655800e+00 * 9.31322574615478515625e-10;   /* pi/2 * 2^-30, exact scaling */
    const uint32_t j = (w >> 30) + ((w >> 29) & 1u);
    const int32_t d = (int32_t)(w - (j << 30));
    double x = SMCMC_FMA((double)d, pio2_30, 0.5 * pio2_30);
    double z = x * x;
    double ps = SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, S6, S5), S4), S3), S2), S1);
    double s = SMCMC_FMA(z * x, ps, x);
    double pc = SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, SMCMC_FMA(z, C6, C5), C4), C3), C2), C1);
    double hz = 0.5 * z;
    double c = (1.0 - hz) + (z * z) * pc;
    const uint32_t swap = j & 1u;
    double rs = swap ? c : s;
    double rc = swap ? s : c;
    /* sign flips through the sign bit: sin negative for j & 2, cos negative for (j + 1) & 2 */
    *sn = smcmc_u2d(smcmc_d2u(rs) ^ ((uint64_t)(j & 2u) << 62));
    *cs = smcmc_u2d(smcmc_d2u(rc) ^ ((uint64_t)((j + 1u) & 2u) << 62));
}

/* sqrt(x) for x well inside the normal range (here 2.3e-10 <= x <= 45): the correctly rounded
 * root.  On the device this is the rsq + Newton sequence the compiler emits for sqrt() without the
 * range scaling and the 0 / inf handling that this argument never needs. */
SMCMC_HD double smcmc_sqrt_mid(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = SMCMC_FMA(-h, g, 0.5);
    g = SMCMC_FMA(g, r, g);
    h = SMCMC_FMA(h, r, h);
    double e = SMCMC_FMA(-g, g, x);
    g = SMCMC_FMA(e, h, g);
    e = SMCMC_FMA(-g, g, x);
    return SMCMC_FMA(e, h, g);
#else
    return __builtin_sqrt(x);
#endif
}

/* ---- Box-Muller pair from two 32-bit words -------------------------------
 * n0 = r cos(theta), n1 = r sin(theta), r = sqrt(-2 log u1), u1 = (w0 + 1/2) 2^-32, theta = 2 pi (w1 + 1/2) 2^-32.
 *
 * The transform is the engine's own (the reference draws from ROOT's TRandom::Gaus, an unrelated rejection method):
 * it has to be one function of the two words on host and device and accurate far below the 2^-32 resolution of its
 * input, not the textbook libm composition.  Built for instruction count (a lone wavefront per SIMD issues one
 * instruction per four cycles whatever its type), division free, two 64-entry tables (include/smcmc_normal_tables.h):
 *   radius  u1 = 2^e m, m in [1, 2); k = top six mantissa bits, t = m r_k - 1 (|t| < 2^-7, r_k = RN(1/c_k));
 *           -2 ln u1 = (-2 ln 2) e + L_k - 2 log1p(t), L_k = RN(2 ln r_k), log1p by its series to t^6
 *           (truncation < 3e-16); the square root is the correctly rounded one;
 *   angle   w1 = q 2^30 + k 2^24 + f: quadrant q, sub-angle theta_k = (k + 1/2) pi / 128 from the table as
 *           {cos, sin}, the rest delta = (f + 1/2 - 2^23) (pi/2) 2^-30, |delta| < pi / 256, by its series
 *           (sin to delta^5, cos to delta^6: truncation < 1e-17), one rotation, then the quadrant.
 * Absolute error of a normal <= ~1e-15 (tests/test_detmath.py compares with the exact formula and checks tails and
 * moments).  The accept test's logarithm (TSimpleMCMC.H:455) and the sigma update's pow (:1772) keep the <= 1 ulp
 * functions above.  SMCMC_NORMAL_PAIR_BODY is the function body, so that a kernel with the tables in LDS can
 * instantiate it with its own loads (smcmc_kernels.hip.h); LT(k, c) / AT(k, c) read component c of entry k. */
#include "smcmc_normal_tables.h"

/* the table entries a pair needs (a kernel fetches them ahead of the arithmetic) */
SMCMC_HD uint32_t smcmc_normal_log_index(uint32_t w0) {
    return ((uint32_t)(smcmc_d2u(smcmc_u01(w0)) >> 32) >> 14) & 63u;
}
SMCMC_HD uint32_t smcmc_normal_angle_index(uint32_t w1) { return (w1 >> 24) & 63u; }

#define SMCMC_NORMAL_PAIR_BODY(LT, AT)                                                                                  \
    /* radius */                                                                                                       \
    const uint64_t ub_ = smcmc_d2u(smcmc_u01(w0));                                                                      \
    const uint32_t uh_ = (uint32_t)(ub_ >> 32);                                                                         \
    const int e_ = (int)(uh_ >> 20) - 1023;                                                                             \
    const uint32_t lk_ = (uh_ >> 14) & 63u;                                                                             \
    const double m_ = smcmc_u2d((ub_ & 0x000fffffffffffffull) | 0x3ff0000000000000ull);                                 \
    const double t_ = SMCMC_FMA(m_, LT(lk_, 0), -1.0);                                                                  \
    /* -2 log1p(t) = t (-2 + t (1 + t (-2/3 + t (1/2 + t (-2/5 + t/3))))) */                                            \
    double q_ = SMCMC_FMA(t_, 0x1.5555555555555p-2, -0x1.999999999999ap-2);                                             \
    q_ = SMCMC_FMA(t_, q_, 0.5);                                                                                        \
    q_ = SMCMC_FMA(t_, q_, -0x1.5555555555555p-1);                                                                      \
    q_ = SMCMC_FMA(t_, q_, 1.0);                                                                                        \
    q_ = SMCMC_FMA(t_, q_, -2.0);                                                                                       \
    const double s_ = SMCMC_FMA(t_, q_, SMCMC_FMA((double)e_, -0x1.62e42fefa39efp+0, LT(lk_, 1)));                      \
    const double r_ = smcmc_sqrt_mid(s_);                                                                               \
    /* angle */                                                                                                        \
    const uint32_t ak_ = (w1 >> 24) & 63u;                                                                              \
    const int32_t f_ = (int32_t)(w1 & 0x00ffffffu) - 0x00800000;                                                        \
    const double d_ = SMCMC_FMA((double)f_, 0x1.921fb54442d18p-30, 0x1.921fb54442d18p-31);   /* (f + 1/2)(pi/2) 2^-30 */ \
    const double z_ = d_ * d_;                                                                                          \
    const double sd_ = SMCMC_FMA(d_ * z_, SMCMC_FMA(z_, 0x1.1111111111111p-7, -0x1.5555555555555p-3), d_);              \
    const double cd_ = SMCMC_FMA(z_, SMCMC_FMA(z_, SMCMC_FMA(z_, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5), -0.5), 1.0); \
    const double ck_ = AT(ak_, 0), sk_ = AT(ak_, 1);                                                                    \
    const double c_ = SMCMC_FMA(-sk_, sd_, ck_ * cd_);                                                                  \
    const double sn_ = SMCMC_FMA(ck_, sd_, sk_ * cd_);                                                                  \
    /* quadrant: q=0 (c,s)  q=1 (-s,c)  q=2 (-c,-s)  q=3 (s,-c) */                                                      \
    const uint32_t qd_ = w1 >> 30;                                                                                      \
    const double rc_ = (qd_ & 1u) ? sn_ : c_;                                                                           \
    const double rs_ = (qd_ & 1u) ? c_ : sn_;                                                                           \
    const double cs_ = smcmc_u2d(smcmc_d2u(rc_) ^ ((uint64_t)((qd_ + 1u) & 2u) << 62));                                 \
    const double ss_ = smcmc_u2d(smcmc_d2u(rs_) ^ ((uint64_t)(qd_ & 2u) << 62));                                        \
    *n0 = r_ * cs_;                                                                                                     \
    *n1 = r_ * ss_;

/* The same pair with the angle's table spread over the half circle: 128 entries, entry 64 + k = (-sin, cos) of entry k
 * (the rotation by pi/2), and the other half circle is the sign of both values, i.e. of the radius.  Bit for bit
 * SMCMC_NORMAL_PAIR_BODY: negation commutes with every rounding in the rotation by delta, so q = 1 gives exactly
 * (-sn, c), and r * (-v) = -(r * v).  It trades the quadrant's compare, four selects and seven integer operations for
 * an AND and an XOR; a kernel with room for the 2 KB table uses it (smcmc_kernels.hip.h).  AT2(c) reads component c of
 * the entry (w1 >> 24) & 127. */
#define SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE(LT, AT2)                                                                      \
    const uint64_t ub_ = smcmc_d2u(smcmc_u01(w0));                                                                      \
    const uint32_t uh_ = (uint32_t)(ub_ >> 32);                                                                         \
    const int e_ = (int)(uh_ >> 20) - 1023;                                                                             \
    const uint32_t lk_ = (uh_ >> 14) & 63u;                                                                             \
    const double m_ = smcmc_u2d((ub_ & 0x000fffffffffffffull) | 0x3ff0000000000000ull);                                 \
    const double t_ = SMCMC_FMA(m_, LT(lk_, 0), -1.0);                                                                  \
    double q_ = SMCMC_FMA(t_, 0x1.5555555555555p-2, -0x1.999999999999ap-2);                                             \
    q_ = SMCMC_FMA(t_, q_, 0.5);                                                                                        \
    q_ = SMCMC_FMA(t_, q_, -0x1.5555555555555p-1);                                                                      \
    q_ = SMCMC_FMA(t_, q_, 1.0);                                                                                        \
    q_ = SMCMC_FMA(t_, q_, -2.0);                                                                                       \
    const double s_ = SMCMC_FMA(t_, q_, SMCMC_FMA((double)e_, -0x1.62e42fefa39efp+0, LT(lk_, 1)));                      \
    const double r_ = smcmc_sqrt_mid(s_);                                                                               \
    const int32_t f_ = (int32_t)(w1 & 0x00ffffffu) - 0x00800000;                                                        \
    const double d_ = SMCMC_FMA((double)f_, 0x1.921fb54442d18p-30, 0x1.921fb54442d18p-31);                              \
    const double z_ = d_ * d_;                                                                                          \
    const double sd_ = SMCMC_FMA(d_ * z_, SMCMC_FMA(z_, 0x1.1111111111111p-7, -0x1.5555555555555p-3), d_);              \
    const double cd_ = SMCMC_FMA(z_, SMCMC_FMA(z_, SMCMC_FMA(z_, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5), -0.5), 1.0); \
    const double ck_ = AT2(0), sk_ = AT2(1);                                                                            \
    const double c_ = SMCMC_FMA(-sk_, sd_, ck_ * cd_);                                                                  \
    const double sn_ = SMCMC_FMA(ck_, sd_, sk_ * cd_);                                                                  \
    const double rs_ = smcmc_u2d(smcmc_d2u(r_) ^ ((uint64_t)(w1 & 0x80000000u) << 32));                                 \
    *n0 = rs_ * c_;                                                                                                     \
    *n1 = rs_ * sn_;
SMCMC_HD uint32_t smcmc_normal_angle_index_halfcircle(uint32_t w1) { return (w1 >> 24) & 127u; }

static const double smcmc_log_table_host[128] = SMCMC_LOG_TABLE_INIT;
static const double smcmc_angle_table_host[128] = SMCMC_ANGLE_TABLE_INIT;
#if defined(__HIPCC__)
static __device__ const double smcmc_log_table_dev[128] = SMCMC_LOG_TABLE_INIT;
static __device__ const double smcmc_angle_table_dev[128] = SMCMC_ANGLE_TABLE_INIT;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SMCMC_LT_DEFAULT(k, c) smcmc_log_table_dev[2u * (k) + (c)]
#define SMCMC_AT_DEFAULT(k, c) smcmc_angle_table_dev[2u * (k) + (c)]
#else
#define SMCMC_LT_DEFAULT(k, c) smcmc_log_table_host[2u * (k) + (c)]
#define SMCMC_AT_DEFAULT(k, c) smcmc_angle_table_host[2u * (k) + (c)]
#endif

#ifdef SMCMC_NORMAL_TEXTBOOK
SMCMC_HD void smcmc_normal_pair(uint32_t w0, uint32_t w1, double* n0, double* n1) {
    const double r = smcmc_sqrt_mid(-2.0 * smcmc_log_pos(smcmc_u01(w0)));
    double sn, cs;
    smcmc_sincos2pi_u32(w1, &sn, &cs);
    *n0 = r * cs;
    *n1 = r * sn;
}
#else
SMCMC_HD void smcmc_normal_pair(uint32_t w0, uint32_t w1, double* n0, double* n1) {
    SMCMC_NORMAL_PAIR_BODY(SMCMC_LT_DEFAULT, SMCMC_AT_DEFAULT)
}
#endif

/* v[i] for a run-time i without indexing memory. */
SMCMC_HD uint32_t smcmc_select_word(smcmc_u32x4 b, uint32_t i) {
    return (i == 0u) ? b.v[0] : (i == 1u) ? b.v[1] : (i == 2u) ? b.v[2] : b.v[3];
}

/* Word index of the Metropolis uniform for dimension D (see header comment). */
SMCMC_HD uint32_t smcmc_accept_word(uint32_t dim) { return 2u * ((dim + 1u) / 2u); }

#endif
