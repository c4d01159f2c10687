// smcmc_hmc_kernel.hip.h -- many-chain Hamiltonian Monte Carlo step, the device side of
// sMCMC::TSimpleHMC::Step() (reference TSimpleHMC.H:279-401) with a fixed step
// length |fMeanEpsilon| and a fixed leapfrog count (SetLeapFrog, TSimpleHMC.H:190):
//   ProposeMomentum (:554-570), KineticEnergy (:535-542), epsilon draw (:297),
//   LeapFrog (:582-651: L steps = L+1 gradient calls), Potential (:411-414), the
//   Hamiltonian test with momentum flip on reject (:346-387) and the acceptance
//   average (:367, 386).
// Every chain is an independent reference chain (HMC shares no adaptive state once
// epsilon and L are fixed).  Layout as in smcmc_panel_kernel.hip.h: a workgroup of W
// wavefronts per 64-chain group, lane = chain, wavefront w owns the components
// i = il * W + w.  Positions and momenta live in HBM as [dim][chain]; only the
// gradient accumulators of the owned components sit in registers.  The gradient of
// the quadratic-form likelihood (TDummyLogLikelihood.H:34-42), g_i -= Error(i,j) q_j,
// is the D x D contraction of the path: the row panel of q goes through LDS, the
// wavefront's slice of Error^T through its own LDS staging area.
// Sums that the reference runs in index order over all components (kinetic energy,
// log-likelihood) are walked by wavefront 0 over LDS gather panels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_kernels.hip.h"
#include "smcmc_panel_kernel.hip.h"

namespace smcmc {

// per-chain columns of the HMC engine that reuse slots of smcmc_lane_f64 / smcmc_lane_i32 (include/smcmc.h)
constexpr int kHmcLaneMeanEpsilon = SMCMC_LANE_SIGMA;         // fMeanEpsilon
constexpr int kHmcLaneReversalLen = SMCMC_LANE_RIGIDITY;      // fReversalLen
constexpr int kHmcLaneLeapfrog = SMCMC_LANE_NEXT_UPDATE;      // fLeapFrogSteps (signed as the reference keeps it)
constexpr int kHmcLaneContributes = SMCMC_LANE_SUCCESSES;     // okLeap && isfinite(fProposedPotential) of the latest step

// max over the 64 lanes of a wavefront
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = (o > v) ? o : v;
    }
    return v;
}

// TSimpleHMC.H:302-323: what a step does to the chain's own step length, leapfrog count and reversal length
__device__ __forceinline__ void hmc_retune_after_leapfrog(int status, double eps, double& mean_eps, int& lfrog,
                                                          double& reversal) {
    if (lfrog > 0) {
        if (status != 2) {
            if (mean_eps > 0 && reversal > mean_eps) {
                const double target = reversal / 8.0;
                const double delta = target - mean_eps;
                if (delta > 0.0) mean_eps += 0.1 * delta;
            }
            if (lfrog < 50) lfrog += 1;
        } else {
            if (reversal < mean_eps) {
                reversal = __builtin_fabs(lfrog * eps);
            } else {
                reversal = 0.95 * reversal;
                reversal += 0.05 * __builtin_fabs(lfrog * eps);
            }
            if (lfrog > 3) lfrog -= 1;
            if (mean_eps > 0) mean_eps *= 0.99;
        }
    }
}

struct HmcParams {
    int nchains, npad, dim, nsteps, leapfrog, init_only;
    int adaptive;          // 1: step length and leapfrog count are per chain (lanes) and retuned every step
                           // (TSimpleHMC.H:302-323, 342-344); one step per launch, the pre-step point kept in qprev
    int gradient_type;     // PotentialGradient's type (TSimpleHMC.H:467-532), GENERIC instantiation only: 0 / 1 / 4 the
                           // likelihood's gradient, 2 the covariant approximation (cov_Eperm, cov_average), 3 finite
                           // differences of the potential (fd_grad), 5 zero
    const double* cov_Eperm;     // type 2: fEstimatedError in the layout of Eperm
    const double* cov_average;   // type 2: fAveragePoint [dim]
    double* fd_grad;             // type 3: the gradient as wavefront 0 assembles it [dim][npad]
    double* p0;            // adaptive: the momentum LeapFrog started from (the reversal test, :633-638) [dim][npad]
    double* qprev;         // adaptive: fAccepted as UpdateCovariance sees it (:338)                     [dim][npad]
    uint32_t step0, chain_offset;
    uint64_t seed;
    double alpha, abs_eps;
    const double* Eperm;   // QUADFORM: [W][dim][CW], Eperm[w][j][il] = Error(il*W + w, j)
    const double* like;    // ROSENBROCK: {b}
    double* q;             // accepted position   [dim][npad]
    double* pm;            // accepted momentum   [dim][npad]
    double* qn;            // proposed position   [dim][npad]
    double* pn;            // proposed momentum   [dim][npad]
    double* lane_f64;      // LOGL = -accepted potential, LOGL_PROPOSED = -proposed potential, ACCEPTANCE
    int32_t* lane_i32;     // NACCEPT, LAST_ACCEPT, TRIALS = step count
};

template <int W, int CW, int LIKE, bool GENERIC = false>
__global__ void __launch_bounds__(W * kWave) hmc_step_kernel(const HmcParams p) {
    __shared__ double rbuf[kPanelRows * kWave];                                   // q rows of the current panel
    __shared__ __attribute__((aligned(16))) double ulds[W * kPanelRows * CW];     // Error^T panels / gather panels
    __shared__ double verdict_f[kWave];
    __shared__ int verdict_i[kWave];
    constexpr int kGd = kGatherJl * W * kWave;
    constexpr int ngather = (CW + kGatherJl - 1) / kGatherJl;

    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int chain = blockIdx.x * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    const int npanels = (D + kPanelRows - 1) / kPanelRows;
    const uint32_t ew = smcmc_accept_word((uint32_t)D);
    const double rb = (LIKE == SMCMC_LIKE_ROSENBROCK) ? p.like[0] : 0.0;

    double* lf = p.lane_f64 + chain;
    int32_t* li = p.lane_i32 + chain;
    double pot_acc = -lf[SMCMC_LANE_LOGL * NP];
    double pot_prop = -lf[SMCMC_LANE_LOGL_PROPOSED * NP];
    double acceptance = lf[SMCMC_LANE_ACCEPTANCE * NP];
    int naccept = li[SMCMC_LANE_NACCEPT * NP];
    int last_accept = li[SMCMC_LANE_LAST_ACCEPT * NP];
    int trials = li[SMCMC_LANE_TRIALS * NP];

    double gr[CW];   // potential gradient of the owned components

    // walks `count` values per chain, dimension order, through wavefront 0:
    // each owner publishes f(il) for its components, wavefront 0 folds them with `fold`
    auto gather = [&](auto&& value_of, auto&& fold) {
        for (int g = 0; g < ngather; ++g) {
            __syncthreads();
            static_for<ngather>([&](auto gc) {
                if (g == decltype(gc)::value) {
#pragma unroll
                    for (int qq = 0; qq < kGatherJl; ++qq) {
                        constexpr int gg = decltype(gc)::value;
                        const int il = gg * kGatherJl + qq;
                        if (il < CW) {
                            double a, b;
                            value_of(il, a, b);
                            ulds[(qq * W + w) * kWave + lane] = a;
                            ulds[kGd + (qq * W + w) * kWave + lane] = b;
                        }
                    }
                }
            });
            __syncthreads();
            if (w == 0) {
                for (int qq = 0; qq < kGatherJl; ++qq)
                    for (int ww = 0; ww < W; ++ww) {
                        const int i = (g * kGatherJl + qq) * W + ww;
                        if (i < D) fold(i, ulds[(qq * W + ww) * kWave + lane], ulds[kGd + (qq * W + ww) * kWave + lane]);
                    }
            }
        }
        __syncthreads();
    };

    // gr[il] = sum_j M(i, j) (qn_j - shift_j), j ascending, for the owned components i = il * W + w; Mperm in the layout
    // of Eperm.  The sum is run as g = 0; g -= M q; g = -g, which rounds exactly like the ascending sum of the products.
    auto contract = [&](const double* Mperm, const double* shift) {
#pragma unroll
        for (int il = 0; il < CW; ++il) gr[il] = 0.0;
        for (int pnl = 0; pnl < npanels; ++pnl) {
            const int j0 = pnl * kPanelRows;
            const int j1 = (j0 + kPanelRows < D) ? j0 + kPanelRows : D;
            __syncthreads();
            for (int r = w; r < j1 - j0; r += W) {
                double v = p.qn[(size_t)(j0 + r) * NP + chain];
                if (shift != nullptr) v = v - shift[j0 + r];
                rbuf[r * kWave + lane] = v;
            }
            {
                const f64x2* src = (const f64x2*)(Mperm + ((size_t)w * D + j0) * CW);
                f64x2* dst = (f64x2*)(ulds + w * (kPanelRows * CW));
                const int npieces = (j1 - j0) * (CW / 2);
                // eight loads in flight per lane (one at a time, each waits out an L2 round trip before its LDS store)
                constexpr int kInFlight = 8;
                for (int k0 = lane; k0 < npieces; k0 += kInFlight * kWave) {
                    f64x2 t[kInFlight];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u)
                        if (k0 + u * kWave < npieces) t[u] = src[k0 + u * kWave];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u)
                        if (k0 + u * kWave < npieces) dst[k0 + u * kWave] = t[u];
                }
            }
            __syncthreads();
            lds_cptr_f64 up = (lds_cptr_f64)(ulds + w * (kPanelRows * CW));
            asm volatile("" : "+v"(up));
            for (int j = j0; j < j1; ++j) {
                const double qj = rbuf[(j - j0) * kWave + lane];
                lds_cptr_f64 erow = up + (j - j0) * CW;
#pragma unroll
                for (int c = 0; c < CW; c += 16) {
                    f64x2 e2[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e2[k] = *(volatile lds_cptr_f64x2)(erow + c + 2 * k);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        gr[c + k] -= e2[k / 2][k & 1] * qj;
                        asm volatile("" : "+v"(gr[c + k]));
                    }
                }
            }
        }
#pragma unroll
        for (int il = 0; il < CW; ++il) gr[il] = -gr[il];
        __syncthreads();
    };

    // the likelihood's own potential gradient at qn for the owned components -> gr[]  (PotentialGradient type 0
    // with the user gradient, TSimpleHMC.H:467-492: the negated gradient of log L)
    auto like_gradient = [&]() {
        __syncthreads();   // every owner has written its part of qn
        if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
#pragma unroll
            for (int il = 0; il < CW; ++il) {
                const int i = il * W + w;
                const double g = (i < D) ? -p.qn[(size_t)i * NP + chain] : 0.0;   // g[i] = -p[i]
                gr[il] = -g;
            }
        } else if constexpr (LIKE == SMCMC_LIKE_ROSENBROCK) {
            // THardLogLikelihood.H:70-91, then g = -g twice (log L -> potential)
#pragma unroll
            for (int il = 0; il < CW; ++il) {
                const int i = il * W + w;
                double g = 0.0;
                if (i < D) {
                    const double pi = p.qn[(size_t)i * NP + chain];
                    if (i == 0) {
                        const double p1 = p.qn[NP + chain];
                        g = -2.0 * (1.0 - pi) - 4.0 * rb * pi * (p1 - pi * pi);
                    } else if (i < D - 1) {
                        const double pm1 = p.qn[(size_t)(i - 1) * NP + chain];
                        const double pp1 = p.qn[(size_t)(i + 1) * NP + chain];
                        g = 2.0 * rb * (pi - pm1 * pm1);
                        g += -2.0 * (1.0 - pi);
                        g += -4.0 * rb * pi * (pp1 - pi * pi);
                    } else {
                        const double pm1 = p.qn[(size_t)(i - 1) * NP + chain];
                        g = +2.0 * rb * (pi - pm1 * pm1);
                    }
                    g = -g;        // THardLogLikelihood.H:88
                }
                gr[il] = -g;       // TSimpleHMC.H:486
            }
        } else {
            // TDummyLogLikelihood.H:34-42: g[i] = 0; g[i] -= Error(i,j)*p[j], j ascending; then TSimpleHMC.H:486
            contract(p.Eperm, nullptr);
        }
    };

    // log L at qn (valid in wavefront 0) and, on the way, the kinetic energy of pn
    auto log_likelihood_at_qn = [&](bool gradient_is_current, double& ke) {
        double lsum = 0.0, prev_q = 0.0;
        gather([&](int il, double& a, double& b) {
                   const int i = il * W + w;
                   a = (i < D) ? p.pn[(size_t)i * NP + chain] : 0.0;
                   b = (i < D) ? p.qn[(size_t)i * NP + chain] : 0.0;
               },
               [&](int i, double a, double b) {
                   ke += a * a / 2.0;
                   if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
                       lsum += -0.5 * b * b;                                   // README.md:57-66
                   } else if constexpr (LIKE == SMCMC_LIKE_ROSENBROCK) {
                       if (i > 0) {                                            // THardLogLikelihood.H:60-64, term i-1
                           const double aa = (1.0 - prev_q);
                           const double bb = b - prev_q * prev_q;
                           lsum -= aa * aa + rb * bb * bb;
                       }
                       prev_q = b;
                   }
               });
        if constexpr (LIKE == SMCMC_LIKE_USER || LIKE == SMCMC_LIKE_ASYM || LIKE == SMCMC_LIKE_HORRIFIC ||
                      LIKE == SMCMC_LIKE_CONSTRAINED) {
            // a likelihood with no gradient of its own (the stress targets, a compiled-in user likelihood): one lane
            // per chain walks the image of qn as the reference's functor walks its vector (the gather's last barrier
            // has made every owner's part of qn visible); it is an HMC target through the finite-difference, the
            // covariant or the zero gradient (TSimpleHMC.H:417-454, 508-529)
            if (w == 0) lsum = serial_loglike<LIKE, true>(p.qn, chain, NP, D, p.like);
        }
        if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
            // log L = -1/2 q^T Error q.  The gradient at the final position is still in gr[]
            // (gr = Error q): the potential is folded from it in dimension order instead of
            // re-running the D^2-term sum of TDummyLogLikelihood.H:24-28 serially.
            if (!gradient_is_current) like_gradient();
            double usum = 0.0;
            gather([&](int il, double& a, double& b) {
                       const int i = il * W + w;
                       a = (i < D) ? p.qn[(size_t)i * NP + chain] : 0.0;
                       b = gr[il];
                   },
                   [&](int, double a, double b) { usum += 0.5 * a * b; });
            lsum = -usum;
        }
        return lsum;
    };

    // PotentialGradient (:467-532) at qn -> gr[]
    auto gradient = [&]() {
        if constexpr (!GENERIC) {
            like_gradient();
        } else {
            if (p.gradient_type == 2) {
                // CovariantGradient (:447-454): grad[i] += fEstimatedError(i,j) * (point[j] - fAveragePoint[j])
                __syncthreads();
                contract(p.cov_Eperm, p.cov_average);
            } else if (p.gradient_type == 3) {
                // FiniteDifferenceGradient (:417-444): two potentials per dimension, du = 0.01.  Wavefront 0 moves
                // the coordinate, receives the potentials and keeps the result in fd_grad.
                const double du = 0.01;
                for (int i = 0; i < D; ++i) {
                    __syncthreads();
                    double* cell = p.qn + (size_t)i * NP + chain;
                    const double keep = (w == 0) ? *cell : 0.0;
                    double work = keep;
                    work -= du;
                    if (w == 0) *cell = work;
                    __syncthreads();
                    double unused = 0.0;
                    const double u1 = -log_likelihood_at_qn(false, unused);
                    work += 2.0 * du;
                    if (w == 0) *cell = work;
                    __syncthreads();
                    const double u2 = -log_likelihood_at_qn(false, unused);
                    if (w == 0) {
                        p.fd_grad[(size_t)i * NP + chain] = 0.5 * (u2 - u1) / du;
                        *cell = keep;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int il = 0; il < CW; ++il) {
                    const int i = il * W + w;
                    gr[il] = (i < D) ? p.fd_grad[(size_t)i * NP + chain] : 0.0;
                }
                __syncthreads();
            } else if (p.gradient_type == 5) {
                __syncthreads();
#pragma unroll
                for (int il = 0; il < CW; ++il) gr[il] = 0.0;
            } else {
                like_gradient();
            }
        }
    };
    // is gr[] the likelihood's own gradient after gradient()?  (the quadratic form folds its potential from it)
    const bool own_gradient = !GENERIC || !(p.gradient_type == 2 || p.gradient_type == 3 || p.gradient_type == 5);

    if (p.init_only) {
        // Start (:210-269): SetPosition's Potential(start) for every chain
#pragma unroll
        for (int il = 0; il < CW; ++il) {
            const int i = il * W + w;
            if (i < D) {
                p.qn[(size_t)i * NP + chain] = p.q[(size_t)i * NP + chain];
                p.pn[(size_t)i * NP + chain] = 0.0;
            }
        }
        __syncthreads();
        double ke = 0.0;
        const double l0 = log_likelihood_at_qn(false, ke);
        if (w == 0 && active) {
            lf[SMCMC_LANE_LOGL * NP] = l0;
            lf[SMCMC_LANE_LOGL_PROPOSED * NP] = l0;
        }
        return;
    }

    // adaptive mode: the chain's own step length, leapfrog count and reversal length (lanes), one step per launch
    double mean_eps = p.adaptive ? lf[kHmcLaneMeanEpsilon * NP] : -p.abs_eps;
    int lfrog = p.adaptive ? li[kHmcLaneLeapfrog * NP] : -p.leapfrog;
    double reversal = p.adaptive ? lf[kHmcLaneReversalLen * NP] : 0.0;
    int contributes = 1;

    for (int s = 0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);   // ++fStepCount, :286
        ++trials;

        // ---- ProposeMomentum (:554-570) for the owned components ----
        const double mix = __builtin_sqrt(1.0 - p.alpha * p.alpha);
#pragma unroll
        for (int il = 0; il < CW; ++il) {
            const int i = il * W + w;
            if (i < D) {
                const double m = p.pm[(size_t)i * NP + chain];
                double v;
                if (p.alpha >= 1.0) {
                    v = m / p.alpha;
                } else {
                    const uint32_t pr = (uint32_t)i >> 1;
                    smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, pr >> 1, SMCMC_STREAM_HMC);
                    double n0, n1;
                    const uint32_t w0 = (pr & 1u) ? blk.v[2] : blk.v[0], w1 = (pr & 1u) ? blk.v[3] : blk.v[1];
                    smcmc_normal_pair(w0, w1, &n0, &n1);
                    const double r = (i & 1) ? n1 : n0;
                    v = p.alpha * m + mix * r;
                }
                p.pn[(size_t)i * NP + chain] = v;
                if (p.adaptive) p.p0[(size_t)i * NP + chain] = v;               // LeapFrog: momentum = pNew (:587)
                p.qn[(size_t)i * NP + chain] = p.q[(size_t)i * NP + chain];   // LeapFrog: qNew = position (:586)
            }
        }
        // ---- initial kinetic energy (:292), dimension order ----
        double ke0 = 0.0;
        gather([&](int il, double& a, double& b) {
                   const int i = il * W + w;
                   a = (i < D) ? p.pn[(size_t)i * NP + chain] : 0.0;
                   b = 0.0;
               },
               [&](int, double a, double) { ke0 += a * a / 2.0; });

        // ---- epsilon (:297-298) ----
        const smcmc_u32x4 eblk = smcmc_draw_block(p.seed, gid, step, ew >> 2, SMCMC_STREAM_HMC);
        const double abs_eps = __builtin_fabs(mean_eps);
        const double lo = 0.9 * abs_eps, hi = 1.1 * abs_eps;
        const double eps = lo + (hi - lo) * smcmc_u01(smcmc_select_word(eblk, ew & 3u));
        const smcmc_u32x4 ablk = smcmc_draw_block(p.seed, gid, step, (ew + 1u) >> 2, SMCMC_STREAM_HMC);
        const double uacc = smcmc_u01(smcmc_select_word(ablk, (ew + 1u) & 3u));

        // ---- LeapFrog (:582-651).  Chains of the group may differ in their leapfrog count: the group runs to
        // the largest one, a chain takes its last (half) kick at its own count and then stands still. ----
        const int L = active ? (lfrog < 0 ? -lfrog : lfrog) : 0;
        const int Lmax = wave_max_i32(L);
        // the reversal test only reaches a chain whose leapfrog count is automatic (hmc_retune_after_leapfrog): with every
        // count of the group fixed its gathers are skipped (lane = chain in every wavefront: workgroup-uniform)
        const bool reversal_wanted = p.adaptive && wave_max_i32((active && lfrog > 0) ? 1 : 0) != 0;
        int status = 1;                                                        // leapStatus
        if (Lmax < 1) {
            // the one-step shortcut (:598-611): qNew += eps*(momentum + pNew)/2 with pNew == momentum
#pragma unroll
            for (int il = 0; il < CW; ++il) {
                const int i = il * W + w;
                if (i < D) {
                    const double m = p.pn[(size_t)i * NP + chain];
                    const double qv = p.qn[(size_t)i * NP + chain];
                    p.qn[(size_t)i * NP + chain] = qv + eps * (m + m) / 2.0;
                }
            }
        } else {
            gradient();                                                        // :615
#pragma unroll
            for (int il = 0; il < CW; ++il) {                                  // :618-620
                const int i = il * W + w;
                if (i < D && L >= 1) {
                    const double m = p.pn[(size_t)i * NP + chain];
                    p.pn[(size_t)i * NP + chain] = m - eps * gr[il] / 2.0;
                }
            }
            for (int ls = 0; ls < Lmax; ++ls) {
                // iteration ls of a chain with L steps: ls < L - 1 the body of :623-639, ls == L - 1 the last
                // position step and half kick of :641-648, afterwards nothing
                const bool live = ls < L, last = ls == L - 1;
#pragma unroll
                for (int il = 0; il < CW; ++il) {
                    const int i = il * W + w;
                    if (i < D && live) {
                        const double m = p.pn[(size_t)i * NP + chain];
                        const double qv = p.qn[(size_t)i * NP + chain];
                        p.qn[(size_t)i * NP + chain] = qv + eps * m;
                    }
                }
                gradient();
#pragma unroll
                for (int il = 0; il < CW; ++il) {
                    const int i = il * W + w;
                    if (i < D && live) {
                        const double m = p.pn[(size_t)i * NP + chain];
                        p.pn[(size_t)i * NP + chain] = last ? m - eps * gr[il] / 2.0 : m - eps * gr[il];
                    }
                }
                if (reversal_wanted && ls < Lmax - 1) {
                    // has the direction reversed (:633-638)?  Only the automatic leapfrog count listens.
                    __syncthreads();
                    double inner = 0.0;
                    gather([&](int il, double& a, double& b) {
                               const int i = il * W + w;
                               a = (i < D) ? p.pn[(size_t)i * NP + chain] : 0.0;
                               b = (i < D) ? p.p0[(size_t)i * NP + chain] : 0.0;
                           },
                           [&](int, double a, double b) { inner += a * b; });
                    if (ls < L - 1 && !(inner >= 0.0)) status = 2;
                }
            }
        }
        __syncthreads();
        if (p.adaptive) hmc_retune_after_leapfrog(status, eps, mean_eps, lfrog, reversal);   // :302-323 (wavefront 0 holds the status)

        // ---- proposed kinetic energy and potential (:326-327), dimension order ----
        double ke1 = 0.0;
        const double lsum = log_likelihood_at_qn(Lmax >= 1 && own_gradient, ke1);

        // ---- Hamiltonian test (:333-387), wavefront 0 decides ----
        if (w == 0) {
            pot_prop = -lsum;
            const double h_prop = pot_prop + ke1;
            const double h_acc = pot_acc + ke0;
            const double delta = h_prop - h_acc;
            const double trial = -smcmc_log_pos(uacc);
            const bool reject = (delta > trial) || !__builtin_isfinite(delta) || !active;
            verdict_i[lane] = reject ? 0 : 1;
            verdict_f[lane] = pot_prop;
        }
        __syncthreads();
        const bool take = verdict_i[lane] != 0;
        pot_prop = verdict_f[lane];
        // UpdateCovariance runs on a finite proposal (okLeap is never zero); otherwise the step length shrinks (:336-344)
        contributes = __builtin_isfinite(pot_prop) ? 1 : 0;
        if (p.adaptive && !contributes && mean_eps > 0) mean_eps = 0.3 * mean_eps;
        if (take) {
            pot_acc = pot_prop;
            acceptance = (acceptance * 4999.0 + 1.0) / 5000.0;                  // :386
            ++naccept;
        } else {
            acceptance = (acceptance * 4999.0) / 5000.0;                        // :367
        }
        last_accept = take ? 1 : 0;
#pragma unroll
        for (int il = 0; il < CW; ++il) {
            const int i = il * W + w;
            if (i < D && active) {
                if (p.adaptive) p.qprev[(size_t)i * NP + chain] = p.q[(size_t)i * NP + chain];   // what :338 folds
                if (take) {                                                     // :380-383
                    p.q[(size_t)i * NP + chain] = p.qn[(size_t)i * NP + chain];
                    p.pm[(size_t)i * NP + chain] = p.pn[(size_t)i * NP + chain];
                } else {                                                        // :364-366
                    p.pm[(size_t)i * NP + chain] = -p.pm[(size_t)i * NP + chain];
                }
            }
        }
        __syncthreads();
    }

    if (active && w == 0) {
        lf[SMCMC_LANE_LOGL * NP] = -pot_acc;
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = -pot_prop;
        lf[SMCMC_LANE_ACCEPTANCE * NP] = acceptance;
        li[SMCMC_LANE_NACCEPT * NP] = naccept;
        li[SMCMC_LANE_LAST_ACCEPT * NP] = last_accept;
        li[SMCMC_LANE_TRIALS * NP] = trials;
        if (p.adaptive) {
            lf[kHmcLaneMeanEpsilon * NP] = mean_eps;
            lf[kHmcLaneReversalLen * NP] = reversal;
            li[kHmcLaneLeapfrog * NP] = lfrog;
            li[kHmcLaneContributes * NP] = contributes;
        }
    }
}

template <int W, int CW>
hipError_t launch_hmc(const HmcParams& p, int like, hipStream_t stream);

}  // namespace smcmc
