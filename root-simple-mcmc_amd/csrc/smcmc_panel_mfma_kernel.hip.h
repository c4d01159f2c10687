// smcmc_panel_mfma_kernel.hip.h -- the Metropolis step for 63 < D <= 512 in the fused order
// (SMCMC_P_EXACT_ARITHMETIC = 0) with the proposal x' = x + sigma U^T r (reference
// TSimpleMCMC.H:709-724) on the FP64 matrix pipe.
//
// The fused order is xp[j] = fma(sigma r_i, U(i,j), xp[j]), i ascending, starting from x[j]: a chain
// of v_mfma_f64_16x16x4_f64 whose C operand starts as x does exactly that (order pinned on the
// hardware by tests/test_gpu_parity.py), so the kernel is bit for bit oracle/ensemble_oracle.c with
// exact = 0, like panel_step_kernel<..., EXACT = false> which it replaces.
//
// Layout (the one of hmc_mfma_kernel): a workgroup of 8 wavefronts advances 32 chains; a matrix
// instruction yields D[row = component j][col = chain]; lane l holds chain l & 15 and components
// (l >> 4) + 4 r of its tiles; the accepted point and the proposal stay in registers in that layout
// for the whole launch.  z = sigma r goes to LDS as z[i][chain] (B operand: the row quad 4 kq ..
// 4 kq + 3); U^T streams from L2 in operand order, Uop[tile][kq][lane] = U(4 kq + (lane >> 4),
// 16 tile + (lane & 15)), only the k-quads on or above the diagonal (kq <= 4 tile + 3).  Tiles are
// dealt to the wavefronts in snake order so that the triangular work is balanced.  StepRMS and the
// likelihood, which sum over all components in index order, are walked by one lane per chain over
// values published through the same LDS array; those lanes also carry the per-chain scalar state
// (UpdateState's scalar half, TSimpleMCMC.H:1723-1776) and run the Metropolis test (:410-463).
//
// LIKE = QUADFORM (TDummyLogLikelihood.H:21-31 at D > 63): log L = -1/2 sum_i p_i (Error p)_i, the rows
// (Error p)_i = sum_j fma(Error(i,j), p_j, .) by a second chain of matrix instructions (Eop in the
// `like` pointer, the layout of hmc_mfma_kernel), the outer sum in dimension order -- the association
// oracle/ensemble_oracle.c calls quadform_rowwise.  To make room for the row sums the accepted point
// is not held in registers across the likelihood: p.x in HBM is kept current (written on accept) and
// read back after the verdict.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_panel_kernel.hip.h"

namespace smcmc {

constexpr int kPmCT = 32;   // chains per workgroup
constexpr int kPmW = 8;     // wavefronts per workgroup

constexpr int kPmPF = 8;    // k-quads of the A operand fetched ahead of the matrix instructions

// k-quads per tile in the operand images (U^T, Error): rounded up to the prefetch depth, zero padded
__host__ __device__ inline int panel_mfma_nkq_padded(int dim) { return ((dim + 3) / 4 + kPmPF - 1) / kPmPF * kPmPF; }
inline size_t panel_mfma_uop_doubles(int dim) {
    const int ntiles = (dim + 15) / 16;
    return (size_t)ntiles * panel_mfma_nkq_padded(dim) * 64;
}

// acc0/acc1 += A(tile rows, k) * z(k, chains 0-15 / 16-31) over k-quads [0, kend), kend a multiple of kPmPF:
// the A operands (one coalesced 512-byte read per k-quad) run kPmPF k-quads ahead of their use.  Used for
// the full-length contractions (Error p, Error q); the triangular proposal, whose tiles are short and whose
// kernel is tighter on registers, measured faster with the plain loop.
template <typename F4>
__device__ __forceinline__ void pm_contract(const double* __restrict__ a_ptr, const double* qs_lane, int kend,
                                            F4& acc0, F4& acc1) {
    double a0[kPmPF], a1[kPmPF];
#pragma unroll
    for (int u = 0; u < kPmPF; ++u) a0[u] = (kend > 0) ? a_ptr[(size_t)u * 64] : 0.0;
    for (int kq0 = 0; kq0 < kend; kq0 += kPmPF) {
        if (kq0 + kPmPF < kend) {
#pragma unroll
            for (int u = 0; u < kPmPF; ++u) a1[u] = a_ptr[(size_t)(kq0 + kPmPF + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < kPmPF; ++u) {
            const double b0 = qs_lane[(size_t)(4 * (kq0 + u)) * 32];
            const double b1 = qs_lane[(size_t)(4 * (kq0 + u)) * 32 + 16];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kPmPF; ++u) a0[u] = a1[u];
    }
}

// TI = 16-component tiles per wavefront: dim <= 128 TI
// VARIANT: PM_PLAIN the common kernel; PM_FORCED the one-step instantiation that proposes the ForceStep point;
// PM_KEEP the one that also stores the proposal of the launch's last step for GetProposed() (both kept out of
// the common kernel, whose register allocation the extra paths disturb: 0.30 -> 0.37 ms/step at D=500)
enum { PM_PLAIN = 0, PM_FORCED = 1, PM_KEEP = 2 };
template <int TI, int LIKE, int VARIANT = PM_PLAIN>
__global__ void __launch_bounds__(kPmW* kWave, 1) panel_mfma_kernel(const PanelParams p) {
    __shared__ double qs[16 * kPmW * TI * kPmCT];   // z[i][chain], later the published values of the ordered sums
    __shared__ double sig[kPmCT], x0s[kPmCT];
    __shared__ int verdict[kPmCT];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int w = tid / kWave;
    const int c = lane & 15, rq = lane >> 4;
    const int base = blockIdx.x * kPmCT;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const int ntiles = (D + 15) / 16, nkq = (D + 3) / 4, nkqp = panel_mfma_nkq_padded(D);
    const uint32_t aw = smcmc_accept_word((uint32_t)D);
    const double rb = (LIKE == SMCMC_LIKE_ROSENBROCK) ? p.like[0] : 0.0;

    const bool summer = (w == 0) && (lane < kPmCT);   // one lane per chain: scalar state, ordered sums, the test
    const int mychain = base + lane;
    const bool active = summer && mychain < p.nchains;

    typedef double f64x4v __attribute__((ext_vector_type(4)));
    f64x4v x[TI][2], xp[TI][2];

    // tile of slot t of this wavefront, snake order: 0..7, 15..8, 16..23, 31..24
    auto tile = [&](int t) { return t * kPmW + ((t & 1) ? (kPmW - 1 - w) : w); };
    auto owns = [&](int t) { return tile(t) < ntiles; };
    auto comp = [&](int t, int r) { return 16 * tile(t) + 4 * r + rq; };
    auto slot = [&](int t, int ct, int r) { return comp(t, r) * kPmCT + 16 * ct + c; };

    auto publish = [&](auto&& value) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            if (!owns(t)) continue;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) qs[slot(t, ct, r)] = (comp(t, r) < D) ? value(t, ct, r) : 0.0;
        }
        __syncthreads();
    };

    // one lane per chain walks qs[i][chain], i = 0 .. n-1, in order: the LDS reads of a few components are issued ahead of the
    // (serially dependent) sums that consume them, the next few are in flight meanwhile -- read one at a time, each sum waited
    // out an LDS round trip and the seven other wavefronts of the workgroup waited with it (a third of the step at D = 500).
    // How far ahead is a register question: at TI = 4 sixteen ahead doubled the spills (93 VGPRs) and the in-launch save of
    // every step, which reads the spilled point back, went from free to +100 us per step; four ahead spills what the plain
    // loop did (45) and is as fast in the walk.
    auto ordered_walk = [&](int n, auto&& f) {
        constexpr int kAhead = (TI >= 4) ? 4 : 8;
        int i0 = 0;
        double v[kAhead], vn[kAhead];
        if (n >= kAhead) {
#pragma unroll
            for (int u = 0; u < kAhead; ++u) v[u] = qs[u * kPmCT + lane];
        }
        for (; i0 + kAhead <= n; i0 += kAhead) {
            const int nx = (i0 + 2 * kAhead <= n) ? i0 + kAhead : i0;   // the next full chunk (or this one again: a harmless re-read)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) vn[u] = qs[(nx + u) * kPmCT + lane];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) f(i0 + u, v[u]);
#pragma unroll
            for (int u = 0; u < kAhead; ++u) v[u] = vn[u];
        }
        for (; i0 < n; ++i0) f(i0, qs[i0 * kPmCT + lane]);
    };

    // per-chain scalar state (summing lanes)
    double logl = 0, sigma = 0, acc_rate = 0, acc_trials = 0, rigid = 0, last_value = 0, last_x0 = 0, step_rms = 0,
           logl_prop = 0;
    int trials = 0, succ = 0, next_update = 0, naccept = 0, rms_trials = 0, last_accept = 0;
    if (summer) {
        const double* lf = p.lane_f64 + mychain;
        const int32_t* li = p.lane_i32 + mychain;
        logl = lf[SMCMC_LANE_LOGL * NP];
        sigma = lf[SMCMC_LANE_SIGMA * NP];
        acc_rate = lf[SMCMC_LANE_ACCEPTANCE * NP];
        acc_trials = lf[SMCMC_LANE_ACCEPTANCE_TRIALS * NP];
        rigid = lf[SMCMC_LANE_RIGIDITY * NP];
        last_value = lf[SMCMC_LANE_LAST_VALUE * NP];
        last_x0 = lf[SMCMC_LANE_LAST_X0 * NP];
        step_rms = lf[SMCMC_LANE_STEP_RMS * NP];
        logl_prop = lf[SMCMC_LANE_LOGL_PROPOSED * NP];
        trials = li[SMCMC_LANE_TRIALS * NP];
        succ = li[SMCMC_LANE_SUCCESSES * NP];
        next_update = li[SMCMC_LANE_NEXT_UPDATE * NP];
        naccept = li[SMCMC_LANE_NACCEPT * NP];
        rms_trials = li[SMCMC_LANE_STEP_RMS_TRIALS * NP];
        last_accept = li[SMCMC_LANE_LAST_ACCEPT * NP];
        x0s[lane] = p.x[mychain];
    }

    // the accepted point, matrix layout
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = comp(t, r);
                // the state streams through once per launch: non-temporal, so that it does not push U out of L2
                x[t][ct][r] = (owns(t) && i < D) ? __builtin_nontemporal_load(&p.x[(size_t)i * NP + base + 16 * ct + c]) : 0.0;
            }
    // rows of z past D stay zero for the whole launch
    for (int k = tid; k < 16 * kPmW * TI * kPmCT; k += kPmW * kWave) qs[k] = 0.0;
    __syncthreads();

    // QUADFORM: v <- 0.5 v (Error v) element by element, v published in qs; the caller sums it in order
    auto quadform_terms = [&](f64x4v (&v)[TI][2]) {
        f64x4v gr[TI][2];
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            gr[t][0] = f64x4v{0.0, 0.0, 0.0, 0.0};
            gr[t][1] = f64x4v{0.0, 0.0, 0.0, 0.0};
        }
        const double* qs_lane = qs + rq * kPmCT + c;
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            if (!owns(t)) continue;
            pm_contract(p.like + lane + (size_t)tile(t) * nkqp * 64, qs_lane, nkqp, gr[t][0], gr[t][1]);
        }
#pragma unroll
        for (int t = 0; t < TI; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) gr[t][ct][r] = 0.5 * v[t][ct][r] * gr[t][ct][r];
        publish([&](int t, int ct, int r) { return gr[t][ct][r]; });
    };

    if (p.init_only) {
        // Start's likelihood call (TSimpleMCMC.H:258) in this kernel's association
        if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
            publish([&](int t, int ct, int r) { return x[t][ct][r]; });
            quadform_terms(x);
            if (summer) {
                double usum = 0.0;
                ordered_walk(D, [&](int, double v) { usum += v; });
                if (active) p.lane_f64[SMCMC_LANE_LOGL * NP + mychain] = -usum;
            }
        }
        return;
    }

    for (int s = 0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);
        uint32_t uword = 0;

        // ForceStep (TSimpleMCMC.H:671-678): the proposal is the forced point, the proposal state is not updated
        constexpr bool forced_now = (VARIANT == PM_FORCED);   // the engine launches that instantiation for the one step
        if (summer && forced_now) {
            const uint32_t gid = p.chain_offset + (uint32_t)mychain;
            const smcmc_u32x4 ablk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            uword = smcmc_select_word(ablk, aw & 3u);
        }

        // ---- A: UpdateState, scalar half (TSimpleMCMC.H:1723-1776), one lane per chain ----
        if (summer && !forced_now) {
            ++trials;
            const double x0 = x0s[lane];
            const bool moved = (logl != last_value) || (x0 != last_x0);
            if (moved) ++succ;
            acc_rate *= acc_trials;
            if (moved) acc_rate = acc_rate + 1.0;
            acc_rate /= acc_trials + 1.0;
            acc_trials = dmin(p.acc_window, acc_trials + 1.0);
            if (rigid < 500.0 && rigid > 0.0) {
                if (__builtin_fabs(acc_rate - p.target) < p.asig) {
                    rigid += 0.5 * rigid / p.acc_window;
                    rigid = dmin(200.0, rigid);
                }
                if (__builtin_fabs(acc_rate - p.target) > 4.0 * p.asig) {
                    rigid -= 1.618 * 0.5 * rigid / p.acc_window;
                    rigid = dmax(2.0, rigid);
                }
            }
            if (rigid > 0 && rigid < 100.0) {
                sigma *= smcmc_pow_small(acc_rate / p.target, dmin(1.0 / 500.0, 1.0 / (rigid * p.acc_window)));
            }
            if (p.per_lane_update && moved && (--next_update) < 1) {
                double up = 0.5 * succ;
                next_update = (int)(p.acc_window + p.max_up - p.max_up / (up + 1.0));
                if (p.acc_w >= 0.0) {
                    acc_trials = dmax(1.0, p.acc_w * acc_trials);
                    acc_trials = dmin(acc_trials, p.acc_wW);
                }
            }
            last_value = logl;
            last_x0 = x0;
            sig[lane] = sigma;
            const uint32_t gid = p.chain_offset + (uint32_t)mychain;
            const smcmc_u32x4 ablk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            uword = smcmc_select_word(ablk, aw & 3u);
        }
        __syncthreads();   // sigma published; the previous step's readers of qs are done

        // ---- B1: z = sigma r (TSimpleMCMC.H:719-722): one Philox block = four rows of one chain ----
        if constexpr (!forced_now) {
            const int ntask = nkq * kPmCT;
            for (int task = tid; task < ntask; task += kPmW * kWave) {
                const int b = task / kPmCT, ch = task - b * kPmCT;
                const uint32_t gid = p.chain_offset + (uint32_t)(base + ch);
                const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, (uint32_t)b, SMCMC_STREAM_STEP);
                double n[4];
                smcmc_normal_pair(blk.v[0], blk.v[1], &n[0], &n[1]);
                smcmc_normal_pair(blk.v[2], blk.v[3], &n[2], &n[3]);
                const double sg = sig[ch];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * b + q < D) qs[(4 * b + q) * kPmCT + ch] = sg * n[q];
            }
        }
        __syncthreads();

        // ---- B2: x' = x + U^T z on the matrix pipe, rows i ascending ----
        if constexpr (forced_now) {
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = comp(t, r);
                        xp[t][ct][r] = (owns(t) && i < D) ? p.forced[(size_t)i * NP + base + 16 * ct + c] : 0.0;
                    }
        } else {
            const double* uop = p.Uperm + lane;
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                xp[t][0] = x[t][0];
                xp[t][1] = x[t][1];
                if (!owns(t)) continue;
                const int jt = tile(t);
                const int kend = (4 * jt + 4 < nkq) ? 4 * jt + 4 : nkq;   // U(i, j) = 0 for i > j
                const double* ut = uop + (size_t)jt * nkqp * 64;
                for (int kq = 0; kq < kend; ++kq) {
                    const double a = ut[(size_t)kq * 64];
                    const double b0 = qs[(4 * kq + rq) * kPmCT + c];
                    const double b1 = qs[(4 * kq + rq) * kPmCT + 16 + c];
                    xp[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, xp[t][0], 0, 0, 0);
                    xp[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, xp[t][1], 0, 0, 0);
                }
            }
        }

        if constexpr (VARIANT != PM_PLAIN) {
            // GetProposed() (TSimpleMCMC.H:514)
            if (p.proposed != nullptr && s + 1 == p.nsteps) {
#pragma unroll
                for (int t = 0; t < TI; ++t)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = comp(t, r), chain = base + 16 * ct + c;
                            if (owns(t) && i < D && chain < p.nchains) p.proposed[(size_t)i * NP + chain] = xp[t][ct][r];
                        }
            }
        }

        // ---- StepRMS (TSimpleMCMC.H:391-406): sqr = fma(d, d, sqr), d = x' - x, dimension order ----
        double sqr = 0.0;
        if (p.step_rms_window > 0) {
            publish([&](int t, int ct, int r) { return xp[t][ct][r] - x[t][ct][r]; });
            if (summer) ordered_walk(D, [&](int, double d) { sqr = SMCMC_FMA(d, d, sqr); });
        }
        // ---- likelihood of the proposal (:410), dimension order ----
        publish([&](int t, int ct, int r) { return xp[t][ct][r]; });
        double prop0 = 0.0;
        if (summer) prop0 = qs[lane];   // component 0 of the proposal
        if constexpr (LIKE == SMCMC_LIKE_QUADFORM) quadform_terms(xp);
        if (summer) {
            double lsum = 0.0;
            if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
                double usum = 0.0;
                ordered_walk(D, [&](int, double v) { usum += v; });
                lsum = -usum;
            } else if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
                ordered_walk(D, [&](int, double pi) { lsum = SMCMC_FMA(-0.5 * pi, pi, lsum); });
            } else {
                double prev = 0.0;
                ordered_walk(D, [&](int i, double nx) {
                    if (i > 0) {
                        const double a = 1.0 - prev;
                        const double b = SMCMC_FMA(-prev, prev, nx);
                        const double tt = SMCMC_FMA(rb * b, b, a * a);
                        lsum -= tt;
                    }
                    prev = nx;
                });
            }
            if (p.step_rms_window > 0) {
                double ms = step_rms * step_rms;
                ms *= rms_trials;
                ms += sqr;
                ms /= rms_trials + 1.0;
                rms_trials = (p.step_rms_window < rms_trials + 1) ? p.step_rms_window : rms_trials + 1;
                step_rms = __builtin_sqrt(ms);
            }
            // ---- Metropolis test (:410-463) ----
            logl_prop = lsum;
            bool take;
            if (p.metropolis == 2) {
                take = true;
            } else if (!__builtin_isfinite(logl_prop) || logl_prop < -0.999999E+30) {
                take = false;
            } else {
                const double delta = logl_prop - logl;
                take = true;
                if (delta < 0.0) {
                    if (p.metropolis == 1) take = false;
                    else {
                        const double trial = smcmc_log_pos(smcmc_u01(uword));
                        if (delta < trial) take = false;
                    }
                }
            }
            take = take && active;
            verdict[lane] = take ? 1 : 0;
            last_accept = take ? 1 : 0;
            if (take) {
                ++naccept;
                logl = logl_prop;
                x0s[lane] = prop0;
            }
        }
        __syncthreads();
        // ---- accept copy (:484-491) ----
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const bool take = verdict[16 * ct + c] != 0;
            const int chain = base + 16 * ct + c;
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
                        // the accepted point was given up for the row sums: HBM holds it
                        const int i = comp(t, r);
                        const bool mine = owns(t) && i < D;
                        if (take) {
                            if (mine && chain < p.nchains) __builtin_nontemporal_store(xp[t][ct][r], &p.x[(size_t)i * NP + chain]);
                            x[t][ct][r] = xp[t][ct][r];
                        } else {
                            x[t][ct][r] = mine ? __builtin_nontemporal_load(&p.x[(size_t)i * NP + chain]) : 0.0;
                        }
                    } else {
                        x[t][ct][r] = take ? xp[t][ct][r] : x[t][ct][r];
                    }
                }
        }
        if (p.save_x != nullptr && ((s + 1) % p.save_stride) == 0) {
            const size_t sl = (size_t)((s + 1) / p.save_stride - 1);
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = comp(t, r), chain = base + 16 * ct + c;
                        // streaming stores: left to allocate in L2 they evict the U operands every step (+170 us per
                        // step at D = 500 for 131 MB that take 22 us to write on their own)
                        if (owns(t) && i < D && chain < p.nchains)
                            __builtin_nontemporal_store(x[t][ct][r], &p.save_x[(sl * (size_t)D + (size_t)i) * NP + chain]);
                    }
            if (active) p.save_logl[sl * NP + mychain] = logl;
        }
    }


#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = comp(t, r), chain = base + 16 * ct + c;
                if (owns(t) && i < D && chain < p.nchains) __builtin_nontemporal_store(x[t][ct][r], &p.x[(size_t)i * NP + chain]);
            }
    if (active) {
        double* lf = p.lane_f64 + mychain;
        int32_t* li = p.lane_i32 + mychain;
        lf[SMCMC_LANE_LOGL * NP] = logl;
        lf[SMCMC_LANE_SIGMA * NP] = sigma;
        lf[SMCMC_LANE_ACCEPTANCE * NP] = acc_rate;
        lf[SMCMC_LANE_ACCEPTANCE_TRIALS * NP] = acc_trials;
        lf[SMCMC_LANE_RIGIDITY * NP] = rigid;
        lf[SMCMC_LANE_LAST_VALUE * NP] = last_value;
        lf[SMCMC_LANE_LAST_X0 * NP] = last_x0;
        lf[SMCMC_LANE_STEP_RMS * NP] = step_rms;
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = logl_prop;
        li[SMCMC_LANE_TRIALS * NP] = trials;
        li[SMCMC_LANE_SUCCESSES * NP] = succ;
        li[SMCMC_LANE_NEXT_UPDATE * NP] = next_update;
        li[SMCMC_LANE_NACCEPT * NP] = naccept;
        li[SMCMC_LANE_STEP_RMS_TRIALS * NP] = rms_trials;
        li[SMCMC_LANE_LAST_ACCEPT * NP] = last_accept;
    }
}

hipError_t launch_panel_mfma(const PanelParams& p, int like, hipStream_t stream);

}  // namespace smcmc
