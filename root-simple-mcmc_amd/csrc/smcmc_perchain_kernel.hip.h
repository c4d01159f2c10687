// smcmc_perchain_kernel.hip.h -- SMCMC_MODE_PER_CHAIN: every lane is a complete reference chain, its own running
// centre, covariance and decomposition included.
//
// This is the reference's default and only mode (one chain = one TProposeAdaptiveStep): UpdateState every step
// (TSimpleMCMC.H:1721-1831) with the running covariance of :1795-1820 -- multiply, add, DIVIDE, j <= i -- and
// UpdateProposal (:1009-1390) whenever the chain's own --fNextUpdate < 1 on an accepted step (:1824-1826).  It is the
// one configuration of the engine whose traffic per chain-step really is O(D^2): a chain's packed covariance and
// decomposition (D (D + 1) / 2 doubles each, 10.2 KB at D = 50) do not fit registers or LDS for 64 chains, so they
// live in HBM as [k][chain] columns -- a wavefront's access to element k of its 64 chains is one coalesced 512-byte
// line run -- and stream through every step: the covariance is read and written, the decomposition read:
//     algorithmic bytes per chain-step = 8 * 3 * D (D + 1) / 2 + 8 * (5 D + ...)        (30.6 KB + 2 KB at D = 50)
// The kernel is built around those three streams: blocks of kPcBlock elements, the next block's loads in flight while
// the current one is consumed; everything else of a chain (point, proposal, likelihood) is O(D) and goes through
// [dim][chain] images in device memory that stay in L2.  Loops are rolled and the dimension is a run-time value.
//
// UpdateProposal inside a launch: the scalar half (trace, sigma rescale, schedule, de-weighting: :1024-1086) per lane;
// the Cholesky decomposition (:1097-1106) by the whole wavefront for one chain at a time -- the chain's covariance is
// gathered into LDS, lane = column, the host's row-ordered A = U^T U in the host's order of roundings -- and the
// factor scattered back to the chain's column.  A pivot that fails stops THAT chain (status kPcNeedsLadder, the step
// it is in recorded); the host runs the fallback ladder (:1134-1389) for it and the next launch lets it catch up:
// chains are independent and every draw is keyed on (chain, step), so a chain that runs a few steps late runs the
// same steps.
//
// Reference-order arithmetic only (compile with -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_kernels.hip.h"
#include "smcmc_panel_kernel.hip.h"   // serial_loglike

namespace smcmc {

constexpr int kPcChunk = 8;           // elements of a stream per chunk (one address register, immediate offsets)
constexpr int kPcDepth = 5;           // chunk buffers in rotation: the loads of kPcDepth - 1 chunks (32 per lane) are in
                                      // flight behind the chunk being consumed.  The streams are read in whole chunks
                                      // and primed unconditionally: both images carry kPcPad elements of padding behind
                                      // the last tile (whose first 64 doubles also take the stores of idle lanes).
constexpr int kPcPad = kPcDepth * kPcChunk + kPcChunk * (kPcChunk + 1) / 2;
constexpr int kPcBatch = 8;           // loads issued together in the O(D) loops over the [dim][chain] images
constexpr int kPcMaxDim = 63;         // one lane per column in the wavefront's Cholesky
constexpr int kPcRPitch = 65;         // row pitch of the LDS matrix of the Cholesky

// SMCMC_LANE_UPDATE_STATUS
enum { kPcOk = 0, kPcNeedsLadder = 1, kPcResume = 2, kPcInvalidTrace = 3 };

struct PerChainParams {
    int nchains, npad, dim;
    int metropolis;
    uint32_t step0;            // fTotalSteps of every chain before this call
    uint32_t target_step;      // every chain runs until its own fTotalSteps reaches this
    uint32_t chain_offset;
    uint64_t seed;
    const double* like;        // QUADFORM: Error^T [dim][dim]; ROSENBROCK {b}; ...
    QuadCsr like_csr;          // QUADFORM with a sparse Error: its non-zero entries (quadform_csr), rowptr == nullptr otherwise
    double target, acc_window, asig, max_up;
    double acc_w, acc_wW;      // acceptance de-weighting: w = 1 - deweight, w * window; acc_w < 0 = off
    double cov_w, cov_wW;      // the same for the covariance / centre trials (:1056-1067)
    double cov_window;
    int cov_frozen;            // SetCovarianceFrozen (:937): the covariance loop is skipped, the centre still runs
    int step_rms_window;
    int has_forced;            // ForceStep pending: step step0 + 1 proposes `forced`
    const double* forced;      // [dim][npad]
    int update_only;           // no steps: UpdateProposal() of every chain (the explicit call of SimpleMCMC.C:254)
    double* x;                 // [dim][npad] accepted point
    double* proposed;          // [dim][npad] the proposal (fProposed), also the image the likelihood walks
    double* last_point;        // [dim][npad] fLastPoint
    double* centre;            // [dim][npad] fCentralPoint
    double* cov;               // fCurrentCov, lower triangle, row major: k = i (i + 1) / 2 + j, j <= i; wavefront tiles
                               // [npad / 64][dim (dim + 1) / 2][64] (pc_tile_index)
    double* ut;                // fDecomposition, wavefront tiles [npad / 64][dim * dim][64]: kk = j (j + 1) / 2 + i holds U(i, j), i <= j (column
                               // packed: a proposal column walks it contiguously); a full matrix (eigen rung of the
                               // ladder, SMCMC_LANE_DECOMP_FULL) keeps U(i, j), j < i, at dim (dim + 1) / 2 + i (i - 1) / 2 + j
    double* lane_f64;
    int32_t* lane_i32;
    double* save_x;            // optional [slot][dim][npad]
    double* save_logl;
    int save_stride;
    int* flag_count;           // += 1 for every chain that stopped for the host
};

// Layout of the two O(D^2) images: wavefront tiles [chain / 64][k][chain % 64].  A wavefront's element k is one
// 512-byte line and its whole stream one contiguous run of `rows` such lines: sequential pages for the address
// translation and open rows for HBM (a [k][chain] image over all chains puts consecutive k of a wavefront
// npad * 8 bytes apart -- a new page for every load).
__host__ __device__ inline size_t pc_tile_index(int k, size_t chain, int rows) {
    return ((chain >> 6) * (size_t)rows + (size_t)k) * kWave + (chain & (kWave - 1));
}

// (row, column) of packed index k = i (i + 1) / 2 + j, j <= i
__device__ __forceinline__ void pc_unpack(int k, int& i, int& j) {
    i = (int)((__builtin_sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= k) ++i;
    while (i * (i + 1) / 2 > k) --i;
    j = k - i * (i + 1) / 2;
}

// TDummyLogLikelihood.H:24-28 for the point held in a column of LDS (v[j * kWave]): logL -= 0.5 p[i] Error(j, i) p[j], i outer,
// j inner, un-fused -- the arithmetic of quadform_serial (smcmc_panel_kernel.hip.h), which walks the [dim][chain] image
// in device memory instead: D^2 loads per chain-step that a lone wavefront has nothing to hide behind.
__device__ __forceinline__ double pc_quadform_lds(const double* v, cptr_f64 et, int D) {
    double logl = 0.0;
    for (int i = 0; i < D; ++i) {
        const double h = 0.5 * v[i * kWave];
        const cptr_f64 erow = et + (size_t)i * D;
        int j = 0;
        for (; j + kPcBatch <= D; j += kPcBatch) {
            double pj[kPcBatch];
#pragma unroll
            for (int u = 0; u < kPcBatch; ++u) pj[u] = v[(j + u) * kWave];
#pragma unroll
            for (int u = 0; u < kPcBatch; ++u) logl -= h * erow[j + u] * pj[u];
        }
        for (; j < D; ++j) logl -= h * erow[j] * v[j * kWave];
    }
    return logl;
}

template <int LIKE>
__global__ void __launch_bounds__(kWave, 1) perchain_step_kernel(const PerChainParams p) {
    extern __shared__ double lds[];   // vec[i * 64 + lane] (the diffs x - c, then sigma * r), or R[64][65] of the Cholesky
    __shared__ int s_ok;

    const int lane = threadIdx.x;
    const int group = blockIdx.x;
    const int chain = group * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const int npk = D * (D + 1) / 2;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    double* const vec = lds + lane;
    double* const gcov = p.cov + (size_t)group * npk * kWave;        // the wavefront's tiles (pc_tile_index)
    double* const gut = p.ut + (size_t)group * D * D * kWave;
    double* const ccov = gcov + lane;
    double* const cut = gut + lane;
    double* const gut_pad = p.ut + (size_t)D * D * NP + lane;       // the padding behind the last tile

    double* lf = p.lane_f64 + chain;
    int32_t* li = p.lane_i32 + chain;
    double logl = lf[SMCMC_LANE_LOGL * NP];
    double sigma = lf[SMCMC_LANE_SIGMA * NP];
    double acc_rate = lf[SMCMC_LANE_ACCEPTANCE * NP];
    double acc_trials = lf[SMCMC_LANE_ACCEPTANCE_TRIALS * NP];
    double rigid = lf[SMCMC_LANE_RIGIDITY * NP];
    double last_value = lf[SMCMC_LANE_LAST_VALUE * NP];
    double last_x0 = lf[SMCMC_LANE_LAST_X0 * NP];
    double step_rms = lf[SMCMC_LANE_STEP_RMS * NP];
    double logl_prop = lf[SMCMC_LANE_LOGL_PROPOSED * NP];
    double centre_trials = lf[SMCMC_LANE_CENTER_TRIALS * NP];
    double cov_trials = lf[SMCMC_LANE_COVARIANCE_TRIALS * NP];
    double sigma_trace = lf[SMCMC_LANE_SIGMA_TRACE * NP];
    int trials = li[SMCMC_LANE_TRIALS * NP];
    int succ = li[SMCMC_LANE_SUCCESSES * NP];
    int next_update = li[SMCMC_LANE_NEXT_UPDATE * NP];
    int naccept = li[SMCMC_LANE_NACCEPT * NP];
    int rms_trials = li[SMCMC_LANE_STEP_RMS_TRIALS * NP];
    int last_accept = li[SMCMC_LANE_LAST_ACCEPT * NP];
    int status = li[SMCMC_LANE_UPDATE_STATUS * NP];
    int ufull = li[SMCMC_LANE_DECOMP_FULL * NP];
    uint32_t tstep = (uint32_t)li[SMCMC_LANE_CHAIN_STEPS * NP];
    int update_count = li[SMCMC_LANE_UPDATE_COUNT * NP];
    int last_path = li[SMCMC_LANE_LAST_UPDATE_PATH * NP];

    bool resume = active && status == kPcResume;   // the host finished this chain's UpdateProposal: the step goes on behind it
    if (resume) status = kPcOk;
    const uint32_t aw = smcmc_accept_word((uint32_t)D);
    const int NB = (D + 3) / 4;

    // UpdateProposal (TSimpleMCMC.H:1009-1106) for the lanes of `want`; `trace` is the lane's covariance trace.
    // Returns with status / ufull / last_path of those lanes set; a lane whose decomposition failed has stopped.
    auto update_proposal = [&](bool want, double trace) {
        if (want) {
            ++update_count;
            if (!(trace > 0)) {                                        // :1025-1028 (the reference throws)
                status = kPcInvalidTrace;
                want = false;
            } else {
                const double scale = __builtin_sqrt(sigma_trace / trace);
                sigma = sigma * scale;                                 // :1042
                sigma_trace = trace;                                   // :1043
                const double up = 0.5 * succ;                          // :1051
                next_update = (int)(p.acc_window + p.max_up - p.max_up / (up + 1.0));   // :1052
                if (p.cov_w >= 0.0) {                                  // :1056-1067
                    cov_trials = dmax(1.0, p.cov_w * cov_trials);
                    cov_trials = dmin(cov_trials, p.cov_wW);
                    centre_trials = dmax(1.0, p.cov_w * centre_trials);
                    centre_trials = dmin(centre_trials, p.cov_wW);
                }
                if (p.acc_w >= 0.0) {                                  // :1081-1086
                    acc_trials = dmax(1.0, p.acc_w * acc_trials);
                    acc_trials = dmin(acc_trials, p.acc_wW);
                }
            }
        }
        // the decomposition, one chain at a time, the whole wavefront on it (lane = column)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");             // every lane's covariance stores are visible
        uint64_t todo = __ballot(want);
        while (todo != 0) {
            const int L = __builtin_ctzll(todo);
            todo &= todo - 1;
            __syncthreads();
            for (int k = lane; k < npk; k += kWave) {
                int i, j;
                pc_unpack(k, i, j);
                const double v = gcov[(size_t)k * kWave + L];
                lds[i * kPcRPitch + j] = v;
                lds[j * kPcRPitch + i] = v;
            }
            if (lane == 0) s_ok = 1;
            __syncthreads();
            // SharedProposal::cholesky (smcmc_proposal.hpp): row c of U from the rows above it
            const int jj = lane;
            for (int c = 0; c < D; ++c) {
                double v = 0.0;
                if (jj >= c && jj < D) {
                    v = lds[c * kPcRPitch + jj];
                    for (int rr = 0; rr < c; ++rr) v -= lds[rr * kPcRPitch + jj] * lds[rr * kPcRPitch + c];
                    if (jj == c) {
                        if (!(v > 0.0) || !__builtin_isfinite(v)) s_ok = 0;
                        else lds[c * kPcRPitch + c] = __builtin_sqrt(v);
                    }
                }
                __syncthreads();
                if (!s_ok) break;
                if (jj > c && jj < D) lds[c * kPcRPitch + jj] = v / lds[c * kPcRPitch + c];
                __syncthreads();
            }
            const bool ok = s_ok != 0;
            if (ok) {
                for (int kk = lane; kk < npk; kk += kWave) {
                    int j, i;
                    pc_unpack(kk, j, i);                               // kk = j (j + 1) / 2 + i, i <= j
                    gut[(size_t)kk * kWave + L] = lds[i * kPcRPitch + j];
                }
            }
            if (lane == L) {
                if (ok) { ufull = 0; last_path = 0; }
                else status = kPcNeedsLadder;                          // the host's ladder takes over (:1134-1389)
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");             // the owners read what the other lanes wrote
    };

    // trace of the lane's covariance, summed in index order (GetCovarianceTrace :961-967)
    auto load_trace = [&]() {
        double t = 0.0;
        for (int i = 0; i < D; ++i) t += ccov[(size_t)(i * (i + 1) / 2 + i) * kWave];
        return t;
    };

    if (p.update_only) {
        update_proposal(active, load_trace());
    } else {
        // a resumed chain still owes the rest of the step its UpdateProposal interrupted -- also when that step was the
        // last one of its launch (tstep already stands at the target then; the loop's end test stops it afterwards)
        bool live = active && status == kPcOk && (resume || tstep < p.target_step);
        while (__any(live)) {
            if (live && !resume) ++tstep;                               // ++fTotalSteps, :376
            const uint64_t step = (uint64_t)tstep;
            const bool forced_now = p.has_forced && tstep == p.step0 + 1u;
            const bool upd = live && !resume && !forced_now;            // UpdateState runs (:706)
            bool moved = false;
            double x0 = p.x[chain];

            if (__any(upd)) {
                // ---- UpdateState, scalar half (TSimpleMCMC.H:1723-1776) ----
                if (upd) {
                    ++trials;
                    moved = (logl != last_value) || (x0 != last_x0);
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
                }
                // ---- running centre (:1780-1788); the diffs x - c stay in LDS for the covariance ----
                __syncthreads();
                for (int i0 = 0; i0 < D; i0 += kPcBatch) {
                    double xv[kPcBatch], cv[kPcBatch];
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q) {
                        const int i = (i0 + q < D) ? i0 + q : D - 1;
                        xv[q] = p.x[(size_t)i * NP + chain];
                        cv[q] = p.centre[(size_t)i * NP + chain];
                    }
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q) {
                        if (i0 + q < D) {
                            double c = cv[q];
                            c *= centre_trials;
                            c += xv[q];
                            c /= centre_trials + 1;
                            if (upd) p.centre[(size_t)(i0 + q) * NP + chain] = c;
                            vec[(i0 + q) * kWave] = xv[q] - c;
                        }
                    }
                }
                if (upd) centre_trials = dmin(p.cov_window, centre_trials + 1.0);
                // ---- running covariance (:1795-1820): the stream ----
                double trace = 0.0;
                if (!p.cov_frozen) {
                    // Chunks of kPcChunk consecutive elements through kPcDepth rotating buffers.  The steady loop has
                    // no conditional memory operation (a lane outside `upd` stores back what it loaded): the waits
                    // the compiler counts out are then exact, and the loads of kPcDepth - 1 chunks stay in flight
                    // behind the chunk whose divisions -- independent chains -- are being worked off.  The
                    // (row, column) of every element is wavefront-uniform bookkeeping.
                    constexpr int CH = kPcChunk;
                    double buf[kPcDepth][CH];
                    const double tv = cov_trials, tv1 = cov_trials + 1.0;
                    const int nch = (npk + CH - 1) / CH;
                    auto fetch = [&](double (&b)[CH], int c) {
                        const double* src = ccov + (size_t)c * (CH * kWave);
#pragma unroll
                        for (int q = 0; q < CH; ++q) b[q] = __builtin_nontemporal_load(src + q * kWave);
                    };
                    int ci = 0, cj = 0;
                    auto consume = [&](double (&b)[CH], int c, auto whole) {
                        constexpr bool WHOLE = decltype(whole)::value;      // every element of the chunk exists
                        const int k0 = c * CH;
                        double* dst = ccov + (size_t)k0 * kWave;
                        double da[CH], db[CH], v[CH];
                        bool dg[CH];
#pragma unroll
                        for (int q = 0; q < CH; ++q) {
                            da[q] = vec[(ci < D ? ci : 0) * kWave];
                            db[q] = vec[(cj < D ? cj : 0) * kWave];
                            dg[q] = cj == ci;
                            cj = dg[q] ? 0 : cj + 1;
                            ci = dg[q] ? ci + 1 : ci;
                        }
#pragma unroll
                        for (int q = 0; q < CH; ++q) {
                            double t = b[q];
                            const double r = da[q] * db[q];
                            t *= tv;
                            t += r;
                            t /= tv1;
                            v[q] = t;
                        }
#pragma unroll
                        for (int q = 0; q < CH; ++q) {
                            if (WHOLE || k0 + q < npk) {
                                __builtin_nontemporal_store(upd ? v[q] : b[q], dst + q * kWave);
                                trace = dg[q] ? trace + v[q] : trace;
                            }
                        }
                    };
#pragma unroll
                    for (int sl = 0; sl < kPcDepth; ++sl) fetch(buf[sl], sl);
                    int c0 = 0;
                    for (; c0 + 2 * kPcDepth <= nch; c0 += kPcDepth) {
#pragma unroll
                        for (int sl = 0; sl < kPcDepth; ++sl) {
                            consume(buf[sl], c0 + sl, std::true_type{});
                            fetch(buf[sl], c0 + sl + kPcDepth);
                        }
                    }
                    for (; c0 < nch; c0 += kPcDepth) {
#pragma unroll
                        for (int sl = 0; sl < kPcDepth; ++sl) {
                            const int c = c0 + sl;
                            if (c < nch) {
                                consume(buf[sl], c, std::false_type{});
                                if (c + kPcDepth < nch) fetch(buf[sl], c + kPcDepth);
                            }
                        }
                    }
                    if (upd) cov_trials = dmin(p.cov_window, cov_trials + 1.0);
                }
                // ---- UpdateProposal when the chain's own schedule says so (:1824-1826) ----
                bool trigger = false;
                if (upd && moved) trigger = (--next_update) < 1;
                if (__any(trigger)) {
                    if (p.cov_frozen) trace = load_trace();
                    update_proposal(trigger, trace);
                    if (status != kPcOk) live = false;                 // this chain waits for the host
                }
            }
            if (live && !forced_now) {                                 // :1829-1830
                last_value = logl;
                last_x0 = x0;
                for (int i0 = 0; i0 < D; i0 += kPcBatch) {
                    double xv[kPcBatch];
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q) xv[q] = p.x[(size_t)((i0 + q < D) ? i0 + q : D - 1) * NP + chain];
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q)
                        if (i0 + q < D) p.last_point[(size_t)(i0 + q) * NP + chain] = xv[q];
                }
            }
            resume = false;

            // ---- the proposal (:709-724) into its image ----
            uint32_t uword = 0;
            // the trial step's square sum (:391-396) and -- ISO_GAUSS -- the likelihood are taken at the column ends of
            // the proposal stream, in the reference's index order, instead of in passes over the images: only when
            // no lane of the wavefront has a full decomposition or a forced step
            const bool fused = !__any(ufull != 0) && !__any(forced_now);
            double sqr = 0.0, lsum = 0.0;
            if (forced_now) {
                for (int i = 0; i < D; ++i)
                    if (live) p.proposed[(size_t)i * NP + chain] = p.forced[(size_t)i * NP + chain];
                const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
                uword = smcmc_select_word(blk, aw & 3u);
            } else {
                __syncthreads();
                for (int b = 0; b < NB; ++b) {
                    const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, (uint32_t)b, SMCMC_STREAM_STEP);
                    if ((uint32_t)b == (aw >> 2)) uword = smcmc_select_word(blk, aw & 3u);
                    double n0, n1, n2, n3;
                    smcmc_normal_pair(blk.v[0], blk.v[1], &n0, &n1);
                    smcmc_normal_pair(blk.v[2], blk.v[3], &n2, &n3);
                    vec[(4 * b) * kWave] = sigma * n0;
                    if (4 * b + 1 < D) vec[(4 * b + 1) * kWave] = sigma * n1;
                    if (4 * b + 2 < D) vec[(4 * b + 2) * kWave] = sigma * n2;
                    if (4 * b + 3 < D) vec[(4 * b + 3) * kWave] = sigma * n3;
                }
                if ((aw >> 2) >= (uint32_t)NB) {
                    const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
                    uword = smcmc_select_word(blk, aw & 3u);
                }
                // column j: x'[j] = x[j] + sum_{i <= j} (sigma r_i) U(i, j), i ascending, un-fused; the columns lie one
                // behind the other in the stream.
                double acc = x0, xcur = x0;
                auto close_column = [&](double xnext) {                // everything of a column's end but the store
                    if (fused) {
                        const double ts = acc - xcur;
                        sqr += ts * ts;
                        if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
                            const double th = -0.5 * acc;
                            lsum += th * acc;
                        }
                    }
                    acc = xnext;
                    xcur = xnext;
                };
                // a lane that is not running leaves its proposal alone: its stores go to the padding of the image
                double* const pdst = live ? p.proposed + chain : gut_pad;
                const size_t pstride = live ? NP : 0;
                // the first columns (shorter than a chunk): their 36 elements and start values loaded up front
                constexpr int CH = kPcChunk;
                constexpr int PRE = CH * (CH + 1) / 2;
                const int P = D < CH ? D : CH;
                const int kP = P * (P + 1) / 2;
                {
                    double pre[PRE], xn[CH];
#pragma unroll
                    for (int k = 0; k < PRE; ++k) pre[k] = __builtin_nontemporal_load(cut + k * kWave);
#pragma unroll
                    for (int j = 0; j < CH; ++j) xn[j] = p.x[(size_t)(j + 1 < D ? j + 1 : D - 1) * NP + chain];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        if (j < P) {
#pragma unroll
                            for (int i = 0; i <= j; ++i) acc += vec[i * kWave] * pre[j * (j + 1) / 2 + i];
                            pdst[(size_t)j * pstride] = acc;
                            close_column(xn[j]);
                        }
                    }
                }
                // the rest in chunks through rotating buffers, as for the covariance.  Columns are now longer than a
                // chunk: a chunk closes at most the column it starts in, and its fetch brings the start value of the
                // column behind that one along.  No conditional memory operation in the steady loop: every chunk
                // stores ONE value to the proposal -- the finished x'[j] of the column it closed, or the running sum
                // of the column still open, which the chunk that closes it overwrites.
                double buf[kPcDepth][CH], xs[kPcDepth];
                const int nch = (npk - kP + CH - 1) / CH;
                int lrow = 0, lcol = P;                                // loader: (row, column) of a chunk's first element
                auto fetch = [&](double (&b)[CH], double& xn, int c) {
                    const double* src = cut + (size_t)(kP + c * CH) * kWave;
#pragma unroll
                    for (int q = 0; q < CH; ++q) b[q] = __builtin_nontemporal_load(src + q * kWave);
                    xn = p.x[(size_t)(lcol + 1 < D ? lcol + 1 : D - 1) * NP + chain];
                    lrow += CH;
                    if (lrow > lcol) {
                        lrow -= lcol + 1;
                        ++lcol;
                    }
                };
                int ci = 0, cj = P;
                auto consume = [&](double (&b)[CH], double xn, int c, auto whole) {
                    constexpr bool WHOLE = decltype(whole)::value;
                    const int k0 = kP + c * CH;
                    double sr[CH];
                    bool end[CH];
                    const int cj0 = cj < D ? cj : D - 1;
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        sr[q] = vec[(ci < D ? ci : 0) * kWave];
                        end[q] = ci == cj;
                        ci = end[q] ? 0 : ci + 1;
                        cj = end[q] ? cj + 1 : cj;
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) sr[q] = sr[q] * b[q];
                    double done = 0.0;
                    bool closed = false;
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        if (WHOLE || k0 + q < npk) {
                            acc += sr[q];
                            if (end[q]) {
                                done = acc;
                                closed = true;
                                close_column(xn);
                            }
                        }
                    }
                    pdst[(size_t)cj0 * pstride] = closed ? done : acc;
                };
#pragma unroll
                for (int sl = 0; sl < kPcDepth; ++sl) fetch(buf[sl], xs[sl], sl);
                int c0 = 0;
                for (; c0 + 2 * kPcDepth <= nch; c0 += kPcDepth) {
#pragma unroll
                    for (int sl = 0; sl < kPcDepth; ++sl) {
                        consume(buf[sl], xs[sl], c0 + sl, std::true_type{});
                        fetch(buf[sl], xs[sl], c0 + sl + kPcDepth);
                    }
                }
                for (; c0 < nch; c0 += kPcDepth) {
#pragma unroll
                    for (int sl = 0; sl < kPcDepth; ++sl) {
                        const int c = c0 + sl;
                        if (c < nch) {
                            consume(buf[sl], xs[sl], c, std::false_type{});
                            if (c + kPcDepth < nch) fetch(buf[sl], xs[sl], c + kPcDepth);
                        }
                    }
                }
                if (__any(ufull != 0)) {
                    // a full decomposition (the eigen rung of the ladder): the rows below the diagonal, which every
                    // x'[j] sees after its upper part (i ascending)
                    for (int i2 = 1; i2 < D; ++i2) {
                        const double sr = vec[i2 * kWave];
                        for (int j2 = 0; j2 < i2; ++j2) {
                            const double u = cut[(size_t)(npk + i2 * (i2 - 1) / 2 + j2) * kWave];
                            double v = p.proposed[(size_t)j2 * NP + chain];
                            v += sr * u;
                            if (live && ufull) p.proposed[(size_t)j2 * NP + chain] = v;
                        }
                    }
                }
            }

            // ---- StepRMS window (:391-406), likelihood (:410), Metropolis test (:432-463), accept copy (:484-491) ----
            if (p.step_rms_window > 0) {
                if (!fused) {
                    sqr = 0.0;
                    for (int i0 = 0; i0 < D; i0 += kPcBatch) {
                        double pv[kPcBatch], xv[kPcBatch];
#pragma unroll
                        for (int q = 0; q < kPcBatch; ++q) {
                            const int i = (i0 + q < D) ? i0 + q : D - 1;
                            pv[q] = p.proposed[(size_t)i * NP + chain];
                            xv[q] = p.x[(size_t)i * NP + chain];
                        }
#pragma unroll
                        for (int q = 0; q < kPcBatch; ++q) {
                            if (i0 + q < D) {
                                const double t = pv[q] - xv[q];
                                sqr += t * t;
                            }
                        }
                    }
                }
                if (live) {
                    double ms = step_rms * step_rms;
                    ms *= rms_trials;
                    ms += sqr;
                    ms /= rms_trials + 1.0;
                    rms_trials = (p.step_rms_window < rms_trials + 1) ? p.step_rms_window : rms_trials + 1;
                    step_rms = __builtin_sqrt(ms);
                }
            }
            double lp;
            if (LIKE == SMCMC_LIKE_ISO_GAUSS && fused) {
                lp = lsum;
            } else if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
                // the D^2-term sum reads the proposal D times over: from LDS (the scaled normals are done with)
                __syncthreads();
                for (int i0 = 0; i0 < D; i0 += kPcBatch) {
                    double pv[kPcBatch];
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q) pv[q] = p.proposed[(size_t)((i0 + q < D) ? i0 + q : D - 1) * NP + chain];
#pragma unroll
                    for (int q = 0; q < kPcBatch; ++q)
                        if (i0 + q < D) vec[(i0 + q) * kWave] = pv[q];
                }
                bool dense = p.like_csr.rowptr == nullptr;
                if (!dense) {
                    lp = quadform_csr<true>([&](int j) { return vec[j * kWave]; }, p.like_csr, D);
                    dense = __any(!__builtin_isfinite(lp));
                }
                if (dense) lp = pc_quadform_lds(vec, as_const(p.like), D);
            } else {
                lp = serial_loglike<LIKE, true>(p.proposed, chain, NP, D, p.like);
            }
            if (live) {
                logl_prop = lp;
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
                last_accept = take ? 1 : 0;
                if (take) {
                    logl = logl_prop;
                    ++naccept;
                    for (int i0 = 0; i0 < D; i0 += kPcBatch) {
                        double pv[kPcBatch];
#pragma unroll
                        for (int q = 0; q < kPcBatch; ++q) pv[q] = p.proposed[(size_t)((i0 + q < D) ? i0 + q : D - 1) * NP + chain];
#pragma unroll
                        for (int q = 0; q < kPcBatch; ++q)
                            if (i0 + q < D) p.x[(size_t)(i0 + q) * NP + chain] = pv[q];
                    }
                }
                if (p.save_x != nullptr && ((tstep - p.step0) % (uint32_t)p.save_stride) == 0) {
                    const size_t slot = (size_t)((tstep - p.step0) / (uint32_t)p.save_stride - 1u);
                    for (int i = 0; i < D; ++i)
                        p.save_x[(slot * (size_t)D + (size_t)i) * NP + chain] = p.x[(size_t)i * NP + chain];
                    p.save_logl[slot * NP + chain] = logl;
                }
            }
            live = live && tstep < p.target_step;
        }
    }

    if (active) {
        if (status == kPcNeedsLadder || status == kPcInvalidTrace) atomicAdd(p.flag_count, 1);
        lf[SMCMC_LANE_LOGL * NP] = logl;
        lf[SMCMC_LANE_SIGMA * NP] = sigma;
        lf[SMCMC_LANE_ACCEPTANCE * NP] = acc_rate;
        lf[SMCMC_LANE_ACCEPTANCE_TRIALS * NP] = acc_trials;
        lf[SMCMC_LANE_RIGIDITY * NP] = rigid;
        lf[SMCMC_LANE_LAST_VALUE * NP] = last_value;
        lf[SMCMC_LANE_LAST_X0 * NP] = last_x0;
        lf[SMCMC_LANE_STEP_RMS * NP] = step_rms;
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = logl_prop;
        lf[SMCMC_LANE_CENTER_TRIALS * NP] = centre_trials;
        lf[SMCMC_LANE_COVARIANCE_TRIALS * NP] = cov_trials;
        lf[SMCMC_LANE_SIGMA_TRACE * NP] = sigma_trace;
        li[SMCMC_LANE_TRIALS * NP] = trials;
        li[SMCMC_LANE_SUCCESSES * NP] = succ;
        li[SMCMC_LANE_NEXT_UPDATE * NP] = next_update;
        li[SMCMC_LANE_NACCEPT * NP] = naccept;
        li[SMCMC_LANE_STEP_RMS_TRIALS * NP] = rms_trials;
        li[SMCMC_LANE_LAST_ACCEPT * NP] = last_accept;
        li[SMCMC_LANE_UPDATE_STATUS * NP] = status;
        li[SMCMC_LANE_DECOMP_FULL * NP] = ufull;
        li[SMCMC_LANE_CHAIN_STEPS * NP] = (int32_t)tstep;
        li[SMCMC_LANE_UPDATE_COUNT * NP] = update_count;
        li[SMCMC_LANE_LAST_UPDATE_PATH * NP] = last_path;
    }
}

// What the host computed once for every chain (InitializeState at Start, RestoreState, the template of an explicit
// ResetProposal): the covariance, the decomposition and the scalar state, handed to every chain.
struct PerChainBroadcast {
    int nchains, npad, dim;
    int reset;                    // 1: ResetProposal() of running chains (:1396-1494): the per-chain parts are kept / recomputed per lane
    const double* cov_packed;     // [dim (dim + 1) / 2] template covariance
    const double* ut;             // [dim * dim] template decomposition in the engine's layout
    const double* centre;         // [dim] (RestoreState) or nullptr: the chain's own fLastPoint / start point
    int decomp_full, last_path, update_count;
    double sigma, sigma_trace, centre_trials, cov_trials, acceptance, acceptance_trials;
    int next_update;
    // reset only
    double sigma_floor, sigma_reset, acc_w, acc_wW, cov_w, cov_wW;
    double* x; double* last_point; double* centre_out; double* cov; double* ut_out;
    double* lane_f64; int32_t* lane_i32;
};

inline size_t perchain_lds_bytes(int dim) {
    const size_t a = sizeof(double) * (size_t)dim * kWave, b = sizeof(double) * (size_t)kWave * kPcRPitch;
    return a > b ? a : b;
}

hipError_t launch_perchain(const PerChainParams& p, int like, hipStream_t stream);
hipError_t launch_perchain_broadcast(const PerChainBroadcast& p, hipStream_t stream);
hipError_t launch_perchain_broadcast_rows(double* dst, const double* values_device, int rows, int nchains, size_t npad, bool tiled,
                                          hipStream_t stream);

// One flagged chain's staging record for the host's fallback ladder (perchain_stage_kernel): the lane scalars (the
// 32-bit ones as doubles), the packed covariance, the decomposition image (scatter only), the centre, the last point.
__host__ __device__ inline size_t pc_stage_stride(int D) {
    return (size_t)SMCMC_LANE_F64_COUNT_ + SMCMC_LANE_I32_COUNT_ + (size_t)D * (D + 1) / 2 + (size_t)D * D + 2 * (size_t)D;
}
struct PerChainStage {
    const int32_t* chains;     // [nflagged] chain indices
    double* stage;             // [nflagged][pc_stage_stride(dim)]
    int npad, dim;
    double* lane_f64; int32_t* lane_i32;
    double* cov; double* ut; double* centre; double* last;
};
hipError_t launch_perchain_stage(const PerChainStage& g, int nflagged, bool scatter, hipStream_t s);

}  // namespace smcmc
