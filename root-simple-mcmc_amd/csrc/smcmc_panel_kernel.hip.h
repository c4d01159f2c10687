// smcmc_panel_kernel.hip.h -- the Metropolis step for dimensions that do not fit one
// wavefront's registers (63 < D <= W * CW).
//
// A workgroup of W wavefronts advances one 64-chain group; lane l of every
// wavefront is chain l of the group.  Wavefront w owns the proposal columns
// j = jl * W + w (interleaved, so the triangular work of x' = x + sigma U^T r,
// reference TSimpleMCMC.H:709-724, is balanced) and keeps them in registers.
//   - the D normals of a step are generated once, in panels of KP rows, each
//     wavefront making its share, and handed to the others through LDS
//   - U is stored per owner, Uperm[w][i][jl] = U(i, jl*W + w); every wavefront copies
//     its slice of the current row panel into LDS and reads it back with broadcast
//     128-bit reads inside a rolled loop over the rows i
//   - the finished proposal is passed through LDS in panels of kGatherJl local
//     columns so that wavefront 0 can run StepRMS, the likelihood and the
//     Metropolis test in the reference's exact summation order
//     (TSimpleMCMC.H:391-406, 410-463); the verdict goes back through LDS and every
//     wavefront commits its own columns (:484-491)
//   - the scalar half of UpdateState (TSimpleMCMC.H:1723-1776) is computed
//     redundantly by every wavefront (it needs sigma), stored by wavefront 0
// The accepted point x[dim][chain] stays in HBM and makes the round trip the
// algorithmic-bytes model of SURVEY.md section 8(d) assumes (16 D + 16 bytes per
// chain-step).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kPanelRows = 32;    // rows of normals per LDS panel
constexpr int kGatherJl = 16;     // local columns per gather panel
constexpr int kPanelCW = 64;      // proposal columns a wavefront keeps in registers

struct PanelParams {
    int nchains, npad, dim;
    int nsteps, metropolis;
    uint32_t step0, chain_offset;
    uint64_t seed;
    const double* Uperm;      // [W][dim][CW]
    const double* like;       // ROSENBROCK: {b}; QUADFORM in reference order: Error^T [dim][dim] (like[i * dim + j] = Error(j, i))
    QuadCsr like_csr;         // ... and its non-zero entries when it is sparse (quadform_csr), rowptr == nullptr otherwise
    double target, acc_window, asig, max_up, acc_w, acc_wW;
    int per_lane_update, step_rms_window, full_u;
    double* x;                // [dim][npad]
    double* lane_f64;
    int32_t* lane_i32;
    double* save_x;           // optional [slot][dim][npad]
    double* save_logl;
    int save_stride;
    int init_only;            // matrix-pipe kernel: only evaluate log L at x into LANE_LOGL (Start, :258)
    int has_forced;           // ForceStep pending (TSimpleMCMC.H:671-678): the first step proposes `forced`
    const double* forced;     // [dim][npad]
    // reference-order kernel only (the engine refuses them in the fused order)
    int special;              // uniform dimensions or a scan are present: the SPECIAL instantiation runs
    const double* uniform;    // [2][dim] lower and upper bounds, then eight 64-bit words: bit j = dimension j is uniform
    int scan_dim;             // fScanDimension (TSimpleMCMC.H:685-704), -1 = off
    int scan_uniform;         // the scanned dimension has a uniform proposal
    double scan_a, scan_b;    // uniform: bounds; Gaussian: centre, sigma
    double* scratch;          // QUADFORM in reference order: [dim][npad], the proposal as the serial likelihood sum reads it
    double* proposed;         // optional [dim][npad]: the proposal of the launch's last step (fProposed, TSimpleMCMC.H:576);
                              // looked at by the SPECIAL instantiation and by the KEEP / FORCED ones of the matrix-pipe kernel
};

template <int W>
__device__ __forceinline__ size_t step_pitch(size_t np) {
    if constexpr (W == 4) asm volatile("" : "+s"(np));
    return np;
}

// The quadratic form of TDummyLogLikelihood.H:21-31 in the reference's order, logL -= 0.5*p[i]*Error(j,i)*p[j] with i
// outer and j inner, is ONE running sum of D^2 terms per chain: nothing inside a chain can be done in parallel, so one
// lane walks it for its chain, row i of Error^T coming through scalar loads.  Two multiplies and the dependent
// subtraction per term: at D = 500 that is 750 000 FP64 instructions per chain-step, more than the proposal takes --
// this is the parity anchor for the likelihood BASELINE config 4 names, not the fast path (the matrix-pipe kernel in
// the fused order is).  `pc` is the lane's column of a [dim][pitch] image of the point: in the step kernel the LDS
// image of 32 chains (the 64 chains of a group do not fit: two passes), at Start the state itself.
// The loads of the point run one group of kQfGroup terms ahead of the arithmetic (two register sets): a term costs two
// independent multiplies and one dependent subtraction, the memory latency of a group is hidden behind the previous one.
constexpr int kQfGroup = 32;
constexpr int kQfLds = 16;   // terms per group of the LDS-fed sum of the step kernel
template <bool EXACT = true, typename PointPtr>
__device__ __forceinline__ double quadform_serial(PointPtr pc, size_t NP, cptr_f64 et, int D) {
    double logl = 0.0;
    const int ngroups = D / kQfGroup;
    for (int i = 0; i < D; ++i) {
        const double h = 0.5 * pc[(size_t)i * NP];
        const cptr_f64 erow = et + (size_t)i * D;
        double cur[kQfGroup], nxt[kQfGroup];
        if (ngroups > 0) {
#pragma unroll
            for (int u = 0; u < kQfGroup; ++u) cur[u] = pc[(size_t)u * NP];
        }
        for (int g = 0; g < ngroups; ++g) {
            const int j0 = g * kQfGroup;
            if (g + 1 < ngroups) {
#pragma unroll
                for (int u = 0; u < kQfGroup; ++u) nxt[u] = pc[(size_t)(j0 + kQfGroup + u) * NP];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kQfGroup; ++u) {
                if constexpr (EXACT) logl -= h * erow[j0 + u] * cur[u];
                else logl = SMCMC_FMA(-(h * erow[j0 + u]), cur[u], logl);   // the fused order of loglike<DP, QUADFORM, false>
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kQfGroup; ++u) cur[u] = nxt[u];
        }
        for (int j = ngroups * kQfGroup; j < D; ++j) {
            if constexpr (EXACT) logl -= h * erow[j] * pc[(size_t)j * NP];
            else logl = SMCMC_FMA(-(h * erow[j]), pc[(size_t)j * NP], logl);
        }
    }
    return logl;
}

// one chain's point in a [dim][npad] image: what a user likelihood for dim > 63 indexes (smcmc_user_loglike_at)
struct ChainColumn {
    const double* base;   // &image[chain]
    size_t pitch;         // npad
    __device__ __forceinline__ double operator[](int i) const { return base[(size_t)i * pitch]; }
};

// likelihoods the large-dimension step kernel evaluates from the proposal's image in device memory (one lane per chain)
template <int LIKE>
constexpr bool kLikeFromImage = (LIKE == SMCMC_LIKE_QUADFORM || LIKE == SMCMC_LIKE_USER || LIKE == SMCMC_LIKE_CONSTRAINED);

template <int LIKE, bool EXACT>
__device__ __forceinline__ double serial_loglike(const double* __restrict__ x, int chain, size_t npad, int D,
                                                 const double* __restrict__ like, QuadCsr csr = QuadCsr{nullptr, nullptr, nullptr, nullptr});

// The dense D^2-term sum of the header-form TDummy for the chains of one pass (see the call in panel_step_kernel): rows
// of Error^T staged through LDS by the whole workgroup one row ahead (rbuf holds two rows, row 0 is there on entry), the
// summing lanes walk pl[j * 32].  Every thread of the workgroup calls it (barriers inside).
typedef __attribute__((address_space(3))) double* lds_ptr_f64;
template <int W, int CW>
__device__ __forceinline__ double panel_quadform_dense_body(const double* __restrict__ like, double* rbuf_generic,
                                                            const double* pl_generic, int D, bool summing) {
    constexpr int kERow = W * CW;
    const lds_ptr_f64 rbuf = (lds_ptr_f64)rbuf_generic;
    const lds_cptr_f64 pl = (lds_cptr_f64)pl_generic;
    double acc = 0.0;
    for (int i = 0; i < D; ++i) {
        // the next row: loaded before the sum (D <= W * kWave: one element per thread), stored after it
        double enext = 0.0;
        if (i + 1 < D && (int)threadIdx.x < D) enext = like[(size_t)(i + 1) * D + threadIdx.x];
        if (summing) {
            const double h = 0.5 * pl[i * 32];
            const lds_cptr_f64 er = rbuf + (i & 1) * kERow;
            // the LDS reads of a group of kQfLds terms are all issued before its arithmetic, the next group's
            // before this group's arithmetic (two register sets): the scheduler, left alone, pairs every read
            // with its use and the sum crawls at one LDS latency per two terms
            constexpr int G = kQfLds;
            const int ngr = D / G;
            double pa[G], pb[G];
            f64x2 ea[G / 2], eb[G / 2];
            auto fetch = [&](int j0, double (&pv)[G], f64x2 (&ev)[G / 2]) {
#pragma unroll
                for (int u = 0; u < G; ++u) pv[u] = pl[(j0 + u) * 32];
#pragma unroll
                for (int u = 0; u < G / 2; ++u) ev[u] = *(lds_cptr_f64x2)(er + j0 + 2 * u);
            };
            auto fold = [&](const double (&pv)[G], const f64x2 (&ev)[G / 2]) {
#pragma unroll
                for (int u = 0; u < G; ++u) acc -= h * ev[u / 2][u & 1] * pv[u];
            };
            if (ngr > 0) fetch(0, pa, ea);
            for (int g = 0; g < ngr; g += 2) {
                if (g + 1 < ngr) fetch((g + 1) * G, pb, eb);
                __builtin_amdgcn_sched_barrier(0);
                fold(pa, ea);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < ngr) fetch((g + 2) * G, pa, ea);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < ngr) fold(pb, eb);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int j = ngr * G; j < D; ++j) acc -= h * er[j] * pl[j * 32];
        }
        if (i + 1 < D && (int)threadIdx.x < D) rbuf[((i + 1) & 1) * kERow + threadIdx.x] = enext;
        __syncthreads();
    }
    return acc;
}
// Out of line in the eight-wavefront kernel: compiled into it, the two register sets of sixteen terms cost that kernel
// (256 registers per lane) 4 KB of scratch per lane whether the loop runs or not -- 3.6 -> 0.64 ms/step at D = 500 with
// the sparse walk, 11.7 -> 9.7 with this loop itself.  The four-wavefront kernel has the registers and keeps its copy of
// the loop inline in the kernel.
template <int W, int CW>
__device__ __attribute__((noinline)) double panel_quadform_dense_call(const double* __restrict__ like, double* rbuf,
                                                                      const double* pl, int D, bool summing) {
    return panel_quadform_dense_body<W, CW>(like, rbuf, pl, D, summing);
}
// SPECIAL = the instantiation that also knows uniform per-dimension proposals and the scan of one dimension
// (kept out of the common kernel: with them in, D=500 went from 0.89 to 2.85 ms/step)
template <int W, int CW, int LIKE, bool EXACT, bool SPECIAL>
__global__ void __launch_bounds__(W * kWave) panel_step_kernel(const PanelParams p) {
    __shared__ double rbuf[kPanelRows * kWave];              // normals of the current panel, [row][lane]
    // U panel of every wavefront, [w][row][jl]; after the last panel of a step the same
    // memory carries the gathered proposal gp[jl][w][lane] and trial step gd = x' - x
    __shared__ __attribute__((aligned(16))) double ulds[W * kPanelRows * CW];
    static_assert(2 * kGatherJl * kWave <= kPanelRows * CW, "gather buffers must fit the U staging area");
    constexpr int kGd = kGatherJl * W * kWave;   // offset of gd inside ulds
    __shared__ double verdict_logl[kWave];
    __shared__ int verdict_take[kWave];

    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int group = blockIdx.x;
    const int chain = group * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    const cptr_f64 likep = as_const(p.like);

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
    int trials = li[SMCMC_LANE_TRIALS * NP];
    int succ = li[SMCMC_LANE_SUCCESSES * NP];
    int next_update = li[SMCMC_LANE_NEXT_UPDATE * NP];
    int naccept = li[SMCMC_LANE_NACCEPT * NP];
    int rms_trials = li[SMCMC_LANE_STEP_RMS_TRIALS * NP];
    int last_accept = li[SMCMC_LANE_LAST_ACCEPT * NP];


    const uint32_t aw = smcmc_accept_word((uint32_t)D);
    const int npanels = (D + kPanelRows - 1) / kPanelRows;
    constexpr int ngather = (CW + kGatherJl - 1) / kGatherJl;

    double xp[CW];
    // PANEL_PROFILE (tools/micro/build_panelprof.sh): cycles per section, printed by workgroup 0 -- never a product build
#ifdef PANEL_PROFILE
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define PN_MARK(k) { const unsigned long long tn_ = __builtin_readcyclecounter(); tsec[k] += tn_ - tlast; tlast = tn_; }
#else
#define PN_MARK(k)
#endif
    for (int s = 0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);
        // W = 4: the row pitch is opaque per step, otherwise the 3 x 64 column addresses are hoisted out of the
        // step loop and live in (spilled) registers for the whole launch (D=200: 0.145 -> 0.129 ms/step).  At
        // W = 8 recomputing them costs more than the spills (D=500: 0.877 -> 0.912 ms/step): left hoisted.
        const size_t NPl = step_pitch<W>(NP);

        // ForceStep (TSimpleMCMC.H:671-678): the proposal is the forced point, the proposal state is not updated
        const bool forced_now = p.has_forced && s == 0;
        // scan of one dimension (TSimpleMCMC.H:685-704): the current point with that dimension redrawn
        const bool scan_now = SPECIAL && p.scan_dim >= 0 && !forced_now;
        const bool no_update = forced_now || scan_now;

        // ---- A: UpdateState, scalar half (TSimpleMCMC.H:1723-1776), every wavefront ----
        if (!no_update) {
        ++trials;
        const double x0 = p.x[chain];
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
        }

        // ---- B: proposal columns of this wavefront ----
        const double* xsrc = forced_now ? p.forced : p.x;
#pragma unroll
        for (int jl = 0; jl < CW; ++jl) {
            const int j = jl * W + w;
            xp[jl] = (j < D) ? xsrc[(size_t)j * (W == 4 ? NPl : NP) + chain] : 0.0;
        }
        if (no_update && w == 0) {
            smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            verdict_logl[lane] = smcmc_u01(smcmc_select_word(blk, aw & 3u));
        }
        const int ncols = (D - w + W - 1) / W;   // local columns of this wavefront that exist (j = jl * W + w < D)
        PN_MARK(0)
        for (int pn = 0; pn < (no_update ? 0 : npanels); ++pn) {
            const int i0 = pn * kPanelRows;
            __syncthreads();                       // the previous panel has been consumed
            PN_MARK(1)
            // normals of rows i0 .. i0+KP-1: Philox block b covers rows 4b..4b+3
            for (int bb = w; bb < kPanelRows / 4; bb += W) {
                const int b = i0 / 4 + bb;
                if (4 * b < D || (uint32_t)b == (aw >> 2)) {
                    smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, (uint32_t)b, SMCMC_STREAM_STEP);
                    if ((uint32_t)b == (aw >> 2)) verdict_logl[lane] = smcmc_u01(smcmc_select_word(blk, aw & 3u));
                    double n[4];
                    smcmc_normal_pair(blk.v[0], blk.v[1], &n[0], &n[1]);
                    smcmc_normal_pair(blk.v[2], blk.v[3], &n[2], &n[3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) rbuf[(4 * bb + q) * kWave + lane] = n[q];
                }
            }
            const int i1 = (i0 + kPanelRows < D) ? i0 + kPanelRows : D;
            PN_MARK(2)
            // this wavefront's slice of the panel, rows i0..i1-1 x CW columns, is contiguous in
            // Uperm: copy it into the wavefront's LDS area with 16-byte pieces
            {
                const f64x2* src = (const f64x2*)(p.Uperm + ((size_t)w * D + i0) * CW);
                f64x2* dst = (f64x2*)(ulds + w * (kPanelRows * CW));
                const int npieces = (i1 - i0) * (CW / 2);
                // eight loads in flight per lane: one at a time, each waits out an L2 round trip before its LDS store
                // (16 round trips per panel, 16 panels per step, nothing else running on the CU behind the barriers)
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
            PN_MARK(3)
            __syncthreads();
            PN_MARK(1)
            lds_cptr_f64 up = (lds_cptr_f64)(ulds + w * (kPanelRows * CW));
            asm volatile("" : "+v"(up));   // LDS addresses live in vector registers
            // (tools/micro/panelprof.py: this loop is 187 000 of the 292 000 cycles of a config-3 step, ~400 cycles per
            // 16-column piece against 128 of arithmetic.  Measured and not kept, all bit-identical (profiles/r04_notes.md
            // section 5): products first / the next piece's reads behind them / additions last: 207 000; the panel cut where
            // the first live piece changes, pieces compile-time, the next piece's reads in flight in a second register
            // set: 161 000 here but the rest of the step slower by as much (512 registers, more scratch), and the
            // eight-wavefront kernel 1.02 instead of 0.78 ms per step.)
            for (int i = i0; i < i1; ++i) {
                const double sr = sigma * rbuf[(i - i0) * kWave + lane];
                // first local column with j = jl*W + w >= i (0 for a full matrix)
                const int jl0 = p.full_u ? 0 : ((i - w + W - 1) / W);
                lds_cptr_f64 urow = up + (i - i0) * CW;
#pragma unroll
                for (int c = 0; c < CW; c += 16) {
                    if (c + 15 >= jl0 && c < ncols) {
                        // 16 columns: eight broadcast 128-bit LDS reads, kept next to their use
                        f64x2 u2[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) u2[k] = *(volatile lds_cptr_f64x2)(urow + c + 2 * k);
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const double u = u2[k / 2][k & 1];
                            if constexpr (EXACT) xp[c + k] += sr * u;
                            else xp[c + k] = SMCMC_FMA(sr, u, xp[c + k]);
                            asm volatile("" : "+v"(xp[c + k]));
                        }
                    }
                }
            }
            PN_MARK(4)
        }
        __syncthreads();   // the U staging area is reused by the gather below
        PN_MARK(1)
        // the Metropolis uniform when its word lies past the last row block
        if (!no_update && (aw >> 2) >= (uint32_t)(npanels * (kPanelRows / 4)) && w == 0) {
            smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            verdict_logl[lane] = smcmc_u01(smcmc_select_word(blk, aw & 3u));
        }

        if (SPECIAL && !forced_now) {
            // word d of the step redraws dimension d; the owner of the column places the value
            auto place = [&](int d, double val) __attribute__((always_inline)) {
                const int ujl = d / W;
#pragma unroll
                for (int jl = 0; jl < CW; ++jl) xp[jl] = (jl == ujl) ? val : xp[jl];
            };
            if (scan_now) {
                const uint32_t sd = (uint32_t)p.scan_dim;
                if ((int)(sd % W) == w) {
                    const smcmc_u32x4 sblk = smcmc_draw_block(p.seed, gid, step, sd >> 2, SMCMC_STREAM_STEP);
                    double val;
                    if (p.scan_uniform) {
                        val = p.scan_a + (p.scan_b - p.scan_a) * smcmc_u01(smcmc_select_word(sblk, sd & 3u));
                    } else {
                        double nc, ns;
                        smcmc_normal_pair(smcmc_select_word(sblk, sd & 2u), smcmc_select_word(sblk, (sd & 2u) + 1u), &nc, &ns);
                        val = p.scan_a + p.scan_b * ((sd & 1u) ? ns : nc);
                    }
                    place((int)sd, val);
                }
            } else {
                // uniform dimensions (TSimpleMCMC.H:711-716): their rows and columns of the device copy
                // of U are zero, so nothing else touched them
                const uint64_t* mask = (const uint64_t*)(p.uniform + 2 * (size_t)D);
#pragma nounroll
                for (int wd = 0; wd < (D + 63) / 64; ++wd) {
                    uint64_t m = mask[wd];
#pragma nounroll
                    while (m != 0) {
                        const uint32_t ud = (uint32_t)(wd * 64 + __builtin_ctzll(m));
                        m &= m - 1;
                        if ((int)(ud % W) != w) continue;
                        const smcmc_u32x4 ublk = smcmc_draw_block(p.seed, gid, step, ud >> 2, SMCMC_STREAM_STEP);
                        const double lo = p.uniform[ud], hi = p.uniform[D + ud];
                        place((int)ud, lo + (hi - lo) * smcmc_u01(smcmc_select_word(ublk, ud & 3u)));
                    }
                }
            }
        }

        if constexpr (SPECIAL) {
            // GetProposed() (TSimpleMCMC.H:514): every wavefront stores its columns of the latest proposal
            if (p.proposed != nullptr && s + 1 == p.nsteps && active) {
#pragma unroll
                for (int jl = 0; jl < CW; ++jl) {
                    const int j = jl * W + w;
                    if (j < D) p.proposed[(size_t)j * NP + chain] = xp[jl];
                }
            }
        }

        if constexpr (kLikeFromImage<LIKE>) {
            // the whole proposal where one lane per chain can read it
#pragma unroll
            for (int jl = 0; jl < CW; ++jl) {
                const int j = jl * W + w;
                if (j < D) p.scratch[(size_t)j * NP + chain] = xp[jl];
            }
        }

        // ---- gather: wavefront 0 walks the proposal in dimension order ----
        double sqr = 0.0, lsum = 0.0, prev_p = 0.0;
        bool outside = false;   // HORRIFIC: a coordinate left the unit box
        const double rb = (LIKE == SMCMC_LIKE_ROSENBROCK) ? likep[0] : 0.0;
        // xp[] must be indexed statically to stay in registers: the panel number selects one of
        // `ngather` fully unrolled write blocks; the (long) serial walk below is emitted once
        // the accepted point's values for a gather round are loaded one round ahead: the second read of the state came in a
        // burst from every CU at once (a tenth of the step), now it is in flight while wavefront 0 walks the previous round
        double xv[kGatherJl];
        auto load_x = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int q = 0; q < kGatherJl; ++q) {
                const int jl = g * kGatherJl + q;
                if (jl < CW) {
                    const int j = jl * W + w;
                    xv[q] = (j < D) ? p.x[(size_t)j * (W == 4 ? NPl : NP) + chain] : 0.0;
                }
            }
        };
        auto write_panel = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int q = 0; q < kGatherJl; ++q) {
                const int jl = g * kGatherJl + q;
                if (jl < CW) {
                    ulds[(q * W + w) * kWave + lane] = xp[jl];
                    ulds[kGd + (q * W + w) * kWave + lane] = xp[jl] - xv[q];
                }
            }
        };
        load_x(std::integral_constant<int, 0>{});
        for (int g = 0; g < ngather && g * kGatherJl * W < D; ++g) {
            __syncthreads();
            static_for<ngather>([&](auto gc) {
                if (g == decltype(gc)::value) {
                    write_panel(gc);
                    if constexpr (decltype(gc)::value + 1 < ngather) {
                        if ((g + 1) * kGatherJl * W < D) load_x(std::integral_constant<int, decltype(gc)::value + 1>{});
                    }
                }
            });
            __syncthreads();
            if (w == 0) {
                for (int q = 0; q < kGatherJl; ++q) {
                    // the W values of this local column are read before the (serial) sums consume them
                    double pjs[W], djs[W];
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) {
                        pjs[ww] = ulds[(q * W + ww) * kWave + lane];
                        djs[ww] = ulds[kGd + (q * W + ww) * kWave + lane];
                    }
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) {
                        const int j = (g * kGatherJl + q) * W + ww;
                        if (j < D) {
                            const double pj = pjs[ww];
                            const double dj = djs[ww];
                            if constexpr (EXACT) sqr += dj * dj;                     // :393-396
                            else sqr = SMCMC_FMA(dj, dj, sqr);
                            if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
                                const double t = -0.5 * pj;
                                if constexpr (EXACT) lsum += t * pj;
                                else lsum = SMCMC_FMA(t, pj, lsum);
                            } else if constexpr (kLikeFromImage<LIKE>) {
                                (void)pj;   // summed below, from the global image
                            } else if constexpr (LIKE == SMCMC_LIKE_ASYM) {
                                const double a = (pj < 0.0) ? pj * likep[1] : pj * likep[0];   // TAsymLogLikelihood.H:24-28
                                lsum += a;
                            } else if constexpr (LIKE == SMCMC_LIKE_HORRIFIC) {
                                outside = outside || (__builtin_fabs(pj) > 1.0);               // THorrificLogLikelihood.H:30-33
                                lsum += pj;
                            } else {
                                // term i = j-1 of THardLogLikelihood.H:60-64 needs p[j-1] and p[j]
                                if (j > 0) {
                                    if constexpr (EXACT) {
                                        const double a = (1.0 - prev_p);
                                        const double b = pj - prev_p * prev_p;
                                        lsum -= a * a + rb * b * b;
                                    } else {
                                        const double a = 1.0 - prev_p;
                                        const double b = SMCMC_FMA(-prev_p, prev_p, pj);
                                        const double t = SMCMC_FMA(rb * b, b, a * a);
                                        lsum -= t;
                                    }
                                }
                                prev_p = pj;
                            }
                        }
                    }
                }
            }
        }

        if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
            // the proposal of 32 chains at a time as pl[j][32] in the (now idle) staging area of U: 256-byte rows, a
            // conflict-free ds_read_b64 per term; lanes 32 * pass .. + 31 of wavefront 0 sum their chains
            // Row i of Error^T (wave-uniform) is staged through LDS by all the threads of the workgroup, one row ahead
            // of the summing lanes (rbuf, free at this point, holds two rows): scalar loads would stall the sum on a
            // scalar-cache miss every eight terms.
            static_assert(W * kPanelRows * CW >= W * CW * 32, "the LDS image of 32 chains must fit the U staging area");
            static_assert(kPanelRows * kWave >= 2 * W * CW, "two rows of Error must fit the normals' buffer");
            for (int pass = 0; pass < 2; ++pass) {
                __syncthreads();
                if ((lane >> 5) == pass) {
                    for (int j = w; j < D; j += W) ulds[j * 32 + (lane & 31)] = p.scratch[(size_t)j * NP + chain];
                }
                for (int k = threadIdx.x; k < D; k += W * kWave) rbuf[k] = p.like[k];
                __syncthreads();
                const bool summing = (w == 0) && ((lane >> 5) == pass);
                const double* pl = ulds + (lane & 31);
                double acc = 0.0;
                // a sparse Error: the summing lanes walk its non-zero entries (quadform_csr: bit for bit the dense sum
                // unless a coordinate is not finite -- then the whole workgroup takes the dense loop below)
                bool sparse_done = false;
                if (p.like_csr.rowptr != nullptr) {
                    if (summing) acc = quadform_csr<EXACT>([&](int j) { return pl[j * 32]; }, p.like_csr, D);
                    sparse_done = __syncthreads_or(summing && !__builtin_isfinite(acc)) == 0;
                    if (!sparse_done) acc = 0.0;
                }
                if constexpr (W >= 8) {
                    // out of line in the eight-wavefront kernel (panel_quadform_dense_call)
                    if (!sparse_done) acc = panel_quadform_dense_call<W, CW>(p.like, rbuf, ulds + (lane & 31), D, summing);
                } else {
                    constexpr int kERow = W * CW;   // doubles per staged row (>= D)
                    for (int i = 0; i < (sparse_done ? 0 : D); ++i) {
                        // the next row: loaded before the sum (D <= W * kWave: one element per thread), stored after it
                        double enext = 0.0;
                        if (i + 1 < D && (int)threadIdx.x < D) enext = p.like[(size_t)(i + 1) * D + threadIdx.x];
                        if (summing) {
                            const double h = 0.5 * pl[i * 32];
                            const double* er = rbuf + (i & 1) * kERow;
                            // the LDS reads of a group of kQfLds terms are all issued before its arithmetic, the next group's
                            // before this group's arithmetic (two register sets): the scheduler, left alone, pairs every read
                            // with its use and the sum crawls at one LDS latency per two terms
                            constexpr int G = kQfLds;
                            const int ngr = D / G;
                            double pa[G], pb[G];
                            f64x2 ea[G / 2], eb[G / 2];
                            auto fetch = [&](int j0, double (&pv)[G], f64x2 (&ev)[G / 2]) {
#pragma unroll
                                for (int u = 0; u < G; ++u) pv[u] = pl[(j0 + u) * 32];
#pragma unroll
                                for (int u = 0; u < G / 2; ++u) ev[u] = *(const f64x2*)(er + j0 + 2 * u);
                            };
                            auto fold = [&](const double (&pv)[G], const f64x2 (&ev)[G / 2]) {
#pragma unroll
                                for (int u = 0; u < G; ++u) acc -= h * ev[u / 2][u & 1] * pv[u];
                            };
                            if (ngr > 0) fetch(0, pa, ea);
                            for (int g = 0; g < ngr; g += 2) {
                                if (g + 1 < ngr) fetch((g + 1) * G, pb, eb);
                                __builtin_amdgcn_sched_barrier(0);
                                fold(pa, ea);
                                __builtin_amdgcn_sched_barrier(0);
                                if (g + 2 < ngr) fetch((g + 2) * G, pa, ea);
                                __builtin_amdgcn_sched_barrier(0);
                                if (g + 1 < ngr) fold(pb, eb);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            for (int j = ngr * G; j < D; ++j) acc -= h * er[j] * pl[j * 32];
                        }
                        if (i + 1 < D && (int)threadIdx.x < D) rbuf[((i + 1) & 1) * kERow + threadIdx.x] = enext;
                        __syncthreads();
                    }
                }
                if (summing) lsum = acc;
            }
        }

        if constexpr (LIKE == SMCMC_LIKE_USER || LIKE == SMCMC_LIKE_CONSTRAINED) {
            // one lane per chain walks the proposal's image (complete and visible since the gather's barriers) as the
            // reference's functor walks its vector
            if (w == 0) lsum = serial_loglike<LIKE, EXACT>(p.scratch, chain, NP, D, p.like, p.like_csr);
        }

        if constexpr (LIKE == SMCMC_LIKE_HORRIFIC) {
            const double sigma = 0.01;                                                         // :27, 34-37
            lsum /= __builtin_sqrt(D * 4.0 / 12.0);
            lsum = -0.5 * lsum * lsum / sigma / sigma;
            lsum = outside ? -1E+30 : lsum;
        }

        PN_MARK(5)
        // ---- wavefront 0: StepRMS, Metropolis test (TSimpleMCMC.H:397-463) ----
        if (w == 0) {
            if (p.step_rms_window > 0) {
                double ms = step_rms * step_rms;
                ms *= rms_trials;
                ms += sqr;
                ms /= rms_trials + 1.0;
                rms_trials = (p.step_rms_window < rms_trials + 1) ? p.step_rms_window : rms_trials + 1;
                step_rms = __builtin_sqrt(ms);
            }
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
                        const double trial = smcmc_log_pos(verdict_logl[lane]);
                        if (delta < trial) take = false;
                    }
                }
            }
            take = take && active;
            verdict_take[lane] = take ? 1 : 0;
            verdict_logl[lane] = take ? logl_prop : logl;
        }
        __syncthreads();
        const bool take = verdict_take[lane] != 0;
        const double new_logl = verdict_logl[lane];
        last_accept = take ? 1 : 0;
        if (take) ++naccept;
        logl = new_logl;
        // QUADFORM: the proposal went to its global image before the likelihood and is taken from there, so that the
        // registers that held it are free during the serial sum
        auto proposal = [&](int jl, int j) {
            if constexpr (kLikeFromImage<LIKE>) return p.scratch[(size_t)j * NP + chain];
            else return xp[jl];
        };
        if (take) {
#pragma unroll
            for (int jl = 0; jl < CW; ++jl) {
                const int j = jl * W + w;
                if (j < D) p.x[(size_t)j * (W == 4 ? NPl : NP) + chain] = proposal(jl, j);
            }
        }
        if (p.save_x != nullptr && ((s + 1) % p.save_stride) == 0 && active) {
            const size_t slot = (size_t)((s + 1) / p.save_stride - 1);
#pragma unroll
            for (int jl = 0; jl < CW; ++jl) {
                const int j = jl * W + w;
                if (j < D)
                    // (streaming store: a save that allocates in L2 evicts U and the state the next step reads)
                    __builtin_nontemporal_store(take ? proposal(jl, j) : p.x[(size_t)j * (W == 4 ? NPl : NP) + chain],
                                                &p.save_x[(slot * (size_t)D + (size_t)j) * (W == 4 ? NPl : NP) + chain]);
            }
            if (w == 0) p.save_logl[slot * (W == 4 ? NPl : NP) + chain] = logl;
        }
        PN_MARK(6)
        __syncthreads();                           // x is final before the next step reads x[0] / its columns
        PN_MARK(7)
    }
#ifdef PANEL_PROFILE
    if (group == 0 && lane == 0 && p.nsteps >= 8)
        printf("panel_step_kernel W=%d D=%d wavefront %d, cycles per step: update+x load %llu | barriers %llu | normals %llu | U copy %llu | "
               "rows %llu | gather+likelihood %llu | verdict+commit %llu | last barrier %llu\n", W, D, w,
               tsec[0] / p.nsteps, tsec[1] / p.nsteps, tsec[2] / p.nsteps, tsec[3] / p.nsteps, tsec[4] / p.nsteps,
               tsec[5] / p.nsteps, tsec[6] / p.nsteps, tsec[7] / p.nsteps);
#endif

    if (active && w == 0) {
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

// log L of one chain's column of x ([dim][npad]) in the reference's summation order (EXACT) or the fused order of
// loglike<DP, LIKE, false>; QUADFORM reads Error^T as a plain [D][D] matrix.
template <int LIKE, bool EXACT>
__device__ __forceinline__ double serial_loglike(const double* __restrict__ x, int chain, size_t npad, int D,
                                                 const double* __restrict__ like, QuadCsr csr) {
    double lsum = 0.0;
    if constexpr (LIKE == SMCMC_LIKE_USER) {
#ifdef SMCMC_USER_LIKELIHOOD_ANY_DIM
        lsum = smcmc_user_loglike_at(ChainColumn{x + chain, npad}, like, D);
#endif
        return lsum;
    }
    if constexpr (LIKE == SMCMC_LIKE_CONSTRAINED) {
        // example4/TConstrainedLikelihood.H:26-46; like = {SummedValues, SummedConstraint, Expected[D], Prior[D]}
        double sum = 0.0;
        for (int i = 0; i < D; ++i) sum += x[(size_t)i * npad + chain];
        sum = (sum - like[0]) / like[1];
        lsum -= 0.5 * sum * sum;
        for (int i = 0; i < D; ++i) {
            double v = x[(size_t)i * npad + chain] - like[2 + i];
            v /= like[2 + D + i];
            lsum -= 0.5 * v * v;
        }
        return lsum;
    }
    if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
        for (int i = 0; i < D; ++i) {
            const double pi = x[(size_t)i * npad + chain];
            const double t = -0.5 * pi;
            if constexpr (EXACT) lsum += t * pi;
            else lsum = SMCMC_FMA(t, pi, lsum);
        }
    } else if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
        bool dense = csr.rowptr == nullptr;
        if (!dense) {
            const double* pc = x + chain;
            lsum = quadform_csr<EXACT>([&](int j) { return pc[(size_t)j * npad]; }, csr, D);
            dense = __any(!__builtin_isfinite(lsum));
        }
        if (dense) lsum = quadform_serial<EXACT>(x + chain, npad, as_const(like), D);
    } else if constexpr (LIKE == SMCMC_LIKE_ASYM) {
        for (int i = 0; i < D; ++i) {
            const double pi = x[(size_t)i * npad + chain];
            lsum += (pi < 0.0) ? pi * like[1] : pi * like[0];
        }
    } else if constexpr (LIKE == SMCMC_LIKE_HORRIFIC) {
        bool outside = false;
        for (int i = 0; i < D; ++i) {
            const double pi = x[(size_t)i * npad + chain];
            outside = outside || (__builtin_fabs(pi) > 1.0);
            lsum += pi;
        }
        const double sigma = 0.01;
        lsum /= __builtin_sqrt(D * 4.0 / 12.0);
        lsum = -0.5 * lsum * lsum / sigma / sigma;
        lsum = outside ? -1E+30 : lsum;
    } else {
        const double rb = like[0];
        double prev = x[chain];
        for (int i = 0; i < D - 1; ++i) {
            const double nx = x[(size_t)(i + 1) * npad + chain];
            if constexpr (EXACT) {
                const double a = (1.0 - prev);
                const double b = nx - prev * prev;
                lsum -= a * a + rb * b * b;
            } else {
                const double a = 1.0 - prev;
                const double b = SMCMC_FMA(-prev, prev, nx);
                const double t = SMCMC_FMA(rb * b, b, a * a);
                lsum -= t;
            }
            prev = nx;
        }
    }
    return lsum;
}

// Start's likelihood call (TSimpleMCMC.H:258) for the large-dimension path: one thread
// per chain walks its column of x in the reference's summation order.
template <int LIKE, bool EXACT>
__global__ void start_loglike_kernel(const double* __restrict__ x, int nchains, size_t npad, int D,
                                     const double* __restrict__ like, double* __restrict__ logl_out) {
    const int chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= nchains) return;
    static_assert(EXACT || LIKE != SMCMC_LIKE_QUADFORM, "the fused order of the quadratic form is the matrix-pipe kernel's");
    logl_out[chain] = serial_loglike<LIKE, EXACT>(x, chain, npad, D, like);
}

template <int W, int CW>
hipError_t launch_panel(const PanelParams& p, int like, bool exact, hipStream_t stream);
hipError_t launch_start_loglike(const double* x, int nchains, size_t npad, int D, const double* like_params,
                                double* logl_out, int like, bool exact, hipStream_t stream);
// a user likelihood at these dimensions (smcmc_user_large.hip, user builds only)
template <int W, int CW>
hipError_t launch_panel_user(const PanelParams& p, hipStream_t stream);
hipError_t launch_start_loglike_user(const double* x, int nchains, size_t npad, int D, const double* like_params,
                                     double* logl_out, hipStream_t stream);

}  // namespace smcmc
