// smcmc_fold_kernel.hip.h -- pooled second moments for the large-dimension path.
//
// For D > 63 the (D+1)(D+2)/2 accumulators of a 64-chain group no longer fit a
// wavefront's registers, so the fold of the current point into the moment sums
// (the batch form of the running covariance, reference TSimpleMCMC.H:1795-1820)
// runs as its own kernel between step launches: one wavefront per (4 x 4 block of
// 16x16 output tiles, chain slice), y = x - c0 from the [dim][chain] state (staged
// through LDS in full cache lines), chains of the slice folded in ascending order by
// chains of v_mfma_f64_16x16x4_f64.
// The accumulators persist in HBM across folds; the slice sums are added in slice
// order by fold_reduce_kernel.  oracle/ensemble_oracle.c mirrors this order with
// moment groups of `slice_chains` chains.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kFoldSlices = 32;   // most chain slices (moment groups) of the large-dimension path
// Slices actually used: as many as give every SIMD of the 256 CUs at most one wavefront (blocks x slices <=
// 1024), a multiple of the workgroup's wavefronts, so that a fold is one round of workgroups with no tail.
inline int fold_slices(int D) {
    const int T = (D + 1 + 15) / 16, TB = (T + 3) / 4, nblocks = TB * (TB + 1) / 2;
    int n = 4 * (256 / nblocks);
    if (n > kFoldSlices) n = kFoldSlices;
    if (n < 4) n = 4;
    return n;
}
constexpr int kFoldBT = 4;        // a wavefront folds a block of kFoldBT x kFoldBT tiles (64 x 64 moments)

// grid = (nblocks, nslices / 4), block = 256.  block -> (bi, bj <= bi) of 4 x 4 tiles: eight operand
// tiles feed sixteen matrix instructions.  The state is read in full cache lines (lane -> row
// lane >> 2, four consecutive chains), one stage of 16 chains ahead of its use, and re-laid out
// through LDS into the operand layout (row lane & 15, chain 4 n + (lane >> 4)).
// Four wavefronts per workgroup (four slices of the same block) and more than half of a CU's LDS per
// workgroup: one workgroup per CU, one wavefront per SIMD -- single-wavefront workgroups stack up on a
// few CUs instead and leave the rest idle.
constexpr int kFoldWaves = 4;
// mask (optional, [npad]): a chain with mask 0 folds nothing this time (the HMC engine: a step whose proposal had a
// non-finite potential skips UpdateCovariance, TSimpleHMC.H:336)
static __global__ void __launch_bounds__(kFoldWaves* kWave) fold_moments_kernel(const double* __restrict__ x, const double* __restrict__ c0,
                                                             int nchains, int npad, int D, int slice_chains,
                                                             double* __restrict__ gacc, const int32_t* __restrict__ mask) {
    constexpr int kS = 18;   // doubles per staged row: 16 chains + 2 (operand reads spread over the banks)
    __shared__ __attribute__((aligned(16))) double st_all[kFoldWaves][2 * kFoldBT][16][kS];
    __shared__ double occupancy_pad[1100];   // pushes the workgroup past 80 KB of LDS: one workgroup per CU
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x / kWave;
    if (threadIdx.x == 0 && D < 0) occupancy_pad[0] = 0.0;   // keeps the array (never true)
    double (*st)[16][kS] = st_all[wv];
    const int slice = blockIdx.y * kFoldWaves + wv;
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
    const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
    const bool diagonal = (bi == bj);
    const size_t NP = (size_t)npad;
    // staging role of this lane: row (lane >> 2) of every operand tile, chains 4 (lane & 3) .. + 3
    const int srow = lane >> 2, sq = lane & 3;
    int rr[2 * kFoldBT];
    double cc[2 * kFoldBT];
#pragma unroll
    for (int op = 0; op < 2 * kFoldBT; ++op) {
        const int tile = (op < kFoldBT) ? kFoldBT * bi + op : kFoldBT * bj + (op - kFoldBT);
        rr[op] = 16 * tile + srow;
        cc[op] = (rr[op] < D) ? c0[rr[op]] : 0.0;
    }
    // tile (ti, tj) of the block: valid when it exists and lies in the lower triangle
    auto valid = [&](int a, int b) { return kFoldBT * bi + a < T && kFoldBT * bj + b <= kFoldBT * bi + a; };
    auto offset = [&](int a, int b) {
        const int ti = kFoldBT * bi + a, tj = kFoldBT * bj + b;
        return (((size_t)slice * ntiles + (size_t)(ti * (ti + 1) / 2 + tj)) * 4) * kWave + lane;
    };
    f64x4 acc[kFoldBT][kFoldBT];
#pragma unroll
    for (int a = 0; a < kFoldBT; ++a)
#pragma unroll
        for (int b = 0; b < kFoldBT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = valid(a, b) ? gacc[offset(a, b) + (size_t)r * kWave] : 0.0;
    const int c_begin = slice * slice_chains;
    const int c_end = (c_begin + slice_chains < npad) ? c_begin + slice_chains : npad;
    const int nops = diagonal ? kFoldBT : 2 * kFoldBT;   // a diagonal block's column operands are its row operands

    typedef f64x2 stage_t[2 * kFoldBT][2];
    stage_t stA, stB;   // two stages of 16 chains in flight ahead of the matrix instructions
    auto fetch = [&](int c, stage_t& stage) {   // y = x - c0 (the constant 1 in row D, 0 above, 0 for chains past the ensemble)
        const int chain = c + 4 * sq;
        bool on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) on[k] = (chain + k < nchains) && (mask == nullptr || mask[chain + k] != 0);
#pragma unroll
        for (int op = 0; op < 2 * kFoldBT; ++op) {
            if (op >= nops) continue;
            f64x2 v0 = {0.0, 0.0}, v1 = {0.0, 0.0};
            if (rr[op] < D) {
                const f64x2* src = (const f64x2*)(x + (size_t)rr[op] * NP + chain);
                v0 = src[0];
                v1 = src[1];
                v0[0] -= cc[op]; v0[1] -= cc[op]; v1[0] -= cc[op]; v1[1] -= cc[op];
            } else if (rr[op] == D) {
                v0[0] = v0[1] = v1[0] = v1[1] = 1.0;
            }
            if (!on[0]) v0[0] = 0.0;
            if (!on[1]) v0[1] = 0.0;
            if (!on[2]) v1[0] = 0.0;
            if (!on[3]) v1[1] = 0.0;
            stage[op][0] = v0;
            stage[op][1] = v1;
        }
    };
    // one stage: registers -> LDS, refill the registers two stages ahead, fold the 16 chains
    // Every wavefront stages through its own LDS region (st_all[wv]) and the slices of a workgroup can differ in
    // length, so the hand-over is ordered within the wavefront only: LDS operations of one wavefront complete in
    // issue order, the fences keep the compiler from moving them across.
    auto wave_lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto consume = [&](int c, stage_t& stage) {
        wave_lds_sync();   // the previous stage has been consumed
#pragma unroll
        for (int op = 0; op < 2 * kFoldBT; ++op) {
            if (op >= nops) continue;
            *(f64x2*)&st[op][srow][4 * sq] = stage[op][0];
            *(f64x2*)&st[op][srow][4 * sq + 2] = stage[op][1];
        }
        wave_lds_sync();
        if (c + 32 < c_end) fetch(c + 32, stage);   // in flight under the matrix instructions
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double av[kFoldBT], bv[kFoldBT];
#pragma unroll
            for (int a = 0; a < kFoldBT; ++a) {
                av[a] = st[a][lane & 15][4 * n + (lane >> 4)];
                bv[a] = diagonal ? av[a] : st[kFoldBT + a][lane & 15][4 * n + (lane >> 4)];
            }
#pragma unroll
            for (int a = 0; a < kFoldBT; ++a)
#pragma unroll
                for (int b = 0; b < kFoldBT; ++b)
                    if (valid(a, b)) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    };
    if (c_begin < c_end) fetch(c_begin, stA);
    if (c_begin + 16 < c_end) fetch(c_begin + 16, stB);
    for (int c = c_begin; c < c_end; c += 32) {
        consume(c, stA);
        if (c + 16 < c_end) consume(c + 16, stB);
    }
#pragma unroll
    for (int a = 0; a < kFoldBT; ++a)
#pragma unroll
        for (int b = 0; b < kFoldBT; ++b)
            if (valid(a, b)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gacc[offset(a, b) + (size_t)r * kWave] = acc[a][b][r];
            }
}

// packed element k = (i, j), j <= i <= D  ->  sum over slices (ascending) of its tile entry
static __global__ void fold_reduce_kernel(const double* __restrict__ gacc, int ntiles, int nslices, int D,
                                   double* __restrict__ moments) {
    const int npk = (D + 1) * (D + 2) / 2;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npk) return;
    int i = (int)((__builtin_sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= k) ++i;
    while (i * (i + 1) / 2 > k) --i;
    const int j = k - i * (i + 1) / 2;
    const int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
    const int reg = ii >> 2, lane = jj + 16 * (ii & 3);   // C/D layout: column = lane & 15, row = (lane >> 4) + 4*reg
    const int tile = ti * (ti + 1) / 2 + tj;
    double s = 0.0;
    for (int sl = 0; sl < nslices; ++sl) s += gacc[(((size_t)sl * ntiles + tile) * 4 + reg) * kWave + lane];
    moments[k] = s;
}

hipError_t launch_fold(const double* x, const double* c0, int nchains, int npad, int D, int slice_chains, int nslices,
                       double* gacc, hipStream_t stream, const int32_t* mask = nullptr);
hipError_t launch_fold_reduce(const double* gacc, int D, int nslices, double* moments, hipStream_t stream);

}  // namespace smcmc
