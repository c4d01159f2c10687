// smcmc_fold_kernel.hip.h -- pooled second moments for the large-dimension path.
//
// For D > 63 the (D+1)(D+2)/2 accumulators of a 64-chain group no longer fit a
// wavefront's registers, so the fold of the current point into the moment sums
// (the batch form of the running covariance, reference TSimpleMCMC.H:1795-1820)
// runs as its own kernel between step launches: one wavefront per (16x16 output
// tile, chain slice), y = x - c0 read straight from the [dim][chain] state, chains
// of the slice folded in ascending order by a chain of v_mfma_f64_16x16x4_f64.
// The accumulators persist in HBM across folds; the slice sums are added in slice
// order by fold_reduce_kernel.  oracle/ensemble_oracle.c mirrors this order with
// moment groups of `slice_chains` chains.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kFoldSlices = 32;   // chain slices (moment groups) of the large-dimension path
constexpr int kFoldBT = 4;        // a wavefront folds a block of kFoldBT x kFoldBT tiles (64 x 64 moments)

// row r of the augmented point y: dims 0..D-1, the constant 1 at r == D, zero above
__device__ __forceinline__ double fold_operand(const double* __restrict__ x, size_t NP, int D, int r, int chain,
                                               int nchains, double c0r) {
    if (chain >= nchains) return 0.0;
    if (r < D) return x[(size_t)r * NP + chain] - c0r;
    return (r == D) ? 1.0 : 0.0;
}

// grid = (nblocks, kFoldSlices), block = 64.  block -> (bi, bj <= bi) of 4 x 4 tiles: eight operand
// reads feed sixteen matrix instructions (a quarter of the state traffic of one tile per wavefront).
static __global__ void __launch_bounds__(kWave) fold_moments_kernel(const double* __restrict__ x, const double* __restrict__ c0,
                                                             int nchains, int npad, int D, int slice_chains,
                                                             double* __restrict__ gacc) {
    const int lane = threadIdx.x;
    const int slice = blockIdx.y;
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
    const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
    const size_t NP = (size_t)npad;
    int ra[kFoldBT], rb[kFoldBT];
    double ca[kFoldBT], cb[kFoldBT];
#pragma unroll
    for (int a = 0; a < kFoldBT; ++a) {
        ra[a] = 16 * (kFoldBT * bi + a) + (lane & 15);
        rb[a] = 16 * (kFoldBT * bj + a) + (lane & 15);
        ca[a] = (ra[a] < D) ? c0[ra[a]] : 0.0;
        cb[a] = (rb[a] < D) ? c0[rb[a]] : 0.0;
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
#pragma unroll 8
    for (int c = c_begin; c < c_end; c += 4) {
        const int chain = c + (lane >> 4);
        double av[kFoldBT], bv[kFoldBT];
#pragma unroll
        for (int a = 0; a < kFoldBT; ++a) {
            av[a] = fold_operand(x, NP, D, ra[a], chain, nchains, ca[a]);
            bv[a] = fold_operand(x, NP, D, rb[a], chain, nchains, cb[a]);
        }
#pragma unroll
        for (int a = 0; a < kFoldBT; ++a)
#pragma unroll
            for (int b = 0; b < kFoldBT; ++b)
                if (valid(a, b)) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
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

hipError_t launch_fold(const double* x, const double* c0, int nchains, int npad, int D, int slice_chains,
                       double* gacc, hipStream_t stream);
hipError_t launch_fold_reduce(const double* gacc, int D, double* moments, hipStream_t stream);

}  // namespace smcmc
