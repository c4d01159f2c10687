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

constexpr int kFoldSlices = 4;   // chain slices (moment groups) of the large-dimension path

// row r of the augmented point y: dims 0..D-1, the constant 1 at r == D, zero above
__device__ __forceinline__ double fold_operand(const double* __restrict__ x, size_t NP, int D, int r, int chain,
                                               int nchains, double c0r) {
    if (chain >= nchains) return 0.0;
    if (r < D) return x[(size_t)r * NP + chain] - c0r;
    return (r == D) ? 1.0 : 0.0;
}

// grid = (ntiles, kFoldSlices), block = 64.  tile -> (ti, tj <= ti).
static __global__ void __launch_bounds__(kWave) fold_moments_kernel(const double* __restrict__ x, const double* __restrict__ c0,
                                                             int nchains, int npad, int D, int slice_chains,
                                                             double* __restrict__ gacc) {
    const int lane = threadIdx.x;
    const int tile = blockIdx.x, slice = blockIdx.y;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    const int ra = 16 * ti + (lane & 15), rb = 16 * tj + (lane & 15);
    const double ca = (ra < D) ? c0[ra] : 0.0, cb = (rb < D) ? c0[rb] : 0.0;
    const size_t NP = (size_t)npad;
    const size_t off = (((size_t)slice * gridDim.x + tile) * 4) * kWave + lane;
    f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = gacc[off + (size_t)r * kWave];
    const int c_begin = slice * slice_chains;
    const int c_end = (c_begin + slice_chains < npad) ? c_begin + slice_chains : npad;
    for (int c = c_begin; c < c_end; c += 16) {
        double a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int chain = c + 4 * k + (lane >> 4);
            a[k] = fold_operand(x, NP, D, ra, chain, (chain < c_end) ? nchains : 0, ca);
            b[k] = fold_operand(x, NP, D, rb, chain, (chain < c_end) ? nchains : 0, cb);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[k], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gacc[off + (size_t)r * kWave] = acc[r];
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
