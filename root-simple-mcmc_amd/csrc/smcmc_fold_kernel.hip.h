// smcmc_fold_kernel.hip.h -- pooled second moments for the large-dimension path: the moment groups and their
// ordered reduction.
//
// For D > 63 the (D+1)(D+2)/2 accumulators of a 64-chain group no longer fit a wavefront's registers, so the fold
// of the current point into the moment sums (the batch form of the running covariance, reference
// TSimpleMCMC.H:1795-1820) runs as its own kernel between step launches (smcmc_fold_ring.hip.h): per chain slice
// (moment group) and 16 x 16 tile of the lower triangle one accumulator tile that persists in HBM, y = x - c0 of the
// slice's chains folded in ascending order by chains of v_mfma_f64_16x16x4_f64.  The slice sums are added in slice
// order by fold_reduce_kernel.  oracle/ensemble_oracle.c mirrors this order with moment groups of `slice_chains`
// chains.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kFoldSlices = 128;  // most chain slices (moment groups) of the large-dimension path
constexpr int kFoldBT = 4;        // geometry of the fold kernel of rounds 1-3 (a wavefront folded 4 x 4 tiles, a
constexpr int kFoldSB = 2;        // workgroup 2 x 2 such blocks): it still DEFINES the number of moment groups below

// super-blocks of that geometry: the lower triangle of the kFoldSB * kFoldBT-tile grid
inline int fold_super_blocks(int D) {
    const int T = (D + 1 + 15) / 16, TB = (T + kFoldBT - 1) / kFoldBT, SB = (TB + kFoldSB - 1) / kFoldSB;
    return SB * (SB + 1) / 2;
}
// The number of moment groups: part of the engine's definition (the groups' sums are added in order, so the count
// shapes the last bits of the pooled moments; the oracle takes it from SMCMC_P_MOMENT_GROUP).  It is what gave the old
// kernel one workgroup per CU and has been kept since, so that results do not move with the kernel.
inline int fold_slices(int D) {
    int n = 256 / fold_super_blocks(D);
    if (n > kFoldSlices) n = kFoldSlices;
    if (n < 1) n = 1;
    return n;
}

// packed element k = (i, j), j <= i <= D  ->  sum over slices of its tile entry, in the order of every moment reduction
// of the engine (reduce_chunks_kernel / reduce_final_kernel for dim <= 63, oracle_ensemble_reduce_moments): slices in
// ascending order within chunks of kReduceChunk = 32, then the chunk sums in ascending order
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
    double total = 0.0;
    for (int sl0 = 0; sl0 < nslices; sl0 += kReduceChunk) {
        double s = 0.0;
        for (int sl = sl0; sl < nslices && sl < sl0 + kReduceChunk; ++sl)
            s += gacc[(((size_t)sl * ntiles + tile) * 4 + reg) * kWave + lane];
        total += s;
    }
    moments[k] = total;
}

hipError_t launch_fold_reduce(const double* gacc, int D, int nslices, double* moments, hipStream_t stream);

}  // namespace smcmc
