// The large-dimension moment fold of rounds 1-3 (fold_moments_kernel), kept for tools/micro/fold_bench.hip only: the
// baseline the round-4 kernel (root-simple-mcmc_amd/csrc/smcmc_fold_ring.hip.h) is timed and bit-compared against.
#pragma once
#include "smcmc_fold_kernel.hip.h"

namespace smcmc {

constexpr int kFoldWaves = kFoldSB * kFoldSB;
constexpr int kFoldOps = 2 * kFoldSB * kFoldBT;   // operand tiles a workgroup stages: its row group and its column group

// grid = (super-blocks, slices), block = 256.  Wavefront (a2, b2) of the workgroup owns block (2 BI + a2, 2 BJ + b2) of
// 4 x 4 tiles: eight operand tiles feed its sixteen matrix instructions per four chains.  The workgroup stages the
// sixteen operand tiles of its row group and column group once for all four wavefronts (round 1 staged eight tiles per
// wavefront and read the state nine times over at D = 500: 1.2 GB per fold, which bound it; now 0.6 GB).  The state is
// read in full cache lines (lane -> row lane >> 2, four consecutive chains), one stage of 16 chains ahead of its use,
// and re-laid out through LDS into the operand layout (row lane & 15, chain 4 n + (lane >> 4)).
// mask (optional, [npad]): a chain with mask 0 folds nothing this time (the HMC engine: a step whose proposal had a
// non-finite potential skips UpdateCovariance, TSimpleHMC.H:336)
static __global__ void __launch_bounds__(kFoldWaves* kWave) fold_moments_kernel(const double* __restrict__ x, const double* __restrict__ c0,
                                                             int nchains, int npad, int D, int slice_chains,
                                                             double* __restrict__ gacc, const int32_t* __restrict__ mask) {
    constexpr int kC = 32;        // chains per stage: 256 contiguous bytes of every staged row (16 chains = 128-byte
                                  // pieces of 64 000 concurrent row streams ran the memory system at a fifth of its rate;
                                  // 64 chains per stage measured no better than 32)
    constexpr int kS = kC + 2;    // doubles per staged row: the chains + 2 (operand reads spread over the banks)
    constexpr int kL = kC / 8;    // 16-byte loads per lane and operand tile (a lane holds kC / 4 consecutive chains of one row)
    __shared__ __attribute__((aligned(16))) double st[kFoldOps][16][kS];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x / kWave;
    const int slice = blockIdx.y;
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2, TB = (T + kFoldBT - 1) / kFoldBT;
    int BI = 0;
    while ((BI + 1) * (BI + 2) / 2 <= (int)blockIdx.x) ++BI;
    const int BJ = (int)blockIdx.x - BI * (BI + 1) / 2;
    const bool sdiag = (BI == BJ);                      // the column group is the row group
    const int a2 = wv / kFoldSB, b2 = wv % kFoldSB;
    const int bi = kFoldSB * BI + a2, bj = kFoldSB * BJ + b2;
    const bool mine = bi < TB && bj <= bi;              // this wavefront has a block (it stages its share either way)
    const bool diagonal = (bi == bj);
    const int nops = sdiag ? kFoldSB * kFoldBT : kFoldOps;
    const int colbase = sdiag ? 0 : kFoldSB * kFoldBT;  // first staged tile of the column group
    const size_t NP = (size_t)npad;
    // staging role of this lane: row (lane >> 2) of the wavefront's share of the operand tiles, kC / 4 consecutive chains
    const int srow = lane >> 2, sq = lane & 3;
    constexpr int kShare = kFoldOps / kFoldWaves;       // operand tiles a wavefront fetches per stage
    int rr[kShare];
    double cc[kShare];
#pragma unroll
    for (int q = 0; q < kShare; ++q) {
        const int op = wv + kFoldWaves * q;
        const int tile = (op < kFoldSB * kFoldBT) ? kFoldSB * kFoldBT * BI + op : kFoldSB * kFoldBT * BJ + (op - kFoldSB * kFoldBT);
        rr[q] = 16 * tile + srow;
        cc[q] = (rr[q] < D) ? c0[rr[q]] : 0.0;
    }
    // tile (ti, tj) of the block: valid when it exists and lies in the lower triangle
    auto valid = [&](int a, int b) { return mine && kFoldBT * bi + a < T && kFoldBT * bj + b <= kFoldBT * bi + a; };
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

    typedef f64x2 stage_t[kShare][kL];
    stage_t stA, stB;   // two stages of kC chains in flight ahead of the matrix instructions
    auto fetch = [&](int c, stage_t& stage) {   // y = x - c0 (the constant 1 in row D, 0 above, 0 for chains past the ensemble)
        const int chain = c + 2 * kL * sq;
        bool on[2 * kL];
#pragma unroll
        for (int k = 0; k < 2 * kL; ++k) on[k] = (chain + k < nchains) && (mask == nullptr || mask[chain + k] != 0);
#pragma unroll
        for (int q = 0; q < kShare; ++q) {
            if (wv + kFoldWaves * q >= nops) continue;
#pragma unroll
            for (int k = 0; k < kL; ++k) {
                f64x2 v = {0.0, 0.0};
                if (rr[q] < D) {
                    v = ((const f64x2*)(x + (size_t)rr[q] * NP + chain))[k];
                    v[0] -= cc[q]; v[1] -= cc[q];
                } else if (rr[q] == D) {
                    v[0] = v[1] = 1.0;
                }
                if (!on[2 * k]) v[0] = 0.0;
                if (!on[2 * k + 1]) v[1] = 0.0;
                stage[q][k] = v;
            }
        }
    };
    // one stage: registers -> LDS (every wavefront its share), refill the registers two stages ahead, fold the 16 chains.
    // All wavefronts of the workgroup walk the same slice, so the barriers are uniform.
    auto consume = [&](int c, stage_t& stage) {
        __syncthreads();   // the previous stage has been consumed by every wavefront
#pragma unroll
        for (int q = 0; q < kShare; ++q) {
            const int op = wv + kFoldWaves * q;
            if (op >= nops) continue;
#pragma unroll
            for (int k = 0; k < kL; ++k) *(f64x2*)&st[op][srow][2 * kL * sq + 2 * k] = stage[q][k];
        }
        __syncthreads();
        if (c + 2 * kC < c_end) fetch(c + 2 * kC, stage);   // in flight under the matrix instructions
        if (!mine) return;
#pragma unroll
        for (int n = 0; n < kC / 4; ++n) {
            double av[kFoldBT], bv[kFoldBT];
#pragma unroll
            for (int a = 0; a < kFoldBT; ++a) {
                av[a] = st[kFoldBT * a2 + a][lane & 15][4 * n + (lane >> 4)];
                bv[a] = diagonal ? av[a] : st[colbase + kFoldBT * b2 + a][lane & 15][4 * n + (lane >> 4)];
            }
#pragma unroll
            for (int a = 0; a < kFoldBT; ++a)
#pragma unroll
                for (int b = 0; b < kFoldBT; ++b)
                    if (valid(a, b)) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    };
    if (c_begin < c_end) fetch(c_begin, stA);
    if (c_begin + kC < c_end) fetch(c_begin + kC, stB);
    for (int c = c_begin; c < c_end; c += 2 * kC) {
        consume(c, stA);
        if (c + kC < c_end) consume(c + kC, stB);
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


}  // namespace smcmc
