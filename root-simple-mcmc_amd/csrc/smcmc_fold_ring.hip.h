// smcmc_fold_ring.hip.h -- pooled second moments for the large-dimension path, round 4.
//
// What it computes is what fold_moments_kernel (smcmc_fold_kernel.hip.h) computed and oracle/ensemble_oracle.c mirrors:
// for every chain slice (moment group) the (D + 1) x (D + 1) lower triangle of sum y y^T, y = (x - c0, 1), the chains of
// the slice in ascending order, the points of consecutive steps in step order, as chains of v_mfma_f64_16x16x4_f64 on
// accumulators that persist in HBM (`gacc`, one 16 x 16 tile = 4 registers x 64 lanes per (slice, tile)); the slices are
// added in order by fold_reduce_kernel.  It is the batch form of the running covariance of TSimpleMCMC.H:1795-1820.
// Same accumulators, same order of fused multiply-adds: the bits are the old kernel's.
//
// What changed is where the operands come from and who computes what:
//  * A workgroup (4 wavefronts, one per SIMD) stages the WHOLE point -- all rows of y for 16 chains -- in LDS, two
//    stages side by side in a row ([row][stage][chain], pitch 34 doubles: operand reads and staging writes are free of
//    bank conflicts).  Any wavefront can then fold any tile, so the tiles of a slice are dealt out as a flat list:
//    a host-made plan gives every workgroup a run of consecutive tiles (row-major over the lower triangle) of one
//    slice, cut evenly over its wavefronts (<= 16 tiles = 128 accumulator registers each).  Config 4: 24 slices x 10
//    workgroups x 52.8 tiles, 13 or 14 per wavefront, against 16 for every wavefront of an 8 x 8 super-block; config
//    3: 64 slices x 4 workgroups, 5 or 6 tiles per wavefront.
//  * The workgroups of a slice read the same rows.  Workgroups are dealt round-robin over the 8 XCDs (block b and
//    b + 8 share one), so the plan is indexed by (b % 8) * (G / 8) + b / 8: the workgroups of a slice sit on ONE XCD
//    and all but the first to touch a line are served by that XCD's L2 (the old grid spread every slice over all
//    eight: 0.67 GB per fold at config 4 from beyond L2, which bound it).
//  * One launch folds up to kFoldMaxSrc points (the ring a multi-step launch leaves, in step order) with the
//    accumulators in registers throughout: they cross HBM once per launch instead of once per step.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "smcmc_fold_kernel.hip.h"
#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kFoldMaxSrc = 16;                 // points one launch folds
constexpr int kFrWaves = 4;                     // wavefronts per workgroup: one per SIMD
constexpr int kFrMaxT = 16;                     // tiles per wavefront (8 accumulator registers each)
constexpr int kFrC = 16;                        // chains per stage: 128-byte runs of every row
constexpr int kFrPitch = 2 * kFrC + 2;          // doubles per LDS row: two stages + 2 (stride = 2 mod 32 bank pairs)

struct FoldPlanEntry {
    int32_t slice, tile0, ntiles, pad;
};

struct FoldRingParams {
    const double* src[kFoldMaxSrc];   // [D][npad] points, folded in this order
    int nsrc;
    const double* c0;                 // [D]
    int nchains, npad, D, slice_chains;
    double* gacc;                     // [slice][tile][4][64]
    const int32_t* mask;              // optional [npad]: a chain with mask 0 folds nothing (TSimpleHMC.H:336)
    const FoldPlanEntry* plan;        // [gridDim.x]
};

// Staging rounds per stage (a round = 32 rows: every thread one 16-byte granule), a compile-time count so that every
// memory operation of the loop is unconditional and the compiler's s_waitcnt counts are exact: the smallest class
// >= ceil(D / 32); a surplus round re-reads row D - 1 and writes rows of the image nobody reads.
constexpr int kFrRoundClasses[] = {4, 7, 10, 13, 16};
inline int fold_ring_rounds(int D) {
    const int need = (D + 31) / 32;
    for (int c : kFrRoundClasses)
        if (c >= need) return c;
    return -1;
}
// LDS image: rows [0, 32 NQ) staged rows of x - c0 (rows >= D of it are never read), then 16 constant rows (row D mod 16
// of them the 1 of y[D], the rest 0: what a tile row >= D reads), then 64 granules nobody reads (where the threads that
// have nothing to write to the constant rows write).
inline size_t fold_ring_lds_bytes(int NQ) {
    return sizeof(double) * ((size_t)(32 * NQ + 16) * kFrPitch) + 1024;
}

// The plan: workgroups per slice so that the largest number of matrix instructions any wavefront issues is as small as
// `budget` workgroups allow; slices in order, a slice's workgroups consecutive; padded to a multiple of 8 entries.
inline std::vector<FoldPlanEntry> fold_ring_plan(int D, int nchains, int npad, int nslices, int slice_chains, int budget) {
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2;
    const int per_wg = kFrWaves * kFrMaxT;
    std::vector<int> kq(nslices, 0), nwg(nslices, 0);
    int total = 0;
    for (int s = 0; s < nslices; ++s) {
        const long long b = (long long)s * slice_chains;
        long long e = std::min<long long>(b + slice_chains, npad);
        e = std::min<long long>(e, ((long long)nchains + kFrC - 1) / kFrC * kFrC);
        if (e > b) {
            kq[s] = (int)((e - b) / 4);
            nwg[s] = (ntiles + per_wg - 1) / per_wg;
            total += nwg[s];
        }
    }
    auto cost = [&](int s, int n) {   // matrix instructions of the busiest wavefront of slice s with n workgroups
        const int t = (ntiles + n - 1) / n;
        return (long long)((t + kFrWaves - 1) / kFrWaves) * kq[s];
    };
    while (total < budget) {
        long long worst = 0;
        for (int s = 0; s < nslices; ++s)
            if (kq[s] > 0) worst = std::max(worst, cost(s, nwg[s]));
        // every slice that bad has to get better, or the largest count does not move
        int need = 0;
        bool stuck = false;
        std::vector<int> grown(nwg);
        for (int s = 0; s < nslices; ++s)
            if (kq[s] > 0 && cost(s, nwg[s]) == worst) {
                int n = nwg[s];
                while (n * kFrWaves < ntiles && cost(s, n) >= worst) ++n;
                if (cost(s, n) >= worst) stuck = true;
                need += n - nwg[s];
                grown[s] = n;
            }
        if (stuck || need == 0 || total + need > budget) break;
        nwg = grown;
        total += need;
    }
    std::vector<FoldPlanEntry> plan;
    for (int s = 0; s < nslices; ++s) {
        const int n = nwg[s];
        for (int w = 0; w < n; ++w) {
            const int base = ntiles / n, rem = ntiles % n;
            FoldPlanEntry e;
            e.slice = s;
            e.tile0 = w * base + std::min(w, rem);
            e.ntiles = base + (w < rem ? 1 : 0);
            e.pad = 0;
            plan.push_back(e);
        }
    }
    while (plan.size() % 8 != 0) plan.push_back(FoldPlanEntry{0, 0, 0, 0});
    return plan;
}

typedef __attribute__((address_space(3))) f64x2* lds_ptr_f64x2;
typedef __attribute__((address_space(3))) double* lds_ptr_f64;
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ lds_cptr_f64 fr_lds_cptr(uint32_t byte_addr) { return (lds_cptr_f64)(uintptr_t)byte_addr; }
__device__ __forceinline__ lds_ptr_f64x2 fr_lds_ptr2(uint32_t byte_addr) { return (lds_ptr_f64x2)(uintptr_t)byte_addr; }

// A buffer descriptor over one source point: the loads then take a 32-bit per-thread offset and a scalar offset, and
// no address arithmetic on the vector pipe (where nothing runs beside a matrix instruction).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fr_make_rsrc(const void* base, uint32_t bytes) {
    // gfx950 raw buffer: DATA_FORMAT = 32 bit (0x20000), no swizzle, stride 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// What a wavefront carries through the loop.
template <int NQ, bool MASKED>
struct FrWave {
    f64x4 acc[kFrMaxT];
    lds_cptr_f64 aA[kFrMaxT], aB[kFrMaxT];   // the operands of tile t in LDS (stage 0, k-quad 0)
    f64x2 raw[NQ];                        // the stage in flight from memory: row srow + 32 q, two chains
    double cc[NQ];                        // c0 of those rows
    uint32_t roff[NQ];                    // byte offsets of those rows (clamped to D - 1) plus the thread's chain pair
    lds_ptr_f64x2 wdst;                   // the thread's granule in row srow, stage 0, in LDS
    lds_ptr_f64x2 wones;                  // ... in the constant 1 row (threads of row 0), or a place nobody reads
    __amdgpu_buffer_rsrc_t rsrc;          // buffer descriptor of the source the next fetch reads (wave-uniform)
    uint32_t soff;                        // byte offset of the first chain of the stage the next fetch brings (wave-uniform)
    uint32_t src_bytes;
    int32_t mraw[2];                      // MASKED: the mask words of the stage in flight
    int chain_raw;                        // first of this thread's two chains in the stage in flight
    // the walk over (source, stage)
    int s_f, stg_f, nst, nsrc, c_begin, spair;
};

template <int NQ, bool MASKED>
__device__ __forceinline__ void fr_fetch_round(FrWave<NQ, MASKED>& w, int q) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, (int)w.roff[q], (int)w.soff, 0);
    w.raw[q] = __builtin_bit_cast(f64x2, v);
}

// Moves the fetch position one stage on (it stays on the last stage at the end: the surplus fetches re-read it).
template <int NQ, bool MASKED>
__device__ __forceinline__ void fr_advance(FrWave<NQ, MASKED>& w, const FoldRingParams& p) {
    int stg = w.stg_f + 1, s = w.s_f;
    if (stg == w.nst) { stg = 0; ++s; }
    if (s < w.nsrc) {
        w.stg_f = stg;
        if (s != w.s_f) {
            w.s_f = s;
            w.rsrc = fr_make_rsrc(p.src[s], w.src_bytes);
        }
    }
    w.soff = (uint32_t)((w.c_begin + w.stg_f * kFrC) * (int)sizeof(double));
}

// y = x - c0 for round q of the stage in `raw` (0 for a chain that folds nothing) into half `par` of its rows.
template <bool SELECT, int NQ, bool MASKED>
__device__ __forceinline__ void fr_put_round(FrWave<NQ, MASKED>& w, int q, int par, bool on0, bool on1) {
    f64x2 v = w.raw[q];
    v[0] -= w.cc[q];
    v[1] -= w.cc[q];
    if (SELECT) {
        if (!on0) v[0] = 0.0;
        if (!on1) v[1] = 0.0;
    }
    w.wdst[(par * kFrC + q * 32 * kFrPitch) / 2] = v;
}

// Staging round Q of the next stage (Q == NQ: the constant 1 row and, MASKED, the mask words of the stage after it).
template <int Q, int NQ, bool MASKED, int PAR>
__device__ __forceinline__ void fr_round(FrWave<NQ, MASKED>& w, const FoldRingParams& p, bool on0, bool on1, int chain_next) {
#ifdef FR_EXP_NO_STAGING
    return;
#endif
    if constexpr (Q < NQ) {
        fr_put_round<MASKED>(w, Q, 1 - PAR, on0, on1);
        fr_fetch_round(w, Q);
    } else {
        f64x2 one = {on0 ? 1.0 : 0.0, on1 ? 1.0 : 0.0};
        w.wones[(1 - PAR) * kFrC / 2] = one;
        if (MASKED) {
            const int2 m = *(const int2*)(p.mask + chain_next);
            w.mraw[0] = m.x;
            w.mraw[1] = m.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int Q0, int Q1, int NQ, bool MASKED, int PAR>
__device__ __forceinline__ void fr_rounds(FrWave<NQ, MASKED>& w, const FoldRingParams& p, bool on0, bool on1, int chain_next) {
    if constexpr (Q0 < Q1) {
        fr_round<Q0, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
        fr_rounds<Q0 + 1, Q1, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
    }
}

// Matrix instruction I of a stage (k-quad I / NT, tile I % NT), the operand reads of instruction I + LA in front of it
// and its share of the staging rounds behind it.  Every index is a compile-time constant.
template <int I, int NT, int NQ, bool MASKED, int PAR, int LA>
__device__ __forceinline__ void fr_step(FrWave<NQ, MASKED>& w, const FoldRingParams& p, double (&ra)[LA], double (&rb)[LA],
                                        bool on0, bool on1, int chain_next) {
    constexpr int N = 4 * NT, R = NQ + 1, so = PAR * kFrC;
    const double a = ra[I % LA], b = rb[I % LA];
    if constexpr (I + LA < N) {
        ra[I % LA] = w.aA[(I + LA) % NT][so + 4 * ((I + LA) / NT)];
        rb[I % LA] = w.aB[(I + LA) % NT][so + 4 * ((I + LA) / NT)];
    }
#ifdef FR_EXP_NO_MFMA
    w.acc[I % NT][0] += a * b;
#elif defined(FR_EXP_ONE_READ)
    w.acc[I % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, w.acc[I % NT], 0, 0, 0);
#else
    w.acc[I % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w.acc[I % NT], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    fr_rounds<(I * R) / N, ((I + 1) * R) / N, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
}
template <int NT, int NQ, bool MASKED, int PAR, int LA, int... I>
__device__ __forceinline__ void fr_steps(std::integer_sequence<int, I...>, FrWave<NQ, MASKED>& w, const FoldRingParams& p,
                                         double (&ra)[LA], double (&rb)[LA], bool on0, bool on1, int chain_next) {
    (fr_step<I, NT, NQ, MASKED, PAR, LA>(w, p, ra, rb, on0, on1, chain_next), ...);
}
template <int NQ, bool MASKED, int PAR, int... Q>
__device__ __forceinline__ void fr_zero_columns(std::integer_sequence<int, Q...>, FrWave<NQ, MASKED>& w, bool on0, bool on1) {
    lds_ptr_f64 z = (lds_ptr_f64)w.wdst + (1 - PAR) * kFrC;
    ((on0 ? (void)0 : (void)(z[Q * 32 * kFrPitch] = 0.0), on1 ? (void)0 : (void)(z[Q * 32 * kFrPitch + 1] = 0.0)), ...);
}

// One stage: 4 k-quads x NT tiles of matrix instructions on half PAR of the image, operands read LA instructions
// ahead; dealt out between them the NQ staging rounds of the NEXT stage (registers -> the other half of every row,
// which nobody reads before the barrier), each followed by the fetch of the same round of the stage after that.
template <int NT, int NQ, bool MASKED, int PAR>
__device__ __forceinline__ void fr_stage(FrWave<NQ, MASKED>& w, const FoldRingParams& p) {
    constexpr int N = 4 * NT;
    constexpr int LA = (N < 8) ? N : 8;
    constexpr int so = PAR * kFrC;
    // flags of the stage in `raw`
    bool on0 = w.chain_raw < p.nchains, on1 = w.chain_raw + 1 < p.nchains;
    if (MASKED) {
        on0 = on0 && w.mraw[0] != 0;
        on1 = on1 && w.mraw[1] != 0;
    }
    // Without a mask the only chains that fold nothing are those past the ensemble in the last stage of a source: their
    // columns are zeroed behind the rounds (rare, wave-uniform), and the rounds carry no selects.
    const bool any_off = !MASKED && __any(!on1) != 0;
    fr_advance(w, p);           // where the fetches of this stage read
    const int chain_next = w.c_begin + w.stg_f * kFrC + 2 * w.spair;
    double ra[LA], rb[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        ra[i] = w.aA[i % NT][so + 4 * (i / NT)];
        rb[i] = w.aB[i % NT][so + 4 * (i / NT)];
    }
    fr_steps<NT, NQ, MASKED, PAR, LA>(std::make_integer_sequence<int, N>{}, w, p, ra, rb, on0, on1, chain_next);
    if (any_off) fr_zero_columns<NQ, MASKED, PAR>(std::make_integer_sequence<int, NQ>{}, w, on0, on1);
    w.chain_raw = chain_next;
    __syncthreads();
}

template <int NT, int NQ, bool MASKED>
__device__ __forceinline__ void fr_run(FrWave<NQ, MASKED>& w, const FoldRingParams& p, int total) {
    for (int it = 0; it < total; it += 2) {
        fr_stage<NT, NQ, MASKED, 0>(w, p);
        if (it + 1 < total) fr_stage<NT, NQ, MASKED, 1>(w, p);
    }
}

template <int NQ, bool MASKED>
static __global__ void __launch_bounds__(kFrWaves* kWave, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) fold_ring_kernel(const FoldRingParams p) {
    extern __shared__ __attribute__((aligned(16))) double st[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
    const int G = (int)gridDim.x;
    const int logical = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const FoldPlanEntry pe = p.plan[logical];
    if (pe.ntiles == 0) return;
    const int D = p.D;
    const int T = (D + 1 + 15) / 16, ntiles_all = T * (T + 1) / 2;
    const size_t NP = (size_t)p.npad;
    // Every wavefront of the workgroup issues NT matrix instructions per k-quad -- the busiest one's count, which sets
    // the workgroup's time either way; a wavefront with fewer tiles folds its last tile once more into a spare
    // accumulator that is not stored.
    const int NT = (pe.ntiles + kFrWaves - 1) / kFrWaves;
    const int base = pe.ntiles / kFrWaves, rem = pe.ntiles % kFrWaves;
    const int first = pe.tile0 + wv * base + (wv < rem ? wv : rem);
    const int nt = base + (wv < rem ? 1 : 0);
    FrWave<NQ, MASKED> w;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)st;
    constexpr int kConstRow = 32 * NQ;     // first of the 16 constant rows
    // a tile row >= D reads a constant row; it keeps its place within the tile, so the 16 rows of an operand read stay on
    // 16 different bank pairs
    auto lds_row = [&](int r) { return r < D ? r : kConstRow + (r & 15); };
    int tix[kFrMaxT];                      // tile numbers (wave-uniform)
    auto goff = [&](int t) { return (((size_t)pe.slice * ntiles_all + (size_t)tix[t]) * 4) * kWave + lane; };
    {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= first) ++ti;
        int tj = first - ti * (ti + 1) / 2;
#pragma unroll
        for (int t = 0; t < kFrMaxT; ++t) {
            w.aA[t] = fr_lds_cptr(lds0 + (uint32_t)((lds_row(16 * ti + (lane & 15)) * kFrPitch + (lane >> 4)) * 8));
            w.aB[t] = fr_lds_cptr(lds0 + (uint32_t)((lds_row(16 * tj + (lane & 15)) * kFrPitch + (lane >> 4)) * 8));
            tix[t] = ti * (ti + 1) / 2 + tj;
            if (t + 1 < nt) {
                if (tj == ti) { ++ti; tj = 0; } else ++tj;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kFrMaxT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) w.acc[t][r] = (t < nt) ? p.gacc[goff(t) + (size_t)r * kWave] : 0.0;

    w.c_begin = pe.slice * p.slice_chains;
    int c_end = (w.c_begin + p.slice_chains < p.npad) ? w.c_begin + p.slice_chains : p.npad;
    {
        const int live = (p.nchains + kFrC - 1) / kFrC * kFrC;       // chains past the ensemble fold zeros: skipped
        if (c_end > live) c_end = live;
    }
    w.nst = (c_end - w.c_begin) / kFrC;
    w.nsrc = p.nsrc;
    const int total = w.nst * p.nsrc;

    // staging role: granule = 16 bytes (two chains) of one row; thread -> rows (thread >> 3) + 32 q, chains 2 (thread & 7), + 1
    const int srow = (int)threadIdx.x >> 3;
    w.spair = (int)threadIdx.x & 7;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = srow + 32 * q, rc = r < D ? r : D - 1;
        w.cc[q] = p.c0[rc];
        w.roff[q] = (uint32_t)(((size_t)rc * NP + 2 * w.spair) * sizeof(double));
    }
    w.wdst = fr_lds_ptr2(lds0 + (uint32_t)((srow * kFrPitch + 2 * w.spair) * 8));
    w.wones = fr_lds_ptr2((srow == 0) ? lds0 + (uint32_t)(((kConstRow + (D & 15)) * kFrPitch + 2 * w.spair) * 8)
                                      : lds0 + (uint32_t)((kConstRow + 16) * kFrPitch * 8) + 16u * (uint32_t)lane);
    w.src_bytes = (uint32_t)((size_t)D * NP * sizeof(double));
    // the constant rows: zero, both halves, once; the 1 row is written with every stage
    for (int k = (int)threadIdx.x; k < 16 * kFrPitch; k += kFrWaves * kWave) st[kConstRow * kFrPitch + k] = 0.0;
    __syncthreads();

    if (total > 0) {
        // stage 0 into half 0 (and its 1 row), stage 1 into the registers
        w.s_f = 0; w.stg_f = 0;
        w.rsrc = fr_make_rsrc(p.src[0], w.src_bytes);
        w.soff = (uint32_t)(w.c_begin * (int)sizeof(double));
        w.chain_raw = w.c_begin + 2 * w.spair;
        if (MASKED) {
            const int2 m = *(const int2*)(p.mask + w.chain_raw);
            w.mraw[0] = m.x; w.mraw[1] = m.y;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) fr_fetch_round(w, q);
        {
            bool on0 = w.chain_raw < p.nchains, on1 = w.chain_raw + 1 < p.nchains;
            if (MASKED) { on0 = on0 && w.mraw[0] != 0; on1 = on1 && w.mraw[1] != 0; }
#pragma unroll
            for (int q = 0; q < NQ; ++q) fr_put_round<true>(w, q, 0, on0, on1);
            f64x2 one = {on0 ? 1.0 : 0.0, on1 ? 1.0 : 0.0};
            w.wones[0] = one;
        }
        fr_advance(w, p);
        w.chain_raw = w.c_begin + w.stg_f * kFrC + 2 * w.spair;
        if (MASKED) {
            const int2 m = *(const int2*)(p.mask + w.chain_raw);
            w.mraw[0] = m.x; w.mraw[1] = m.y;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) fr_fetch_round(w, q);
        __syncthreads();
        switch (NT) {
#define SMCMC_FR_CASE(n) case n: fr_run<n, NQ, MASKED>(w, p, total); break;
            SMCMC_FR_CASE(1) SMCMC_FR_CASE(2) SMCMC_FR_CASE(3) SMCMC_FR_CASE(4) SMCMC_FR_CASE(5) SMCMC_FR_CASE(6)
            SMCMC_FR_CASE(7) SMCMC_FR_CASE(8) SMCMC_FR_CASE(9) SMCMC_FR_CASE(10) SMCMC_FR_CASE(11) SMCMC_FR_CASE(12)
            SMCMC_FR_CASE(13) SMCMC_FR_CASE(14) SMCMC_FR_CASE(15) SMCMC_FR_CASE(16)
#undef SMCMC_FR_CASE
            default: break;
        }
    }
#pragma unroll
    for (int t = 0; t < kFrMaxT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.gacc[goff(t) + (size_t)r * kWave] = w.acc[t][r];
        }
}

// Host side (smcmc_fold_inst.hip): the plan of an engine on its device, and the launch.
struct FoldRing {
    FoldPlanEntry* d_plan = nullptr;
    int nwg = 0;                      // workgroups of a launch = plan entries (a multiple of 8)
};
hipError_t fold_ring_prepare(FoldRing& fr, int D, int nchains, int npad, int nslices, int slice_chains);
void fold_ring_release(FoldRing& fr);
// p.plan is taken from fr; p.nsrc points in p.src are folded in order (1 <= nsrc <= kFoldMaxSrc)
hipError_t launch_fold_ring(const FoldRing& fr, FoldRingParams p, hipStream_t stream);

}  // namespace smcmc
