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
//  * A workgroup stages the WHOLE point -- all rows of y for 16 chains -- in LDS, two
//    stages side by side in a row ([row][stage][chain], pitch 34 doubles: operand reads and staging writes are free of
//    bank conflicts).  Any wavefront can then fold any tile, so the tiles of a slice are dealt out as a flat list:
//    a host-made plan gives every workgroup a run of consecutive tiles (row-major over the lower triangle) of one
//    slice, cut evenly over its wavefronts (<= 8 tiles = 64 accumulator registers each).  Config 4: 24 slices x 10
//    workgroups x 52.8 tiles, 6 or 7 per wavefront (two wavefronts per SIMD: 13.2 tiles per SIMD against 16 for the 8 x 8 super-blocks); config
//    3: 64 slices x 4 workgroups, 2 or 3 tiles per wavefront.
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
#include <cstring>
#include <utility>
#include <vector>

#include "smcmc_fold_kernel.hip.h"
#include "smcmc_kernels.hip.h"

namespace smcmc {

constexpr int kFoldMaxSrc = 16;                 // points one launch folds
constexpr int kFrWaves = 8;                     // wavefronts per workgroup: two per SIMD
constexpr int kFrMaxT = 8;                      // tiles per wavefront (8 accumulator registers each)
constexpr int kFrRows = kFrWaves * kWave / 8;   // rows a staging round covers: every thread one 16-byte granule
constexpr int kFrC = 16;                        // chains per stage: 128-byte runs of every row
constexpr int kFrPitch = 2 * kFrC + 2;          // doubles per LDS row: two stages + 2 (stride = 2 mod 32 bank pairs)

constexpr int kFrMaxFoot = 36;                  // operand tiles a workgroup can stage: all 33 of dim 512 and a margin

// One workgroup's work: tiles pos0 .. pos0 + ntiles - 1 of the plan's tile order, of one slice; the operand tiles they
// need (their rows and their columns, `foot`, ascending) are what the workgroup stages, tile foot[k] in rows 16 k .. of
// its LDS image (slot_of is the inverse).
struct FoldPlanEntry {
    int32_t slice, pos0, ntiles, nfoot;
    uint8_t foot[kFrMaxFoot];
    uint8_t slot_of[kFrMaxFoot];
};

struct FoldRingParams {
    const double* src[kFoldMaxSrc];   // [D][npad] points, folded in this order
    int nsrc;
    const double* c0;                 // [D]
    int nchains, npad, D, slice_chains;
    double* gacc;                     // [slice][tile][4][64]
    const int32_t* mask;              // optional [npad]: a chain with mask 0 folds nothing (TSimpleHMC.H:336)
    const FoldPlanEntry* plan;        // [gridDim.x]
    const uint16_t* order;            // [tiles of the lower triangle] the plan's tile order: row << 8 | column
};

// Staging rounds per stage (a round = kFrRows rows = two operand tiles), a compile-time count so that every memory
// operation of the loaders' loop is unconditional and the compiler's s_waitcnt counts are exact: the smallest class that
// covers the largest footprint of the plan; a surplus round re-reads row D - 1 and writes rows of the image nobody reads.
constexpr int kFrRoundClasses[] = {2, 4, 5, 7, 9};
inline int fold_ring_rounds(int max_foot) {
    const int need = (16 * max_foot + kFrRows - 1) / kFrRows;
    for (int c : kFrRoundClasses)
        if (c >= need) return c;
    return -1;
}
// LDS image: rows [0, kFrRows NQ) the staged operand tiles (rows of x - c0; a row >= D of the point is never read there), then 16 constant rows (row D mod 16
// of them the 1 of y[D], the rest 0: what a tile row >= D reads), then 64 granules nobody reads (where the threads that
// have nothing to write to the constant rows write).
inline size_t fold_ring_lds_bytes(int NQ) {
    return sizeof(double) * ((size_t)(kFrRows * NQ + 16) * kFrPitch) + 1024;
}

// The tile order of a plan: bands of `bh` tile rows, within a band column by column, within a column down the rows.
// A run of consecutive tiles then needs few operand tiles: the band's rows and a short range of columns (a run of 53
// tiles in bands of 7: 15 of the 32 operand tiles of dim 500, where a run in row-major order needs nearly all of them).
inline std::vector<uint16_t> fold_ring_order(int T, int bh) {
    std::vector<uint16_t> o;
    for (int r0 = 0; r0 < T; r0 += bh) {
        const int r1 = std::min(T, r0 + bh);
        for (int tj = 0; tj < r1; ++tj)
            for (int ti = std::max(r0, tj); ti < r1; ++ti) o.push_back((uint16_t)(ti << 8 | tj));
    }
    return o;
}

struct FoldPlan {
    std::vector<FoldPlanEntry> wg;     // a multiple of 8 entries (empty ones at the end)
    std::vector<uint16_t> order;
    int band = 0, max_foot = 0, rounds = 0, max_tiles = 0;   // rounds: the class fold_ring_kernel is launched with
};

// The plan: workgroups per slice so that the largest number of matrix instructions any wavefront issues is as small as
// `budget` workgroups allow; slices in order, a slice's workgroups consecutive (they read the same rows: one XCD, see
// the kernel); the band height that gives the smallest footprints.
inline FoldPlan fold_ring_plan(int D, int nchains, int npad, int nslices, int slice_chains, int budget) {
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
    auto build = [&](int bh) {
        FoldPlan pl;
        pl.band = bh;
        pl.order = fold_ring_order(T, bh);
        long long foot_sum = 0;
        for (int s = 0; s < nslices; ++s) {
            const int n = nwg[s];
            for (int w = 0; w < n; ++w) {
                const int base = ntiles / n, rem = ntiles % n;
                FoldPlanEntry e;
                std::memset(&e, 0, sizeof(e));
                e.slice = s;
                e.pos0 = w * base + std::min(w, rem);
                e.ntiles = base + (w < rem ? 1 : 0);
                bool used[kFrMaxFoot] = {};
                for (int k = 0; k < e.ntiles; ++k) {
                    const uint16_t t = pl.order[e.pos0 + k];
                    used[t >> 8] = used[t & 255] = true;
                }
                for (int t = 0; t < T; ++t)
                    if (used[t]) {
                        e.slot_of[t] = (uint8_t)e.nfoot;
                        e.foot[e.nfoot++] = (uint8_t)t;
                    }
                pl.max_foot = std::max(pl.max_foot, e.nfoot);
                pl.max_tiles = std::max(pl.max_tiles, (e.ntiles + kFrWaves - 1) / kFrWaves);
                foot_sum += e.nfoot;
                pl.wg.push_back(e);
            }
        }
        pl.rounds = fold_ring_rounds(pl.max_foot);
        FoldPlanEntry none;
        std::memset(&none, 0, sizeof(none));
        while (pl.wg.size() % 8 != 0) pl.wg.push_back(none);
        return std::make_pair(pl, foot_sum);
    };
    // the band height: fewest staging rounds first (they are compiled in: every workgroup runs the class of the largest
    // footprint), then the least staged in all
    FoldPlan best;
    long long best_sum = -1;
    for (int bh = 2; bh <= std::min(T, 16); ++bh) {
        auto cand = build(bh);
        if (best_sum < 0 || cand.first.rounds < best.rounds ||
            (cand.first.rounds == best.rounds && cand.second < best_sum)) {
            best = cand.first;
            best_sum = cand.second;
        }
    }
    if (best_sum < 0) best = build(1).first;     // T == 1
    return best;
}

typedef __attribute__((address_space(3))) f64x2* lds_ptr_f64x2;
typedef __attribute__((address_space(3))) double* lds_ptr_f64;

__device__ __forceinline__ lds_cptr_f64 fr_lds_cptr(uint32_t byte_addr) { return (lds_cptr_f64)(uintptr_t)byte_addr; }
__device__ __forceinline__ lds_ptr_f64x2 fr_lds_ptr2(uint32_t byte_addr) { return (lds_ptr_f64x2)(uintptr_t)byte_addr; }

// A buffer descriptor over one source point: the loads then take a 32-bit per-thread offset and a scalar offset, and
// no address arithmetic on the vector pipe.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fr_make_rsrc(const void* base, uint32_t bytes) {
    // gfx950 raw buffer: DATA_FORMAT = 32 bit (0x20000), no swizzle, stride 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// What a wavefront carries through the loop: its accumulators and operand addresses, and its share of the staging.
template <int NT, int NQ, bool MASKED>
struct FrWave {
    static constexpr int N = 4 * NT;                       // matrix instructions per stage: 4 k-quads x NT tiles
    static constexpr int LA = (N / 2 < 8) ? N / 2 : 8;     // operands are read LA instructions ahead
    f64x4 acc[NT];
    lds_cptr_f64 aA[NT], aB[NT];          // the operands of tile t in LDS (stage 0, k-quad 0)
    double ra[LA], rb[LA];                // the operands of the next LA instructions
    f64x2 raw[NQ];                        // the stage in flight from memory: image row srow + kFrRows q, two chains
    double cc[NQ];                        // c0 of those rows
    uint32_t roff[NQ];                    // byte offsets of those rows plus the thread's chain pair
    lds_ptr_f64x2 wdst;                   // the thread's granule in image row srow, stage 0, in LDS
    lds_ptr_f64x2 wones;                  // ... in the constant 1 row (threads of row 0), or a place nobody reads
    __amdgpu_buffer_rsrc_t rsrc;          // buffer descriptor of the source the next fetch reads (wave-uniform)
    uint32_t soff;                        // byte offset of the first chain of the stage the next fetch brings (wave-uniform)
    uint32_t src_bytes;
    int32_t mraw[2];                      // MASKED: the mask words of the stage in flight
    int chain_raw;                        // first of this thread's two chains in the stage in flight
    int s_f, stg_f, nst, nsrc, c_begin, spair;   // the walk over (source, stage)
};

template <class W>
__device__ __forceinline__ void fr_fetch_round(W& w, int q) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, (int)w.roff[q], (int)w.soff, 0);
    w.raw[q] = __builtin_bit_cast(f64x2, v);
}

// Moves the fetch position one stage on (it stays on the last stage at the end: the surplus fetches re-read it).
template <class W>
__device__ __forceinline__ void fr_advance(W& w, const FoldRingParams& p) {
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
template <bool SELECT, class W>
__device__ __forceinline__ void fr_put_round(W& w, int q, int par, bool on0, bool on1) {
    f64x2 v = w.raw[q];
    v[0] -= w.cc[q];
    v[1] -= w.cc[q];
    if (SELECT) {
        if (!on0) v[0] = 0.0;
        if (!on1) v[1] = 0.0;
    }
    w.wdst[(par * kFrC + q * kFrRows * kFrPitch) / 2] = v;
}

// Staging round Q of the next stage (Q == NQ: the constant 1 row and, MASKED, the mask words of the stage after it).
template <int Q, int NQ, bool MASKED, int PAR, class W>
__device__ __forceinline__ void fr_round(W& w, const FoldRingParams& p, bool on0, bool on1, int chain_next) {
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
template <int Q0, int Q1, int NQ, bool MASKED, int PAR, class W>
__device__ __forceinline__ void fr_rounds(W& w, const FoldRingParams& p, bool on0, bool on1, int chain_next) {
    if constexpr (Q0 < Q1) {
        fr_round<Q0, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
        fr_rounds<Q0 + 1, Q1, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
    }
}
template <int NQ, int PAR, class W, int... Q>
__device__ __forceinline__ void fr_zero_columns(std::integer_sequence<int, Q...>, W& w, bool on0, bool on1) {
    lds_ptr_f64 z = (lds_ptr_f64)w.wdst + (1 - PAR) * kFrC;
    ((on0 ? (void)0 : (void)(z[Q * kFrRows * kFrPitch] = 0.0), on1 ? (void)0 : (void)(z[Q * kFrRows * kFrPitch + 1] = 0.0)), ...);
}

// Matrix instruction I of a stage on half PAR of the image (k-quad I / NT, tile I % NT).  In front of it the operand
// reads of instruction I + LA -- of the NEXT stage, in the other half, for the last LA instructions: the workgroup's
// barrier stands in front of the first of those reads, so the matrix pipe has LA instructions' worth of operands in
// registers while the wavefronts meet and the first reads of the new stage are on their way.  Behind the instructions
// in front of the barrier, dealt out evenly, the staging rounds of the next stage (registers -> the other half of every
// row), each followed by the fetch of the same round of the stage after that.  Every index is a compile-time constant
// and every memory operation unconditional.
template <int I, int NT, int NQ, bool MASKED, int PAR>
__device__ __forceinline__ void fr_step(FrWave<NT, NQ, MASKED>& w, const FoldRingParams& p, bool on0, bool on1, bool any_off,
                                        int chain_next) {
    constexpr int N = FrWave<NT, NQ, MASKED>::N, LA = FrWave<NT, NQ, MASKED>::LA, M = N - LA, R = NQ + 1;
    // The operand registers are a ring over the instructions of ALL stages: stage s starts at slot s N mod LA, and
    // N mod LA is 0 or LA / 2 (N is a multiple of 4, LA = 8 unless it divides N), so the start alternates with PAR.
    static_assert((2 * N) % LA == 0, "the ring closes over two stages");
    constexpr int S = (I + (N % LA) * PAR) % LA;
    const double a = w.ra[S], b = w.rb[S];
    if constexpr (I == M) {
        if (any_off) fr_zero_columns<NQ, PAR>(std::make_integer_sequence<int, NQ>{}, w, on0, on1);
        __syncthreads();
    }
    constexpr int J = (I + LA) % N;                        // the instruction whose operands are read now
    constexpr int so = ((I + LA < N) ? PAR : 1 - PAR) * kFrC;
    w.ra[S] = w.aA[J % NT][so + 4 * (J / NT)];
    w.rb[S] = w.aB[J % NT][so + 4 * (J / NT)];
    w.acc[I % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w.acc[I % NT], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I < M) fr_rounds<(I * R) / M, ((I + 1) * R) / M, NQ, MASKED, PAR>(w, p, on0, on1, chain_next);
}

template <int NT, int NQ, bool MASKED, int PAR, int... I>
__device__ __forceinline__ void fr_stage(std::integer_sequence<int, I...>, FrWave<NT, NQ, MASKED>& w, const FoldRingParams& p) {
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
    (fr_step<I, NT, NQ, MASKED, PAR>(w, p, on0, on1, any_off, chain_next), ...);
    w.chain_raw = chain_next;
}

// Tiles first .. first + nt - 1 of the plan's order, of slice pe.slice; NT >= nt matrix instructions per k-quad: the
// workgroup's busiest wavefront sets its pace either way, and a wavefront with fewer tiles folds its last one once more
// into a spare accumulator that is not stored.
template <int NT, int NQ, bool MASKED>
__device__ __forceinline__ void fr_run(const FoldRingParams& p, const FoldPlanEntry& pe, double* st, int first, int nt, int nst,
                                       int total, int lane) {
    constexpr int kConstRow = kFrRows * NQ;   // first of the 16 constant rows
    typedef FrWave<NT, NQ, MASKED> W;
    const int D = p.D;
    const int T = (D + 1 + 15) / 16, ntiles_all = T * (T + 1) / 2;
    const size_t NP = (size_t)p.npad;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)st;
    W w;
    // row r' of operand tile t: in the tile's slot of the image -- or, for a row >= D of the point, a constant row; it
    // keeps its place within the tile, so the 16 rows of an operand read stay on 16 different bank pairs
    auto lds_row = [&](int t, int rr) { return 16 * t + rr < D ? 16 * (int)pe.slot_of[t] + rr : kConstRow + rr; };
    int tix[NT];                           // tile numbers (wave-uniform)
    auto goff = [&](int t) { return (((size_t)pe.slice * ntiles_all + (size_t)tix[t]) * 4) * kWave + lane; };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = first + (t < nt ? t : (nt > 0 ? nt - 1 : 0));      // past its tiles: the last one again
        const int tt = (nt > 0) ? (int)p.order[k] : 0;
        const int ti = tt >> 8, tj = tt & 255;
        w.aA[t] = fr_lds_cptr(lds0 + (uint32_t)((lds_row(ti, lane & 15) * kFrPitch + (lane >> 4)) * 8));
        w.aB[t] = fr_lds_cptr(lds0 + (uint32_t)((lds_row(tj, lane & 15) * kFrPitch + (lane >> 4)) * 8));
        tix[t] = ti * (ti + 1) / 2 + tj;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) w.acc[t][r] = (t < nt) ? p.gacc[goff(t) + (size_t)r * kWave] : 0.0;

    w.c_begin = pe.slice * p.slice_chains;
    w.nst = nst;
    w.nsrc = p.nsrc;
    // staging role: granule = 16 bytes (two chains) of one row of the image; thread -> image rows (thread >> 3) + kFrRows q,
    // chains 2 (thread & 7), + 1.  Image row r holds row r & 15 of operand tile foot[r >> 4]; rows behind the footprint and
    // rows >= D of the point re-read row D - 1 (nobody reads what they write).
    const int srow = (int)threadIdx.x >> 3;
    w.spair = (int)threadIdx.x & 7;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int ri = srow + kFrRows * q, slot = ri >> 4;
        int r = (slot < pe.nfoot) ? 16 * (int)pe.foot[slot] + (ri & 15) : D - 1;
        if (r >= D) r = D - 1;
        w.cc[q] = p.c0[r];
        w.roff[q] = (uint32_t)(((size_t)r * NP + 2 * w.spair) * sizeof(double));
    }
    w.wdst = fr_lds_ptr2(lds0 + (uint32_t)((srow * kFrPitch + 2 * w.spair) * 8));
    w.wones = fr_lds_ptr2((srow == 0) ? lds0 + (uint32_t)(((kConstRow + (D & 15)) * kFrPitch + 2 * w.spair) * 8)
                                      : lds0 + (uint32_t)((kConstRow + 16) * kFrPitch * 8) + 16u * (uint32_t)lane);
    w.src_bytes = (uint32_t)((size_t)D * NP * sizeof(double));
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
    __syncthreads();                       // stage 0 is in LDS
#pragma unroll
    for (int i = 0; i < W::LA; ++i) {
        w.ra[i] = w.aA[i % NT][4 * (i / NT)];
        w.rb[i] = w.aB[i % NT][4 * (i / NT)];
    }
    for (int it = 0; it < total; it += 2) {
        fr_stage<NT, NQ, MASKED, 0>(std::make_integer_sequence<int, W::N>{}, w, p);
        if (it + 1 < total) fr_stage<NT, NQ, MASKED, 1>(std::make_integer_sequence<int, W::N>{}, w, p);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.gacc[goff(t) + (size_t)r * kWave] = w.acc[t][r];
        }
}

template <int NQ, bool MASKED>
static __global__ void __launch_bounds__(kFrWaves* kWave, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) fold_ring_kernel(const FoldRingParams p) {
    extern __shared__ __attribute__((aligned(16))) double st[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
    const int G = (int)gridDim.x;
    const int logical = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    const FoldPlanEntry pe = p.plan[logical];
    if (pe.ntiles == 0) return;
    constexpr int kConstRow = kFrRows * NQ;
    const int c_begin = pe.slice * p.slice_chains;
    int c_end = (c_begin + p.slice_chains < p.npad) ? c_begin + p.slice_chains : p.npad;
    {
        const int live = (p.nchains + kFrC - 1) / kFrC * kFrC;       // chains past the ensemble fold zeros: skipped
        if (c_end > live) c_end = live;
    }
    const int nst = (c_end - c_begin) / kFrC;
    const int total = nst * p.nsrc;
    // the constant rows: zero, both halves, once; the 1 row is written with every stage
    for (int k = (int)threadIdx.x; k < 16 * kFrPitch; k += kFrWaves * kWave) st[kConstRow * kFrPitch + k] = 0.0;
    __syncthreads();
    if (total <= 0) return;
    // Every wavefront of the workgroup issues NT matrix instructions per k-quad (the busiest one's count)
    const int NT = (pe.ntiles + kFrWaves - 1) / kFrWaves;
    const int base = pe.ntiles / kFrWaves, rem = pe.ntiles % kFrWaves;
    const int first = pe.pos0 + wv * base + (wv < rem ? wv : rem);
    const int nt = base + (wv < rem ? 1 : 0);
    switch (NT) {
#define SMCMC_FR_CASE(n) case n: fr_run<n, NQ, MASKED>(p, pe, st, first, nt, nst, total, lane); break;
        SMCMC_FR_CASE(1) SMCMC_FR_CASE(2) SMCMC_FR_CASE(3) SMCMC_FR_CASE(4) SMCMC_FR_CASE(5) SMCMC_FR_CASE(6)
        SMCMC_FR_CASE(7) SMCMC_FR_CASE(8)
#undef SMCMC_FR_CASE
        default: break;
    }
    static_assert(kFrMaxT == 8, "one case per tile count");
}

// Host side (smcmc_fold_inst.hip): the plan of an engine on its device, and the launch.
struct FoldRing {
    FoldPlanEntry* d_plan = nullptr;
    uint16_t* d_order = nullptr;
    int nwg = 0;                      // workgroups of a launch = plan entries (a multiple of 8)
    int rounds = 0;                   // the staging-round class of the plan (the kernel instantiation)
};
hipError_t fold_ring_prepare(FoldRing& fr, int D, int nchains, int npad, int nslices, int slice_chains);
void fold_ring_release(FoldRing& fr);
// p.plan is taken from fr; p.nsrc points in p.src are folded in order (1 <= nsrc <= kFoldMaxSrc)
hipError_t launch_fold_ring(const FoldRing& fr, FoldRingParams p, hipStream_t stream);

}  // namespace smcmc
