// smcmc_kernels.hip.h -- gfx950 kernels of the many-chain Metropolis step.
//
// One chain per lane, one 64-chain group per wavefront, one wavefront per
// workgroup.  A launch advances every chain by `nsteps` calls of
// sMCMC::TSimpleMCMC::Step() (reference TSimpleMCMC.H:370-496) without touching
// HBM between the steps: the accepted point lives in LDS as x[dim][lane]
// (conflict-free column per lane), the proposal in registers.
//   A  scalar half of TProposeAdaptiveStep::UpdateState (TSimpleMCMC.H:1723-1776)
//      per lane; in POOLED mode the current point is folded into the group's
//      second-moment accumulator with v_mfma_f64_16x16x4_f64 whose operands are
//      read straight from the LDS image of x (y = x - c0 formed on the fly),
//      replacing the per-chain running covariance of TSimpleMCMC.H:1780-1820
//   B  proposal x' = x + sigma U^T r (TSimpleMCMC.H:709-724), r from Philox +
//      Box-Muller (include/smcmc_detmath.h), U through wave-uniform scalar loads;
//      StepRMS window (:391-406); likelihood (:410); Metropolis test (:432-463);
//      accept copy (:484-491) as an exec-masked LDS write
// EXACT = the reference's un-fused operation order (mul, then add); !EXACT =
// fused multiply-add.  Both are reproduced bit for bit by oracle/ensemble_oracle.c.
//
// Runtime dim <= DP: state rows >= dim are zero, U / Error are zero padded, so the
// padded lanes of the arithmetic only ever add +-0 and need no guards.
//
// Compile with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "smcmc.h"
#include "smcmc_detmath.h"

namespace smcmc {

// Register-array sizes the step kernel is instantiated for, smallest first (an
// engine of dimension D runs the smallest DP >= D).  D + 1 <= 64 keeps the
// moment contraction within 4 x 4 tiles.  build.py reads this list.
#define SMCMC_FOR_EACH_DP(X) X(7) X(15) X(31) X(47) X(50) X(63)

constexpr int kWave = 64;
constexpr int kXStride = 66;   // doubles per LDS row: stride = 2 (mod 32) keeps both the lane=chain
                               // ds_read/ds_write_b64 and the MFMA-operand ds_read_b64 conflict-free

typedef double f64x4 __attribute__((ext_vector_type(4)));

// Wave-uniform read-only tables (U, Error, c0) are read through the constant
// address space so that the loads become scalar (s_load into SGPRs, one fetch per
// wavefront through the scalar cache) instead of 64 identical vector loads.
typedef const __attribute__((address_space(4))) double* cptr_f64;
typedef const __attribute__((address_space(3))) double* lds_cptr_f64;
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f64x2* lds_cptr_f64x2;
__device__ __forceinline__ cptr_f64 as_const(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (cptr_f64)p;
#pragma clang diagnostic pop
}


// smcmc_normal_pair (smcmc_detmath.h) with its two tables in LDS: the index differs from lane to lane, which LDS serves
// in a few passes and a global load in up to 64 cache-line requests.  lt / at: 64 entries of two doubles each.
// The two entries come in ahead of the arithmetic (normal_tables_fetch): an LDS read returns in issue order, so a
// table read issued behind the prefetch of the next U piece would wait for the whole piece.
struct NormalTables { f64x2 le, ae; };
// Hand-placed LDS reads (inline assembly, waited for by hand: load_piece below says why) are only used where the
// compiler leaves them alone.  It sees an assembly read's destination as an ordinarily defined value and may copy or
// spill it at once -- before the data has arrived.  The families up to 31 dimensions, which it allocates for two
// wavefronts per SIMD, are short of registers and did exactly that (v_accvgpr_write right behind the read); they keep
// reads the compiler can see and count.  build.py runs inflight_check.py over every step-kernel listing and fails on
// a compiler instruction that touches a register with an assembly read in flight.
template <int DP>
constexpr bool kAsmReads = DP >= 47;
template <bool ASM>
__device__ __forceinline__ NormalTables normal_tables_fetch(uint32_t w0, uint32_t w1, uint32_t lt_base, uint32_t at_base) {
    NormalTables t;
    const uint32_t la = lt_base + 16u * smcmc_normal_log_index(w0), aa = at_base + 16u * smcmc_normal_angle_index_halfcircle(w1);
    if constexpr (ASM) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(t.le) : "v"(la));
        asm volatile("ds_read_b128 %0, %1" : "=v"(t.ae) : "v"(aa));
    } else {
        t.le = *(volatile lds_cptr_f64x2)(uintptr_t)la;
        t.ae = *(volatile lds_cptr_f64x2)(uintptr_t)aa;
    }
    return t;
}
template <int N, bool ASM>
__device__ __forceinline__ void normal_tables_ready(NormalTables& a, NormalTables& b) {
    if constexpr (ASM) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.le), "+v"(a.ae), "+v"(b.le), "+v"(b.ae) : "n"(N));
}
template <int N, bool ASM>
__device__ __forceinline__ void normal_tables_ready(NormalTables& a) {
    if constexpr (ASM) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.le), "+v"(a.ae) : "n"(N));
}
__device__ __forceinline__ void normal_pair_lds(uint32_t w0, uint32_t w1, const NormalTables& t, double* n0, double* n1) {
#ifdef SMCMC_NORMAL_TEXTBOOK
    (void)t;   // the frozen-definition build (include/smcmc_detmath.h): the textbook pair, no tables
    smcmc_normal_pair(w0, w1, n0, n1);
    return;
#endif
#define SMCMC_LT_LDS(k, c) (t.le[c])
#define SMCMC_AT_LDS(c) (t.ae[c])
    SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE(SMCMC_LT_LDS, SMCMC_AT_LDS)
#undef SMCMC_LT_LDS
#undef SMCMC_AT_LDS
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

constexpr int kPiece = 16;   // columns of U consumed per scheduling region (8 x ds_read_b128)

template <int DP>
struct Geo {
    static constexpr int T = (DP + 1 + 15) / 16;   // 16-row tiles covering dims 0..DP-1 plus the ones row DP
    static constexpr int NT = T * (T + 1) / 2;     // lower-triangular tiles
    static constexpr int NB = (DP + 3) / 4;        // Philox blocks of 4 normals
    static constexpr int ROWS_MOMENTS = DP + 1;    // LDS rows of x: the dims and the ones row (zero rows are synthesized)
    // When at most 4 rows spill into the last 16-row tile (D = 50: rows 48, 49 and the ones row), that tile
    // row is folded with v_mfma_f64_4x4x4_4b_f64 instead (a quarter of the matrix-pipe time): its four
    // 4x4 blocks are (strip rows) x (columns 16 t + 4 blk ..), which lands exactly where register 0 of
    // tile (T-1, t) of the 16x16 scheme would, so the stored layout does not change.
    static constexpr bool STRIP = ((DP + 1) % 16 != 0) && ((DP + 1) % 16 <= 4);
    static constexpr int T16 = STRIP ? T - 1 : T;  // tile rows folded with 16x16x4
    static constexpr int NT16 = T16 * (T16 + 1) / 2;
};

// LDS image of the decomposition: row i keeps columns j0(i)..DP-1 (j0 = i rounded
// down to even for the triangular factor, 0 for a full matrix), padded to an even
// length so that every row starts 16-byte aligned (ds_read_b128 = two columns).
template <int DP, bool FULLU>
struct ULayout {
    static constexpr int DPE = DP + (DP & 1);       // DP rounded up to even
    static constexpr int j0(int i) { return FULLU ? 0 : (i & ~1); }
    static constexpr int len(int i) { return DPE - j0(i); }
    // closed form of sum_{r<i} len(r) (no loop: must fold once the caller's loops unroll)
    static constexpr int off(int i) {
        return FULLU ? i * DPE
                     : i * DPE - ((i & 1) ? 2 * (i / 2) * (i / 2) : 2 * (i / 2) * (i / 2 - 1));
    }
    static constexpr int SIZE = off(DP);
};

// The pieces (row, first column) that the rows of Philox block B contribute, in
// the order they are consumed.
template <int DP, bool FULLU, int B>
struct UPieces {
    typedef ULayout<DP, FULLU> UL;
    static constexpr int rows() { return (4 * B + 4 <= DP) ? 4 : (DP - 4 * B); }
    static constexpr int per_row(int i) { return (UL::len(i) + kPiece - 1) / kPiece; }
    static constexpr int count() {
        int n = 0;
        for (int q = 0; q < rows(); ++q) n += per_row(4 * B + q);
        return n;
    }
    static constexpr int COUNT = count();
    // pieces of the blocks before B (all rows < 4B)
    static constexpr int first_global() {
        int n = 0;
        for (int i = 0; i < 4 * B; ++i) n += per_row(i);
        return n;
    }
    static constexpr int row(int r) {
        int i = 4 * B;
        while (r >= per_row(i)) { r -= per_row(i); ++i; }
        return i;
    }
    static constexpr int col(int r) {
        int i = 4 * B;
        while (r >= per_row(i)) { r -= per_row(i); ++i; }
        return UL::j0(i) + r * kPiece;
    }
};

// kPiece columns of row i starting at column c (c - j0(i) is even: 16-byte aligned reads).
//
// The reads are inline assembly and so is the wait for them.  A lone wavefront pays an issue slot (four cycles) for
// every instruction of any kind; left to the compiler, every use of a loaded pair gets a wait with a count of its own
// -- eight per piece, some 600 s_waitcnt per step, 7 % of the step.  LDS reads return in issue order, so ONE
// `s_waitcnt lgkmcnt(n)` placed after the next piece's n reads covers the whole current piece (piece_ready).  The
// compiler does not count reads it cannot see: its own waits then over-wait (safe), these wait for everything older
// (safe whatever else is in flight).
template <int DP, bool FULLU, int i, int c, int k = 0>
__device__ __forceinline__ void load_piece(uint32_t ubase, f64x2 (&dst)[kPiece / 2]) {
    typedef ULayout<DP, FULLU> UL;
    if constexpr (k < kPiece / 2) {
        if constexpr (c + 2 * k < UL::DPE) {
            constexpr int off = (UL::off(i) + (c - UL::j0(i)) + 2 * k) * 8;
            if constexpr (kAsmReads<DP>) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[k]) : "v"(ubase), "n"(off));
            else dst[k] = *(lds_cptr_f64x2)(uintptr_t)(ubase + (uint32_t)off);
        }
        load_piece<DP, FULLU, i, c, k + 1>(ubase, dst);
    }
}
// LDS reads load_piece issues for the piece at column c
template <int DP, bool FULLU, int c>
constexpr int piece_reads() {
    int n = 0;
    for (int k = 0; k < kPiece / 2; ++k) n += (c + 2 * k < ULayout<DP, FULLU>::DPE) ? 1 : 0;
    return n;
}
// every LDS read older than the N youngest has returned; `piece` is the piece those older reads filled (its first M pairs)
template <int N, int M, bool ASM>
__device__ __forceinline__ void piece_ready(f64x2 (&q)[kPiece / 2]) {
    static_assert(kPiece == 16 && N >= 0 && N < 16 && M >= 1 && M <= 8, "eight register pairs per piece, a 4-bit counter");
    if constexpr (!ASM) return;
    else if constexpr (M == 8) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "n"(N));
    else if constexpr (M == 7) asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]) : "n"(N));
    else if constexpr (M == 6) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]) : "n"(N));
    else if constexpr (M == 5) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]) : "n"(N));
    else if constexpr (M == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(N));
    else if constexpr (M == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) : "n"(N));
    else if constexpr (M == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(q[0]), "+v"(q[1]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q[0]) : "n"(N));
}

// The matrix-pipe operands of one k-quad (four chains) of the moment fold, fetched one k-quad ahead: T raw values of the
// accepted point per lane (plus the strip's), inline assembly like the piece reads -- issued in the middle of a piece,
// covered by the next piece's piece_ready, consumed pieces later (operands_ready ties them behind that wait).  Read and
// used on the spot they cost the LDS latency sixteen times per step.
template <int T, int KK, int t = 0>
__device__ __forceinline__ void fetch_operands(const uint32_t (&xaddr)[T], double (&raw)[T]) {
    if constexpr (t < T) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(raw[t]) : "v"(xaddr[t]), "n"(32 * KK));
        fetch_operands<T, KK, t + 1>(xaddr, raw);
    }
}
template <int KK>
__device__ __forceinline__ void fetch_operand(uint32_t addr, double& raw) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(raw) : "v"(addr), "n"(32 * KK));
}
template <int T, int t = 0>
__device__ __forceinline__ void operands_ready(double (&raw)[T], double& raws) {
    if constexpr (t < T) {
        asm volatile("" : "+v"(raw[t]));
        operands_ready<T, t + 1>(raw, raws);
    } else {
        asm volatile("" : "+v"(raws));
    }
}

// row ti of lower-triangular tile number `tile` (tile = ti (ti + 1) / 2 + tj, tj <= ti)
constexpr int tile_row(int tile) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    return ti;
}

struct StepParams {
    int nchains, npad, dim;
    int nsteps, metropolis;
    uint32_t step0;            // fTotalSteps before this launch
    uint32_t chain_offset;
    uint64_t seed;
    const double* U;           // [DP][DP], zero padded
    const double* like;        // QUADFORM: Error [DP][DP] zero padded; ROSENBROCK: {b}
    const int32_t* like_rowptr;   // QUADFORM with a sparse Error: its non-zero entries (QuadCsr below), else nullptr
    const int32_t* like_cols;
    const double* like_vals;
    const int32_t* like_rows;
    const double* c0;          // [DP] centre the moments are taken about
    double target, acc_window, asig, max_up;
    double acc_w, acc_wW;      // acceptance de-weighting: w = 1 - deweight, w*window; acc_w < 0 = off
    int per_lane_update;       // FROZEN mode: per-chain UpdateProposal schedule
    int step_rms_window;
    int has_forced;            // ForceStep pending: the first step proposes `forced`
    const double* forced;      // [DP][npad]
    double* x;                 // [DP][npad], rows >= dim stay zero
    double* lane_f64;          // [SMCMC_LANE_F64_COUNT_][npad]
    int32_t* lane_i32;         // [SMCMC_LANE_I32_COUNT_][npad]
    double* gacc;              // [group][NT][4][64] moment accumulators
    double* save_x;            // optional [slot][DP][npad]
    double* save_logl;         // optional [slot][npad]
    int save_stride;
    uint64_t uniform_mask;     // bit d: dimension d has a uniform proposal (TSimpleMCMC.H:711-716)
    const double* uniform;     // [2][DP]: lower bounds, upper bounds
    int scan_dim;              // fScanDimension (TSimpleMCMC.H:685-704), -1 = off
    int scan_uniform;          // the scanned dimension has a uniform proposal
    double scan_a, scan_b;     // uniform: bounds; Gaussian: centre, sigma
    double* proposed;          // optional [DP][npad]: the proposal of the launch's last step (fProposed, TSimpleMCMC.H:576);
                               // the SPECIAL instantiation and the fused-order kernels look at it
    int zero;                  // always 0; makes table addresses depend on the step so that the
                               // compiler does not hoist (and then spill) whole tables out of the loop
};

__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }

// A user likelihood compiled into the library (SMCMC_LIKE_USER): the header named by
// SMCMC_USER_LIKELIHOOD defines
//     template <int DP> __device__ double smcmc_user_loglike(const double (&p)[DP], smcmc::cptr_f64 params, int D);
// p[0..D) is the point (entries past D are zero), params what smcmc_set_likelihood_params handed over.
// For 63 < dim <= 512 the point does not fit one lane's registers; a header that also defines
//     #define SMCMC_USER_LIKELIHOOD_ANY_DIM 1
//     template <class Point> __device__ double smcmc_user_loglike_at(const Point& p, const double* params, int D);
// (p[i], 0 <= i < D, reads coordinate i of the chain's point from its [dim][chain] image in device memory) is served at
// those dimensions too, in reference-order arithmetic.
#ifdef SMCMC_USER_LIKELIHOOD
}  // namespace smcmc
#include SMCMC_USER_LIKELIHOOD
namespace smcmc {
#endif

// log-likelihood of the point p[0..D) held in registers.
template <int DP, int LIKE, bool EXACT>
__device__ __forceinline__ double loglike(const double (&p)[DP], cptr_f64 prm, int D) {
    double logl = 0.0;
    if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
        // README.md:57-66: logL += - 0.5*p[i]*p[i]   (padded dims add -0)
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            double t = -0.5 * p[i];
            if constexpr (EXACT) logl += t * p[i];
            else logl = SMCMC_FMA(t, p[i], logl);
        }
    } else if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
        // TDummyLogLikelihood.H:24-28: logL -= 0.5*p[i]*Error(j,i)*p[j].  prm holds the
        // TRANSPOSE of Error (prm[i][j] = Error(j,i)) so that the inner loop walks a
        // contiguous row; rows are fetched in 16-column pieces as for U.
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const double h = 0.5 * p[i];
#pragma unroll
            for (int jc = 0; jc < DP; jc += 16) {
                cptr_f64 ep = prm + i * DP + jc;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = jc + jj;
                    if (j < DP) {
                        const double e = ep[jj];
                        if constexpr (EXACT) logl -= h * e * p[j];
                        else logl = SMCMC_FMA(-(h * e), p[j], logl);
                    }
                }
            }
        }
    } else if constexpr (LIKE == SMCMC_LIKE_USER) {
#ifdef SMCMC_USER_LIKELIHOOD
        logl = smcmc_user_loglike<DP>(p, prm, D);
#endif
    } else if constexpr (LIKE == SMCMC_LIKE_ASYM) {
        // TAsymLogLikelihood.H:20-31 (the same arithmetic in both orders: there is nothing to fuse)
        const double positive = prm[0], negative = prm[1];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                double a = p[i];
                a = (a < 0.0) ? a * negative : a * positive;
                logl += a;
            }
        }
    } else if constexpr (LIKE == SMCMC_LIKE_HORRIFIC) {
        // THorrificLogLikelihood.H:26-38: -1E+30 as soon as a coordinate leaves the unit box
        bool outside = false;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                outside = outside || (__builtin_fabs(p[i]) > 1.0);
                logl += p[i];
            }
        }
        const double sigma = 0.01;
        const double natural = __builtin_sqrt(D * 4.0 / 12.0);
        logl /= natural;
        logl = -0.5 * logl * logl / sigma / sigma;
        logl = outside ? -1E+30 : logl;
    } else if constexpr (LIKE == SMCMC_LIKE_CONSTRAINED) {
        // example4/TConstrainedLikelihood.H:26-46; prm = {SummedValues, SummedConstraint, Expected[D], Prior[D]}
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < D) sum += p[i];
        sum = (sum - prm[0]) / prm[1];
        logl -= 0.5 * sum * sum;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                double v = p[i] - prm[2 + i];
                v /= prm[2 + D + i];
                logl -= 0.5 * v * v;
            }
        }
    } else {
        // THardLogLikelihood.H:60-64
        const double rb = prm[0];
#pragma unroll
        for (int i = 0; i < DP - 1; ++i) {
            if (i < D - 1) {
                if constexpr (EXACT) {
                    double a = (1.0 - p[i]);
                    double b = p[i + 1] - p[i] * p[i];
                    logl -= a * a + rb * b * b;
                } else {
                    double a = 1.0 - p[i];
                    double b = SMCMC_FMA(-p[i], p[i], p[i + 1]);
                    double t = SMCMC_FMA(rb * b, b, a * a);
                    logl -= t;
                }
            }
        }
    }
    return logl;
}

// SPECIAL = the variant that also knows uniform per-dimension proposals, the scan of one dimension
// (TSimpleMCMC.H:685-716) and the read-back of the proposed point (:514); kept out of the common kernels, whose
// register allocation it disturbs.
// The quadratic form (TDummyLogLikelihood.H:24-28) of a point held in a column of LDS, xq[j * kXStride]:
// the D^2-term running sum needs every p[j] for every i, which a register array can only give to fully
// unrolled code (50 x 50 terms: minutes of compile time per kernel) -- a rolled outer loop indexes it
// dynamically and the compiler moves it to scratch memory.  From LDS the outer loop can stay rolled.
template <int DP, bool EXACT>
__device__ __forceinline__ double loglike_quadform_lds(const double* xq, cptr_f64 prm) {
    double logl = 0.0;
#pragma nounroll
    for (int i = 0; i < DP; ++i) {
        const double h = 0.5 * xq[i * kXStride];
        cptr_f64 ep = prm + i * DP;
#pragma unroll
        for (int j = 0; j < DP; ++j) {
            const double e = ep[j];
            const double pj = xq[j * kXStride];
            if constexpr (EXACT) logl -= h * e * pj;
            else logl = SMCMC_FMA(-(h * e), pj, logl);
        }
    }
    return logl;
}

// The same sum over the entries of Error that are not zero (rows of Error^T in compressed form, made by the host when
// the matrix is sparse -- the reference's own TDummy matrix is the identity plus one correlated pair: 52 entries of 2 500
// at D = 50).  A skipped term is (h * 0) * p[j] = +-0 for finite h and p[j], and `logl -= +-0` leaves logl as it is (logl
// starts at +0 and x - y is never -0 unless x is): bit for bit the dense sum.  With a non-finite coordinate the skipped
// terms would be NaN -- but then the diagonal term of that coordinate (the host insists on a full diagonal) makes this
// sum non-finite too, and the caller falls back on the dense one.
struct QuadCsr {
    const int32_t* rowptr;     // nullptr: no compressed form, use the dense sum; else rowptr[0] = number of entries, padded to
                               // a multiple of kQuadChunk with zero entries (a zero entry is a skipped term: +-0)
    const int32_t* cols;       // [nnz] column j of entry k (Error^T(i, j) = Error(j, i)); entries row by row, j ascending
    const double* vals;        // [nnz]
    const int32_t* rows;       // [nnz] row i of entry k
};
constexpr int kQuadChunk = 8;  // entries fetched together: three scalar loads per chunk instead of three dependent ones per entry
typedef const __attribute__((address_space(4))) int32_t* cptr_i32;
__device__ __forceinline__ cptr_i32 as_const(const int32_t* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (cptr_i32)p;
#pragma clang diagnostic pop
}
template <bool EXACT, typename Point>
__device__ __forceinline__ double quadform_csr(const Point& point, const QuadCsr& q, int D) {
    const cptr_i32 rw = as_const(q.rows), cl = as_const(q.cols);
    const cptr_f64 vl = as_const(q.vals);
    const int nnz = as_const(q.rowptr)[0];
    double logl = 0.0;
    for (int k0 = 0; k0 < nnz; k0 += kQuadChunk) {
        int ri[kQuadChunk], ci[kQuadChunk];
        double e[kQuadChunk], pi[kQuadChunk], pj[kQuadChunk];
#pragma unroll
        for (int u = 0; u < kQuadChunk; ++u) {
            ri[u] = rw[k0 + u];
            ci[u] = cl[k0 + u];
            e[u] = vl[k0 + u];
        }
#pragma unroll
        for (int u = 0; u < kQuadChunk; ++u) {
            pi[u] = point(ri[u]);
            pj[u] = point(ci[u]);
        }
#pragma unroll
        for (int u = 0; u < kQuadChunk; ++u) {
            const double h = 0.5 * pi[u];
            if constexpr (EXACT) logl -= h * e[u] * pj[u];
            else logl = SMCMC_FMA(-(h * e[u]), pj[u], logl);
        }
    }
    (void)D;
    return logl;
}

template <int DP, int LIKE, bool EXACT, bool FULLU, bool MOMENTS, bool SPECIAL>
__global__ void __launch_bounds__(kWave, 2) step_kernel(const StepParams p) {
    constexpr int T = Geo<DP>::T;
    constexpr int NT = Geo<DP>::NT;
    constexpr int NB = Geo<DP>::NB;
    constexpr int ROWS = MOMENTS ? Geo<DP>::ROWS_MOMENTS : DP;
    typedef ULayout<DP, FULLU> UL;
    __shared__ double xs[ROWS * kXStride];   // accepted point, x[row][lane]
    __shared__ __attribute__((aligned(16))) double us[UL::SIZE];   // decomposition, every lane reads the same word
    __shared__ __attribute__((aligned(16))) double ntab[384];      // tables of the normal transform: log [64][2], angle over the half circle [128][2]

    const int lane = threadIdx.x;
    const int group = blockIdx.x;
    const int chain = group * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    const cptr_f64 c0 = as_const(p.c0);
    const cptr_f64 likep = as_const(p.like);
    double* const xcol = xs + lane;

    // Lanes past the last chain hold x = c0 so that their y = x - c0 is +0 in the
    // moment contraction; they never accept and never store.
#pragma unroll
    for (int d = 0; d < DP; ++d) xcol[d * kXStride] = active ? p.x[(size_t)d * NP + chain] : c0[d];

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

    // stage U: lane-strided copy of each kept row segment
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        for (int k = lane; k < UL::len(i); k += kWave) {
            const int j = UL::j0(i) + k;
            us[UL::off(i) + k] = (j < DP) ? p.U[i * DP + j] : 0.0;
        }
    }

    for (int k = lane; k < 128; k += kWave) ntab[k] = smcmc_log_table_dev[k];
    for (int k = lane; k < 64; k += kWave) {
        // entry 64 + k is entry k turned by pi / 2: (-sin, cos) (SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE)
        const double c = smcmc_angle_table_dev[2 * k], sn = smcmc_angle_table_dev[2 * k + 1];
        ntab[128 + 2 * k] = c;
        ntab[128 + 2 * k + 1] = sn;
        ntab[128 + 128 + 2 * k] = -sn;
        ntab[128 + 128 + 2 * k + 1] = c;
    }
    const uint32_t ltab = (uint32_t)(uintptr_t)(lds_cptr_f64)ntab, atab = ltab + 128u * 8u;   // LDS byte addresses

    constexpr bool STRIP = MOMENTS && Geo<DP>::STRIP;
    constexpr int NT16 = Geo<DP>::NT16;
    f64x4 acc[MOMENTS ? NT16 : 1];
    double accs[STRIP ? T : 1];    // strip accumulators: (row 16 T16 + (lane >> 4), column 16 t + (lane & 15))
    double c0r[MOMENTS ? T : 1];   // c0 of the tile rows this lane feeds to the matrix pipe
    int xrow[MOMENTS ? T : 1];     // LDS row (clamped to the ones row) of those tile rows
    double c0s = 0.0;              // the same for the strip rows 16 T16 + (lane & 3)
    int xsrow = 0;
    if constexpr (MOMENTS) {
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[t][r] = p.gacc[(((size_t)group * NT + t) * 4 + r) * kWave + lane];
        if constexpr (STRIP) {
#pragma unroll
            for (int t = 0; t < T; ++t) accs[t] = p.gacc[(((size_t)group * NT + NT16 + t) * 4) * kWave + lane];
            const int r = 16 * Geo<DP>::T16 + (lane & 3);
            c0s = (r < DP) ? p.c0[r] : 0.0;
            xsrow = ((r <= DP) ? r : DP) * kXStride + (lane >> 4);
        }
        // row DP carries the constant 1 (sum y and the point count come out of the
        // same contraction); tile rows above it are zero and are not stored
        xcol[DP * kXStride] = active ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int r = 16 * t + (lane & 15);
            c0r[t] = (r < DP) ? p.c0[r] : 0.0;
            xrow[t] = ((r <= DP) ? r : DP) * kXStride + (lane >> 4);
        }
    }
    __syncthreads();

    // operand prefetch of the fold (fetch_operands): only where a k-quad's matrix instructions span several pieces, only
    // with the other hand-placed reads (kAsmReads), only in the common kernels (triangular decomposition, no uniform
    // dimensions / scan) and not in the 63-dimension family: there the kernels are short of registers and the compiler
    // spilled the prefetched operands behind their reads in one variant or another (the build's listing check
    // refused them)
    constexpr bool OPF = MOMENTS && T >= 3 && kAsmReads<DP> && DP <= 50 && !FULLU && !SPECIAL;
    double raw[OPF ? T : 1], raws = 0.0;
    uint32_t xaddr[OPF ? T : 1], xsaddr = 0;
    if constexpr (OPF) {
#pragma unroll
        for (int t = 0; t < T; ++t) xaddr[t] = (uint32_t)(uintptr_t)(lds_cptr_f64)(xs + xrow[t]);
        if constexpr (STRIP) xsaddr = (uint32_t)(uintptr_t)(lds_cptr_f64)(xs + xsrow);
    }

    double xp[DP];
    const uint32_t aw = smcmc_accept_word((uint32_t)D);

    // StepRMS window, likelihood, Metropolis test and accept copy of one step
    // (TSimpleMCMC.H:391-406, 410-491) for the proposal held in xp.
    auto finish_step = [&](int s, uint32_t uword) {
        // QUADFORM: the proposal and the accepted point trade places -- the likelihood reads the proposal from
        // the LDS column, the registers keep the accepted point to put back on a reject
        constexpr bool SWAP = (LIKE == SMCMC_LIKE_QUADFORM);
        if constexpr (SPECIAL || !EXACT) {
            // GetProposed() (TSimpleMCMC.H:514): the proposal of the latest step, accepted or not (in the reference
            // order only the SPECIAL instantiation carries the store; the fused kernels all do)
            if (p.proposed != nullptr && s + 1 == p.nsteps && active) {
#pragma unroll
                for (int d = 0; d < DP; ++d) p.proposed[(size_t)d * NP + chain] = xp[d];
            }
        }
        if constexpr (SWAP) {
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const double t = xcol[d * kXStride];
                xcol[d * kXStride] = xp[d];
                xp[d] = t;
            }
        }
        if (p.step_rms_window > 0) {
            double sqr = 0.0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                double t = SWAP ? xcol[d * kXStride] - xp[d] : xp[d] - xcol[d * kXStride];
                if constexpr (EXACT) sqr += t * t;
                else sqr = SMCMC_FMA(t, t, sqr);
            }
            double ms = step_rms * step_rms;
            ms *= rms_trials;
            ms += sqr;
            ms /= rms_trials + 1.0;
            rms_trials = (p.step_rms_window < rms_trials + 1) ? p.step_rms_window : rms_trials + 1;
            step_rms = __builtin_sqrt(ms);
        }
        if constexpr (SWAP) {
            bool dense = p.like_rowptr == nullptr;
            if (!dense) {
                const QuadCsr csr = {p.like_rowptr + p.zero * (s + 1), p.like_cols, p.like_vals, p.like_rows};
                logl_prop = quadform_csr<EXACT>([&](int j) { return xcol[j * kXStride]; }, csr, D);
                dense = __any(!__builtin_isfinite(logl_prop));
            }
            if (dense) logl_prop = loglike_quadform_lds<DP, EXACT>(xcol, likep + p.zero * (s + 1));
        }
        else logl_prop = loglike<DP, LIKE, EXACT>(xp, likep + p.zero * (s + 1), D);
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
        last_accept = take ? 1 : 0;
        if (take) {
            logl = logl_prop;
            ++naccept;
        }
        if (take != SWAP) {   // plain: commit the proposal on accept; swapped: put the accepted point back on reject
#pragma unroll
            for (int d = 0; d < DP; ++d) xcol[d * kXStride] = xp[d];
        }
        if constexpr (MOMENTS) __syncthreads();   // the next step's operand reads see this step's accepts
        if (p.save_x != nullptr && ((s + 1) % p.save_stride) == 0 && active) {
            const size_t slot = (size_t)((s + 1) / p.save_stride - 1);
#pragma unroll
            for (int d = 0; d < DP; ++d) p.save_x[(slot * (size_t)DP + (size_t)d) * NP + chain] = xcol[d * kXStride];
            p.save_logl[slot * NP + chain] = logl;
        }
    };

    int s0 = 0;
    if (p.has_forced && p.nsteps > 0) {
        // ForceStep (TSimpleMCMC.H:671-678): the proposal is the forced point and the
        // proposal state is not updated
#pragma unroll
        for (int d = 0; d < DP; ++d) xp[d] = p.forced[(size_t)d * NP + chain];
        smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, (uint64_t)(p.step0 + 1u), aw >> 2, SMCMC_STREAM_STEP);
        finish_step(0, smcmc_select_word(blk, aw & 3u));
        s0 = 1;
    }

    if (SPECIAL && p.scan_dim >= 0) {
        // scan of one dimension (TSimpleMCMC.H:685-704): the proposal is the current point
        // with that dimension redrawn; the proposal state is not updated
        const uint32_t sd = (uint32_t)p.scan_dim;
#pragma nounroll
        for (int s = s0; s < p.nsteps; ++s) {
            const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);
            const smcmc_u32x4 sblk = smcmc_draw_block(p.seed, gid, step, sd >> 2, SMCMC_STREAM_STEP);
            double val;
            if (p.scan_uniform) {
                val = p.scan_a + (p.scan_b - p.scan_a) * smcmc_u01(smcmc_select_word(sblk, sd & 3u));
            } else {
                double nc, ns;
                smcmc_normal_pair(smcmc_select_word(sblk, sd & 2u), smcmc_select_word(sblk, (sd & 2u) + 1u), &nc, &ns);
                val = p.scan_a + p.scan_b * ((sd & 1u) ? ns : nc);
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) xp[d] = ((uint32_t)d == sd) ? val : xcol[d * kXStride];
            const smcmc_u32x4 ablk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            finish_step(s, smcmc_select_word(ablk, aw & 3u));
        }
        s0 = p.nsteps;
    }

    for (int s = s0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);   // ++fTotalSteps, :376

        // ---- A: UpdateState, scalar half (TSimpleMCMC.H:1723-1776) ----
        ++trials;
        const double x0 = xcol[0];
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
            sigma *= smcmc_pow_small(acc_rate / p.target,
                                     dmin(1.0 / 500.0, 1.0 / (rigid * p.acc_window)));
        }
        if constexpr (!MOMENTS) {
            if (p.per_lane_update && moved && (--next_update) < 1) {
                // per-chain UpdateProposal on a frozen covariance (TSimpleMCMC.H:1824-1826,
                // 1050-1052, 1081-1086); trace unchanged => sigma and U unchanged
                double up = 0.5 * succ;
                next_update = (int)(p.acc_window + p.max_up - p.max_up / (up + 1.0));
                if (p.acc_w >= 0.0) {
                    acc_trials = dmax(1.0, p.acc_w * acc_trials);
                    acc_trials = dmin(acc_trials, p.acc_wW);
                }
            }
        }
        last_value = logl;
        last_x0 = x0;

        // ---- B: proposal (TSimpleMCMC.H:709-724) ----
        if constexpr (OPF) {
            fetch_operands<T, 0>(xaddr, raw);
            if constexpr (STRIP) fetch_operand<0>(xsaddr, raws);
        }
#pragma unroll
        for (int d = 0; d < DP; ++d) xp[d] = xcol[d * kXStride];
        uint32_t uword = 0;
        // The LDS image of U never changes, so the compiler would hoist all of its reads out
        // of the step loop and spill them; reading through a pointer it cannot see through
        // keeps them inside the step.
        uint32_t up = (uint32_t)(uintptr_t)(lds_cptr_f64)us;   // LDS byte address of the decomposition
        asm volatile("" : "+v"(up));
        double ma[MOMENTS ? T : 1];   // matrix-pipe operands of the chain quad being folded
        double ms = 0.0;              // ... and the strip rows' operand
        // The Philox block and the table entries of the NEXT block of four normals are made during the last piece of the
        // current one (kAsmReads: the table reads then have a whole piece to come back in, and sit in the registers the
        // idle `nxt` buffer leaves free); two sets in turn, indexed at compile time -- never copied while in flight.
        constexpr bool PIPE = kAsmReads<DP>;
        smcmc_u32x4 blks[2];
        NormalTables tabs[2][2];
        auto draw_and_fetch = [&](auto bn) {
            constexpr int b1 = decltype(bn)::value;
            blks[b1 & 1] = smcmc_draw_block(p.seed, gid, step, (uint32_t)b1, SMCMC_STREAM_STEP);
            tabs[b1 & 1][0] = normal_tables_fetch<kAsmReads<DP>>(blks[b1 & 1].v[0], blks[b1 & 1].v[1], ltab, atab);
            if constexpr (4 * b1 + 2 < DP)
                tabs[b1 & 1][1] = normal_tables_fetch<kAsmReads<DP>>(blks[b1 & 1].v[2], blks[b1 & 1].v[3], ltab, atab);
        };
        if constexpr (PIPE) draw_and_fetch(std::integral_constant<int, 0>{});
        static_for<NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            // U rows 4b..4b+3 are consumed in pieces of kPiece columns; the LDS reads of a
            // piece are issued one piece ahead of its use (the first one before the random
            // numbers are made), and each piece is its own scheduling region: the scheduler
            // otherwise issues every read of the step up front and spills hundreds of registers.
            typedef UPieces<DP, FULLU, b> PC;
            f64x2 cur[kPiece / 2], nxt[kPiece / 2];

            if constexpr (!PIPE) draw_and_fetch(bc);
            smcmc_u32x4& blk = blks[b & 1];
            if ((uint32_t)b == (aw >> 2)) uword = smcmc_select_word(blk, aw & 3u);
            // LDS reads of the block, in this order (they return in issue order): the table entries of its two pairs of
            // normals (issued here, or during the last piece of the block before), then the first piece of U, which
            // stays in flight under the normals' arithmetic
            NormalTables& t0 = tabs[b & 1][0];
            NormalTables& t1 = tabs[b & 1][1];
            {
                constexpr int i0 = PC::row(0), c0p = PC::col(0);
                load_piece<DP, FULLU, i0, c0p>(up, cur);
                if constexpr (4 * b + 2 < DP) normal_tables_ready<piece_reads<DP, FULLU, c0p>(), kAsmReads<DP>>(t0, t1);
                else normal_tables_ready<piece_reads<DP, FULLU, c0p>(), kAsmReads<DP>>(t0);
            }
            double n[4];
            normal_pair_lds(blk.v[0], blk.v[1], t0, &n[0], &n[1]);
            if constexpr (4 * b + 2 < DP) normal_pair_lds(blk.v[2], blk.v[3], t1, &n[2], &n[3]);
            else { n[2] = 0.0; n[3] = 0.0; }
            double sr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sr[q] = sigma * n[q];
            __builtin_amdgcn_sched_barrier(0);

            static_for<PC::COUNT>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int i = PC::row(r), c = PC::col(r);
                if constexpr (r + 1 < PC::COUNT) {
                    constexpr int i1 = PC::row(r + 1), c1 = PC::col(r + 1);
                    load_piece<DP, FULLU, i1, c1>(up, nxt);
                    piece_ready<piece_reads<DP, FULLU, c1>(), piece_reads<DP, FULLU, c>(), kAsmReads<DP>>(cur);
                } else {
                    piece_ready<0, piece_reads<DP, FULLU, c>(), kAsmReads<DP>>(cur);
                    // the last piece of the block: the next block's random words and its table reads
                    if constexpr (PIPE && b + 1 < NB) draw_and_fetch(std::integral_constant<int, b + 1>{});
                }
                const double srow = sr[i - 4 * b];
#pragma unroll
                for (int k = 0; k < kPiece; ++k) {
                    const int j = c + k;
                    if (j < DP && (FULLU || j >= i)) {
                        const double u = cur[k / 2][k & 1];
                        if constexpr (EXACT) xp[j] += srow * u;
                        else xp[j] = SMCMC_FMA(srow, u, xp[j]);
                        // pins this update between the (ordered) LDS reads around it; without it
                        // the arithmetic of the whole step sinks below all the reads, which then
                        // have to be spilled
                        asm volatile("" : "+v"(xp[j]));
                    }
                }
                if constexpr (MOMENTS) {
                    // The group's second moments: the 16 x NT matrix instructions of the 64-chain contraction are
                    // dealt out over the pieces of the step (1-2 per piece) and pinned BEHIND the piece's vector work.
                    // An FP64 matrix instruction holds the vector pipe for its 64 cycles but not the issue of LDS
                    // reads, waits and scalar instructions: what follows it here is exactly that -- the prefetched
                    // operands of the next k-quad, then the next piece's eight reads and its wait -- and runs in
                    // its shadow.  (Left to the scheduler it goes to the top of the piece, in front of 32 vector
                    // instructions that then wait for it.)  Chains fold in ascending order.
                    constexpr int g = PC::first_global() + r;          // piece number within the step
                    constexpr int G = UPieces<DP, FULLU, NB - 1>::first_global() + UPieces<DP, FULLU, NB - 1>::COUNT;
                    constexpr int NM = 16 * NT;
                    constexpr int m_lo = (int)(((long)g * NM) / G), m_hi = (int)(((long)(g + 1) * NM) / G);
                    if constexpr (m_hi > m_lo) __builtin_amdgcn_sched_barrier(0);
                    static_for<m_hi - m_lo>([&](auto mc) {
                        constexpr int m = m_lo + decltype(mc)::value;
                        constexpr int kk = m / NT, tile = m % NT;
                        if constexpr (tile == 0) {
                            if constexpr (OPF) operands_ready<T>(raw, raws);   // fetched a k-quad ago, a piece_ready since
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                if constexpr (OPF) ma[t] = raw[t] - c0r[t];
                                else ma[t] = xs[xrow[t] + 4 * kk] - c0r[t];
                                if (16 * t + 15 > DP) ma[t] = (16 * t + (lane & 15) <= DP) ? ma[t] : 0.0;   // rows past the ones row
                            }
                            if constexpr (STRIP) {
                                if constexpr (OPF) ms = raws - c0s;
                                else ms = xs[xsrow + 4 * kk] - c0s;
                                ms = (16 * Geo<DP>::T16 + (lane & 3) <= DP) ? ms : 0.0;
                            }
                        }
                        if constexpr (!STRIP || tile < NT16) {
                            constexpr int ti = tile_row(tile), tj = tile - ti * (ti + 1) / 2;
                            acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[ti], ma[tj], acc[tile], 0, 0, 0);
                            asm volatile("" : "+a"(acc[tile]));   // keeps the instruction in this piece
                        } else {
                            // A: strip rows (lane & 3) x chains (lane >> 4), the same for the four blocks;
                            // B: the column operand of tile column t, block (lane >> 2) & 3
                            constexpr int t = tile - NT16;
                            accs[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(ms, ma[t], accs[t], 0, 0, 0);
                            asm volatile("" : "+a"(accs[t]));
                        }
                        if constexpr (tile == 0 && OPF && kk + 1 < 16) {
                            // (behind the k-quad's first matrix instruction: in its shadow) the next k-quad's first
                            // matrix instruction sits in a later piece
                            static_assert(((long)(kk + 1) * NT * G) / NM > g, "operand prefetch needs a piece boundary");
                            fetch_operands<T, kk + 1>(xaddr, raw);
                            if constexpr (STRIP) fetch_operand<kk + 1>(xsaddr, raws);
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (r + 1 < PC::COUNT) {
#pragma unroll
                    for (int k = 0; k < kPiece / 2; ++k) cur[k] = nxt[k];
                }
            });
        });
        if ((aw >> 2) >= (uint32_t)NB) {
            smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
            uword = smcmc_select_word(blk, aw & 3u);
        }
        if (SPECIAL && p.uniform_mask != 0) {
            // uniform dimensions (TSimpleMCMC.H:711-716): word d of the step makes the draw; their
            // rows and columns of the device copy of U are zero, so nothing else touched them
            const cptr_f64 ub = as_const(p.uniform);
            uint64_t m = p.uniform_mask;
#pragma nounroll
            while (m != 0) {
                const uint32_t ud = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const smcmc_u32x4 ublk = smcmc_draw_block(p.seed, gid, step, ud >> 2, SMCMC_STREAM_STEP);
                const double lo = ub[ud], hi = ub[DP + ud];
                const double val = lo + (hi - lo) * smcmc_u01(smcmc_select_word(ublk, ud & 3u));
#pragma unroll
                for (int d = 0; d < DP; ++d) xp[d] = ((uint32_t)d == ud) ? val : xp[d];
            }
        }
        finish_step(s, uword);
    }

    if constexpr (MOMENTS) {
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.gacc[(((size_t)group * NT + t) * 4 + r) * kWave + lane] = acc[t][r];
        if constexpr (STRIP) {
#pragma unroll
            for (int t = 0; t < T; ++t) p.gacc[(((size_t)group * NT + NT16 + t) * 4) * kWave + lane] = accs[t];
        }
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < DP; ++d) p.x[(size_t)d * NP + chain] = xcol[d * kXStride];
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

// Moment reduction, two ordered levels (the order is part of the engine's definition and
// is mirrored by oracle/ensemble_oracle.c): groups are summed in ascending order within
// chunks of kReduceChunk groups, then the chunk sums in ascending order.
constexpr int kReduceChunk = 32;

// offset of packed element k = (i, j), j <= i <= D, inside one group's tiles
template <int DP>
__device__ __forceinline__ size_t packed_to_tile_offset(int k, int D) {
    int i = (int)((__builtin_sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= k) ++i;
    while (i * (i + 1) / 2 > k) --i;
    const int j = k - i * (i + 1) / 2;
    const int ri = (i == D) ? DP : i, rj = (j == D) ? DP : j;   // the ones row sits at row DP of the tiles
    const int ti = ri >> 4, tj = rj >> 4, ii = ri & 15, jj = rj & 15;
    // C/D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4*reg
    const int reg = ii >> 2;
    const int lane = jj + 16 * (ii & 3);
    return ((size_t)(ti * (ti + 1) / 2 + tj) * 4 + reg) * kWave + lane;
}

// level 1: thread (k, chunk) -> chunk_sums[chunk][k]; it leaves ZERO in the accumulator entries it read (the next window
// accumulates into them -- rounds 1-3 cleared all 21 MB of accumulators with a memset per window; entries nobody reads,
// the lower halves of the diagonal tiles, keep growing unread).
// (Both levels in one launch -- the last workgroup of a range of k adding the chunk sums -- was built and measured:
// 36 us instead of 14.  Workgroups on different XCDs only see each other's stores after an agent-scope release, which
// on this chip writes the XCD's L2 back, once per workgroup.)
template <int DP>
__global__ void reduce_chunks_kernel(double* __restrict__ gacc, int ngroups, int D, double* __restrict__ chunk_sums) {
    constexpr int NT = Geo<DP>::NT;
    const int npk = (D + 1) * (D + 2) / 2;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int chunk = blockIdx.y;
    if (k >= npk) return;
    const size_t off = packed_to_tile_offset<DP>(k, D);
    const size_t gstride = (size_t)NT * 4 * kWave;
    const int g0 = chunk * kReduceChunk;
    const int g1 = (g0 + kReduceChunk < ngroups) ? g0 + kReduceChunk : ngroups;
    double v[kReduceChunk];
#pragma unroll
    for (int q = 0; q < kReduceChunk; ++q) v[q] = (g0 + q < g1) ? gacc[(size_t)(g0 + q) * gstride + off] : 0.0;
#pragma unroll
    for (int q = 0; q < kReduceChunk; ++q)
        if (g0 + q < g1) gacc[(size_t)(g0 + q) * gstride + off] = 0.0;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < kReduceChunk; ++q)
        if (g0 + q < g1) s += v[q];
    chunk_sums[(size_t)chunk * npk + k] = s;
}

// level 2: thread k -> moments[k]
template <int DP>
__global__ void reduce_final_kernel(const double* __restrict__ chunk_sums, int nchunks, int npk,
                                    double* __restrict__ moments) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npk) return;
    // (the loads go out in batches of 16 -- one at a time this launch was 9 us of load latency --, the additions stay in
    // chunk order)
    double s = 0.0;
    for (int c0 = 0; c0 < nchunks; c0 += 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (c0 + q < nchunks) ? chunk_sums[(size_t)(c0 + q) * npk + k] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (c0 + q < nchunks) s += v[q];
    }
    moments[k] = s;
}

// Host-callable launchers, one translation unit per (DP, likelihood) (smcmc_inst.hip).
template <int DP, int LIKE> hipError_t launch_step_like(const StepParams& p, bool exact, bool fullu, bool moments,
                                                        bool special, hipStream_t stream);
template <int DP> hipError_t launch_reduce(double* gacc, int ngroups, int D, double* chunk_sums, double* moments,
                                           hipStream_t stream);

template <int DP>
inline hipError_t launch_step(const StepParams& p, int like, bool exact, bool fullu, bool moments, bool special,
                              hipStream_t stream) {
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return launch_step_like<DP, SMCMC_LIKE_ISO_GAUSS>(p, exact, fullu, moments, special, stream);
        case SMCMC_LIKE_QUADFORM: return launch_step_like<DP, SMCMC_LIKE_QUADFORM>(p, exact, fullu, moments, special, stream);
        case SMCMC_LIKE_ROSENBROCK: return launch_step_like<DP, SMCMC_LIKE_ROSENBROCK>(p, exact, fullu, moments, special, stream);
        // the stress likelihoods are built for two register-array sizes only (kStressDP)
        case SMCMC_LIKE_ASYM:
            if constexpr (DP == 31 || DP == 63) return launch_step_like<DP, SMCMC_LIKE_ASYM>(p, exact, fullu, moments, special, stream);
            else return hipErrorInvalidValue;
        case SMCMC_LIKE_HORRIFIC:
            if constexpr (DP == 31 || DP == 63) return launch_step_like<DP, SMCMC_LIKE_HORRIFIC>(p, exact, fullu, moments, special, stream);
            else return hipErrorInvalidValue;
        case SMCMC_LIKE_CONSTRAINED:
            if constexpr (DP == 31 || DP == 63) return launch_step_like<DP, SMCMC_LIKE_CONSTRAINED>(p, exact, fullu, moments, special, stream);
            else return hipErrorInvalidValue;
#ifdef SMCMC_USER_LIKELIHOOD
        case SMCMC_LIKE_USER: return launch_step_like<DP, SMCMC_LIKE_USER>(p, exact, fullu, moments, special, stream);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
