// smcmc_perchain_wave.hip.h -- SMCMC_MODE_PER_CHAIN with ONE CHAIN PER WAVEFRONT: the reference's own mode (every
// chain its own running centre, covariance and decomposition, UpdateState every step, TSimpleMCMC.H:1721-1831, and
// UpdateProposal on the chain's own schedule, :1009-1106) for ensembles too small to fill the chip with one chain per
// lane -- down to the single chain of SimpleMCMC.C (BASELINE config 1).
//
// perchain_step_kernel (smcmc_perchain_kernel.hip.h) gives a chain one lane and streams its O(D^2) state through HBM
// every step: 33 KB per chain-step at D = 50, HBM-bound from ~40 000 chains up, and below that one wavefront's own pace
// (233 us per step at D = 50 whatever the number of chains: 4 096 chains leave 15 of 16 SIMDs idle, one chain runs at
// 4 300 steps/s against 3.7e5 on one host core).  Here the 64 lanes of a wavefront share ONE chain and the chain's
// state stays on chip for the whole launch:
//   * the packed covariance in registers, element k = lane + 64 r (20 per lane at D = 50); its update -- multiply,
//     multiply, add, DIVIDE per element, as the reference has it -- is 20 independent elements per lane;
//   * the decomposition in registers too, lane j its column U(0 .. j, j); lane j builds
//     x'[j] = x[j] + sum_{i <= j} (sigma r_i) U(i, j), i ascending, un-fused -- the reference's order -- with sigma r_i
//     read from lane i by v_readlane;
//   * lane i owns x[i], the running centre c[i], fLastPoint[i] and draws r_i;
//   * what the reference sums in index order (the trial step's square sum, the likelihood, the covariance trace): every
//     lane forms its own term, and the terms are added one after the other in that order by every lane alike
//     (v_readlane, no memory: the dependent chain is the additions alone);
//   * UpdateProposal's Cholesky runs in place on the LDS image (lane = column, the host's row-ordered A = U^T U in the
//     host's order of roundings); a failed pivot stops the chain for the host's ladder exactly as in the other kernel.
// The HBM images, the per-chain scalar columns and the stop / resume protocol are those of perchain_step_kernel (a
// launch reads the chain's state at its start and writes it back at its end), so the host side -- ladder, broadcast,
// restore, getters -- does not know which kernel ran, and every chain is bit for bit oracle.Chain either way.
//
// A launch can leave a per-step record of one chain (PerChainRecord): what TSimpleMCMC::Step() shows its caller after
// every step (fAccepted, fProposed, the two likelihoods, StepRMS, the accept flag, the Adaptive* scalars SaveStep
// fills).  include/TSimpleMCMC_amd.H runs Step() ahead in launches of many steps and serves the calls from it.
//
// Reference-order arithmetic only (compile with -ffp-contract=off).
#pragma once

#include "smcmc_perchain_kernel.hip.h"

namespace smcmc {

// Per-step record of one chain: rec[(step - step0 - 1) * stride + ...]: [0, D) fAccepted, [D, 2 D) fProposed, [2 D, 3 D) the
// diagonal of the covariance (GetCovarianceTrace is its sum in index order: the reader adds it up when somebody asks),
// then the scalars below.
enum {
    kPcRecLogl = 0, kPcRecLoglProposed, kPcRecStepRms, kPcRecLastAccept, kPcRecTrials, kPcRecSuccesses, kPcRecNextUpdate,
    kPcRecAcceptance, kPcRecAcceptanceTrials, kPcRecSigma, kPcRecCenterTrials, kPcRecCovarianceTrials, kPcRecTrace,
    kPcRecTotalSteps, kPcRecStatus, kPcRecScalars
};
static_assert(kPcRecScalars == SMCMC_REC_COUNT_ && kPcRecTrace == SMCMC_REC_COVARIANCE_TRACE && kPcRecStatus == SMCMC_REC_UPDATE_STATUS,
              "the record of smcmc_step_recorded (smcmc.h)");
struct PerChainRecord {
    double* rec;     // nullptr: no record
    int chain;
    int stride;      // doubles per step: >= 3 dim + kPcRecScalars
};

constexpr int kPwBatch = 8;          // LDS values read ahead of the additions of an ordered sum

// registers of packed covariance per lane for dimension D
inline int perchain_wave_elements(int D) {
    const int need = (D * (D + 1) / 2 + kWave - 1) / kWave;
    return need <= 4 ? 4 : need <= 12 ? 12 : need <= 20 ? 20 : 32;
}

// lane i's value of v in every lane (i wave-uniform): two v_readlane_b32
__device__ __forceinline__ double pw_readlane(double v, int i) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), i);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}

// What the reference's loops over the dimensions do -- s = t[0]; s += t[1]; ... one addition after the other in index
// order -- for terms that sit one per lane.  tools/micro/ordered_sum.hip (cycles per term, one wavefront): the chain of
// dependent v_add_f64 alone 5; terms through v_readlane 21 (compile-time lane) to 42 (run-time lane); through LDS, every
// lane reading every term (broadcast), 19 in batches of eight and 12 with ALL the reads issued ahead of the chain.  So:
// the terms go to LDS, every lane fetches all DMAX of them (ds_read_b128) and then adds.  Entries from D on are +0.0
// (the caller's): a sum that starts at +0.0 is never -0.0, so adding them changes nothing, and the chain needs no
// per-term condition.
template <int DMAX>
__device__ __forceinline__ void pw_fetch_all(const double* arr, double (&v)[(DMAX + 1) / 2 * 2]) {
    const lds_cptr_f64x2 a2 = (lds_cptr_f64x2)(uintptr_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double*)arr;
#pragma unroll
    for (int k = 0; k < (DMAX + 1) / 2; ++k) {
        const f64x2 w = a2[k];
        v[2 * k] = w[0];
        v[2 * k + 1] = w[1];
    }
}
template <int DMAX>
__device__ __forceinline__ double pw_sum_lds(const double* arr, int D, double s0 = 0.0) {
    double v[(DMAX + 1) / 2 * 2];
    pw_fetch_all<DMAX>(arr, v);
    double s = s0;
#pragma unroll
    for (int i0 = 0; i0 < DMAX; i0 += 8) {
        if (i0 < D) {                   // wave-uniform; the chunk's entries from D on are +0.0
#pragma unroll
            for (int i = i0; i < i0 + 8 && i < DMAX; ++i) s += v[i];
        }
    }
    return s;
}
// t: this lane's term; lanes from n on contribute nothing.  scratch: kWave doubles of LDS, 16-byte aligned.
template <int DMAX>
__device__ __forceinline__ double pw_ordered_sum(double t, int n, double* scratch, double s0 = 0.0) {
    __syncthreads();
    scratch[threadIdx.x] = ((int)threadIdx.x < n) ? t : 0.0;
    __syncthreads();
    return pw_sum_lds<DMAX>(scratch, n, s0);
}

// log L of the point in LDS (p[0 .. D)), in the reference's summation order: the arithmetic of serial_loglike<LIKE, true>
// (pi: this lane's coordinate of the same point)
template <int LIKE, int DMAX>
__device__ __forceinline__ double pw_loglike(const double* p, double pi, int D, const double* __restrict__ like, const QuadCsr& csr,
                                             double* scratch) {
    double lsum = 0.0;
    if constexpr (LIKE == SMCMC_LIKE_ISO_GAUSS) {
        // README.md:57-66: logL += -0.5 p[i] p[i], i ascending: every lane its own term, the terms added in order
        const double t = -0.5 * pi;
        lsum = pw_ordered_sum<DMAX>(t * pi, D, scratch);
    } else if constexpr (LIKE == SMCMC_LIKE_QUADFORM) {
        bool dense = csr.rowptr == nullptr;
        if (!dense) {
            lsum = quadform_csr<true>([&](int j) { return p[j]; }, csr, D);
            dense = !__builtin_isfinite(lsum);
        }
        if (dense) {
            // TDummyLogLikelihood.H:24-28: logL -= 0.5 p[i] Error(j, i) p[j], i outer, j inner, un-fused
            const cptr_f64 et = as_const(like);
            lsum = 0.0;
            for (int i = 0; i < D; ++i) {
                const double h = 0.5 * p[i];
                const cptr_f64 erow = et + (size_t)i * D;
                int j = 0;
                for (; j + kPwBatch <= D; j += kPwBatch) {
                    double pj[kPwBatch];
#pragma unroll
                    for (int u = 0; u < kPwBatch; ++u) pj[u] = p[j + u];
#pragma unroll
                    for (int u = 0; u < kPwBatch; ++u) lsum -= h * erow[j + u] * pj[u];
                }
                for (; j < D; ++j) lsum -= h * erow[j] * p[j];
            }
        }
    } else if constexpr (LIKE == SMCMC_LIKE_ASYM) {
        // TAsymLogLikelihood.H:20-31: logL += p[i] * (p[i] < 0 ? like[1] : like[0]), i ascending
        lsum = pw_ordered_sum<DMAX>((pi < 0.0) ? pi * like[1] : pi * like[0], D, scratch);
    } else if constexpr (LIKE == SMCMC_LIKE_HORRIFIC) {
        // THorrificLogLikelihood.H:26-38 (the arithmetic of serial_loglike<HORRIFIC, true>)
        const bool outside = __any(((int)threadIdx.x < D) && (__builtin_fabs(pi) > 1.0)) != 0;
        lsum = pw_ordered_sum<DMAX>(pi, D, scratch);
        const double sigma = 0.01;
        lsum /= __builtin_sqrt(D * 4.0 / 12.0);
        lsum = -0.5 * lsum * lsum / sigma / sigma;
        lsum = outside ? -1E+30 : lsum;
    } else if constexpr (LIKE == SMCMC_LIKE_CONSTRAINED) {
        // example4/TConstrainedLikelihood.H:26-46; like = {SummedValues, SummedConstraint, Expected[D], Prior[D]}
        double sum = pw_ordered_sum<DMAX>(pi, D, scratch);
        sum = (sum - like[0]) / like[1];
        lsum -= 0.5 * sum * sum;
        const int il = ((int)threadIdx.x < D) ? (int)threadIdx.x : 0;
        double v = pi - like[2 + il];
        v /= like[2 + D + il];
        lsum = pw_ordered_sum<DMAX>(-(0.5 * v * v), D, scratch, lsum);     // x - t and x + (-t) are the same rounding
#ifdef SMCMC_USER_LIKELIHOOD
    } else if constexpr (LIKE == SMCMC_LIKE_USER) {
        // the user's function of the whole point (smcmc_user_loglike<DP>, smcmc_kernels.hip.h): every lane takes the point
        // from LDS (p, zero past D) into a register array and evaluates it -- 64 times the same value, one evaluation's time
        constexpr int DPU = (DMAX + 1) / 2 * 2;
        double v[DPU];
        pw_fetch_all<DMAX>(p, v);
        lsum = smcmc_user_loglike<DPU>(v, as_const(like), D);
#endif
    } else {
        static_assert(LIKE == SMCMC_LIKE_ROSENBROCK, "the likelihoods the wave kernel serves");
        // THardLogLikelihood.H:57-67: logL -= a a + 100 b b with a = 1 - p[i], b = p[i + 1] - p[i]^2, i ascending:
        // lane i its own term (its neighbour's coordinate by a lane shift), the terms subtracted in order
        const double rb = like[0];
        const double nx = __shfl_down(pi, 1);
        const double a = (1.0 - pi);
        const double b = nx - pi * pi;
        const double term = a * a + rb * b * b;
        lsum = pw_ordered_sum<DMAX>(-term, D - 1, scratch);     // x - t and x + (-t) are the same rounding
    }
    return lsum;
}

// largest dimension of a class of packed-covariance registers (perchain_wave_elements): the registers of a lane's column
// of the decomposition
template <int NE>
constexpr int kPwMaxDim = NE == 4 ? 22 : NE == 12 ? 38 : NE == 20 ? 50 : 63;

// grid = nchains workgroups of one wavefront.  NE: registers of packed covariance per lane (perchain_wave_elements).
// PAIRED (NE = 4, D <= 22, ensembles of more wavefronts than the chip has SIMDs): held to 256 registers so that two
// wavefronts share a SIMD: + 11 - 17 % from 16 384 chains on, and 4 % slower for a single chain (hence the switch).  The
// larger classes pay 3 - 5 x for that (336 - 1 179 scratch accesses): never paired.
template <int LIKE, int NE, bool PAIRED = false>
__attribute__((amdgpu_waves_per_eu(PAIRED ? 2 : 1)))
__global__ void __launch_bounds__(kWave) perchain_wave_kernel(const PerChainParams p, const PerChainRecord rec) {
    constexpr int DMAX = kPwMaxDim<NE>;
    __shared__ double ul[2048];          // UpdateProposal's workspace: the decomposition, column packed (U(i, j), i <= j, at j (j + 1) / 2 + i)
    __shared__ __attribute__((aligned(16))) double dv[kWave];   // x - c of UpdateState; the diagonal of the covariance while a trace is summed
    __shared__ __attribute__((aligned(16))) double zv[kWave];   // sigma r_i of the step, for every lane to read
    __shared__ __attribute__((aligned(16))) double sv[kWave];   // the terms of an ordered sum
    __shared__ __attribute__((aligned(16))) double ps[kWave];   // QUADFORM / USER: the proposal where every lane can read all of it
    // One step's record (PerChainRecord) is put together here and leaves in three coalesced stores; behind it one slot
    // per lane for the writes of the elements that are not on the diagonal (so that no write needs a predicate).
    constexpr int kRecDump = 3 * kWave + 16;
    __shared__ __attribute__((aligned(16))) double recl[kRecDump + kWave];
    __shared__ __attribute__((aligned(16))) double ntab[384];   // tables of the normal transform (as in step_kernel)

    const int lane = threadIdx.x;
    const int chain = blockIdx.x;
    if (chain >= p.nchains) return;
    const int D = p.dim;
    const int npk = D * (D + 1) / 2;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    const bool mine = lane < D;          // this lane owns a coordinate
    const size_t own = (size_t)(mine ? lane : D - 1) * NP + chain;

    for (int k = lane; k < 128; k += kWave) ntab[k] = smcmc_log_table_dev[k];
    {
        // entry 64 + k is entry k turned by pi / 2: (-sin, cos) (SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE)
        const double c = smcmc_angle_table_dev[2 * lane], sn = smcmc_angle_table_dev[2 * lane + 1];
        ntab[128 + 2 * lane] = c;
        ntab[128 + 2 * lane + 1] = sn;
        ntab[128 + 128 + 2 * lane] = -sn;
        ntab[128 + 128 + 2 * lane + 1] = c;
    }
    const uint32_t ltab = (uint32_t)(uintptr_t)(lds_cptr_f64)ntab, atab = ltab + 128u * 8u;   // LDS byte addresses

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

    // the chain's state on chip
    double xi = p.x[own], ci = p.centre[own], lasti = p.last_point[own], xpi = p.proposed[own];
    double cov[NE];
    uint32_t ij[NE];                     // (row << 8 | column) of element k = lane + 64 r; bits 16..: where the element goes in
                                         // recl when a step is recorded (its slot of the diagonal, or the lane's dump slot)
    // Which of this lane's registers hold an element at all, and which a diagonal one: as BIT masks, tested where they
    // are used.  (As conditions on `lane + 64 r < npk` they are loop-invariant lane masks: the compiler keeps all 2 NE of
    // them in SGPR pairs across the step loop, runs out of SGPRs and moves them through VGPR lanes at every use --
    // a thousand v_readlane / v_writelane in the first build.)  A register past the end computes on two entries of LDS
    // nobody writes; it is never stored.
    uint32_t valid_bits = 0, diag_bits = 0;
#pragma unroll
    for (int r = 0; r < NE; ++r) {
        const int k = lane + kWave * r;
        int i = 62, j = 63;
        if (k < npk) {
            pc_unpack(k, i, j);
            valid_bits |= 1u << r;
            if (i == j) diag_bits |= 1u << r;
        }
        ij[r] = (uint32_t)(i << 8 | j) | (uint32_t)((k < npk && i == j) ? 2 * D + i : kRecDump + lane) << 16;
        cov[r] = (k < npk) ? p.cov[pc_tile_index(k, (size_t)chain, npk)] : 0.0;
    }
    const int colj = lane * (lane + 1) / 2;    // start of this lane's column of the decomposition
    // This lane's column of the decomposition, U(i, lane) for i <= lane, and ZERO above: the proposal loop then needs no
    // predicate (a term sigma r_i * 0 behind the column's own terms adds +-0 to a sum that is not -0).
    double ucol[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
        const bool has = mine && i <= lane;
        const double u = p.ut[pc_tile_index(has ? colj + i : 0, (size_t)chain, D * D)];
        ucol[i] = has ? u : 0.0;
    }
    if (lane == 0) ul[2047] = 0.0;       // what a lane reads for the entries above its column when it reloads ucol from LDS
    __syncthreads();

    bool resume = status == kPcResume;   // the host finished this chain's UpdateProposal: the step goes on behind it
    if (resume) status = kPcOk;
    const uint32_t aw = smcmc_accept_word((uint32_t)D);

    // trace of the covariance, summed in index order (GetCovarianceTrace :961-967)
    auto trace_now = [&]() {
        __syncthreads();
        sv[lane] = 0.0;
        __syncthreads();
        uint32_t db = diag_bits;
        asm volatile("" : "+v"(db));     // (tested here, not hoisted out of the step loop as NE lane masks)
#pragma unroll
        for (int r = 0; r < NE; ++r)
            if ((db >> r) & 1u) sv[ij[r] & 255u] = cov[r];
        __syncthreads();
        return pw_sum_lds<DMAX>(sv, D);
    };

    // UpdateProposal (TSimpleMCMC.H:1009-1106); `trace` is the covariance trace.  A failed pivot leaves
    // status = kPcNeedsLadder (the host's ladder takes over, :1134-1389).
    auto update_proposal = [&](double trace) {
        ++update_count;
        if (!(trace > 0)) {                                            // :1025-1028 (the reference throws)
            status = kPcInvalidTrace;
            return;
        }
        const double scale = __builtin_sqrt(sigma_trace / trace);
        sigma = sigma * scale;                                         // :1042
        sigma_trace = trace;                                           // :1043
        const double up = 0.5 * succ;                                  // :1051
        next_update = (int)(p.acc_window + p.max_up - p.max_up / (up + 1.0));   // :1052
        if (p.cov_w >= 0.0) {                                          // :1056-1067
            cov_trials = dmax(1.0, p.cov_w * cov_trials);
            cov_trials = dmin(cov_trials, p.cov_wW);
            centre_trials = dmax(1.0, p.cov_w * centre_trials);
            centre_trials = dmin(centre_trials, p.cov_wW);
        }
        if (p.acc_w >= 0.0) {                                          // :1081-1086
            acc_trials = dmax(1.0, p.acc_w * acc_trials);
            acc_trials = dmin(acc_trials, p.acc_wW);
        }
        // the decomposition in place in LDS: A(c, j) = cov(j, c) sits where U(c, j) will (the same packed index)
        __syncthreads();
        {
            uint32_t vb = valid_bits;
            asm volatile("" : "+v"(vb));
#pragma unroll
            for (int r = 0; r < NE; ++r)
                if ((vb >> r) & 1u) ul[lane + kWave * r] = cov[r];
        }
        __syncthreads();
        bool ok = true;
        // SharedProposal::cholesky (smcmc_proposal.hpp): row c of U from the rows above it
        for (int c = 0; c < D; ++c) {
            const int colc = c * (c + 1) / 2;
            double v = 0.0;
            if (lane >= c && lane < D) {
                v = ul[colj + c];
                for (int rr = 0; rr < c; ++rr) v -= ul[colj + rr] * ul[colc + rr];
            }
            const double piv = pw_readlane(v, c);
            if (!(piv > 0.0) || !__builtin_isfinite(piv)) {
                ok = false;
                break;
            }
            const double sq = __builtin_sqrt(piv);
            if (lane == c) ul[colj + c] = sq;
            else if (lane > c && lane < D) ul[colj + c] = v / sq;
            __syncthreads();
        }
        if (ok) {
            for (int k = lane; k < npk; k += kWave) p.ut[pc_tile_index(k, (size_t)chain, D * D)] = ul[k];
#pragma unroll
            for (int i = 0; i < DMAX; ++i) ucol[i] = ul[(mine && i <= lane) ? colj + i : 2047];
            ufull = 0;
            last_path = 0;
        } else {
            status = kPcNeedsLadder;
        }
        __syncthreads();
    };

#ifdef PW_PROFILE
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define PW_MARK(k) { const unsigned long long tn_ = __builtin_readcyclecounter(); tsec[k] += tn_ - tlast; tlast = tn_; }
#else
#define PW_MARK(k)
#endif
    if (p.update_only) {
        update_proposal(trace_now());
    } else {
        bool live = status == kPcOk && (resume || tstep < p.target_step);
        while (live) {
            if (!resume) ++tstep;                                       // ++fTotalSteps, :376
            const uint64_t step = (uint64_t)tstep;
            const bool forced_now = p.has_forced && tstep == p.step0 + 1u;
            const bool upd = !resume && !forced_now;                    // UpdateState runs (:706)
            bool moved = false;
            const double x0 = pw_readlane(xi, 0);
            PW_MARK(0)
            if (upd) {
                // ---- UpdateState, scalar half (TSimpleMCMC.H:1723-1776) ----
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
                // ---- running centre (:1780-1788) ----
                {
                    double c = ci;
                    c *= centre_trials;
                    c += xi;
                    c /= centre_trials + 1;
                    ci = c;
                    dv[lane] = xi - c;
                }
                centre_trials = dmin(p.cov_window, centre_trials + 1.0);
                PW_MARK(1)
                // ---- running covariance about the updated centre (:1795-1820) ----
                if (!p.cov_frozen) {
                    __syncthreads();
                    const double tv = cov_trials, tv1 = cov_trials + 1.0;
                    double da[NE], db[NE];
#pragma unroll
                    for (int r = 0; r < NE; ++r) {
                        da[r] = dv[(ij[r] >> 8) & 255u];
                        db[r] = dv[ij[r] & 255u];
                    }
#pragma unroll
                    for (int r = 0; r < NE; ++r) {
                        double t = cov[r];
                        const double rr = da[r] * db[r];
                        t *= tv;
                        t += rr;
                        t /= tv1;
                        cov[r] = t;
                    }
                    cov_trials = dmin(p.cov_window, cov_trials + 1.0);
                }
                PW_MARK(2)
                // ---- UpdateProposal when the chain's own schedule says so (:1824-1826) ----
                bool trigger = false;
                if (moved) trigger = (--next_update) < 1;
                if (trigger) {
                    update_proposal(trace_now());
                    if (status != kPcOk) live = false;                  // this chain waits for the host
                }
            }
            if (live && !forced_now) {                                  // :1829-1830
                last_value = logl;
                last_x0 = x0;
                lasti = xi;
            }
            resume = false;
            if (!live) break;

            PW_MARK(3)
            // ---- the proposal (:709-724) ----
            uint32_t uword;
            {
                const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, aw >> 2, SMCMC_STREAM_STEP);
                uword = smcmc_select_word(blk, aw & 3u);
            }
            if (forced_now) {
                xpi = p.forced[own];
            } else {
                // lane i draws r_i: normal i of the step is word pair (i & 2) of Philox block i / 4
                const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, (uint32_t)(lane >> 2), SMCMC_STREAM_STEP);
                const uint32_t w0 = (lane & 2) ? blk.v[2] : blk.v[0], w1 = (lane & 2) ? blk.v[3] : blk.v[1];
                const NormalTables nt = normal_tables_fetch<false>(w0, w1, ltab, atab);
                double n0, n1;
                normal_pair_lds(w0, w1, nt, &n0, &n1);
                const double z = sigma * ((lane & 1) ? n1 : n0);
                __syncthreads();
                zv[lane] = mine ? z : 0.0;
                __syncthreads();
                PW_MARK(4)
                // column j: x'[j] = x[j] + sum_{i <= j} (sigma r_i) U(i, j), i ascending, un-fused; every lane reads all
                // the sigma r_i from LDS ahead of its chain
                double acc = xi;
                {
                    double zz[(DMAX + 1) / 2 * 2];
                    pw_fetch_all<DMAX>(zv, zz);
#pragma unroll
                    for (int i0 = 0; i0 < DMAX; i0 += 8) {
                        if (i0 < D) {                                   // (wave-uniform; the entries past D are zero)
#pragma unroll
                            for (int i = i0; i < i0 + 8 && i < DMAX; ++i) {
                                const double t = zz[i] * ucol[i];
                                acc += t;
                            }
                        }
                    }
                }
                if (ufull) {
                    // a full decomposition (the eigen rung of the ladder): the rows below the diagonal, which every
                    // x'[j] sees after its upper part (i ascending)
                    for (int i2 = 1; i2 < D; ++i2) {
                        const double zi = zv[i2];
                        if (mine && lane < i2) {
                            const double u = p.ut[pc_tile_index(npk + i2 * (i2 - 1) / 2 + lane, (size_t)chain, D * D)];
                            acc += zi * u;
                        }
                    }
                }
                xpi = acc;
            }
            PW_MARK(5)

            // ---- StepRMS window (:391-406), likelihood (:410), Metropolis test (:432-463), accept copy (:484-491) ----
            if (p.step_rms_window > 0) {
                const double ts = xpi - xi;
                const double sqr = pw_ordered_sum<DMAX>(ts * ts, D, sv);
                double ms = step_rms * step_rms;
                ms *= rms_trials;
                ms += sqr;
                ms /= rms_trials + 1.0;
                rms_trials = (p.step_rms_window < rms_trials + 1) ? p.step_rms_window : rms_trials + 1;
                step_rms = __builtin_sqrt(ms);
            }
            PW_MARK(6)
            if constexpr (LIKE == SMCMC_LIKE_QUADFORM || LIKE == SMCMC_LIKE_USER) {
                __syncthreads();
                ps[lane] = mine ? xpi : 0.0;
                __syncthreads();
            }
            logl_prop = pw_loglike<LIKE, DMAX>(ps, xpi, D, p.like, p.like_csr, sv);
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
                xi = xpi;
            }
            if (p.save_x != nullptr && ((tstep - p.step0) % (uint32_t)p.save_stride) == 0) {
                const size_t slot = (size_t)((tstep - p.step0) / (uint32_t)p.save_stride - 1u);
                if (mine) p.save_x[(slot * (size_t)D + (size_t)lane) * NP + chain] = xi;
                if (lane == 0) p.save_logl[slot * NP + chain] = logl;
            }
            if (rec.rec != nullptr && chain == rec.chain) {          // (one chain per workgroup: the same for all 64 lanes)
                double* r = rec.rec + (size_t)(tstep - p.step0 - 1u) * rec.stride;
                if (mine) {
                    recl[lane] = xi;
                    recl[D + lane] = xpi;
                }
#pragma unroll
                for (int q = 0; q < NE; ++q) recl[ij[q] >> 16] = cov[q];   // the diagonal to [2 D, 3 D), everything else to the dump slot
                if (lane == 0) {
                    double* s = recl + 3 * D;
                    s[kPcRecLogl] = logl; s[kPcRecLoglProposed] = logl_prop; s[kPcRecStepRms] = step_rms;
                    s[kPcRecLastAccept] = last_accept; s[kPcRecTrials] = trials; s[kPcRecSuccesses] = succ;
                    s[kPcRecNextUpdate] = next_update; s[kPcRecAcceptance] = acc_rate; s[kPcRecAcceptanceTrials] = acc_trials;
                    s[kPcRecSigma] = sigma; s[kPcRecCenterTrials] = centre_trials; s[kPcRecCovarianceTrials] = cov_trials;
                    s[kPcRecTrace] = 0.0;     // (the reader sums the diagonal)
                    s[kPcRecTotalSteps] = (double)tstep; s[kPcRecStatus] = status;
                }
                __syncthreads();
                for (int k = lane; k < rec.stride; k += kWave) r[k] = recl[k];
            }
            live = tstep < p.target_step;
            PW_MARK(7)
        }
    }
#ifdef PW_PROFILE
    if (rec.rec != nullptr && chain == rec.chain && lane == 0)
        for (int k = 0; k < 8; ++k) rec.rec[k] = (double)tsec[k];
#endif

    // the chain's state back to its images
    if (mine) {
        p.x[own] = xi;
        p.centre[own] = ci;
        p.last_point[own] = lasti;
        p.proposed[own] = xpi;
    }
#pragma unroll
    for (int r = 0; r < NE; ++r) {
        const int k = lane + kWave * r;
        if (k < npk) p.cov[pc_tile_index(k, (size_t)chain, npk)] = cov[r];
    }
    if (lane == 0) {
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

// the likelihoods the wave kernel serves
inline bool perchain_wave_serves(int like) {
#ifdef SMCMC_USER_LIKELIHOOD
    if (like == SMCMC_LIKE_USER) return true;      // (a library built with a user likelihood: build.py --user-likelihood)
#endif
    return like == SMCMC_LIKE_ISO_GAUSS || like == SMCMC_LIKE_QUADFORM || like == SMCMC_LIKE_ROSENBROCK ||
           like == SMCMC_LIKE_ASYM || like == SMCMC_LIKE_HORRIFIC || like == SMCMC_LIKE_CONSTRAINED;
}

hipError_t launch_perchain_wave(const PerChainParams& p, const PerChainRecord& rec, int like, hipStream_t stream);

}  // namespace smcmc
