// smcmc_vaat_kernel.hip.h -- many independent sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> chains
// (reference TProposeVAATStep.H:22-307 inside TSimpleMCMC::Step, TSimpleMCMC.H:370-496): one coordinate changes per
// step, drawn from a shuffled queue of the dimensions; every dimension keeps its own proposal width, adapted to a 44 %
// acceptance.  Nothing is shared between chains, so chain c is the reference chain on the random stream
// (seed, chain_offset + c) -- SMCMC_STREAM_VAAT of include/smcmc_detmath.h.
//
// One lane per chain.  The per-dimension state fSigma / fAcceptance / fAcceptanceTrials / fNextIndex is indexed by a
// different dimension in every lane:
//   * vaat_step_kernel<DP, LIKE, EXACT> (dim <= 63): the point stays in registers for the likelihood (loglike<> of
//     smcmc_kernels.hip.h, the same code the adaptive-step kernels run), the per-dimension state sits in LDS as
//     [dimension][lane] -- a lane-varying row, a fixed column: every access is bank-conflict free -- loaded when the
//     launch starts and written back when it ends;
//   * vaat_large_kernel<LIKE, EXACT> (64 <= dim <= 512): the point and the state stay in HBM as [dimension][chain]; a
//     step is one gather per state array plus the likelihood's walk over the chain's column (serial_loglike of
//     smcmc_panel_kernel.hip.h: the reference's summation order, which is what a full re-evaluation per step means --
//     the reference calls the whole likelihood too, TSimpleMCMC.H:410).
// The queue length is the same for every chain (each step pops one index, all chains shuffle in the same steps), so
// the host carries it and the shuffle is a uniform branch.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_kernels.hip.h"

namespace smcmc {

struct VaatParams {
    int nchains, npad, dim, nsteps;
    int queue_len;             // entries left in fNextIndex when the launch starts (uniform over the chains)
    int shuffle_only;          // the explicit UpdateProposal() of SimpleVAAT.C:44: refill + shuffle, no step
    int init_only;             // Start (TSimpleMCMC.H:258): logL of the point in x, nothing else
    int restart;               // init_only on a chain that was started before: InitializeState returns at once
                               // (TProposeVAATStep.H:197), so fLastValue keeps what it had
    uint32_t step0;            // fTotalSteps before this launch
    uint32_t chain_offset;
    uint64_t seed;
    const double* like;        // as the adaptive-step kernels take it (QUADFORM: Error^T, padded for dim <= 63)
    const int32_t* ptype;      // [dim] fProposalType[].type
    const double* param1;      // [dim]
    const double* param2;      // [dim]
    int acc_window;            // fAcceptanceWindow (an int in the reference, :277)
    double rigidity, target;
    int step_rms_window;
    double* x;                 // [DP or dim][npad]
    double* lane_f64;          // LOGL, LAST_VALUE, STEP_RMS, LOGL_PROPOSED, SMCMC_VAAT_LANE_PROPOSED_VALUE
    int32_t* lane_i32;         // TRIALS, SUCCESSES, NACCEPT, STEP_RMS_TRIALS, LAST_ACCEPT, SMCMC_VAAT_LANE_LAST_INDEX
    double* sigma;             // [dim][npad] fSigma
    double* acceptance;        // [dim][npad] fAcceptance
    int32_t* acc_trials;       // [dim][npad] fAcceptanceTrials
    uint16_t* queue;           // [dim][npad] fNextIndex
    double* save_x;            // optional [slot][dim][npad]: the accepted point after every save_stride-th step
    double* save_logl;         // optional [slot][npad]
    int save_stride;
};

constexpr int kVaatLaneLastIndex = SMCMC_LANE_NEXT_UPDATE;
constexpr int kVaatLaneProposedValue = SMCMC_LANE_LAST_X0;   // fProposed[fLastIndex] of the latest step

// UpdateState's per-index half (TProposeVAATStep.H:237-254) on values already fetched
__device__ __forceinline__ void vaat_adapt(int& at, double& acc, double& sg, bool accepted, int window, double rigidity,
                                           double target) {
    ++at;                                                               // :238
    const int m = (window < at) ? window : at;
    acc *= 1.0 * m;                                                     // :239-240
    if (accepted) acc += 1.0;
    acc /= 1.0 + 1.0 * m;                                               // :242-243
    if (at > 0.1 * window && rigidity > 0 && rigidity < 100.0) {        // :245-247
        double v = sg;
        const double ratio = acc / target;
        const double expo = dmin(1.0 / 500.0, 1.0 / (rigidity * window));
        v *= (ratio > 0.0) ? smcmc_pow_small(ratio, expo) : 0.0;        // pow(0, y > 0) = 0
        sg = dmax(v, 1.0E-4);                                           // :253
    }
}

// the proposed value of coordinate idx (TProposeVAATStep.H:60-78)
template <bool EXACT>
__device__ __forceinline__ double vaat_propose(int type, double prm1, double prm2, double cur, double sg,
                                               const smcmc_u32x4& blk) {
    if (type == 1) {
        const double u = smcmc_u01(blk.v[2]);
        return prm1 + (prm2 - prm1) * u;                               // gRandom->Uniform(a, b)
    }
    double width = 1.0;                                                 // "expectedVariance": Gaus()'s sigma
    if (type == 0 && prm1 > 0) width = prm1;
    double n0, n1;
    smcmc_normal_pair(blk.v[0], blk.v[1], &n0, &n1);
    const double g = 0.0 + width * n0;                                  // gRandom->Gaus(0.0, width)
    if constexpr (EXACT) return cur + sg * g;
    else return SMCMC_FMA(sg, g, cur);
}

// TSimpleMCMC::Step's StepRMS window (TSimpleMCMC.H:391-406); only one coordinate moved, the other terms are +0
template <bool EXACT>
__device__ __forceinline__ void vaat_step_rms(double t, double& step_rms, int& step_rms_trials, int window) {
    double sqr = 0.0;
    if constexpr (EXACT) sqr += t * t;
    else sqr = SMCMC_FMA(t, t, sqr);
    double ms = step_rms * step_rms;
    ms *= step_rms_trials;
    ms += sqr;
    ms /= step_rms_trials + 1.0;
    step_rms_trials = (window < step_rms_trials + 1) ? window : step_rms_trials + 1;
    step_rms = __builtin_sqrt(ms);
}

// the Metropolis test (TSimpleMCMC.H:432-463)
__device__ __forceinline__ bool vaat_accepts(double lp, double value, uint32_t accept_word) {
    if (!__builtin_isfinite(lp) || lp < -0.999999E+30) return false;
    const double delta = lp - value;
    if (delta < 0.0) {
        const double trial = smcmc_log_pos(smcmc_u01(accept_word));
        if (delta < trial) return false;
    }
    return true;
}

template <int DP, int LIKE, bool EXACT>
__global__ void __launch_bounds__(kWave) vaat_step_kernel(const VaatParams p) {
    __shared__ double s_sigma[DP * kWave];
    __shared__ double s_acc[DP * kWave];
    __shared__ int32_t s_at[DP * kWave];
    __shared__ uint8_t s_q[DP * kWave];
    __shared__ double s_prm1[DP], s_prm2[DP];
    __shared__ int32_t s_type[DP];

    const int lane = threadIdx.x;
    const int chain = blockIdx.x * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    const cptr_f64 likep = as_const(p.like);

    for (int d = 0; d < D; ++d) {
        s_sigma[d * kWave + lane] = p.sigma[(size_t)d * NP + chain];
        s_acc[d * kWave + lane] = p.acceptance[(size_t)d * NP + chain];
        s_at[d * kWave + lane] = p.acc_trials[(size_t)d * NP + chain];
        s_q[d * kWave + lane] = (uint8_t)p.queue[(size_t)d * NP + chain];
    }
    for (int d = lane; d < D; d += kWave) {
        s_type[d] = p.ptype[d];
        s_prm1[d] = p.param1[d];
        s_prm2[d] = p.param2[d];
    }
    __syncthreads();

    double* lf = p.lane_f64 + chain;
    int32_t* li = p.lane_i32 + chain;
    int qlen = p.queue_len;

    // UpdateProposal (:177-195) with the Uniform() draws of `step`
    auto shuffle = [&](uint64_t step) {
        for (int i = 0; i < D; ++i) s_q[i * kWave + lane] = (uint8_t)i;
        smcmc_u32x4 blk;
        for (int i = 0; i < D; ++i) {
            const uint32_t word = 4u + (uint32_t)i;
            if ((word & 3u) == 0u || i == 0) blk = smcmc_draw_block(p.seed, gid, step, word >> 2, SMCMC_STREAM_VAAT);
            const int s = (int)((double)D * smcmc_u01(smcmc_select_word(blk, word & 3u)));
            const uint8_t a = s_q[i * kWave + lane], b = s_q[s * kWave + lane];
            s_q[i * kWave + lane] = b;
            s_q[s * kWave + lane] = a;
        }
        qlen = D;
    };

    if (p.shuffle_only) {
        if (qlen == 0) {
            shuffle((uint64_t)p.step0);
            for (int d = 0; d < D; ++d) p.queue[(size_t)d * NP + chain] = s_q[d * kWave + lane];
            if (active) li[kVaatLaneLastIndex * NP] = -1;                // :183
        }
        return;
    }

    double x[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = p.x[(size_t)d * NP + chain];   // rows >= dim are zero
    if (p.init_only) {
        const double l0 = loglike<DP, LIKE, EXACT>(x, likep, D);
        lf[SMCMC_LANE_LOGL * NP] = l0;
        if (!p.restart) lf[SMCMC_LANE_LAST_VALUE * NP] = l0;              // fLastValue = value (:207)
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = l0;                           // fProposedLogLikelihood = L(start)
        return;
    }
    double logl = lf[SMCMC_LANE_LOGL * NP];
    double last_value = lf[SMCMC_LANE_LAST_VALUE * NP];
    double logl_proposed = lf[SMCMC_LANE_LOGL_PROPOSED * NP];
    double step_rms = lf[SMCMC_LANE_STEP_RMS * NP];
    double proposed_value = lf[kVaatLaneProposedValue * NP];
    int trials = li[SMCMC_LANE_TRIALS * NP];
    int successes = li[SMCMC_LANE_SUCCESSES * NP];
    int naccept = li[SMCMC_LANE_NACCEPT * NP];
    int step_rms_trials = li[SMCMC_LANE_STEP_RMS_TRIALS * NP];
    int last_accept = li[SMCMC_LANE_LAST_ACCEPT * NP];
    int last_index = li[kVaatLaneLastIndex * NP];

    for (int s = 0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);   // ++fTotalSteps, TSimpleMCMC.H:376
        // ---- UpdateState (:219-255) ----
        ++trials;
        const bool accepted = (logl != last_value);                     // :225-226
        if (accepted) ++successes;
        last_value = logl;
        if (last_index >= 0) {
            const int k = last_index * kWave + lane;
            int at = s_at[k];
            double acc = s_acc[k], sg = s_sigma[k];
            vaat_adapt(at, acc, sg, accepted, p.acc_window, p.rigidity, p.target);
            s_at[k] = at; s_acc[k] = acc; s_sigma[k] = sg;
        }
        // ---- operator() (:52-78) ----
        if (qlen == 0) shuffle(step);                                   // :55
        const int idx = s_q[(qlen - 1) * kWave + lane];                 // :58-59
        --qlen;
        last_index = idx;
        const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, 0u, SMCMC_STREAM_VAAT);
        double cur = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) cur = (d == idx) ? x[d] : cur;
        const double newv = vaat_propose<EXACT>(s_type[idx], s_prm1[idx], s_prm2[idx], cur, s_sigma[idx * kWave + lane], blk);
        proposed_value = newv;
        // ---- TSimpleMCMC::Step around it (TSimpleMCMC.H:391-491) ----
        if (p.step_rms_window > 0) vaat_step_rms<EXACT>(newv - cur, step_rms, step_rms_trials, p.step_rms_window);
#pragma unroll
        for (int d = 0; d < DP; ++d) x[d] = (d == idx) ? newv : x[d];
        const double lp = loglike<DP, LIKE, EXACT>(x, likep, D);        // :410
        logl_proposed = lp;
        const bool take = active && vaat_accepts(lp, logl, blk.v[3]);
        if (take) {                                                     // :484-487
            logl = lp;
            ++naccept;
        } else {
#pragma unroll
            for (int d = 0; d < DP; ++d) x[d] = (d == idx) ? cur : x[d];
        }
        last_accept = take ? 1 : 0;
        if (p.save_x != nullptr && (s + 1) % p.save_stride == 0 && active) {
            const size_t slot = (size_t)((s + 1) / p.save_stride - 1);
#pragma unroll
            for (int d = 0; d < DP; ++d)
                if (d < D) p.save_x[(slot * D + d) * NP + chain] = x[d];
            if (p.save_logl != nullptr) p.save_logl[slot * NP + chain] = logl;
        }
    }

    if (active) {
#pragma unroll
        for (int d = 0; d < DP; ++d)
            if (d < D) p.x[(size_t)d * NP + chain] = x[d];
        lf[SMCMC_LANE_LOGL * NP] = logl;
        lf[SMCMC_LANE_LAST_VALUE * NP] = last_value;
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = logl_proposed;
        lf[SMCMC_LANE_STEP_RMS * NP] = step_rms;
        lf[kVaatLaneProposedValue * NP] = proposed_value;
        li[SMCMC_LANE_TRIALS * NP] = trials;
        li[SMCMC_LANE_SUCCESSES * NP] = successes;
        li[SMCMC_LANE_NACCEPT * NP] = naccept;
        li[SMCMC_LANE_STEP_RMS_TRIALS * NP] = step_rms_trials;
        li[SMCMC_LANE_LAST_ACCEPT * NP] = last_accept;
        li[kVaatLaneLastIndex * NP] = last_index;
    }
    for (int d = 0; d < D; ++d) {
        p.sigma[(size_t)d * NP + chain] = s_sigma[d * kWave + lane];
        p.acceptance[(size_t)d * NP + chain] = s_acc[d * kWave + lane];
        p.acc_trials[(size_t)d * NP + chain] = s_at[d * kWave + lane];
        p.queue[(size_t)d * NP + chain] = s_q[d * kWave + lane];
    }
}

template <int DP, int LIKE> hipError_t launch_vaat_like(const VaatParams& p, bool exact, hipStream_t stream);

template <int DP>
inline hipError_t launch_vaat(const VaatParams& p, int like, bool exact, hipStream_t stream) {
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return launch_vaat_like<DP, SMCMC_LIKE_ISO_GAUSS>(p, exact, stream);
        case SMCMC_LIKE_QUADFORM: return launch_vaat_like<DP, SMCMC_LIKE_QUADFORM>(p, exact, stream);
        case SMCMC_LIKE_ROSENBROCK: return launch_vaat_like<DP, SMCMC_LIKE_ROSENBROCK>(p, exact, stream);
        case SMCMC_LIKE_ASYM:
            if constexpr (DP == 31 || DP == 63) return launch_vaat_like<DP, SMCMC_LIKE_ASYM>(p, exact, stream);
            else return hipErrorInvalidValue;
        case SMCMC_LIKE_HORRIFIC:
            if constexpr (DP == 31 || DP == 63) return launch_vaat_like<DP, SMCMC_LIKE_HORRIFIC>(p, exact, stream);
            else return hipErrorInvalidValue;
        case SMCMC_LIKE_CONSTRAINED:
            if constexpr (DP == 31 || DP == 63) return launch_vaat_like<DP, SMCMC_LIKE_CONSTRAINED>(p, exact, stream);
            else return hipErrorInvalidValue;
#ifdef SMCMC_USER_LIKELIHOOD
        case SMCMC_LIKE_USER: return launch_vaat_like<DP, SMCMC_LIKE_USER>(p, exact, stream);
#endif
        default: return hipErrorInvalidValue;
    }
}

// dim > 63 (smcmc_vaat_large.hip)
hipError_t launch_vaat_large(const VaatParams& p, int like, bool exact, hipStream_t stream);
hipError_t launch_vaat_large_user(const VaatParams& p, bool exact, hipStream_t stream);   // smcmc_user_large.hip

}  // namespace smcmc
