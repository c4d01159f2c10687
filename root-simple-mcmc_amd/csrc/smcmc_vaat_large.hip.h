// smcmc_vaat_large.hip.h -- TProposeVAATStep chains for 64 <= dim <= 512 (see smcmc_vaat_kernel.hip.h): the point and the
// per-dimension state stay in HBM as [dimension][chain]; one lane per chain.
#pragma once

#include "smcmc_vaat_kernel.hip.h"
#include "smcmc_panel_kernel.hip.h"

namespace smcmc {

template <int LIKE, bool EXACT>
__global__ void __launch_bounds__(kWave) vaat_large_kernel(const VaatParams p) {
    const int lane = threadIdx.x;
    const int chain = blockIdx.x * kWave + lane;
    const bool active = chain < p.nchains;
    const int D = p.dim;
    const size_t NP = (size_t)p.npad;
    const uint32_t gid = p.chain_offset + (uint32_t)chain;
    double* lf = p.lane_f64 + chain;
    int32_t* li = p.lane_i32 + chain;
    uint16_t* q = p.queue + chain;
    int qlen = p.queue_len;

    // UpdateProposal (TProposeVAATStep.H:177-195) with the Uniform() draws of `step`
    auto shuffle = [&](uint64_t step) {
        for (int i = 0; i < D; ++i) q[(size_t)i * NP] = (uint16_t)i;
        smcmc_u32x4 blk;
        for (int i = 0; i < D; ++i) {
            const uint32_t word = 4u + (uint32_t)i;
            if ((word & 3u) == 0u) blk = smcmc_draw_block(p.seed, gid, step, word >> 2, SMCMC_STREAM_VAAT);
            const int s = (int)((double)D * smcmc_u01(smcmc_select_word(blk, word & 3u)));
            const uint16_t a = q[(size_t)i * NP], b = q[(size_t)s * NP];
            q[(size_t)i * NP] = b;
            q[(size_t)s * NP] = a;
        }
        qlen = D;
    };

    if (p.shuffle_only) {
        if (qlen == 0) {
            shuffle((uint64_t)p.step0);
            if (active) li[kVaatLaneLastIndex * NP] = -1;
        }
        return;
    }

    if (p.init_only) {
        const double l0 = serial_loglike<LIKE, EXACT>(p.x, chain, NP, D, p.like);
        lf[SMCMC_LANE_LOGL * NP] = l0;
        if (!p.restart) lf[SMCMC_LANE_LAST_VALUE * NP] = l0;
        lf[SMCMC_LANE_LOGL_PROPOSED * NP] = l0;
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
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);
        ++trials;
        const bool accepted = (logl != last_value);
        if (accepted) ++successes;
        last_value = logl;
        if (last_index >= 0) {
            const size_t k = (size_t)last_index * NP + chain;
            int at = p.acc_trials[k];
            double acc = p.acceptance[k], sg = p.sigma[k];
            vaat_adapt(at, acc, sg, accepted, p.acc_window, p.rigidity, p.target);
            p.acc_trials[k] = at; p.acceptance[k] = acc; p.sigma[k] = sg;
        }
        if (qlen == 0) shuffle(step);
        const int idx = q[(size_t)(qlen - 1) * NP];
        --qlen;
        last_index = idx;
        const smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, 0u, SMCMC_STREAM_VAAT);
        double* cell = p.x + (size_t)idx * NP + chain;
        const double cur = *cell;
        const double newv = vaat_propose<EXACT>(p.ptype[idx], p.param1[idx], p.param2[idx], cur,
                                                p.sigma[(size_t)idx * NP + chain], blk);
        proposed_value = newv;
        if (p.step_rms_window > 0) vaat_step_rms<EXACT>(newv - cur, step_rms, step_rms_trials, p.step_rms_window);
        *cell = newv;
        const double lp = serial_loglike<LIKE, EXACT>(p.x, chain, NP, D, p.like);
        logl_proposed = lp;
        const bool take = active && vaat_accepts(lp, logl, blk.v[3]);
        if (take) {
            logl = lp;
            ++naccept;
        } else {
            *cell = cur;
        }
        last_accept = take ? 1 : 0;
        if (p.save_x != nullptr && (s + 1) % p.save_stride == 0 && active) {
            const size_t slot = (size_t)((s + 1) / p.save_stride - 1);
            for (int d = 0; d < D; ++d) p.save_x[(slot * D + d) * NP + chain] = p.x[(size_t)d * NP + chain];
            if (p.save_logl != nullptr) p.save_logl[slot * NP + chain] = logl;
        }
    }

    if (active) {
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
}

}  // namespace smcmc
