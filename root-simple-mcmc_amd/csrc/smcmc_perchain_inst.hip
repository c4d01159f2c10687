// smcmc_perchain_inst.hip -- instantiations of the per-chain adaptive step (smcmc_perchain_kernel.hip.h): one kernel
// per likelihood, the dimension is a run-time value.
#include "smcmc_perchain_kernel.hip.h"

namespace smcmc {

template <int LIKE>
static hipError_t go_perchain(const PerChainParams& p, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(perchain_step_kernel<LIKE>), dim3(p.npad / kWave), dim3(kWave), perchain_lds_bytes(p.dim), s, p);
    return hipGetLastError();
}

hipError_t launch_perchain(const PerChainParams& p, int like, hipStream_t s) {
    // operand shapes the kernel's indexing assumes
    if (p.dim < 1 || p.dim > kPcMaxDim || p.npad < kWave || p.npad % kWave != 0 || p.nchains < 1 || p.nchains > p.npad)
        return hipErrorInvalidValue;
    if (!p.x || !p.proposed || !p.last_point || !p.centre || !p.cov || !p.ut || !p.lane_f64 || !p.lane_i32 || !p.flag_count)
        return hipErrorInvalidValue;
    if (p.save_x && p.save_stride < 1) return hipErrorInvalidValue;
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_perchain<SMCMC_LIKE_ISO_GAUSS>(p, s);
        case SMCMC_LIKE_QUADFORM: return go_perchain<SMCMC_LIKE_QUADFORM>(p, s);
        case SMCMC_LIKE_ROSENBROCK: return go_perchain<SMCMC_LIKE_ROSENBROCK>(p, s);
        case SMCMC_LIKE_ASYM: return go_perchain<SMCMC_LIKE_ASYM>(p, s);
        case SMCMC_LIKE_HORRIFIC: return go_perchain<SMCMC_LIKE_HORRIFIC>(p, s);
        case SMCMC_LIKE_CONSTRAINED: return go_perchain<SMCMC_LIKE_CONSTRAINED>(p, s);
        default: return hipErrorInvalidValue;
    }
}

__global__ void perchain_broadcast_kernel(const PerChainBroadcast p) {
    const int chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= p.nchains) return;
    const int D = p.dim, npk = D * (D + 1) / 2;
    const size_t NP = (size_t)p.npad;
    double* lf = p.lane_f64 + chain;
    int32_t* li = p.lane_i32 + chain;
    for (int k = 0; k < npk; ++k) p.cov[pc_tile_index(k, (size_t)chain, npk)] = p.cov_packed[k];
    const int nu = p.decomp_full ? D * D : npk;
    for (int k = 0; k < nu; ++k) p.ut_out[pc_tile_index(k, (size_t)chain, D * D)] = p.ut[k];
    li[SMCMC_LANE_DECOMP_FULL * NP] = p.decomp_full;
    li[SMCMC_LANE_LAST_UPDATE_PATH * NP] = p.last_path;
    li[SMCMC_LANE_UPDATE_STATUS * NP] = kPcOk;
    if (!p.reset) {
        // Start (InitializeState :1679-1714) / Restore (RestoreState :1501-1612): the same numbers for every chain
        for (int i = 0; i < D; ++i) {
            const double xi = p.x[(size_t)i * NP + chain];
            p.last_point[(size_t)i * NP + chain] = xi;
            p.centre_out[(size_t)i * NP + chain] = p.centre ? p.centre[i] : xi;
        }
        lf[SMCMC_LANE_CENTER_TRIALS * NP] = p.centre_trials;
        lf[SMCMC_LANE_COVARIANCE_TRIALS * NP] = p.cov_trials;
        lf[SMCMC_LANE_SIGMA_TRACE * NP] = p.sigma_trace;
        li[SMCMC_LANE_UPDATE_COUNT * NP] = p.update_count;
        return;
    }
    // ResetProposal (:1396-1494) followed by UpdateProposal(true) on the template covariance
    li[SMCMC_LANE_TRIALS * NP] = 0;                                                       // :1405-1406
    li[SMCMC_LANE_SUCCESSES * NP] = 0;
    double sigma = lf[SMCMC_LANE_SIGMA * NP];
    if (sigma < p.sigma_floor) sigma = p.sigma_reset;                                     // :1408-1410
    lf[SMCMC_LANE_ACCEPTANCE * NP] = p.acceptance;                                        // :1481
    double acc_trials = p.acceptance_trials;                                              // :1482
    for (int i = 0; i < D; ++i) p.centre_out[(size_t)i * NP + chain] = p.last_point[(size_t)i * NP + chain];   // :1484-1485
    double centre_trials = dmax(lf[SMCMC_LANE_CENTER_TRIALS * NP], 1.0);                  // :1491
    double cov_trials = lf[SMCMC_LANE_COVARIANCE_TRIALS * NP];
    // UpdateProposal(true) :1493: the trace is the one fSigmaTrace was just set to, so sigma keeps its value
    const double scale = __builtin_sqrt(p.sigma_trace / p.sigma_trace);
    sigma = sigma * scale;
    if (p.cov_w >= 0.0) {
        cov_trials = dmax(1.0, p.cov_w * cov_trials);
        cov_trials = dmin(cov_trials, p.cov_wW);
        centre_trials = dmax(1.0, p.cov_w * centre_trials);
        centre_trials = dmin(centre_trials, p.cov_wW);
    }
    if (p.acc_w >= 0.0) {
        acc_trials = dmax(1.0, p.acc_w * acc_trials);
        acc_trials = dmin(acc_trials, p.acc_wW);
    }
    lf[SMCMC_LANE_SIGMA * NP] = sigma;
    lf[SMCMC_LANE_SIGMA_TRACE * NP] = p.sigma_trace;
    lf[SMCMC_LANE_ACCEPTANCE_TRIALS * NP] = acc_trials;
    lf[SMCMC_LANE_CENTER_TRIALS * NP] = centre_trials;
    lf[SMCMC_LANE_COVARIANCE_TRIALS * NP] = cov_trials;
    li[SMCMC_LANE_NEXT_UPDATE * NP] = p.next_update;
    li[SMCMC_LANE_UPDATE_COUNT * NP] = li[SMCMC_LANE_UPDATE_COUNT * NP] + 1;
}

// values[rows] into every chain's column of a [rows][npad] image (tiled: a wavefront-tiled image, pc_tile_index)
__global__ void perchain_broadcast_rows_kernel(double* dst, const double* __restrict__ values, int rows, int nchains, size_t npad,
                                               int tiled) {
    const size_t chain = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= (size_t)nchains) return;
    for (int r = 0; r < rows; ++r) dst[tiled ? pc_tile_index(r, chain, rows) : (size_t)r * npad + chain] = values[r];
}

hipError_t launch_perchain_broadcast_rows(double* dst, const double* values_device, int rows, int nchains, size_t npad, bool tiled,
                                          hipStream_t s) {
    if (!dst || !values_device || rows < 1 || nchains < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(perchain_broadcast_rows_kernel, dim3((nchains + 255) / 256), dim3(256), 0, s, dst, values_device, rows, nchains,
                       npad, tiled ? 1 : 0);
    return hipGetLastError();
}

// The host's fallback ladder works on whole chains: the flagged chains' columns of every image go to / come from one
// contiguous staging record per chain (PerChainStage: the layout), one workgroup per chain.
__global__ void __launch_bounds__(256) perchain_stage_kernel(PerChainStage g, int scatter) {
    const size_t chain = (size_t)g.chains[blockIdx.x], NP = g.npad;
    const int D = g.dim, npk = D * (D + 1) / 2;
    double* rec = g.stage + (size_t)blockIdx.x * pc_stage_stride(D);
    double* f64 = rec, *i32 = f64 + SMCMC_LANE_F64_COUNT_, *cov = i32 + SMCMC_LANE_I32_COUNT_, *ut = cov + npk,
            *centre = ut + D * D, *last = centre + D;
    for (int k = threadIdx.x; k < SMCMC_LANE_F64_COUNT_; k += blockDim.x) {
        if (scatter) g.lane_f64[(size_t)k * NP + chain] = f64[k]; else f64[k] = g.lane_f64[(size_t)k * NP + chain];
    }
    for (int k = threadIdx.x; k < SMCMC_LANE_I32_COUNT_; k += blockDim.x) {
        if (scatter) g.lane_i32[(size_t)k * NP + chain] = (int32_t)i32[k]; else i32[k] = (double)g.lane_i32[(size_t)k * NP + chain];
    }
    for (int k = threadIdx.x; k < npk; k += blockDim.x) {
        if (scatter) g.cov[pc_tile_index(k, chain, npk)] = cov[k]; else cov[k] = g.cov[pc_tile_index(k, chain, npk)];
    }
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        if (scatter) g.centre[(size_t)k * NP + chain] = centre[k];
        else { centre[k] = g.centre[(size_t)k * NP + chain]; last[k] = g.last[(size_t)k * NP + chain]; }
    }
    if (scatter)
        for (int k = threadIdx.x; k < D * D; k += blockDim.x) g.ut[pc_tile_index(k, chain, D * D)] = ut[k];
}

hipError_t launch_perchain_stage(const PerChainStage& g, int nflagged, bool scatter, hipStream_t s) {
    if (nflagged < 1 || g.dim < 1 || g.dim > kPcMaxDim || !g.chains || !g.stage) return hipErrorInvalidValue;
    hipLaunchKernelGGL(perchain_stage_kernel, dim3(nflagged), dim3(256), 0, s, g, scatter ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_perchain_broadcast(const PerChainBroadcast& p, hipStream_t s) {
    if (p.dim < 1 || p.dim > kPcMaxDim || p.nchains < 1 || p.nchains > p.npad) return hipErrorInvalidValue;
    hipLaunchKernelGGL(perchain_broadcast_kernel, dim3((p.nchains + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace smcmc
