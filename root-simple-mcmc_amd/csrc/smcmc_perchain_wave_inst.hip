// smcmc_perchain_wave_inst.hip -- instantiations of the one-chain-per-wavefront adaptive step
// (smcmc_perchain_wave.hip.h): likelihood x registers of packed covariance per lane.
#include "smcmc_perchain_wave.hip.h"

namespace smcmc {

template <int LIKE, int NE, bool PAIRED = false>
static hipError_t go_wave(const PerChainParams& p, const PerChainRecord& rec, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(perchain_wave_kernel<LIKE, NE, PAIRED>), dim3(p.nchains), dim3(kWave), 0, s, p, rec);
    return hipGetLastError();
}

template <int LIKE>
static hipError_t go_wave_like(const PerChainParams& p, const PerChainRecord& rec, hipStream_t s) {
    switch (perchain_wave_elements(p.dim)) {
        case 4: return (p.nchains > 1024) ? go_wave<LIKE, 4, true>(p, rec, s) : go_wave<LIKE, 4>(p, rec, s);   // 1 024 SIMDs
        case 12: return go_wave<LIKE, 12>(p, rec, s);
        case 20: return go_wave<LIKE, 20>(p, rec, s);
        default: return go_wave<LIKE, 32>(p, rec, s);
    }
}

hipError_t launch_perchain_wave(const PerChainParams& p, const PerChainRecord& rec, int like, hipStream_t s) {
    // operand shapes the kernel's indexing assumes
    if (p.dim < 1 || p.dim > kPcMaxDim || p.npad < kWave || p.npad % kWave != 0 || p.nchains < 1 || p.nchains > p.npad)
        return hipErrorInvalidValue;
    if (!p.x || !p.proposed || !p.last_point || !p.centre || !p.cov || !p.ut || !p.lane_f64 || !p.lane_i32 || !p.flag_count)
        return hipErrorInvalidValue;
    if (p.save_x && p.save_stride < 1) return hipErrorInvalidValue;
    if (rec.rec && (rec.chain < 0 || rec.chain >= p.nchains || rec.stride < 3 * p.dim + kPcRecScalars)) return hipErrorInvalidValue;
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_wave_like<SMCMC_LIKE_ISO_GAUSS>(p, rec, s);
        case SMCMC_LIKE_QUADFORM: return go_wave_like<SMCMC_LIKE_QUADFORM>(p, rec, s);
        case SMCMC_LIKE_ROSENBROCK: return go_wave_like<SMCMC_LIKE_ROSENBROCK>(p, rec, s);
        case SMCMC_LIKE_ASYM: return go_wave_like<SMCMC_LIKE_ASYM>(p, rec, s);
        case SMCMC_LIKE_HORRIFIC: return go_wave_like<SMCMC_LIKE_HORRIFIC>(p, rec, s);
        case SMCMC_LIKE_CONSTRAINED: return go_wave_like<SMCMC_LIKE_CONSTRAINED>(p, rec, s);
#ifdef SMCMC_USER_LIKELIHOOD
        case SMCMC_LIKE_USER: return go_wave_like<SMCMC_LIKE_USER>(p, rec, s);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
