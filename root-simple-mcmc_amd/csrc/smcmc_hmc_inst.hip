// smcmc_hmc_inst.hip -- instantiations of the HMC step kernel for one workgroup shape;
// built once per -DSMCMC_PANEL_W=<wavefronts per chain group>.
#include "smcmc_hmc_kernel.hip.h"

#ifndef SMCMC_PANEL_W
#error "compile with -DSMCMC_PANEL_W=<4|8>"
#endif

namespace smcmc {

template <int W, int CW, int LIKE>
static hipError_t go_hmc(const HmcParams& p, hipStream_t s) {
    // PotentialGradient types 2 / 3 / 5 (TSimpleHMC.H:467-532) live in the GENERIC instantiation
    const bool generic = p.gradient_type == 2 || p.gradient_type == 3 || p.gradient_type == 5;
    if (generic) {
        if (p.gradient_type == 2 && (p.cov_Eperm == nullptr || p.cov_average == nullptr)) return hipErrorInvalidValue;
        if (p.gradient_type == 3 && p.fd_grad == nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_step_kernel<W, CW, LIKE, true>), dim3(p.npad / kWave), dim3(W * kWave), 0, s, p);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_step_kernel<W, CW, LIKE, false>), dim3(p.npad / kWave), dim3(W * kWave), 0, s, p);
    }
    return hipGetLastError();
}

// the likelihoods without a gradient of their own: the GENERIC instantiation only (Start's potential included)
template <int W, int CW, int LIKE>
static hipError_t go_hmc_no_gradient(const HmcParams& p, hipStream_t s) {
    if (!p.init_only) {
        if (!(p.gradient_type == 2 || p.gradient_type == 3 || p.gradient_type == 5)) return hipErrorInvalidValue;
        if (p.gradient_type == 2 && (p.cov_Eperm == nullptr || p.cov_average == nullptr)) return hipErrorInvalidValue;
        if (p.gradient_type == 3 && p.fd_grad == nullptr) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_step_kernel<W, CW, LIKE, true>), dim3(p.npad / kWave), dim3(W * kWave), 0, s, p);
    return hipGetLastError();
}

template <>
hipError_t launch_hmc<SMCMC_PANEL_W, kPanelCW>(const HmcParams& p, int like, hipStream_t s) {
    constexpr int W = SMCMC_PANEL_W, CW = kPanelCW;
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_hmc<W, CW, SMCMC_LIKE_ISO_GAUSS>(p, s);
        case SMCMC_LIKE_QUADFORM: return go_hmc<W, CW, SMCMC_LIKE_QUADFORM>(p, s);
        case SMCMC_LIKE_ROSENBROCK: return go_hmc<W, CW, SMCMC_LIKE_ROSENBROCK>(p, s);
        case SMCMC_LIKE_ASYM: return go_hmc_no_gradient<W, CW, SMCMC_LIKE_ASYM>(p, s);
        case SMCMC_LIKE_HORRIFIC: return go_hmc_no_gradient<W, CW, SMCMC_LIKE_HORRIFIC>(p, s);
        case SMCMC_LIKE_CONSTRAINED: return go_hmc_no_gradient<W, CW, SMCMC_LIKE_CONSTRAINED>(p, s);
#ifdef SMCMC_USER_LIKELIHOOD_ANY_DIM
        case SMCMC_LIKE_USER: return go_hmc_no_gradient<W, CW, SMCMC_LIKE_USER>(p, s);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
