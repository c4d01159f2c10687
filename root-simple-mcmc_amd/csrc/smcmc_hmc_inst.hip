// smcmc_hmc_inst.hip -- instantiations of the HMC step kernel for one workgroup shape;
// built once per -DSMCMC_PANEL_W=<wavefronts per chain group>.
#include "smcmc_hmc_kernel.hip.h"

#ifndef SMCMC_PANEL_W
#error "compile with -DSMCMC_PANEL_W=<4|8>"
#endif

namespace smcmc {

template <int W, int CW, int LIKE>
static hipError_t go_hmc(const HmcParams& p, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_step_kernel<W, CW, LIKE>), dim3(p.npad / kWave), dim3(W * kWave), 0, s, p);
    return hipGetLastError();
}

template <>
hipError_t launch_hmc<SMCMC_PANEL_W, kPanelCW>(const HmcParams& p, int like, hipStream_t s) {
    constexpr int W = SMCMC_PANEL_W, CW = kPanelCW;
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_hmc<W, CW, SMCMC_LIKE_ISO_GAUSS>(p, s);
        case SMCMC_LIKE_QUADFORM: return go_hmc<W, CW, SMCMC_LIKE_QUADFORM>(p, s);
        case SMCMC_LIKE_ROSENBROCK: return go_hmc<W, CW, SMCMC_LIKE_ROSENBROCK>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
