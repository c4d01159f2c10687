// smcmc_panel_mfma_inst.hip -- the matrix-pipe form of the large-dimension Metropolis step (fused order).
#include "smcmc_panel_mfma_kernel.hip.h"

namespace smcmc {

template <int LIKE, int VARIANT>
static hipError_t go_panel_mfma_f(const PanelParams& p, hipStream_t s) {
    const dim3 grid(p.npad / kPmCT), block(kPmW * kWave);
    if (p.dim <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_mfma_kernel<1, LIKE, VARIANT>), grid, block, 0, s, p);
    else if (p.dim <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_mfma_kernel<2, LIKE, VARIANT>), grid, block, 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_mfma_kernel<4, LIKE, VARIANT>), grid, block, 0, s, p);
    return hipGetLastError();
}

// p.has_forced: the launch must be a single step (the engine cuts it so); it runs the FORCED instantiation.
// p.proposed: the KEEP instantiation stores the proposal of the launch's last step.
template <int LIKE>
static hipError_t go_panel_mfma(const PanelParams& p, hipStream_t s) {
    if (p.has_forced) return (p.nsteps == 1) ? go_panel_mfma_f<LIKE, PM_FORCED>(p, s) : hipErrorInvalidValue;
    if (p.proposed != nullptr && !p.init_only) return go_panel_mfma_f<LIKE, PM_KEEP>(p, s);
    return go_panel_mfma_f<LIKE, PM_PLAIN>(p, s);
}

hipError_t launch_panel_mfma(const PanelParams& p, int like, hipStream_t s) {
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_panel_mfma<SMCMC_LIKE_ISO_GAUSS>(p, s);
        case SMCMC_LIKE_QUADFORM: return go_panel_mfma<SMCMC_LIKE_QUADFORM>(p, s);
        case SMCMC_LIKE_ROSENBROCK: return go_panel_mfma<SMCMC_LIKE_ROSENBROCK>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
