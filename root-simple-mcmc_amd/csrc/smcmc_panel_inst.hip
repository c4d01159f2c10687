// smcmc_panel_inst.hip -- instantiations of the large-dimension step kernel for one
// workgroup shape; built once per -DSMCMC_PANEL_W=<wavefronts per chain group>.
#include "smcmc_panel_kernel.hip.h"
#include "smcmc_fold_kernel.hip.h"

#ifndef SMCMC_PANEL_W
#error "compile with -DSMCMC_PANEL_W=<4|8>"
#endif

namespace smcmc {

template <int W, int CW, int LIKE, bool SPECIAL>
static hipError_t go_panel(const PanelParams& p, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_step_kernel<W, CW, LIKE, true, SPECIAL>), dim3(p.npad / kWave),
                       dim3(W * kWave), 0, s, p);
    return hipGetLastError();
}

template <>
hipError_t launch_panel<SMCMC_PANEL_W, kPanelCW>(const PanelParams& p, int like, bool exact, hipStream_t s) {
    constexpr int W = SMCMC_PANEL_W, CW = kPanelCW;
    if (!exact) return hipErrorInvalidValue;   // the fused order is smcmc_panel_mfma_kernel.hip.h
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS:
            return p.special ? go_panel<W, CW, SMCMC_LIKE_ISO_GAUSS, true>(p, s)
                             : go_panel<W, CW, SMCMC_LIKE_ISO_GAUSS, false>(p, s);
        case SMCMC_LIKE_ROSENBROCK:
            return p.special ? go_panel<W, CW, SMCMC_LIKE_ROSENBROCK, true>(p, s)
                             : go_panel<W, CW, SMCMC_LIKE_ROSENBROCK, false>(p, s);
        case SMCMC_LIKE_ASYM:
            return p.special ? go_panel<W, CW, SMCMC_LIKE_ASYM, true>(p, s) : go_panel<W, CW, SMCMC_LIKE_ASYM, false>(p, s);
        case SMCMC_LIKE_HORRIFIC:
            return p.special ? go_panel<W, CW, SMCMC_LIKE_HORRIFIC, true>(p, s)
                             : go_panel<W, CW, SMCMC_LIKE_HORRIFIC, false>(p, s);
        case SMCMC_LIKE_QUADFORM:
            if (p.scratch == nullptr) return hipErrorInvalidValue;
            return p.special ? go_panel<W, CW, SMCMC_LIKE_QUADFORM, true>(p, s)
                             : go_panel<W, CW, SMCMC_LIKE_QUADFORM, false>(p, s);
        case SMCMC_LIKE_CONSTRAINED:
            if (p.scratch == nullptr) return hipErrorInvalidValue;
            return p.special ? go_panel<W, CW, SMCMC_LIKE_CONSTRAINED, true>(p, s)
                             : go_panel<W, CW, SMCMC_LIKE_CONSTRAINED, false>(p, s);
        default: return hipErrorInvalidValue;
    }
}

#if SMCMC_PANEL_W == 4
hipError_t launch_start_loglike(const double* x, int nchains, size_t npad, int D, const double* like_params,
                                double* logl_out, int like, bool exact, hipStream_t s) {
    const dim3 grid((nchains + 255) / 256), block(256);
    if (like == SMCMC_LIKE_ISO_GAUSS) {
        if (exact) hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_ISO_GAUSS, true>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_ISO_GAUSS, false>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
    } else if (like == SMCMC_LIKE_ROSENBROCK) {
        if (exact) hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_ROSENBROCK, true>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_ROSENBROCK, false>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
    } else if (like == SMCMC_LIKE_ASYM) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_ASYM, true>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
    } else if (like == SMCMC_LIKE_HORRIFIC) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_HORRIFIC, true>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
    } else if (like == SMCMC_LIKE_CONSTRAINED) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_CONSTRAINED, true>), grid, block, 0, s, x, nchains, npad, D, like_params, logl_out);
    } else if (like == SMCMC_LIKE_QUADFORM && exact) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_QUADFORM, true>), dim3((nchains + 63) / 64), dim3(64), 0, s, x, nchains, npad, D, like_params, logl_out);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif

}  // namespace smcmc
