// smcmc_user_large.hip -- a user likelihood (SMCMC_LIKE_USER) at 63 < dim <= 512: the large-dimension step kernel, Start's
// likelihood call and the variable-at-a-time kernel instantiated on smcmc_user_loglike_at.  Part of a library built with
// `build.py --user-likelihood <header>` whose header defines SMCMC_USER_LIKELIHOOD_ANY_DIM; one object per
// -DSMCMC_PANEL_W=<4|8> (the W = 4 object also carries the two single-wavefront kernels).
#include "smcmc_vaat_large.hip.h"

#if !defined(SMCMC_PANEL_W) || !defined(SMCMC_USER_LIKELIHOOD)
#error "compile with -DSMCMC_PANEL_W=<4|8> -DSMCMC_USER_LIKELIHOOD=<header>"
#endif

namespace smcmc {

#ifdef SMCMC_USER_LIKELIHOOD_ANY_DIM

template <>
hipError_t launch_panel_user<SMCMC_PANEL_W, kPanelCW>(const PanelParams& p, hipStream_t s) {
    constexpr int W = SMCMC_PANEL_W, CW = kPanelCW;
    if (p.scratch == nullptr) return hipErrorInvalidValue;
    if (p.special)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_step_kernel<W, CW, SMCMC_LIKE_USER, true, true>), dim3(p.npad / kWave),
                           dim3(W * kWave), 0, s, p);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(panel_step_kernel<W, CW, SMCMC_LIKE_USER, true, false>), dim3(p.npad / kWave),
                           dim3(W * kWave), 0, s, p);
    return hipGetLastError();
}

#if SMCMC_PANEL_W == 4
hipError_t launch_start_loglike_user(const double* x, int nchains, size_t npad, int D, const double* like_params,
                                     double* logl_out, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(start_loglike_kernel<SMCMC_LIKE_USER, true>), dim3((nchains + 255) / 256), dim3(256),
                       0, s, x, nchains, npad, D, like_params, logl_out);
    return hipGetLastError();
}

hipError_t launch_vaat_large_user(const VaatParams& p, bool exact, hipStream_t s) {
    if (exact) hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_large_kernel<SMCMC_LIKE_USER, true>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_large_kernel<SMCMC_LIKE_USER, false>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    return hipGetLastError();
}
#endif

#else   // the header serves dim <= 63 only

template <>
hipError_t launch_panel_user<SMCMC_PANEL_W, kPanelCW>(const PanelParams&, hipStream_t) { return hipErrorNotSupported; }
#if SMCMC_PANEL_W == 4
hipError_t launch_start_loglike_user(const double*, int, size_t, int, const double*, double*, hipStream_t) {
    return hipErrorNotSupported;
}
hipError_t launch_vaat_large_user(const VaatParams&, bool, hipStream_t) { return hipErrorNotSupported; }
#endif

#endif

}  // namespace smcmc
