// smcmc_vaat_large.hip -- the large-dimension variable-at-a-time kernel (smcmc_vaat_large.hip.h) on the library's likelihoods
#include "smcmc_vaat_large.hip.h"

namespace smcmc {

template <int LIKE>
static hipError_t go_vaat_large(const VaatParams& p, bool exact, hipStream_t s) {
    if (exact) hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_large_kernel<LIKE, true>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_large_kernel<LIKE, false>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_vaat_large(const VaatParams& p, int like, bool exact, hipStream_t s) {
    switch (like) {
        case SMCMC_LIKE_ISO_GAUSS: return go_vaat_large<SMCMC_LIKE_ISO_GAUSS>(p, exact, s);
        case SMCMC_LIKE_QUADFORM: return go_vaat_large<SMCMC_LIKE_QUADFORM>(p, exact, s);
        case SMCMC_LIKE_ROSENBROCK: return go_vaat_large<SMCMC_LIKE_ROSENBROCK>(p, exact, s);
        case SMCMC_LIKE_ASYM: return go_vaat_large<SMCMC_LIKE_ASYM>(p, exact, s);
        case SMCMC_LIKE_HORRIFIC: return go_vaat_large<SMCMC_LIKE_HORRIFIC>(p, exact, s);
        case SMCMC_LIKE_CONSTRAINED: return go_vaat_large<SMCMC_LIKE_CONSTRAINED>(p, exact, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace smcmc
