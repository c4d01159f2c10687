// smcmc_hmc_mfma_inst.hip -- the matrix-pipe HMC kernel (quadratic-form likelihood, fused order).
#include "smcmc_hmc_mfma_kernel.hip.h"

namespace smcmc {

hipError_t launch_hmc_mfma(const HmcParams& p, hipStream_t s) {
    const dim3 grid(p.npad / kMfCT), block(kMfW * kWave);
    if (p.dim <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<1>), grid, block, 0, s, p);
    else if (p.dim <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<4>), grid, block, 0, s, p);
    return hipGetLastError();
}

// the same layout in the reference's operation order (gradient on the vector pipe)
hipError_t launch_hmc_matrix_exact(const HmcParams& p, hipStream_t s) {
    const dim3 grid(p.npad / kMfCT), block(kMfW * kWave);
    if (p.dim <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<1, false>), grid, block, 0, s, p);
    else if (p.dim <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<2, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_mfma_kernel<4, false>), grid, block, 0, s, p);
    return hipGetLastError();
}

}  // namespace smcmc
