// smcmc_inst.hip -- explicit instantiations of the step kernels for one
// (register-array size, likelihood) pair; built once per
// -DSMCMC_DP=<n> -DSMCMC_LIKE=<k> (see root-simple-mcmc_amd/build.py).
#include "smcmc_kernels.hip.h"
#include "smcmc_vaat_kernel.hip.h"

#if !defined(SMCMC_DP) || !defined(SMCMC_LIKE)
#error "compile with -DSMCMC_DP=<padded dimension> -DSMCMC_LIKE=<likelihood id>"
#endif

namespace smcmc {

template <int DP, int LIKE, bool EXACT, bool FULLU, bool MOM, bool SPECIAL>
static hipError_t go(const StepParams& p, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(step_kernel<DP, LIKE, EXACT, FULLU, MOM, SPECIAL>), dim3(p.npad / kWave),
                       dim3(kWave), 0, s, p);
    return hipGetLastError();
}

template <>
hipError_t launch_step_like<SMCMC_DP, SMCMC_LIKE>(const StepParams& p, bool exact, bool fullu, bool mom, bool special,
                                                  hipStream_t s) {
    constexpr int DP = SMCMC_DP, LIKE = SMCMC_LIKE;
    if (special) {
        // uniform dimensions / scan: reference-order arithmetic only
        if (!exact) return hipErrorNotSupported;
        if (fullu) return mom ? go<DP, LIKE, true, true, true, true>(p, s) : go<DP, LIKE, true, true, false, true>(p, s);
        return mom ? go<DP, LIKE, true, false, true, true>(p, s) : go<DP, LIKE, true, false, false, true>(p, s);
    }
    if (fullu) return mom ? go<DP, LIKE, true, true, true, false>(p, s) : go<DP, LIKE, true, true, false, false>(p, s);
    if (exact) return mom ? go<DP, LIKE, true, false, true, false>(p, s) : go<DP, LIKE, true, false, false, false>(p, s);
    return mom ? go<DP, LIKE, false, false, true, false>(p, s) : go<DP, LIKE, false, false, false, false>(p, s);
}

#if SMCMC_LIKE != 3 || defined(SMCMC_USER_LIKELIHOOD)   // 3 = SMCMC_LIKE_USER (an enumerator, invisible to the preprocessor)
// the variable-at-a-time chains on the same likelihood code (smcmc_vaat_kernel.hip.h)
template <>
hipError_t launch_vaat_like<SMCMC_DP, SMCMC_LIKE>(const VaatParams& p, bool exact, hipStream_t s) {
    constexpr int DP = SMCMC_DP, LIKE = SMCMC_LIKE;
    if (exact) hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_step_kernel<DP, LIKE, true>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_step_kernel<DP, LIKE, false>), dim3(p.npad / kWave), dim3(kWave), 0, s, p);
    return hipGetLastError();
}
#endif

#if SMCMC_LIKE == 0
template <>
hipError_t launch_reduce<SMCMC_DP>(double* gacc, int ngroups, int D, double* chunk_sums, double* moments,
                                   hipStream_t s) {
    const int npk = (D + 1) * (D + 2) / 2;
    const int nchunks = (ngroups + kReduceChunk - 1) / kReduceChunk;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(reduce_chunks_kernel<SMCMC_DP>), dim3((npk + 63) / 64, nchunks), dim3(64), 0, s,
                       gacc, ngroups, D, chunk_sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(reduce_final_kernel<SMCMC_DP>), dim3((npk + 63) / 64), dim3(64), 0, s, chunk_sums, nchunks, npk, moments);
    return hipGetLastError();
}
#endif

}  // namespace smcmc
