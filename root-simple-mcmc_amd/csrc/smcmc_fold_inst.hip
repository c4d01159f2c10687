// smcmc_fold_inst.hip -- the large-dimension moment fold: instantiations of fold_ring_kernel (one per class of staging
// rounds, with and without a chain mask), the plan of an engine, the ordered reduction of the moment groups.
#include "smcmc_fold_ring.hip.h"

namespace smcmc {

namespace {

template <int NQ, bool MASKED>
hipError_t go_fold_ring(const FoldRingParams& p, int nwg, hipStream_t stream) {
    // more dynamic LDS than the default limit: raised per device the first time the instantiation runs there
    static bool raised[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!raised[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fold_ring_kernel<NQ, MASKED>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_ring_lds_bytes(NQ));
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fold_ring_kernel<NQ, MASKED>), dim3(nwg), dim3(kFrWaves * kWave),
                       fold_ring_lds_bytes(NQ), stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t fold_ring_prepare(FoldRing& fr, int D, int nchains, int npad, int nslices, int slice_chains) {
    fold_ring_release(fr);
    if ((D + 1 + 15) / 16 > kFrMaxFoot - 2) return hipErrorInvalidValue;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const FoldPlan plan = fold_ring_plan(D, nchains, npad, nslices, slice_chains, cus > 0 ? cus : 256);
    if (plan.rounds < 0 || plan.max_tiles > kFrMaxT) return hipErrorInvalidValue;
    e = hipMalloc(&fr.d_plan, sizeof(FoldPlanEntry) * plan.wg.size());
    if (e != hipSuccess) return e;
    e = hipMemcpy(fr.d_plan, plan.wg.data(), sizeof(FoldPlanEntry) * plan.wg.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    e = hipMalloc(&fr.d_order, sizeof(uint16_t) * plan.order.size());
    if (e != hipSuccess) return e;
    e = hipMemcpy(fr.d_order, plan.order.data(), sizeof(uint16_t) * plan.order.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    fr.nwg = (int)plan.wg.size();
    fr.rounds = plan.rounds;
    return hipSuccess;
}

void fold_ring_release(FoldRing& fr) {
    if (fr.d_plan) (void)hipFree(fr.d_plan);
    if (fr.d_order) (void)hipFree(fr.d_order);
    fr.d_plan = nullptr;
    fr.d_order = nullptr;
    fr.nwg = 0;
    fr.rounds = 0;
}

hipError_t launch_fold_ring(const FoldRing& fr, FoldRingParams p, hipStream_t stream) {
    if (!fr.d_plan || fr.nwg < 8 || p.nsrc < 1 || p.nsrc > kFoldMaxSrc) return hipErrorInvalidValue;
    p.plan = fr.d_plan;
    p.order = fr.d_order;
    switch (fr.rounds) {
#define SMCMC_FR_GO(n) \
    case n: return p.mask ? go_fold_ring<n, true>(p, fr.nwg, stream) : go_fold_ring<n, false>(p, fr.nwg, stream);
        SMCMC_FR_GO(2) SMCMC_FR_GO(4) SMCMC_FR_GO(5) SMCMC_FR_GO(7) SMCMC_FR_GO(9)
#undef SMCMC_FR_GO
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fold_reduce(const double* gacc, int D, int nslices, double* moments, hipStream_t s) {
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2;
    const int npk = (D + 1) * (D + 2) / 2;
    hipLaunchKernelGGL(fold_reduce_kernel, dim3((npk + 255) / 256), dim3(256), 0, s, gacc, ntiles, nslices, D,
                       moments);
    return hipGetLastError();
}

}  // namespace smcmc
