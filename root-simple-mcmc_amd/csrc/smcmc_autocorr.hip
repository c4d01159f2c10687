// smcmc_autocorr.hip -- the lagged-product sums behind the autocorrelation of a saved trace
// (MakeAutocorrelation.C:108-148: a(lag) = (E[x_t x_{t-lag}] - mean^2) / var per dimension), taken
// on the device so that the trace (slots x dim x chains doubles, gigabytes at 65 536 chains) never
// crosses PCIe.  One wavefront owns 64 chains of one dimension and walks the slots once: like the
// reference's ring buffer (:106-124), the last 64 values of every chain sit in LDS ([slot & 63][lane],
// conflict free), the 64 lag accumulators of a chain in registers.  HBM-bound by construction (every
// trace value is read once, 8 bytes; 64 multiply-adds against LDS per value).
// The sums may be taken about a reference point so that E[x x] - mean^2 cancels less; that changes a(lag)
// only through the edges of the lagged sums (O(lag / slots)); the macro itself uses the origin.  Summation order is fixed: slots ascending per chain,
// a butterfly over the 64 chains of a wavefront, wavefronts ascending -- the same bits on every run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"

namespace {

constexpr int kLags = SMCMC_AUTOCORR_LAGS;   // lags 0 .. 63
constexpr int kWave = 64;

__global__ void __launch_bounds__(kWave) autocorr_partial_kernel(const double* __restrict__ trace, int nslots, int dim,
                                                                 size_t dim_stride, int nchains, size_t npad,
                                                                 const double* __restrict__ centre,
                                                                 double* __restrict__ partial) {
    __shared__ double ring[kLags * kWave];
    const int lane = threadIdx.x;
    const int d = blockIdx.x;
    const int cb = blockIdx.y;
    const size_t chain = (size_t)cb * kWave + lane;
    const bool active = chain < (size_t)nchains;
    const double c0 = centre[d];
    for (int k = 0; k < kLags; ++k) ring[k * kWave + lane] = 0.0;
    double acc[kLags];
#pragma unroll
    for (int k = 0; k < kLags; ++k) acc[k] = 0.0;
    double sum = 0.0;
    const double* src = trace + (size_t)d * npad + chain;
    const size_t slot_stride = dim_stride * npad;
    for (int t = 0; t < nslots; ++t) {
        const double v = active ? src[(size_t)t * slot_stride] - c0 : 0.0;
        ring[(t & (kLags - 1)) * kWave + lane] = v;
        sum += v;
#pragma unroll
        for (int k = 0; k < kLags; ++k) acc[k] += v * ring[((t - k) & (kLags - 1)) * kWave + lane];
    }
    // butterfly over the wavefront's chains
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sum += __shfl_xor(sum, off, kWave);
#pragma unroll
        for (int k = 0; k < kLags; ++k) acc[k] += __shfl_xor(acc[k], off, kWave);
    }
    if (lane == 0) {
        double* out = partial + ((size_t)cb * dim + d) * (kLags + 1);
#pragma unroll
        for (int k = 0; k < kLags; ++k) out[k] = acc[k];
        out[kLags] = sum;
    }
}

// out[k][d] = sum over the chain blocks, ascending
__global__ void autocorr_reduce_kernel(const double* __restrict__ partial, int nblocks, int dim, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= dim * (kLags + 1)) return;
    const int d = idx / (kLags + 1), k = idx % (kLags + 1);
    double s = 0.0;
    for (int cb = 0; cb < nblocks; ++cb) s += partial[((size_t)cb * dim + d) * (kLags + 1) + k];
    out[(size_t)k * dim + d] = s;
}

}  // namespace

extern "C" int smcmc_autocorrelation_sums(const double* trace_device, int nslots, int dim, int dim_stride, int nchains,
                                          int nchains_padded, const double* centre, double* sum, double* lagged,
                                          void* stream) {
    if (!trace_device || !sum || !lagged) return SMCMC_ERR_INVALID;
    if (nslots < 1 || dim < 1 || dim_stride < dim || nchains < 1 || nchains_padded < nchains || nchains_padded % kWave != 0)
        return SMCMC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return SMCMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const int nblocks = nchains_padded / kWave;
    const size_t nout = (size_t)dim * (kLags + 1);
    double *d_centre = nullptr, *d_partial = nullptr, *d_out = nullptr;
    int status = SMCMC_ERR_HIP;
    do {
        if (hipMalloc(&d_centre, sizeof(double) * dim) != hipSuccess) break;
        if (hipMalloc(&d_partial, sizeof(double) * nout * nblocks) != hipSuccess) break;
        if (hipMalloc(&d_out, sizeof(double) * nout) != hipSuccess) break;
        if (centre) {
            if (hipMemcpyAsync(d_centre, centre, sizeof(double) * dim, hipMemcpyHostToDevice, s) != hipSuccess) break;
        } else if (hipMemsetAsync(d_centre, 0, sizeof(double) * dim, s) != hipSuccess) break;
        hipLaunchKernelGGL(autocorr_partial_kernel, dim3(dim, nblocks), dim3(kWave), 0, s, trace_device, nslots, dim,
                           (size_t)dim_stride, nchains, (size_t)nchains_padded, d_centre, d_partial);
        if (hipGetLastError() != hipSuccess) break;
        hipLaunchKernelGGL(autocorr_reduce_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, d_partial, nblocks,
                           dim, d_out);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpyAsync(lagged, d_out, sizeof(double) * dim * kLags, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipMemcpyAsync(sum, d_out + (size_t)dim * kLags, sizeof(double) * dim, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        status = SMCMC_OK;
    } while (false);
    (void)hipFree(d_centre); (void)hipFree(d_partial); (void)hipFree(d_out);
    return status;
}
