// smcmc_autocorr.hip -- the lagged-product sums behind the autocorrelation of a saved trace
// (MakeAutocorrelation.C:108-148: a(lag) = (E[x_t x_{t-lag}] - mean^2) / var per dimension), taken
// on the device so that the trace (slots x dim x chains doubles, gigabytes at 65 536 chains) never
// crosses PCIe.  One lane owns one chain of one dimension; a wavefront covers 64 chains and 32 of the
// 64 lags (two passes over the trace, the second mostly out of L2).  Everything sits in registers with
// static indices: the 16 values of the current block of slots, a sliding window of the 47 values the
// 32 lags reach back to (the macro's ring buffer, :106-124), 32 accumulators: 512 fused multiply-adds per
// 16 loaded values and no LDS.  Measured on the headline trace (512 slots x 50 x 65 536, 13.4 GB):
// 5.1 ms for lags 0..31 (2.6 TB/s), 11.1 ms for lags 32..63 (it loads the lagged block as well);
// an LDS ring with 64 accumulators per lane took 36.5 ms (five instructions per lag step, one
// wavefront per SIMD).  Bounds: 1.7 ms of HBM time per pass, 2.7 ms of FP64 vector time in all
// (64 multiply-adds per 8-byte value at 78.6 TFLOP/s); what is left is load latency at two to three
// wavefronts per SIMD (the next block is not prefetched yet).
// The sums may be taken about a reference point so that E[x x] - mean^2 cancels less; that changes a(lag)
// only through the edges of the lagged sums (O(lag / slots)); the macro itself uses the origin.
// Summation order is fixed: slots ascending per chain, a butterfly over the 64 chains of a wavefront,
// wavefronts ascending -- the same bits on every run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"

namespace {

constexpr int kLags = SMCMC_AUTOCORR_LAGS;   // lags 0 .. 63
constexpr int kWave = 64;
constexpr int kPassLags = 32;                // lags per wavefront
constexpr int kBlock = 16;                   // slots per register block
constexpr int kWin = kBlock + kPassLags - 1; // y[t0 - K0 - 31 .. t0 - K0 + 15]
static_assert(kLags == 2 * kPassLags, "two passes cover the lags");

template <int K0>
__global__ void __launch_bounds__(kWave) autocorr_partial_kernel(const double* __restrict__ trace, int nslots, int dim,
                                                                 size_t dim_stride, int nchains, size_t npad,
                                                                 const double* __restrict__ centre,
                                                                 double* __restrict__ partial) {
    const int lane = threadIdx.x;
    const int cb = blockIdx.x;   // neighbouring workgroups read neighbouring 512-byte pieces of a trace row
    const int d = blockIdx.y;
    const size_t chain = (size_t)cb * kWave + lane;
    const bool active = chain < (size_t)nchains;
    const double c0 = centre[d];
    const double* src = trace + (size_t)d * npad + chain;
    const size_t slot_stride = dim_stride * npad;
    auto y = [&](int t) __attribute__((always_inline)) {
        return (active && t >= 0 && t < nslots) ? src[(size_t)t * slot_stride] - c0 : 0.0;
    };
    double win[kWin], acc[kPassLags], v[kBlock];
#pragma unroll
    for (int i = 0; i < kWin; ++i) win[i] = 0.0;
#pragma unroll
    for (int k = 0; k < kPassLags; ++k) acc[k] = 0.0;
    double sum = 0.0;
    for (int t0 = 0; t0 < nslots; t0 += kBlock) {
#pragma unroll
        for (int j = 0; j < kBlock; ++j) {
            v[j] = y(t0 + j);
            win[kPassLags - 1 + j] = (K0 == 0) ? v[j] : y(t0 - K0 + j);
        }
        if (K0 == 0) {
#pragma unroll
            for (int j = 0; j < kBlock; ++j) sum += v[j];
        }
        // acc[kk] += y[t0 + j] * y[t0 + j - K0 - kk]
#pragma unroll
        for (int j = 0; j < kBlock; ++j)
#pragma unroll
            for (int kk = 0; kk < kPassLags; ++kk) acc[kk] = __builtin_fma(v[j], win[kPassLags - 1 + j - kk], acc[kk]);
#pragma unroll
        for (int i = 0; i < kPassLags - 1; ++i) win[i] = win[i + kBlock];
    }
    // butterfly over the wavefront's chains
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sum += __shfl_xor(sum, off, kWave);
#pragma unroll
        for (int k = 0; k < kPassLags; ++k) acc[k] += __shfl_xor(acc[k], off, kWave);
    }
    if (lane == 0) {
        double* out = partial + ((size_t)cb * dim + d) * (kLags + 1);
#pragma unroll
        for (int k = 0; k < kPassLags; ++k) out[K0 + k] = acc[k];
        if (K0 == 0) out[kLags] = sum;
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
        hipLaunchKernelGGL(autocorr_partial_kernel<0>, dim3(nblocks, dim), dim3(kWave), 0, s, trace_device, nslots, dim,
                           (size_t)dim_stride, nchains, (size_t)nchains_padded, d_centre, d_partial);
        if (hipGetLastError() != hipSuccess) break;
        hipLaunchKernelGGL(autocorr_partial_kernel<kPassLags>, dim3(nblocks, dim), dim3(kWave), 0, s, trace_device, nslots,
                           dim, (size_t)dim_stride, nchains, (size_t)nchains_padded, d_centre, d_partial);
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
