// smcmc_selftest.hip -- device-side probes behind smcmc_selftest_* (include/smcmc.h).
//
// They exist so the parity tests can pin two hardware facts the engine relies on:
//   1. include/smcmc_detmath.h evaluates bit for bit the same on gfx950 as on the
//      host (IEEE + - * / sqrt fma, no contraction);
//   2. a chain of v_mfma_f64_16x16x4_f64 accumulates its K products in ascending k
//      with one fused multiply-add each -- the order the pooled moments are
//      defined in (oracle/ensemble_oracle.c).
#include <hip/hip_runtime.h>

#include <vector>

#include "smcmc.h"
#include "smcmc_detmath.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void detmath_kernel(int kind, int n, const double* __restrict__ x, const double* __restrict__ y,
                               double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    const double b = y ? y[i] : 0.0;
    double r = 0.0, s, c;
    switch (kind) {
        case 0: r = smcmc_log(a); break;
        case 1: r = smcmc_exp(a); break;
        case 2: smcmc_sincos2pi(a, &s, &c); r = s; break;
        case 3: smcmc_sincos2pi(a, &s, &c); r = c; break;
        case 4: r = smcmc_pow_small(a, b); break;
        case 5: r = __builtin_sqrt(a); break;
        case 6: r = a / b; break;
        case 7: smcmc_normal_pair((uint32_t)a, (uint32_t)b, &c, &s); r = c; break;   // words held as doubles
        case 8: smcmc_normal_pair((uint32_t)a, (uint32_t)b, &c, &s); r = s; break;
        case 9: r = smcmc_sqrt_mid(a); break;
        default: break;
    }
    out[i] = r;
}

// c[16][16] (row-major) = sum_k a[i][k] * b[k][j], a: [16][K], b: [K][16]; one wavefront.
__global__ void __launch_bounds__(64) mfma_chain_kernel(int K, const double* __restrict__ a,
                                                        const double* __restrict__ b, double* __restrict__ c) {
    const int lane = threadIdx.x;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double av = a[(lane & 15) * K + k0 + (lane >> 4)];    // A[i = lane&15][k = lane>>4]
        const double bv = b[(k0 + (lane >> 4)) * 16 + (lane & 15)]; // B[k = lane>>4][j = lane&15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    // C/D: column = lane & 15, row = (lane >> 4) + 4*reg
#pragma unroll
    for (int r = 0; r < 4; ++r) c[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

// c[4][16] = sum_k a[4][K] b[K][16] with v_mfma_f64_4x4x4_4b_f64: four 4x4 blocks per instruction,
// block blk = (lane >> 2) & 3 covering columns 4 blk .. 4 blk + 3, the same four rows for every block.
__global__ void __launch_bounds__(64) mfma_strip_kernel(int K, const double* __restrict__ a,
                                                        const double* __restrict__ b, double* __restrict__ c) {
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double av = a[(lane & 3) * K + k0 + (lane >> 4)];     // A[i = lane&3][k = lane>>4], any block
        const double bv = b[(k0 + (lane >> 4)) * 16 + (lane & 15)]; // B[k = lane>>4][j = lane&15]
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0);
    }
    c[(lane >> 4) * 16 + (lane & 15)] = acc;                        // D[i = lane>>4][j = lane&15]
}

}  // namespace

#define ST_TRY(expr)                                   \
    do {                                               \
        if ((expr) != hipSuccess) { rc = SMCMC_ERR_HIP; goto done; } \
    } while (0)

extern "C" int smcmc_selftest_detmath(int device, int kind, int n, const double* x, const double* y, double* out) {
    if (n <= 0 || !x || !out || kind < 0 || kind > 9) return SMCMC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    int rc = SMCMC_OK;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    ST_TRY(hipSetDevice(device));
    ST_TRY(hipMalloc(&dx, bytes));
    ST_TRY(hipMalloc(&dout, bytes));
    ST_TRY(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
    if (y) {
        ST_TRY(hipMalloc(&dy, bytes));
        ST_TRY(hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, kind, n, dx, dy, dout);
    ST_TRY(hipGetLastError());
    ST_TRY(hipDeviceSynchronize());
    ST_TRY(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
done:
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return rc;
}

extern "C" int smcmc_selftest_mfma(int device, int K, const double* a, const double* b, double* c) {
    if (K <= 0 || (K & 3) || !a || !b || !c) return SMCMC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    int rc = SMCMC_OK;
    double *da = nullptr, *db = nullptr, *dc = nullptr;
    const size_t ab = sizeof(double) * 16 * (size_t)K;
    ST_TRY(hipSetDevice(device));
    ST_TRY(hipMalloc(&da, ab));
    ST_TRY(hipMalloc(&db, ab));
    ST_TRY(hipMalloc(&dc, sizeof(double) * 256));
    ST_TRY(hipMemcpy(da, a, ab, hipMemcpyHostToDevice));
    ST_TRY(hipMemcpy(db, b, ab, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_chain_kernel, dim3(1), dim3(64), 0, nullptr, K, da, db, dc);
    ST_TRY(hipGetLastError());
    ST_TRY(hipDeviceSynchronize());
    ST_TRY(hipMemcpy(c, dc, sizeof(double) * 256, hipMemcpyDeviceToHost));
done:
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    return rc;
}

extern "C" int smcmc_selftest_mfma_strip(int device, int K, const double* a, const double* b, double* c) {
    if (K <= 0 || (K & 3) || !a || !b || !c) return SMCMC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    int rc = SMCMC_OK;
    double *da = nullptr, *db = nullptr, *dc = nullptr;
    ST_TRY(hipSetDevice(device));
    ST_TRY(hipMalloc(&da, sizeof(double) * 4 * (size_t)K));
    ST_TRY(hipMalloc(&db, sizeof(double) * 16 * (size_t)K));
    ST_TRY(hipMalloc(&dc, sizeof(double) * 64));
    ST_TRY(hipMemcpy(da, a, sizeof(double) * 4 * (size_t)K, hipMemcpyHostToDevice));
    ST_TRY(hipMemcpy(db, b, sizeof(double) * 16 * (size_t)K, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_strip_kernel, dim3(1), dim3(64), 0, nullptr, K, da, db, dc);
    ST_TRY(hipGetLastError());
    ST_TRY(hipDeviceSynchronize());
    ST_TRY(hipMemcpy(c, dc, sizeof(double) * 64, hipMemcpyDeviceToHost));
done:
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    return rc;
}
