// Micro-benchmark: how fast can one wavefront add D lane values ONE AFTER THE OTHER in index order (the reference's
// loops over the dimensions: the trial step's square sum, the likelihood, the covariance trace)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/ordered_sum.hip -o /tmp/ordered_sum && /tmp/ordered_sum
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ double rl(double v, int i) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), i);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, int D, int reps) {
    __shared__ double lds[64];
    const int lane = threadIdx.x;
    double t = 1.0 + 1e-3 * lane + out[lane];
    double total = 0.0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        double s = 0.0;
        if (MODE == 0) {            // readlane, run-time index, batches of 8
            int i = 0;
            for (; i + 8 <= D; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = rl(t, i + u);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; i < D; ++i) s += rl(t, i);
        } else if (MODE == 1) {     // readlane, compile-time index (D = 48 terms)
#pragma unroll
            for (int i = 0; i < 48; ++i) s += rl(t, i);
        } else if (MODE == 2) {     // LDS: every lane reads every term (broadcast), batches of 8
            lds[lane] = t;
            __syncthreads();
            int i = 0;
            for (; i + 8 <= D; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = lds[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; i < D; ++i) s += lds[i];
            __syncthreads();
        } else if (MODE == 3) {     // LDS, all D terms fetched up front (D <= 48), then the chain of additions
            lds[lane] = t;
            __syncthreads();
            double v[48];
#pragma unroll
            for (int u = 0; u < 48; ++u) v[u] = lds[u];
#pragma unroll
            for (int u = 0; u < 48; ++u) s += v[u];
            __syncthreads();
        } else if (MODE == 4) {     // the chain of additions alone (no data movement): 48 dependent v_add_f64
#pragma unroll
            for (int u = 0; u < 48; ++u) s += t;
        }
        total += s;
        t += 1e-9 * s;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[lane] = total;
    if (lane == 0 && blockIdx.x == 0) out[64] = (double)(t1 - t0);
}

template <int MODE>
static void run(const char* name, int grid, int D) {
    double* d;
    hipMalloc(&d, 65 * sizeof(double));
    hipMemset(d, 0, 65 * sizeof(double));
    const int reps = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, D, reps);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, D, reps);
    hipDeviceSynchronize();
    double h[65];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("%-64s grid %4d: %.1f cycles per term\n", name, grid, h[64] / reps / D);
    hipFree(d);
}

int main() {
    for (int grid : {1, 1024}) {
        run<0>("v_readlane, run-time index, batches of 8", grid, 48);
        run<1>("v_readlane, compile-time index", grid, 48);
        run<2>("LDS broadcast reads, batches of 8", grid, 48);
        run<3>("LDS broadcast reads, all 48 up front", grid, 48);
        run<4>("48 dependent v_add_f64 alone", grid, 48);
    }
    return 0;
}
