// Micro-benchmark: the cost of writing / reading a [500][32768] double state (131 MB) with the access patterns of the
// large-dimension kernels.  hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern A: the matrix-pipe kernel's: workgroup = 32 chains, 8 wavefronts; a store instruction writes 4 rows x 16 chains
__global__ void __launch_bounds__(512) pat_mfma(double* x, int D, size_t NP, int write) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const int base = blockIdx.x * 32;
    double acc = 0.0;
    for (int t = w; t < (D + 15) / 16; t += 8)
        for (int ct = 0; ct < 2; ++ct)
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + rq + 4 * r;
                if (i < D) {
                    if (write) x[(size_t)i * NP + base + 16 * ct + c] = (double)i;
                    else acc += x[(size_t)i * NP + base + 16 * ct + c];
                }
            }
    if (!write && acc == 123.456) x[0] = acc;
}
// pattern B: full 512-byte rows per wavefront: wavefront w of a 64-chain group writes rows w, w+8, ...
__global__ void __launch_bounds__(512) pat_rows(double* x, int D, size_t NP, int write) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int chain = blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int i = w; i < D; i += 8) {
        if (write) x[(size_t)i * NP + chain] = (double)i;
        else acc += x[(size_t)i * NP + chain];
    }
    if (!write && acc == 123.456) x[0] = acc;
}
// pattern C: a plain streaming pass
__global__ void __launch_bounds__(256) pat_stream(double* x, size_t n, int write) {
    double acc = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        if (write) x[k] = 1.0; else acc += x[k];
    }
    if (!write && acc == 123.456) x[0] = acc;
}

template <typename F> float timed(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int k = 0; k < 5; ++k) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5.0f * 1000.0f;
}

int main() {
    const int D = 500; const size_t NP = 32768; const size_t n = (size_t)D * NP;
    double* x; hipMalloc(&x, n * sizeof(double)); hipMemset(x, 0, n * sizeof(double));
    const double mb = n * 8 / 1e6;
    // the same through a ring of eight states (1 GB: past the 256 MB infinity cache, so HBM itself is measured)
    double* ring; hipMalloc(&ring, 8 * n * sizeof(double)); hipMemset(ring, 0, 8 * n * sizeof(double));
    for (int write = 0; write < 2; ++write) {
        int k = 0;
        float a = timed([&] { hipLaunchKernelGGL(pat_mfma, dim3(NP / 32), dim3(512), 0, 0, ring + (size_t)(k++ % 8) * n, D, NP, write); });
        k = 0;
        float b = timed([&] { hipLaunchKernelGGL(pat_rows, dim3(NP / 64), dim3(512), 0, 0, ring + (size_t)(k++ % 8) * n, D, NP, write); });
        k = 0;
        float c = timed([&] { hipLaunchKernelGGL(pat_stream, dim3(2048), dim3(256), 0, 0, ring + (size_t)(k++ % 8) * n, n, write); });
        printf("ring %s 131 MB: matrix-pipe pattern %.0f us (%.2f TB/s), 512-byte rows %.0f us (%.2f TB/s), streaming %.0f us (%.2f TB/s)\n",
               write ? "write" : "read ", a, mb / a, b, mb / b, c, mb / c);
    }
    for (int write = 0; write < 2; ++write) {
        float a = timed([&] { hipLaunchKernelGGL(pat_mfma, dim3(NP / 32), dim3(512), 0, 0, x, D, NP, write); });
        float b = timed([&] { hipLaunchKernelGGL(pat_rows, dim3(NP / 64), dim3(512), 0, 0, x, D, NP, write); });
        float c = timed([&] { hipLaunchKernelGGL(pat_stream, dim3(2048), dim3(256), 0, 0, x, n, write); });
        printf("%s 131 MB: matrix-pipe pattern %.0f us (%.2f TB/s), 512-byte rows %.0f us (%.2f TB/s), streaming %.0f us (%.2f TB/s)\n",
               write ? "write" : "read ", a, mb / a, b, mb / b, c, mb / c);
    }
    return 0;
}
