// Micro-benchmark: the decomposition as a VALU operand -- from LDS (ds_read_b128: two doubles per instruction, a VGPR pair
// each) or from the scalar cache (s_load_dwordx16: eight doubles per instruction, read by v_mul_f64 as an SGPR operand)?
// One wavefront per SIMD (1024 workgroups of 64 threads), the headline kernel's inner pattern: per 8-column piece one
// multiply + add per column (reference order: un-fused), the next piece's values fetched while the current one is used.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/scalar_u.hip -o /tmp/scalar_u && /tmp/scalar_u
// Prints cycles per piece (at 2.4 GHz) for both sources, with the table at 10.2 KB (D = 50) walked the way a step walks it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) double* cptr_f64;
constexpr int kTable = 1280;   // doubles: the triangular factor at D = 50 (1 275), padded
constexpr int kPieces = kTable / 8;

// Both kernels: a rolled loop over PAIRS of pieces (piece A feeds x[0..7], piece B x[8..15]: static register indices),
// the next pair's values requested before the current pair is used.
typedef const __attribute__((address_space(3))) f64x2* lds_cptr_f64x2;

__global__ void __launch_bounds__(64) k_lds(const double* __restrict__ u, double* out, int steps, double s0) {
    __shared__ __attribute__((aligned(16))) double us[kTable + 16];
    for (int k = threadIdx.x; k < kTable + 16; k += 64) us[k] = u[k];
    __syncthreads();
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = j;
    double s = s0 + threadIdx.x;
    for (int it = 0; it < steps; ++it) {
        uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)us;
        asm volatile("" : "+v"(base));
        f64x2 a[8], b[8];          // two buffers used in turn (no copies): the loop body is two pairs of pieces
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = *(lds_cptr_f64x2)(uintptr_t)(base + 16u * q);
#pragma nounroll
        for (int pp = 0; pp < kPieces / 4; ++pp) {
            base += 128u;
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = *(lds_cptr_f64x2)(uintptr_t)(base + 16u * q);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x[k] += s * a[k / 2][k & 1];
                asm volatile("" : "+v"(x[k]));
            }
            base += 128u;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = *(lds_cptr_f64x2)(uintptr_t)(base + 16u * q);     // (the last one reads the padding)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x[k] += s * b[k / 2][k & 1];
                asm volatile("" : "+v"(x[k]));
            }
        }
        s += 1e-9;
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += x[j];
    if (acc == 123.456) out[0] = acc;
}

__global__ void __launch_bounds__(64) k_smem(const double* __restrict__ u, double* out, int steps, double s0) {
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = j;
    double s = s0 + threadIdx.x;
    for (int it = 0; it < steps; ++it) {
        // the table does not change: an opaque (scalar) offset keeps its loads where they are written
        uint32_t off = 0;
        asm volatile("" : "+s"(off));
        double a[16], b[16];       // two buffers used in turn (no copies)
        {
            cptr_f64 p0 = (cptr_f64)(u + off);
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = p0[k];
        }
#pragma nounroll
        for (int pp = 0; pp < kPieces / 4; ++pp) {
            // scalar loads return in any order: the only wait is lgkmcnt(0), so the next pair's loads have to be issued
            // BEHIND the wait for the current pair -- their address is made to depend on the first use of the current one
            x[0] += s * a[0];
            off += 16;
            asm volatile("" : "+v"(x[0]), "+s"(off));
            {
                cptr_f64 np = (cptr_f64)(u + off);
#pragma unroll
                for (int k = 0; k < 16; ++k) b[k] = np[k];
            }
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                x[k] += s * a[k];
                asm volatile("" : "+v"(x[k]));
            }
            x[0] += s * b[0];
            off += 16;
            asm volatile("" : "+v"(x[0]), "+s"(off));
            {
                cptr_f64 np = (cptr_f64)(u + off);                                                 // (the last one reads the padding)
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = np[k];
            }
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                x[k] += s * b[k];
                asm volatile("" : "+v"(x[k]));
            }
        }
        s += 1e-9;
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += x[j];
    if (acc == 123.456) out[0] = acc;
}

template <typename K> void run(const char* name, K k, const double* u, double* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 2000;
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, u, out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, u, out, steps, 1.0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ns_piece = ms * 1e6 / steps / kPieces;
    printf("%-8s %7.2f ns per 8-column piece = %6.1f cycles at 2.4 GHz  (%6.1f us per 1 280-value step; 16 multiply + add "
           "instructions per piece alone are 64 cycles)\n", name, ns_piece, ns_piece * 2.4, ms * 1e3 / steps);
}

int main() {
    std::vector<double> h(kTable + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 / (1.0 + i);
    double *u, *out;
    hipMalloc(&u, h.size() * sizeof(double));
    hipMalloc(&out, 2 * sizeof(double));
    hipMemcpy(u, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    run("lds", k_lds, u, out);
    run("smem", k_smem, u, out);
    run("lds", k_lds, u, out);
    run("smem", k_smem, u, out);
    return 0;
}
