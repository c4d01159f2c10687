// Micro-benchmark: what does the CU's LDS pipe charge for the reads the step kernels feed their multiply / add pairs with?
// A workgroup of NW wavefronts (one per SIMD up to 4), every wavefront reading ITS OWN row of doubles; cycles per read
// instruction seen by one wavefront, for
//   b128 broadcast (all lanes one address: 2 useful doubles per instruction -- what panel_step_kernel / step_kernel do),
//   b64 broadcast, b128 / b64 with every lane its own address (64 * 2 / 64 useful doubles),
// alone (reads only) and with the 2 multiplies + 2 additions per double pair the kernels do per b128 (does the arithmetic
// hide behind the pipe or add to it?), and -- the alternative -- one per-lane b64 read + v_readlane pairs feeding the
// same arithmetic through SGPRs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/lds_bcast.hip -o root-simple-mcmc_amd/build/micro/lds_bcast
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f64x2* lds_c2;
typedef __attribute__((address_space(3))) const double* lds_c1;

constexpr int kRow = 2048;   // doubles per wavefront's row (16 KB)

__device__ __forceinline__ double rl(double v, int i) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), i);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}

// MODE 0: b128 broadcast, reads only   1: b128 broadcast + arithmetic   2: b64 broadcast reads only
//      3: b128 per-lane addresses, reads only   4: b64 per-lane, reads only   5: per-lane b64 + readlane + arithmetic
//      6: arithmetic alone (16 mul + 16 add per piece, no reads)
//      7: b128 broadcast, the NEXT piece's reads issued before this piece's arithmetic (two register sets), mul / add adjacent
//      8: as 7, all 16 products first, then the 16 additions
//      9: as 7, product k + 1 issued before addition k
template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) k(double* out, int reps, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(16))) double lds[NW * kRow];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NW * kRow; i += NW * 64) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = out[lane] + q;
    const double sr = 1.0 + 1e-6 * lane;
    lds_c1 base = (lds_c1)(lds + w * kRow);
    asm volatile("" : "+v"(base));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int p = 0; p < (MODE >= 7 ? 0 : kRow / 16); ++p) {           // one "piece": 16 doubles
            lds_c1 a = base + p * 16;
            if constexpr (MODE == 0 || MODE == 1) {
                f64x2 u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u[q] = *(volatile lds_c2)(a + 2 * q);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if constexpr (MODE == 1) { acc[q] += sr * u[q / 2][q & 1]; asm volatile("" : "+v"(acc[q])); }
                    else { double t = u[q / 2][q & 1]; asm volatile("" :: "v"(t)); }
                }
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { double t = *(volatile lds_c1)(a + q); asm volatile("" :: "v"(t)); }
            } else if constexpr (MODE == 3) {
                // every lane its own 16 bytes: 8 instructions cover 8 * 128 doubles; per piece-equivalent count the same 8
#pragma unroll
                for (int q = 0; q < 8; ++q) { f64x2 t = *(volatile lds_c2)(base + ((p * 8 + q) & 15) * 128 + 2 * lane); asm volatile("" :: "v"(t)); }
            } else if constexpr (MODE == 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { double t = *(volatile lds_c1)(base + ((p * 8 + q) & 31) * 64 + lane); asm volatile("" :: "v"(t)); }
            } else if constexpr (MODE == 5) {
                // one per-lane b64 read brings 64 doubles = four pieces; this piece uses lanes (p & 3) * 16 ..
                double v = *(volatile lds_c1)(base + (p >> 2) * 64 + lane);
                const int l0 = (p & 3) * 16;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    double u;
                    switch (p & 3) { case 0: u = rl(v, q); break; case 1: u = rl(v, 16 + q); break; case 2: u = rl(v, 32 + q); break; default: u = rl(v, 48 + q); }
                    acc[q] += sr * u;
                    asm volatile("" : "+v"(acc[q]));
                }
                (void)l0;
            } else if constexpr (MODE == 6) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[q] += sr * acc[(q + 1) & 15]; asm volatile("" : "+v"(acc[q])); }
            }
        }
        if constexpr (MODE >= 7) {
            f64x2 ua[8], ub[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ua[q] = *(volatile lds_c2)(base + 2 * q);
            auto arith = [&](f64x2 (&u)[8]) __attribute__((always_inline)) {
                if constexpr (MODE == 7) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { acc[q] += sr * u[q / 2][q & 1]; asm volatile("" : "+v"(acc[q])); }
                } else if constexpr (MODE == 8) {
                    double t[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) t[q] = sr * u[q / 2][q & 1];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 16; ++q) { acc[q] += t[q]; asm volatile("" : "+v"(acc[q])); }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    double t = sr * u[0][0];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        double tn = (q + 1 < 16) ? sr * u[(q + 1) / 2][(q + 1) & 1] : 0.0;
                        asm volatile("" : "+v"(tn));
                        acc[q] += t;
                        asm volatile("" : "+v"(acc[q]));
                        t = tn;
                    }
                }
            };
#pragma unroll 1
            for (int p = 0; p < kRow / 16; p += 2) {
                lds_c1 a1 = base + (p + 1) * 16, a2 = base + ((p + 2) & (kRow / 16 - 1)) * 16;
#pragma unroll
                for (int q = 0; q < 8; ++q) ub[q] = *(volatile lds_c2)(a1 + 2 * q);
                arith(ua);
#pragma unroll
                for (int q = 0; q < 8; ++q) ua[q] = *(volatile lds_c2)(a2 + 2 * q);
                arith(ub);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cycles[w] = t1 - t0;
}

template <int MODE, int NW>
void run(const char* name, double* d_out, unsigned long long* d_cyc, int blocks) {
    const int reps = 20;
    hipMemset(d_out, 0, sizeof(double) * 4096 * 256);
    k<MODE, NW><<<blocks, NW * 64>>>(d_out, 2, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, NW><<<blocks, NW * 64>>>(d_out, reps, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8] = {0};
    hipMemcpy(c, d_cyc, sizeof(unsigned long long) * NW, hipMemcpyDeviceToHost);
    const double pieces = (double)reps * (kRow / 16);
    printf("%-58s %d wavefront(s)/CU: %7.1f cycles per 16-double piece (wavefront 0; %.3f ms)\n", name, NW, c[0] / pieces, ms);
}

int main() {
    double* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, sizeof(double) * 4096 * 256);
    hipMalloc(&d_cyc, 64);
    const int blocks = 256;   // one workgroup per CU
#define BOTH(MODE, name) run<MODE, 1>(name, d_out, d_cyc, blocks); run<MODE, 4>(name, d_out, d_cyc, blocks); run<MODE, 8>(name, d_out, d_cyc, blocks);
    BOTH(0, "8 x ds_read_b128 broadcast, reads only")
    BOTH(1, "8 x ds_read_b128 broadcast + 16 mul + 16 add")
    BOTH(2, "16 x ds_read_b64 broadcast, reads only")
    BOTH(3, "8 x ds_read_b128 per-lane addresses, reads only")
    BOTH(4, "8 x ds_read_b64 per-lane addresses, reads only")
    BOTH(5, "1/4 per-lane ds_read_b64 + 32 v_readlane + 16 mul + 16 add")
    BOTH(6, "16 mul + 16 add alone")
    BOTH(7, "b128 broadcast, next piece's reads ahead, mul / add adjacent")
    BOTH(8, "b128 broadcast, next piece's reads ahead, 16 mul then 16 add")
    BOTH(9, "b128 broadcast, next piece's reads ahead, mul k+1 before add k")
    return 0;
}
