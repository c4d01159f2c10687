// Micro-benchmark: which instruction classes run under an FP64 matrix instruction on gfx950?
// One wavefront per SIMD (1024 workgroups of 64 threads), a loop of fixed instruction sequences, HIP events around it.
// hipcc --offload-arch=gfx950 -O2 tools/micro/pipe_overlap.hip -o /tmp/pipe_overlap && /tmp/pipe_overlap
// Prints wall cycles per loop iteration at the clock the run reports (s_memtime ticks are printed next to it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)

#define MFMA(n) "v_mfma_f64_16x16x4_f64 %" #n ", %8, %9, %" #n "\n"
#define XOR1 "v_xor_b32 %10, %10, %11\n"
#define ADD1 "v_add_u32 %10, %10, %11\n"
#define MAD1 "v_mad_u64_u32 %12, vcc, %10, %11, %12\n"
#define FMA1 "v_fma_f64 %13, %13, %9, %8\n"
#define MUL1 "v_mul_f64 %13, %13, %9\n"
#define ADDD "v_add_f64 %13, %13, %8\n"
#define DSR  "ds_read_b128 %14, %15\n"

#define OPERANDS                                                                                         \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
    : "v"(x), "v"(y), "v"(i0), "v"(i1), "v"(w64), "v"(z), "v"(ld), "v"(laddr)

#define BODY(name, text)                                                                                 \
    __global__ void __launch_bounds__(64) name(double* out, int iters, double x, double y) {            \
        __shared__ double lds[1024];                                                                     \
        lds[threadIdx.x] = x;                                                                            \
        f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;          \
        uint32_t i0 = threadIdx.x, i1 = 0x9E3779B9u;                                                     \
        uint64_t w64 = threadIdx.x;                                                                      \
        double z = x;                                                                                    \
        f64x4 ld4 = a0; typedef double d2 __attribute__((ext_vector_type(2))); d2 ld = {0, 0};           \
        uint32_t laddr = (uint32_t)(uintptr_t)(lds) + 16 * (threadIdx.x & 3);                            \
        uint64_t t0 = __builtin_readcyclecounter();                                                      \
        for (int it = 0; it < iters; ++it) {                                                             \
            asm volatile(text OPERANDS : "vcc");                                                         \
        }                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        uint64_t t1 = __builtin_readcyclecounter();                                                      \
        double s = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3] + (double)i0 + (double)w64 + z + ld[0] + ld4[0]; \
        if (s == 123.456) out[0] = s;                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (double)(t1 - t0);                             \
    }

// 8 matrix instructions on 8 accumulators
BODY(k_mfma, MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7))
BODY(k_xor, R8(R16(XOR1)))
BODY(k_mad, R8(R4(MAD1)))
BODY(k_fma, R8(R16(FMA1)))
BODY(k_muladd, R8(R8(MUL1 ADDD)))
BODY(k_dsr, R8(R8(DSR)))
// one matrix instruction, then 16 cheap integer instructions (64 cycles if they issue at 4 each), eight times
BODY(k_mfma_xor, MFMA(0) R16(XOR1) MFMA(1) R16(XOR1) MFMA(2) R16(XOR1) MFMA(3) R16(XOR1) MFMA(4) R16(XOR1) MFMA(5) R16(XOR1) MFMA(6) R16(XOR1) MFMA(7) R16(XOR1))
BODY(k_mfma_mad, MFMA(0) R4(MAD1) MFMA(1) R4(MAD1) MFMA(2) R4(MAD1) MFMA(3) R4(MAD1) MFMA(4) R4(MAD1) MFMA(5) R4(MAD1) MFMA(6) R4(MAD1) MFMA(7) R4(MAD1))
BODY(k_mfma_fma, MFMA(0) R16(FMA1) MFMA(1) R16(FMA1) MFMA(2) R16(FMA1) MFMA(3) R16(FMA1) MFMA(4) R16(FMA1) MFMA(5) R16(FMA1) MFMA(6) R16(FMA1) MFMA(7) R16(FMA1))
BODY(k_mfma_dsr, MFMA(0) R8(DSR) MFMA(1) R8(DSR) MFMA(2) R8(DSR) MFMA(3) R8(DSR) MFMA(4) R8(DSR) MFMA(5) R8(DSR) MFMA(6) R8(DSR) MFMA(7) R8(DSR))
// the Philox round's mix: 2 wide multiplies + 6 narrow integer instructions, per matrix instruction two rounds
#define ROUND MAD1 MAD1 XOR1 XOR1 XOR1 XOR1 ADD1 ADD1
BODY(k_round, R8(ROUND ROUND))
BODY(k_mfma_round, MFMA(0) ROUND ROUND MFMA(1) ROUND ROUND MFMA(2) ROUND ROUND MFMA(3) ROUND ROUND MFMA(4) ROUND ROUND MFMA(5) ROUND ROUND MFMA(6) ROUND ROUND MFMA(7) ROUND ROUND)

template <typename K> void run(const char* name, K k, double* out, int per_iter_note) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, 100, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double h[2]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-16s %8.1f ns/iter  (%7.1f cycles at 2.4 GHz)  counter ticks/iter %.1f   [%d instr groups]\n", name, ms * 1e6 / iters,
           ms * 1e6 / iters * 2.4, h[1] / iters, per_iter_note);
}

int main() {
    double* out; hipMalloc(&out, 64);
    run("mfma x8", k_mfma, out, 8);
    run("xor x128", k_xor, out, 128);
    run("mad64 x32", k_mad, out, 32);
    run("fma64 x128", k_fma, out, 128);
    run("mul+add x64", k_muladd, out, 128);
    run("ds_read_b128 x64", k_dsr, out, 64);
    run("mfma+16xor x8", k_mfma_xor, out, 8);
    run("mfma+4mad x8", k_mfma_mad, out, 8);
    run("mfma+16fma x8", k_mfma_fma, out, 8);
    run("mfma+8dsr x8", k_mfma_dsr, out, 8);
    run("philox rnd x16", k_round, out, 16);
    run("mfma+2rnd x8", k_mfma_round, out, 8);
    return 0;
}
