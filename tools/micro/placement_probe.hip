// Micro-benchmark: where do the two wavefronts of a 128-thread workgroup land?  1024 workgroups with 39 KB of LDS each
// (four per CU, as the SPLIT step kernel), every wavefront records HW_ID; the host counts, per SIMD, how many
// wavefronts of index 0 and of index 1 it hosts at the same time.
// hipcc --offload-arch=gfx950 -O2 tools/micro/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

__global__ void __launch_bounds__(128) probe(uint32_t* out, int spin) {
    __shared__ double pad[39384 / 8];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
    // keep the workgroup resident long enough for the whole grid to be in flight
    double x = pad[threadIdx.x];
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc & 0xf;
    }
    if (x == 123.0) out[0] = 0;
}

int main() {
    const int nwg = 1024;
    uint32_t* d;
    (void)hipMalloc(&d, nwg * 4 * sizeof(uint32_t));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(128), 0, 0, d, 200000);
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> h(nwg * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    std::map<uint32_t, int> w0, w1;
    int same_simd = 0, adjacent = 0, same_slot = 0;
    for (int g = 0; g < nwg; ++g) {
        const uint32_t a = h[(g * 2) * 2], b = h[(g * 2 + 1) * 2], xa = h[(g * 2) * 2 + 1];
        const uint32_t simd_a = (a >> 4) & 3, simd_b = (b >> 4) & 3;
        const uint32_t key_a = (xa << 20) | (a & 0xfff0), key_b = (h[(g * 2 + 1) * 2 + 1] << 20) | (b & 0xfff0);
        w0[key_a]++; w1[key_b]++;
        same_simd += simd_a == simd_b;
        adjacent += (simd_a ^ simd_b) == 1;
        same_slot += (a & 15) == (b & 15);
        if (g < 12) printf("wg %d: wave0 cu %u simd %u slot %u | wave1 cu %u simd %u slot %u | xcc %u\n", g, (a >> 8) & 15, simd_a, a & 15,
                           (b >> 8) & 15, simd_b, b & 15, xa);
    }
    int balanced = 0, two_zero = 0, two_one = 0, other = 0;
    std::map<uint32_t, int> all;
    for (auto& kv : w0) all[kv.first] += 0;
    for (auto& kv : w1) all[kv.first] += 0;
    for (auto& kv : all) {
        const int n0 = w0.count(kv.first) ? w0[kv.first] : 0, n1 = w1.count(kv.first) ? w1[kv.first] : 0;
        if (n0 == 1 && n1 == 1) ++balanced; else if (n0 == 2 && n1 == 0) ++two_zero; else if (n0 == 0 && n1 == 2) ++two_one; else ++other;
    }
    printf("workgroups: both wavefronts on one SIMD %d, on neighbouring SIMDs %d, in the same slot %d (of %d)\n", same_simd, adjacent, same_slot, nwg);
    printf("SIMDs seen %zu: one wave0 + one wave1: %d, two wave0: %d, two wave1: %d, other: %d\n", all.size(), balanced, two_zero, two_one, other);
    // the kernel's key: (simd ^ slot) & 1 differs between the two wavefronts of a workgroup?
    int differ = 0;
    for (int g = 0; g < nwg; ++g) {
        const uint32_t a = h[(g * 2) * 2], b = h[(g * 2 + 1) * 2];
        differ += (((a ^ (a >> 4)) & 1) != ((b ^ (b >> 4)) & 1));
    }
    printf("workgroups whose wavefronts get different role keys: %d of %d\n", differ, nwg);
    return 0;
}
