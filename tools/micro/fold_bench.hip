// Micro-benchmark + bitwise check of the large-dimension moment fold: fold_moments_kernel (rounds 1-3) against
// fold_ring_kernel (round 4) on BASELINE config 3 / config 4 shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iroot-simple-mcmc_amd/csrc \
//         tools/micro/fold_bench.hip -o /tmp/fold_bench && /tmp/fold_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "smcmc_fold_ring.hip.h"
#include "fold_moments_old.hip.h"

using namespace smcmc;

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                        \
        }                                                                                        \
    } while (0)

#include "../../root-simple-mcmc_amd/csrc/smcmc_fold_inst.hip"

static void run_case(const char* name, int D, int nchains, int nsrc, bool masked, int reps, bool same = false) {
    const int npad = (nchains + 63) / 64 * 64;
    const int ngroups = npad / 64;
    const int nslices = fold_slices(D);
    const int slice_chains = ((ngroups + nslices - 1) / nslices) * 64;
    const int T = (D + 1 + 15) / 16, ntiles = T * (T + 1) / 2;
    const size_t state = (size_t)D * npad;
    const size_t gacc_n = (size_t)nslices * ntiles * 4 * 64;
    std::mt19937_64 rng(12345 + D);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> hx(state * nsrc), hc0(D);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hc0) v = 0.1 * nd(rng);
    std::vector<int32_t> hmask(npad, 1);
    if (masked)
        for (int c = 0; c < npad; ++c) hmask[c] = (rng() % 5) != 0;
    double *dx, *dc0, *ga, *gb;
    int32_t* dmask;
    CK(hipMalloc(&dx, sizeof(double) * state * nsrc));
    CK(hipMalloc(&dc0, sizeof(double) * D));
    CK(hipMalloc(&ga, sizeof(double) * gacc_n));
    CK(hipMalloc(&gb, sizeof(double) * gacc_n));
    CK(hipMalloc(&dmask, sizeof(int32_t) * npad));
    CK(hipMemcpy(dx, hx.data(), sizeof(double) * state * nsrc, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc0, hc0.data(), sizeof(double) * D, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmask, hmask.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    auto plan = fold_ring_plan(D, nchains, npad, nslices, slice_chains, prop.multiProcessorCount);
    FoldRing fr;
    CK(fold_ring_prepare(fr, D, nchains, npad, nslices, slice_chains));
    int maxt = plan.max_tiles, used = 0;
    double foot = 0;
    for (auto& e : plan.wg)
        if (e.ntiles) { ++used; foot += e.nfoot; }
    foot /= used;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto old_folds = [&]() {
        for (int k = 0; k < nsrc; ++k)
            hipLaunchKernelGGL(fold_moments_kernel, dim3(fold_super_blocks(D), nslices), dim3(kFoldWaves * kWave), 0, s,
                               dx + (same ? 0 : state * k), dc0, nchains, npad, D, slice_chains, ga, masked ? dmask : nullptr);
    };
    FoldRingParams p;
    std::memset(&p, 0, sizeof(p));
    for (int k = 0; k < nsrc; ++k) p.src[k] = dx + (same ? 0 : state * k);
    p.nsrc = nsrc; p.c0 = dc0; p.nchains = nchains; p.npad = npad; p.D = D; p.slice_chains = slice_chains;
    p.gacc = gb; p.mask = masked ? dmask : nullptr;
    auto new_folds = [&]() { CK(launch_fold_ring(fr, p, s)); };
    // bitwise check: two rounds each on zeroed accumulators
    CK(hipMemsetAsync(ga, 0, sizeof(double) * gacc_n, s));
    CK(hipMemsetAsync(gb, 0, sizeof(double) * gacc_n, s));
    old_folds(); old_folds();
    new_folds(); new_folds();
    CK(hipStreamSynchronize(s));
    std::vector<double> ha(gacc_n), hb(gacc_n);
    CK(hipMemcpy(ha.data(), ga, sizeof(double) * gacc_n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), gb, sizeof(double) * gacc_n, hipMemcpyDeviceToHost));
    size_t bad = 0, nonzero = 0;
    for (size_t i = 0; i < gacc_n; ++i) {
        if (std::memcmp(&ha[i], &hb[i], 8) != 0) ++bad;
        if (ha[i] != 0.0) ++nonzero;
    }
    float ms_old = 0, ms_new = 0;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) old_folds();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_old, e0, e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) new_folds();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_new, e0, e1));
    }
    const double kq = (double)npad / 4.0;
    const double floor_us = (double)ntiles * kq * 64.0 / 1024.0 / 2.4e3;
    std::printf("%-28s D=%d chains=%d nsrc=%d mask=%d slices=%d x %d  plan: %d wg, <=%d tiles/wave, bands of %d, footprint %.1f (<= %d) of %d tiles, %d rounds | mismatches %zu of %zu (%zu nonzero)"
                " | per fold: old %.1f us, new %.1f us (matrix-pipe floor %.1f us)\n",
                name, D, nchains, nsrc, (int)masked, nslices, slice_chains, used, maxt, plan.band, foot, plan.max_foot, T, plan.rounds, bad, gacc_n, nonzero,
                1e3 * ms_old / reps / nsrc, 1e3 * ms_new / reps / nsrc, floor_us);
    std::fflush(stdout);
    CK(hipFree(dx)); CK(hipFree(dc0)); CK(hipFree(ga)); CK(hipFree(gb)); CK(hipFree(dmask)); fold_ring_release(fr);
    CK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
    const char* only = argc > 2 ? argv[2] : nullptr;   // run the cases whose name contains this
    struct Case { const char* name; int D, nchains, nsrc; bool masked; bool same = false; };
    const Case cases[] = {{"small ragged", 70, 200, 3, false}, {"small masked", 100, 1000, 1, true},
                          {"config 3, one point", 200, 16384, 1, false}, {"config 3, ring of 8", 200, 16384, 8, false},
                          {"config 4 share, one point", 500, 32768, 1, false}, {"config 4 share, ring of 8", 500, 32768, 8, false},
                          {"config 4 share, ring of 16", 500, 32768, 16, false},
                          {"config 4 share, one point 8 times", 500, 32768, 8, false, true},
                          {"config 4 share + 64 chains (row pitch not a power of two), ring of 8", 500, 32768 + 64, 8, false},
                          {"config 3 + 64 chains, ring of 8", 200, 16384 + 64, 8, false},
                          {"D=512 max", 512, 8192, 2, false}, {"hmc config 5 masked", 500, 8192, 1, true}};
    for (const Case& c : cases)
        if (!only || std::strstr(c.name, only)) run_case(c.name, c.D, c.nchains, c.nsrc, c.masked, reps, c.same);
    return 0;
}
