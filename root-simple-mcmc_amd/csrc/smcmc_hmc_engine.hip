// smcmc_hmc_engine.hip -- host engine behind the smcmc_hmc_* entry points of
// include/smcmc.h: the many-chain form of sMCMC::TSimpleHMC (reference
// TSimpleHMC.H:119-973).  With a fixed step length and leapfrog count (SetMeanEpsilon(<0) +
// SetLeapFrog(n)) the chains share nothing and a launch runs any number of steps.  Otherwise
// (the reference's default) every chain retunes its own step length and leapfrog count as it
// goes (:302-323, 342-344) and the covariance-driven retuning (:665-858) is pooled over the
// ensemble: one step per launch, the accepted points folded into moment sums on the device
// (smcmc_fold_kernel.hip.h), the pooled running covariance and UpdateErrorMatrix on the host
// (smcmc_hmc_shared.hpp) every `sync_every` steps, the new step length / leapfrog count applied
// per chain on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smcmc.h"
#include "smcmc_hmc_kernel.hip.h"
#include "smcmc_hmc_mfma_kernel.hip.h"
#include "smcmc_fold_ring.hip.h"
#include "smcmc_hmc_shared.hpp"

using namespace smcmc;

struct smcmc_hmc {
    int dim = 0, nchains = 0, npad = 0, likelihood = 0, device = 0, W = 4;
    uint64_t seed = 0;
    uint32_t chain_offset = 0, step_count = 0;
    bool started = false;
    bool exact = true;             // false: fused order, the quadratic-form gradient on the matrix pipe
    bool use_mfma = false;
    bool use_matrix_exact = false;   // quadratic form, reference order: the matrix layout on the vector pipe
    double alpha = 0.0;            // fAlpha, TSimpleHMC.H:133
    double mean_epsilon = 0.05;    // fMeanEpsilon, set by Start (:229)
    int leapfrog = 10;             // fLeapFrogSteps (:133); SetLeapFrog(n) stores -n (:190)
    hipStream_t stream = nullptr;
    std::vector<double> like_params;
    double *d_q = nullptr, *d_pm = nullptr, *d_qn = nullptr, *d_pn = nullptr, *d_E = nullptr, *d_like = nullptr;
    double* d_lane_f64 = nullptr;
    int32_t* d_lane_i32 = nullptr;
    // pooled tuning (adaptive step length / leapfrog count, or track_cov)
    HmcShared* shared = nullptr;
    bool track_cov = false;        // keep the running covariance even with a fixed step and count
    int sync_every = 1, steps_in_window = 0;
    int steps_reduced = 0;   // steps whose moments are in d_moments, waiting for hmc_apply (between reduce and apply)
    int fold_nslices = 0, slice_chains = 0;
    smcmc::FoldRing fold;   // the fold kernel's plan for this ensemble
    double *d_p0 = nullptr, *d_qprev = nullptr, *d_gacc = nullptr, *d_moments = nullptr, *d_zero = nullptr;
    double* h_moments = nullptr;   // pinned: the packed moments come back every sync
    // PotentialGradient types 2 / 3 / 5 (TSimpleHMC.H:467-532): the GENERIC instantiation of hmc_step_kernel
    int gradient_type = 0;
    double *d_Eperm = nullptr;     // QUADFORM: Error in hmc_step_kernel's layout (d_E holds the matrix kernels')
    double *d_covE = nullptr, *d_cov_avg = nullptr, *d_fd_grad = nullptr;
    bool cov_dirty = true;         // fEstimatedError / fAveragePoint changed since the last upload
    // the running average point / covariance on the device (hmc_absorb_* kernels): the host copy in *shared follows
    // on demand (hmc_pull) or when UpdateErrorMatrix decides to run
    double *d_avg = nullptr, *d_exxt = nullptr, *d_hcov = nullptr, *d_hscal = nullptr;
    double* h_hscal = nullptr;     // pinned: {n, average trials, covariance trials, trace}
    bool host_stale = false;       // the device holds newer average / covariance than *shared
    bool shared_on_device = false; // hmc_push has run since the host last (re)initialised *shared
    std::string error;
};

namespace {

int hfail(smcmc_hmc* h, int status, const std::string& msg) {
    if (h) h->error = msg;
    return status;
}

#define HMC_TRY(h, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return hfail((h), SMCMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct HmcDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit HmcDeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~HmcDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    HmcDeviceGuard(const HmcDeviceGuard&) = delete;
    HmcDeviceGuard& operator=(const HmcDeviceGuard&) = delete;
};
#define HMC_ON_DEVICE(h) HmcDeviceGuard device_guard_((h)->device)

size_t hmc_npacked(const smcmc_hmc* h) { return (size_t)(h->dim + 1) * (h->dim + 2) / 2; }
size_t hmc_gacc_doubles(const smcmc_hmc* h) {
    const size_t T = (size_t)(h->dim + 1 + 15) / 16;
    return (size_t)fold_slices(h->dim) * (T * (T + 1) / 2) * 4 * kWave;
}

// the chains retune themselves (TSimpleHMC.H:302-345, 833-847) unless both the step length and the count are fixed
bool hmc_adaptive(const smcmc_hmc* h) { return h->mean_epsilon > 0.0 || h->leapfrog > 0; }
// likelihoods without a gradient of their own (the reference's functors throw / return false, TAsymLogLikelihood.H:34-36,
// TSimpleHMC.H:85-89): HMC targets through PotentialGradient types 2 / 3 / 5 only
bool hmc_no_own_gradient(int like) {
    return like == SMCMC_LIKE_USER || like == SMCMC_LIKE_ASYM || like == SMCMC_LIKE_HORRIFIC || like == SMCMC_LIKE_CONSTRAINED;
}
size_t hmc_like_doubles(int dim) { return (size_t)dim * dim + 2 * (size_t)dim + 8; }
bool hmc_generic_gradient(const smcmc_hmc* h) { return h->gradient_type == 2 || h->gradient_type == 3 || h->gradient_type == 5; }
// the covariant gradient reads the running covariance: it has to be kept
bool hmc_tracking(const smcmc_hmc* h) { return hmc_adaptive(h) || h->track_cov || h->gradient_type == 2; }

// What an UpdateErrorMatrix that went through does to every chain (TSimpleHMC.H:833-847)
__global__ void hmc_retune_kernel(double* lane_f64, int32_t* lane_i32, int npad, int nchains, double max_scale,
                                  double min_scale, double orbit, int dim) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    double eps = lane_f64[(size_t)kHmcLaneMeanEpsilon * npad + c];
    int L = lane_i32[(size_t)kHmcLaneLeapfrog * npad + c];
    if (eps > 0) {                                                       // :835-839
        eps = 0.2 * max_scale;
        if (eps > 0.5 * min_scale) eps = 0.5 * min_scale;
        if (eps < 0.05 * max_scale) eps = 0.05 * max_scale;
    }
    if (L > 0) {                                                         // :841-848
        const double target = 0.4 * orbit;
        L = (int)(target / __builtin_fabs(eps));
        L = 2 * (L / 2 + 1);
        if (L > 3 * dim) L = 3 * dim;
        if (eps > 0) eps = target / L;
    }
    lane_f64[(size_t)kHmcLaneMeanEpsilon * npad + c] = eps;
    lane_i32[(size_t)kHmcLaneLeapfrog * npad + c] = L;
}

template <typename T>
__global__ void hmc_fill_lane_kernel(T* col, int nchains, T v) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchains) col[c] = v;
}

HmcParams hmc_params(smcmc_hmc* h, int nsteps, int init_only) {
    HmcParams p;
    std::memset(&p, 0, sizeof(p));
    p.nchains = h->nchains; p.npad = h->npad; p.dim = h->dim; p.nsteps = nsteps;
    p.leapfrog = std::abs(h->leapfrog);                    // :300
    p.init_only = init_only;
    p.step0 = h->step_count; p.chain_offset = h->chain_offset; p.seed = h->seed;
    p.alpha = h->alpha; p.abs_eps = std::fabs(h->mean_epsilon);
    p.Eperm = h->d_E; p.like = h->d_like;
    p.q = h->d_q; p.pm = h->d_pm; p.qn = h->d_qn; p.pn = h->d_pn;
    p.lane_f64 = h->d_lane_f64; p.lane_i32 = h->d_lane_i32;
    p.p0 = h->d_p0; p.qprev = h->d_qprev;
    p.gradient_type = h->gradient_type;
    if (hmc_generic_gradient(h)) {
        p.Eperm = h->d_Eperm;
        p.cov_Eperm = h->d_covE; p.cov_average = h->d_cov_avg; p.fd_grad = h->d_fd_grad;
    }
    return p;
}

// M [dim][dim] row-major -> hmc_step_kernel's layout: out[w][j][il] = M(il*W + w, j)
std::vector<double> hmc_permute(const smcmc_hmc* h, const double* M) {
    const int D = h->dim, W = h->W;
    std::vector<double> perm((size_t)W * D * kPanelCW, 0.0);
    for (int w = 0; w < W; ++w)
        for (int j = 0; j < D; ++j)
            for (int il = 0; il < kPanelCW; ++il) {
                const int i = il * W + w;
                if (i < D) perm[((size_t)w * D + j) * kPanelCW + il] = M[(size_t)i * D + j];
            }
    return perm;
}

// buffers of the GENERIC gradient types, allocated when one is first asked for; the estimated error matrix and the
// average point go up again whenever the pooled update changed them
int hmc_generic_buffers(smcmc_hmc* h) {
    const int D = h->dim;
    const size_t perm_bytes = sizeof(double) * (size_t)h->W * D * kPanelCW;
    if (h->likelihood == SMCMC_LIKE_QUADFORM && !h->d_Eperm) {
        HMC_TRY(h, hipMalloc(&h->d_Eperm, perm_bytes));
        const std::vector<double> perm = hmc_permute(h, h->like_params.data());
        HMC_TRY(h, hipMemcpyAsync(h->d_Eperm, perm.data(), perm_bytes, hipMemcpyHostToDevice, h->stream));
        HMC_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (h->gradient_type == 3 && !h->d_fd_grad) {
        HMC_TRY(h, hipMalloc(&h->d_fd_grad, sizeof(double) * (size_t)h->npad * D));
        HMC_TRY(h, hipMemsetAsync(h->d_fd_grad, 0, sizeof(double) * (size_t)h->npad * D, h->stream));
    }
    if (h->gradient_type == 2) {
        if (!h->d_covE) {
            HMC_TRY(h, hipMalloc(&h->d_covE, perm_bytes));
            HMC_TRY(h, hipMalloc(&h->d_cov_avg, sizeof(double) * D));
            h->cov_dirty = true;
        }
        if (h->cov_dirty) {
            const std::vector<double> perm = hmc_permute(h, h->shared->error.data());
            HMC_TRY(h, hipMemcpyAsync(h->d_covE, perm.data(), perm_bytes, hipMemcpyHostToDevice, h->stream));
            if (h->shared_on_device)   // the running average lives on the device
                HMC_TRY(h, hipMemcpyAsync(h->d_cov_avg, h->d_avg, sizeof(double) * D, hipMemcpyDeviceToDevice, h->stream));
            else
                HMC_TRY(h, hipMemcpyAsync(h->d_cov_avg, h->shared->average.data(), sizeof(double) * D, hipMemcpyHostToDevice, h->stream));
            HMC_TRY(h, hipStreamSynchronize(h->stream));   // the staging vector goes out of scope
            h->cov_dirty = false;
        }
    }
    return SMCMC_OK;
}

template <typename T>
int hmc_fill_lane(smcmc_hmc* h, T* col, T v) {
    const int threads = 256;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hmc_fill_lane_kernel<T>), dim3((h->nchains + threads - 1) / threads), dim3(threads), 0,
                       h->stream, col, h->nchains, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("lane fill launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

// buffers of the pooled tuning, allocated when it is first needed
int hmc_tracking_buffers(smcmc_hmc* h) {
    if (h->d_gacc) return SMCMC_OK;
    const size_t vec = sizeof(double) * (size_t)h->npad * h->dim;
    HMC_TRY(h, hipMalloc(&h->d_p0, vec));
    HMC_TRY(h, hipMalloc(&h->d_qprev, vec));
    HMC_TRY(h, hipMalloc(&h->d_gacc, sizeof(double) * hmc_gacc_doubles(h)));
    HMC_TRY(h, smcmc::fold_ring_prepare(h->fold, h->dim, h->nchains, h->npad, h->fold_nslices, h->slice_chains));
    HMC_TRY(h, hipMalloc(&h->d_moments, sizeof(double) * hmc_npacked(h)));
    HMC_TRY(h, hipMalloc(&h->d_zero, sizeof(double) * h->dim));
    HMC_TRY(h, hipHostMalloc((void**)&h->h_moments, sizeof(double) * hmc_npacked(h), hipHostMallocDefault));
    HMC_TRY(h, hipMalloc(&h->d_avg, sizeof(double) * h->dim));
    HMC_TRY(h, hipMalloc(&h->d_exxt, sizeof(double) * (size_t)h->dim * h->dim));
    HMC_TRY(h, hipMalloc(&h->d_hcov, sizeof(double) * (size_t)h->dim * h->dim));
    HMC_TRY(h, hipMalloc(&h->d_hscal, sizeof(double) * 8));
    HMC_TRY(h, hipHostMalloc((void**)&h->h_hscal, sizeof(double) * 8, hipHostMallocDefault));
    h->shared_on_device = false;
    HMC_TRY(h, hipMemsetAsync(h->d_p0, 0, vec, h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_qprev, 0, vec, h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * hmc_gacc_doubles(h), h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_moments, 0, sizeof(double) * hmc_npacked(h), h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_zero, 0, sizeof(double) * h->dim, h->stream));
    return SMCMC_OK;
}

// ---- UpdateCovariance (TSimpleHMC.H:665-695) fed with a batch, on the device: the arithmetic of HmcShared::absorb ----
enum { kHsN = 0, kHsAverageTrials, kHsCovTrials, kHsTrace, kHsCount };

__global__ void hmc_absorb_average_kernel(const double* M, int D, double* avg, const double* scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const double* S1 = M + (size_t)D * (D + 1) / 2;
    const double n = S1[D];
    if (!(n > 0.0)) return;
    const double trials = scal[kHsAverageTrials];
    double v = avg[i];                                                   // :671-677
    v *= trials;
    v += S1[i];
    v /= trials + n;
    avg[i] = v;
}

__global__ void __launch_bounds__(256) hmc_absorb_cov_kernel(const double* M, int D, const double* avg, double* exxt, double* cov,
                                                             const double* scal) {
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= D || j > i) return;
    const double n = M[(size_t)D * (D + 1) / 2 + D];
    if (!(n > 0.0)) return;
    const double trials = scal[kHsCovTrials];
    double v = exxt[(size_t)i * D + j];                                  // :681-691
    v *= trials;
    v += M[(size_t)i * (i + 1) / 2 + j];
    v /= trials + n;
    exxt[(size_t)i * D + j] = v;
    exxt[(size_t)j * D + i] = v;
    const double c = v - avg[i] * avg[j];
    cov[(size_t)i * D + j] = c;
    cov[(size_t)j * D + i] = c;
}

// trial counts (:678-679, 692-693) and the trace UpdateErrorMatrix looks at (:708-711): one wavefront
__global__ void __launch_bounds__(64) hmc_absorb_scalars_kernel(const double* M, int D, const double* cov, double* scal,
                                                                double cov_window) {
    __shared__ double diag[512];
    for (int d = threadIdx.x; d < D; d += 64) diag[d] = cov[(size_t)d * D + d];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double n = M[(size_t)D * (D + 1) / 2 + D];
    scal[kHsN] = n;
    if (!(n > 0.0)) return;
    scal[kHsAverageTrials] = __builtin_fmin(cov_window, scal[kHsAverageTrials] + n);
    scal[kHsCovTrials] = __builtin_fmin(cov_window, scal[kHsCovTrials] + n);
    double trace = 0.0;
    for (int d = 0; d < D; ++d) trace += __builtin_fabs(diag[d]);
    scal[kHsTrace] = trace;
}

// host copy of the running average / covariance -> device (Start, or after the host changed them)
int hmc_push(smcmc_hmc* h) {
    const HmcShared& S = *h->shared;
    const size_t D = (size_t)h->dim;
    double sc[kHsCount] = {0.0, S.averageTrials, S.covTrials, 0.0};
    HMC_TRY(h, hipMemcpyAsync(h->d_avg, S.average.data(), D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipMemcpyAsync(h->d_exxt, S.exxt.data(), D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipMemcpyAsync(h->d_hcov, S.cov.data(), D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipMemcpyAsync(h->d_hscal, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    h->host_stale = false;
    return SMCMC_OK;
}

// device -> host copy, when somebody asks for fAveragePoint / fEstimatedCovariance
int hmc_pull(smcmc_hmc* h) {
    if (!h->host_stale) return SMCMC_OK;
    HmcShared& S = *h->shared;
    const size_t D = (size_t)h->dim;
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    HMC_TRY(h, hipMemcpy(S.average.data(), h->d_avg, D * sizeof(double), hipMemcpyDeviceToHost));
    HMC_TRY(h, hipMemcpy(S.exxt.data(), h->d_exxt, D * D * sizeof(double), hipMemcpyDeviceToHost));
    HMC_TRY(h, hipMemcpy(S.cov.data(), h->d_hcov, D * D * sizeof(double), hipMemcpyDeviceToHost));
    h->host_stale = false;
    return SMCMC_OK;
}

// The moment groups of the steps since the last update summed (in group order) into the packed vector M, which is what
// crosses ranks when the ensemble is sharded (smcmc_hmc_export_moments / import).
int hmc_reduce(smcmc_hmc* h) {
    // d_moments holds ONE reduction: a second one (an explicit smcmc_hmc_reduce_moments, or the sync a step triggers
    // between a caller's reduce / import and its apply) would overwrite moments that no update has absorbed yet
    if (h->steps_reduced > 0)
        return hfail(h, SMCMC_ERR_LOGIC, "moments of an earlier smcmc_hmc_reduce_moments are waiting for smcmc_hmc_apply_moments");
    h->steps_reduced += h->steps_in_window;
    h->steps_in_window = 0;
    hipError_t e = launch_fold_reduce(h->d_gacc, h->dim, h->fold_nslices, h->d_moments, h->stream);
    if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("fold reduce launch: ") + hipGetErrorString(e));
    HMC_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * hmc_gacc_doubles(h), h->stream));
    return SMCMC_OK;
}

int hmc_apply(smcmc_hmc* h);

// The pooled UpdateCovariance + UpdateErrorMatrix (TSimpleHMC.H:337-341) for the steps since the last one
int hmc_sync(smcmc_hmc* h) {
    if (h->steps_in_window == 0) return SMCMC_OK;
    int st = hmc_reduce(h);
    if (st) return st;
    return hmc_apply(h);
}

// the running averages absorb the batch M and UpdateErrorMatrix decides (every rank of a sharded ensemble the same)
int hmc_apply(smcmc_hmc* h) {
    const int steps = h->steps_reduced;
    h->steps_reduced = 0;
    if (steps == 0) return SMCMC_OK;
    hipError_t e;
    // the running averages absorb the batch on the device; four scalars come back for UpdateErrorMatrix's decision
    HmcShared& S = *h->shared;
    if (!h->shared_on_device) {
        int pst = hmc_push(h);
        if (pst) return pst;
        h->shared_on_device = true;
    }
    const int D = h->dim, t16 = (D + 15) / 16;
    hipLaunchKernelGGL(hmc_absorb_average_kernel, dim3((D + 255) / 256), dim3(256), 0, h->stream, (const double*)h->d_moments, D,
                       h->d_avg, (const double*)h->d_hscal);
    hipLaunchKernelGGL(hmc_absorb_cov_kernel, dim3(t16, t16), dim3(256), 0, h->stream, (const double*)h->d_moments, D,
                       (const double*)h->d_avg, h->d_exxt, h->d_hcov, (const double*)h->d_hscal);
    hipLaunchKernelGGL(hmc_absorb_scalars_kernel, dim3(1), dim3(64), 0, h->stream, (const double*)h->d_moments, D,
                       (const double*)h->d_hcov, h->d_hscal, S.covWindow);
    e = hipGetLastError();
    if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("absorb launch: ") + hipGetErrorString(e));
    HMC_TRY(h, hipMemcpyAsync(h->h_hscal, h->d_hscal, sizeof(double) * kHsCount, hipMemcpyDeviceToHost, h->stream));
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    if (!(h->h_hscal[kHsN] > 0.0)) return SMCMC_OK;
    S.stepCount = (int)h->step_count;
    S.leapfrogZero = (h->leapfrog == 0);
    S.absorbedOnDevice(steps, h->h_hscal[kHsAverageTrials], h->h_hscal[kHsCovTrials]);
    h->host_stale = true;
    h->cov_dirty = true;
    bool updated = false;
    if (S.wantsUpdate(h->h_hscal[kHsTrace])) {
        // the O(D^3) part (eigenvalues, repair, inverse) stays on the host; it needs the covariance there
        int pst = hmc_pull(h);
        if (pst) return pst;
        S.finishUpdate();
        updated = true;
    }
    if (updated) {
        const int threads = 256;
        hipLaunchKernelGGL(hmc_retune_kernel, dim3((h->nchains + threads - 1) / threads), dim3(threads), 0, h->stream,
                           h->d_lane_f64, h->d_lane_i32, h->npad, h->nchains, S.maxScale, S.minScale, S.orbitLength,
                           h->dim);
        e = hipGetLastError();
        if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("retune launch: ") + hipGetErrorString(e));
    }
    return SMCMC_OK;
}

hipError_t hmc_dispatch(smcmc_hmc* h, const HmcParams& p) {
    const bool generic = hmc_generic_gradient(h) && !p.init_only;
    if (h->use_mfma && !generic) return launch_hmc_mfma(p, h->stream);
    if (h->use_matrix_exact && !generic) return launch_hmc_matrix_exact(p, h->stream);
    return (h->W == 4) ? launch_hmc<4, kPanelCW>(p, h->likelihood, h->stream)
                       : launch_hmc<8, kPanelCW>(p, h->likelihood, h->stream);
}

}  // namespace

extern "C" {

int smcmc_hmc_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset, int device,
                     smcmc_hmc** out) {
    if (!out) return SMCMC_ERR_INVALID;
    *out = nullptr;
    if (dim < 1 || nchains < 1) return SMCMC_ERR_INVALID;
    if (likelihood < SMCMC_LIKE_ISO_GAUSS || likelihood > SMCMC_LIKE_CONSTRAINED) return SMCMC_ERR_INVALID;
#ifndef SMCMC_USER_LIKELIHOOD_ANY_DIM
    // a compiled-in user likelihood is an HMC target when its header has the form that walks a point in device memory
    if (likelihood == SMCMC_LIKE_USER) return SMCMC_ERR_UNSUPPORTED;
#endif
    if (likelihood == SMCMC_LIKE_ROSENBROCK && dim < 2) return SMCMC_ERR_INVALID;
    if (dim > 8 * kPanelCW) return SMCMC_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return SMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    smcmc_hmc* h = new (std::nothrow) smcmc_hmc();
    if (!h) return SMCMC_ERR_RUNTIME;
    h->dim = dim; h->nchains = nchains; h->likelihood = likelihood; h->seed = seed;
    h->chain_offset = chain_offset; h->device = device;
    h->W = (dim <= 4 * kPanelCW) ? 4 : 8;
    h->npad = (nchains + kWave - 1) / kWave * kWave;
    h->fold_nslices = fold_slices(dim);
    h->slice_chains = ((h->npad / kWave + h->fold_nslices - 1) / h->fold_nslices) * kWave;
    h->shared = new HmcShared(dim);
    *out = h;
    HMC_ON_DEVICE(h);
    const size_t vec = sizeof(double) * (size_t)h->npad * dim;
    HMC_TRY(h, hipMalloc(&h->d_q, vec));
    HMC_TRY(h, hipMalloc(&h->d_pm, vec));
    HMC_TRY(h, hipMalloc(&h->d_qn, vec));
    HMC_TRY(h, hipMalloc(&h->d_pn, vec));
    const size_t e_doubles = std::max({(size_t)h->W * dim * kPanelCW, hmc_mfma_eop_doubles(dim), hmc_exact_ex_doubles(dim)});
    HMC_TRY(h, hipMalloc(&h->d_E, sizeof(double) * e_doubles));
    HMC_TRY(h, hipMalloc(&h->d_like, sizeof(double) * hmc_like_doubles(dim)));
    HMC_TRY(h, hipMalloc(&h->d_lane_f64, sizeof(double) * (size_t)h->npad * SMCMC_LANE_F64_COUNT_));
    HMC_TRY(h, hipMalloc(&h->d_lane_i32, sizeof(int32_t) * (size_t)h->npad * SMCMC_LANE_I32_COUNT_));
    HMC_TRY(h, hipMemset(h->d_q, 0, vec));
    HMC_TRY(h, hipMemset(h->d_pm, 0, vec));
    HMC_TRY(h, hipMemset(h->d_qn, 0, vec));
    HMC_TRY(h, hipMemset(h->d_pn, 0, vec));
    HMC_TRY(h, hipMemset(h->d_E, 0, sizeof(double) * e_doubles));
    HMC_TRY(h, hipMemset(h->d_like, 0, sizeof(double) * hmc_like_doubles(dim)));
    HMC_TRY(h, hipMemset(h->d_lane_f64, 0, sizeof(double) * (size_t)h->npad * SMCMC_LANE_F64_COUNT_));
    HMC_TRY(h, hipMemset(h->d_lane_i32, 0, sizeof(int32_t) * (size_t)h->npad * SMCMC_LANE_I32_COUNT_));
    return SMCMC_OK;
}

int smcmc_hmc_destroy(smcmc_hmc* h) {
    if (!h) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    if (h->d_q) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_q); (void)hipFree(h->d_pm); (void)hipFree(h->d_qn); (void)hipFree(h->d_pn);
    (void)hipFree(h->d_E); (void)hipFree(h->d_like); (void)hipFree(h->d_lane_f64); (void)hipFree(h->d_lane_i32);
    (void)hipFree(h->d_p0); (void)hipFree(h->d_qprev); (void)hipFree(h->d_gacc); (void)hipFree(h->d_moments);
    smcmc::fold_ring_release(h->fold);
    (void)hipFree(h->d_zero); (void)hipFree(h->d_Eperm); (void)hipFree(h->d_covE); (void)hipFree(h->d_cov_avg);
    (void)hipFree(h->d_avg); (void)hipFree(h->d_exxt); (void)hipFree(h->d_hcov); (void)hipFree(h->d_hscal);
    (void)hipHostFree(h->h_hscal);
    (void)hipFree(h->d_fd_grad);
    (void)hipHostFree(h->h_moments);
    delete h->shared;
    delete h;
    return SMCMC_OK;
}

const char* smcmc_hmc_last_error(const smcmc_hmc* h) { return h ? h->error.c_str() : "null engine"; }

int smcmc_hmc_set_stream(smcmc_hmc* h, void* hip_stream) {
    if (!h) return SMCMC_ERR_INVALID;
    h->stream = (hipStream_t)hip_stream;
    return SMCMC_OK;
}

int smcmc_hmc_set_likelihood_params(smcmc_hmc* h, const double* params, int count) {
    if (!h || count < 0 || (count > 0 && !params)) return SMCMC_ERR_INVALID;
    h->like_params.assign(params, params + count);
    return SMCMC_OK;
}

int smcmc_hmc_set_exact_arithmetic(smcmc_hmc* h, int exact) {
    if (!h) return SMCMC_ERR_INVALID;
    if (h->started) return hfail(h, SMCMC_ERR_LOGIC, "choose the arithmetic before Start");
    h->exact = exact != 0;
    return SMCMC_OK;
}
int smcmc_hmc_set_alpha(smcmc_hmc* h, double a) { if (!h) return SMCMC_ERR_INVALID; h->alpha = a; return SMCMC_OK; }
// SetMeanEpsilon / SetLeapFrog reach every chain's own copy (TSimpleHMC.H:181, 190)
int smcmc_hmc_set_mean_epsilon(smcmc_hmc* h, double e) {
    if (!h) return SMCMC_ERR_INVALID;
    h->mean_epsilon = e;
    if (!h->started) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    return hmc_fill_lane<double>(h, h->d_lane_f64 + (size_t)kHmcLaneMeanEpsilon * h->npad, e);
}
int smcmc_hmc_set_leapfrog(smcmc_hmc* h, int n) {
    if (!h) return SMCMC_ERR_INVALID;
    h->leapfrog = -n;
    if (!h->started) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    return hmc_fill_lane<int32_t>(h, h->d_lane_i32 + (size_t)kHmcLaneLeapfrog * h->npad, (int32_t)-n);
}
// chain 0's fMeanEpsilon / fLeapFrogSteps (each chain retunes its own unless they are fixed)
int smcmc_hmc_get_mean_epsilon(smcmc_hmc* h, double* e) {
    if (!h || !e) return SMCMC_ERR_INVALID;
    *e = h->mean_epsilon;
    if (!h->started) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    HMC_TRY(h, hipMemcpy(e, h->d_lane_f64 + (size_t)kHmcLaneMeanEpsilon * h->npad, sizeof(double), hipMemcpyDeviceToHost));
    return SMCMC_OK;
}
int smcmc_hmc_get_leapfrog(smcmc_hmc* h, int* steps) {
    if (!h || !steps) return SMCMC_ERR_INVALID;
    *steps = h->leapfrog;
    if (!h->started) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    int32_t v = 0;
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    HMC_TRY(h, hipMemcpy(&v, h->d_lane_i32 + (size_t)kHmcLaneLeapfrog * h->npad, sizeof(int32_t), hipMemcpyDeviceToHost));
    *steps = v;
    return SMCMC_OK;
}
int smcmc_hmc_set_sync_interval(smcmc_hmc* h, int steps) {
    if (!h || steps < 1) return SMCMC_ERR_INVALID;
    h->sync_every = steps;
    return SMCMC_OK;
}
// Step(save, gradientType) (TSimpleHMC.H:279, 467-532).  0, 1 and 4 are the likelihood's own gradient here (every
// device likelihood has one); 2 the covariant approximation from the pooled running covariance (which is then kept
// whatever the tuning); 3 finite differences of the potential; 5 zero.
int smcmc_hmc_set_gradient_type(smcmc_hmc* h, int type) {
    if (!h || type < 0 || type > 5) return SMCMC_ERR_INVALID;
    if (!h->exact && (type == 2 || type == 3 || type == 5))
        return hfail(h, SMCMC_ERR_UNSUPPORTED, "gradient types 2, 3 and 5 run in reference-order arithmetic only");
    h->gradient_type = type;
    return SMCMC_OK;
}
int smcmc_hmc_get_gradient_type(const smcmc_hmc* h) { return h ? h->gradient_type : -1; }
int smcmc_hmc_set_track_covariance(smcmc_hmc* h, int on) {
    if (!h) return SMCMC_ERR_INVALID;
    h->track_cov = on != 0;
    return SMCMC_OK;
}
int smcmc_hmc_moment_group(const smcmc_hmc* h) { return h ? h->slice_chains : 0; }
int smcmc_hmc_get_tuning(smcmc_hmc* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    const HmcShared& S = *h->shared;
    out[0] = S.curTrace; out[1] = S.orbitLength; out[2] = S.updateCount; out[3] = S.covTrials;
    out[4] = S.averageTrials; out[5] = S.stepsRemaining; out[6] = S.stepsSinceUpdate; out[7] = S.maxScale;
    out[8] = S.minScale; out[9] = S.estTrace;
    return SMCMC_OK;
}
int smcmc_hmc_get_average_point(smcmc_hmc* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    { int pst = hmc_pull(h); if (pst) return pst; }
    std::copy(h->shared->average.begin(), h->shared->average.end(), out);
    return SMCMC_OK;
}
int smcmc_hmc_get_covariance(smcmc_hmc* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    { int pst = hmc_pull(h); if (pst) return pst; }
    std::copy(h->shared->cov.begin(), h->shared->cov.end(), out);
    return SMCMC_OK;
}

int smcmc_hmc_start(smcmc_hmc* h, const double* x0, int broadcast) {
    if (!h || !x0) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    if (h->likelihood == SMCMC_LIKE_QUADFORM) {
        if ((int)h->like_params.size() != D * D)
            return hfail(h, SMCMC_ERR_INVALID, "QUADFORM needs dim*dim likelihood parameters (the Error matrix)");
        h->use_mfma = !h->exact && D <= kMfDimMax;
        h->use_matrix_exact = h->exact && D <= kMfDimMax;
        if (h->use_matrix_exact) {
            // Ex[(tile * D + j) * 16 + 4 rq + r] = Error(16 tile + 4 r + rq, j)
            const int ntiles = (D + 15) / 16;
            std::vector<double> ex(hmc_exact_ex_doubles(D), 0.0);
            for (int it = 0; it < ntiles; ++it)
                for (int j = 0; j < D; ++j)
                    for (int rq = 0; rq < 4; ++rq)
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * it + 4 * r + rq;
                            if (i < D) ex[((size_t)it * D + j) * 16 + 4 * rq + r] = h->like_params[(size_t)i * D + j];
                        }
            HMC_TRY(h, hipMemcpyAsync(h->d_E, ex.data(), ex.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HMC_TRY(h, hipStreamSynchronize(h->stream));
        }
        if (h->use_mfma) {
            // Eop[(tile * nkq + kq) * 64 + lane] = Error(16 tile + (lane & 15), 4 kq + (lane >> 4)): the A operand
            // of every matrix instruction as one contiguous 512-byte read
            const int ntiles = (D + 15) / 16, nkq = (D + 3) / 4, nkqp = panel_mfma_nkq_padded(D);
            std::vector<double> eop(hmc_mfma_eop_doubles(D), 0.0);
            for (int it = 0; it < ntiles; ++it)
                for (int kq = 0; kq < nkq; ++kq)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 16 * it + (l & 15), j = 4 * kq + (l >> 4);
                        if (i < D && j < D) eop[((size_t)it * nkqp + kq) * 64 + l] = h->like_params[(size_t)i * D + j];
                    }
            HMC_TRY(h, hipMemcpyAsync(h->d_E, eop.data(), eop.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HMC_TRY(h, hipStreamSynchronize(h->stream));
        }
        // Eperm[w][j][il] = Error(il*W + w, j): the rows a wavefront owns, contiguous per source column j
        const std::vector<double> perm = hmc_permute(h, h->like_params.data());
        if (h->d_Eperm) {
            HMC_TRY(h, hipMemcpyAsync(h->d_Eperm, perm.data(), perm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HMC_TRY(h, hipStreamSynchronize(h->stream));
        }
        if (!h->use_mfma && !h->use_matrix_exact) {
            HMC_TRY(h, hipMemcpyAsync(h->d_E, perm.data(), perm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HMC_TRY(h, hipStreamSynchronize(h->stream));
        }
    } else {
        h->use_mfma = false;
        h->use_matrix_exact = false;
    }
    if (hmc_no_own_gradient(h->likelihood)) {
        std::vector<double> prm = h->like_params;
        if (h->likelihood == SMCMC_LIKE_ASYM) {
            if (prm.empty()) prm = {-1.0, 100.0};                                  // TAsymLogLikelihood.H:17-18
            if (prm.size() != 2) return hfail(h, SMCMC_ERR_INVALID, "ASYM takes {positiveSlope, negativeSlope}");
        } else if (h->likelihood == SMCMC_LIKE_CONSTRAINED) {
            if ((int)prm.size() != 2 + 2 * D)
                return hfail(h, SMCMC_ERR_INVALID,
                             "CONSTRAINED needs {SummedValues, SummedConstraint, ExpectedValues[dim], PriorConstraints[dim]}");
        } else if (prm.size() > hmc_like_doubles(D)) {
            return hfail(h, SMCMC_ERR_INVALID, "a user likelihood takes at most dim^2 + 2 dim + 8 parameters");
        }
        if (!prm.empty())
            HMC_TRY(h, hipMemcpyAsync(h->d_like, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HMC_TRY(h, hipStreamSynchronize(h->stream));
    } else {
        double b = 100.0;
        if (h->likelihood == SMCMC_LIKE_ROSENBROCK && !h->like_params.empty()) b = h->like_params[0];
        HMC_TRY(h, hipMemcpyAsync(h->d_like, &b, sizeof(double), hipMemcpyHostToDevice, h->stream));
        HMC_TRY(h, hipStreamSynchronize(h->stream));
    }
    std::vector<double> x(NP * D, 0.0);
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < N; ++c) x[(size_t)d * NP + c] = broadcast ? x0[d] : x0[(size_t)d * N + c];
    HMC_TRY(h, hipMemcpyAsync(h->d_q, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_pm, 0, sizeof(double) * NP * D, h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_lane_f64, 0, sizeof(double) * NP * SMCMC_LANE_F64_COUNT_, h->stream));
    HMC_TRY(h, hipMemsetAsync(h->d_lane_i32, 0, sizeof(int32_t) * NP * SMCMC_LANE_I32_COUNT_, h->stream));
    h->step_count = 0;                                   // :211
    h->mean_epsilon = 0.05;                              // :229
    HmcParams p = hmc_params(h, 0, 1);
    hipError_t e = hmc_dispatch(h, p);
    if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("hmc start launch: ") + hipGetErrorString(e));
    // fCurrentAcceptance = fTargetAcceptance = 0.65 (:234-235)
    std::vector<double> acc(NP, 0.0);
    for (int c = 0; c < N; ++c) acc[c] = 0.65;
    HMC_TRY(h, hipMemcpyAsync(h->d_lane_f64 + (size_t)SMCMC_LANE_ACCEPTANCE * NP, acc.data(), NP * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    // every chain's own fMeanEpsilon = 0.05 (:229), fLeapFrogSteps as the constructor / SetLeapFrog left it, fReversalLen = 0
    int st = hmc_fill_lane<double>(h, h->d_lane_f64 + (size_t)kHmcLaneMeanEpsilon * NP, h->mean_epsilon);
    if (st) return st;
    st = hmc_fill_lane<int32_t>(h, h->d_lane_i32 + (size_t)kHmcLaneLeapfrog * NP, (int32_t)h->leapfrog);
    if (st) return st;
    // the running covariance starts from chain 0's point (:236-266)
    std::vector<double> p0(D);
    for (int d = 0; d < D; ++d) p0[d] = x[(size_t)d * NP];
    h->shared->start(p0.data());
    h->shared_on_device = false;   // pushed again at the next pooled update
    h->host_stale = false;
    h->cov_dirty = true;
    h->steps_in_window = 0;
    if (h->d_gacc) HMC_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * hmc_gacc_doubles(h), h->stream));
    h->started = true;
    return SMCMC_OK;
}

int smcmc_hmc_step(smcmc_hmc* h, int nsteps) {
    if (!h) return SMCMC_ERR_INVALID;
    if (!h->started) return hfail(h, SMCMC_ERR_INVALID, "Must initialize starting point");   // :280-284
    if (nsteps <= 0) return SMCMC_OK;
    HMC_ON_DEVICE(h);
    if (hmc_no_own_gradient(h->likelihood) && !hmc_generic_gradient(h))
        return hfail(h, SMCMC_ERR_RUNTIME, "the likelihood has no gradient (TSimpleHMC.H:85-89: its functor returns false): "
                                           "choose gradient type 2 (covariant), 3 (finite differences) or 5 (none)");
    if (hmc_generic_gradient(h)) {
        if (!h->exact) return hfail(h, SMCMC_ERR_UNSUPPORTED, "gradient types 2, 3 and 5 run in reference-order arithmetic only");
        int gst = hmc_generic_buffers(h);
        if (gst) return gst;
    }
    if (!hmc_tracking(h)) {
        // fixed step length and leapfrog count: the chains share nothing, one launch runs all the steps
        HmcParams p = hmc_params(h, nsteps, 0);
        hipError_t e = hmc_dispatch(h, p);
        if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("hmc step launch: ") + hipGetErrorString(e));
        h->step_count += (uint32_t)nsteps;
        return SMCMC_OK;
    }
    int st = hmc_tracking_buffers(h);
    if (st) return st;
    for (int s = 0; s < nsteps; ++s) {
        if (h->gradient_type == 2 && h->cov_dirty) {
            st = hmc_generic_buffers(h);
            if (st) return st;
        }
        HmcParams p = hmc_params(h, 1, 0);
        p.adaptive = 1;
        hipError_t e = hmc_dispatch(h, p);
        if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("hmc step launch: ") + hipGetErrorString(e));
        h->step_count += 1u;
        // UpdateCovariance (:338): the point each chain stood on, if its proposal's potential was finite (:336)
        {
            smcmc::FoldRingParams fp;
            std::memset(&fp, 0, sizeof(fp));
            fp.src[0] = h->d_qprev; fp.nsrc = 1; fp.c0 = h->d_zero;
            fp.nchains = h->nchains; fp.npad = h->npad; fp.D = h->dim; fp.slice_chains = h->slice_chains;
            fp.gacc = h->d_gacc; fp.mask = h->d_lane_i32 + (size_t)kHmcLaneContributes * h->npad;
            e = smcmc::launch_fold_ring(h->fold, fp, h->stream);
        }
        if (e != hipSuccess) return hfail(h, SMCMC_ERR_HIP, std::string("fold launch: ") + hipGetErrorString(e));
        if (++h->steps_in_window >= h->sync_every) {
            st = hmc_sync(h);
            if (st) return st;
        }
    }
    return SMCMC_OK;
}

// the pooled update now, whatever the interval (a partial window at the end of a run)
int smcmc_hmc_sync(smcmc_hmc* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    if (!h->d_gacc) return SMCMC_OK;
    return hmc_sync(h);
}

// The same in pieces, for an ensemble sharded over engines / ranks: reduce, export, (sum over ranks), import, apply.
int smcmc_hmc_moments_size(const smcmc_hmc* h) { return h ? (int)(((size_t)h->dim + 1) * ((size_t)h->dim + 2) / 2) : 0; }

int smcmc_hmc_reduce_moments(smcmc_hmc* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    int st = hmc_tracking_buffers(h);
    if (st) return st;
    return hmc_reduce(h);
}

int smcmc_hmc_export_moments(smcmc_hmc* h, double* dst_device) {
    if (!h || !dst_device || !h->d_moments) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipMemcpyAsync(dst_device, h->d_moments, sizeof(double) * (size_t)smcmc_hmc_moments_size(h), hipMemcpyDeviceToDevice,
                              h->stream));
    return SMCMC_OK;
}

int smcmc_hmc_import_moments(smcmc_hmc* h, const double* src_device) {
    if (!h || !src_device || !h->d_moments) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipMemcpyAsync(h->d_moments, src_device, sizeof(double) * (size_t)smcmc_hmc_moments_size(h), hipMemcpyDeviceToDevice,
                              h->stream));
    return SMCMC_OK;
}

int smcmc_hmc_apply_moments(smcmc_hmc* h) {
    if (!h || !h->started || !h->d_moments) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    return hmc_apply(h);
}

int smcmc_hmc_read_state(smcmc_hmc* h, double* q, double* momentum, double* logl) {
    if (!h) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    if (q) HMC_TRY(h, hipMemcpy2D(q, (size_t)N * sizeof(double), h->d_q, NP * sizeof(double), (size_t)N * sizeof(double),
                                  (size_t)D, hipMemcpyDeviceToHost));
    if (momentum) HMC_TRY(h, hipMemcpy2D(momentum, (size_t)N * sizeof(double), h->d_pm, NP * sizeof(double),
                                         (size_t)N * sizeof(double), (size_t)D, hipMemcpyDeviceToHost));
    if (logl) HMC_TRY(h, hipMemcpy(logl, h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, (size_t)N * sizeof(double),
                                   hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_hmc_copy_positions(smcmc_hmc* h, double* dst_device) {
    if (!h || !dst_device || !h->started) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipMemcpyAsync(dst_device, h->d_q, sizeof(double) * (size_t)h->dim * h->npad, hipMemcpyDeviceToDevice, h->stream));
    return SMCMC_OK;
}

int smcmc_hmc_nchains_padded(const smcmc_hmc* h) { return h ? h->npad : 0; }

int smcmc_hmc_read_lane_f64(smcmc_hmc* h, int field, double* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_F64_COUNT_) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    HMC_TRY(h, hipMemcpy(out, h->d_lane_f64 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(double),
                         hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_hmc_read_lane_i32(smcmc_hmc* h, int field, int32_t* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_I32_COUNT_) return SMCMC_ERR_INVALID;
    HMC_ON_DEVICE(h);
    HMC_TRY(h, hipStreamSynchronize(h->stream));
    HMC_TRY(h, hipMemcpy(out, h->d_lane_i32 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(int32_t),
                         hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

}  // extern "C"
