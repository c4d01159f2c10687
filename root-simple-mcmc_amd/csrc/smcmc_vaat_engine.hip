// smcmc_vaat_engine.hip -- host side of the smcmc_vaat_* entry points (include/smcmc.h): N independent
// sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> chains (TProposeVAATStep.H:22-307) on one device.  The chains share
// only settings (proposal types, acceptance window, rigidity); all adaptive state is per chain and per dimension and
// lives on the device (smcmc_vaat_kernel.hip.h).  No CPU fallback: without a device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smcmc.h"
#include "smcmc_vaat_kernel.hip.h"

using namespace smcmc;

struct smcmc_vaat {
    int dim = 0, nchains = 0, npad = 0, likelihood = 0, device = 0;
    int dp = 0;                    // register-array size of the kernel family (dim <= 63), else dim
    bool large = false;            // dim > 63: vaat_large_kernel
    uint64_t seed = 0;
    uint32_t chain_offset = 0, total_steps = 0;
    bool started = false, initialized = false;   // fStateInitialized (:199-200)
    bool exact = true;
    hipStream_t stream = nullptr;
    std::vector<double> like_params;
    std::vector<int32_t> ptype;
    std::vector<double> param1, param2;
    int acc_window = -1;           // fAcceptanceWindow (:26); InitializeState makes it 100 (:211)
    double rigidity = 2.0;         // fAcceptanceRigidity (:27)
    double target = 0.44;          // fTargetAcceptance (:30)
    int step_rms_window = 0;
    int queue_len = 0;             // entries left in fNextIndex (the same for every chain)
    double *d_x = nullptr, *d_like = nullptr, *d_lane_f64 = nullptr, *d_sigma = nullptr, *d_acceptance = nullptr;
    double *d_param1 = nullptr, *d_param2 = nullptr;
    int32_t *d_lane_i32 = nullptr, *d_acc_trials = nullptr, *d_ptype = nullptr;
    uint16_t* d_queue = nullptr;
    std::string error;
};

namespace {

int vfail(smcmc_vaat* h, int status, const std::string& msg) {
    if (h) h->error = msg;
    return status;
}

#define VAAT_TRY(h, expr)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return vfail((h), SMCMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct VaatDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit VaatDeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~VaatDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    VaatDeviceGuard(const VaatDeviceGuard&) = delete;
    VaatDeviceGuard& operator=(const VaatDeviceGuard&) = delete;
};
#define VAAT_ON_DEVICE(h) VaatDeviceGuard device_guard_((h)->device)

#define SMCMC_DP_ENTRY(n) n,
constexpr int kVaatDPList[] = {SMCMC_FOR_EACH_DP(SMCMC_DP_ENTRY)};
#undef SMCMC_DP_ENTRY

bool vaat_stress(int like) { return like == SMCMC_LIKE_ASYM || like == SMCMC_LIKE_HORRIFIC || like == SMCMC_LIKE_CONSTRAINED; }

int vaat_pick_dp(int dim, int like) {
    for (int dp : kVaatDPList)
        if (dim <= dp && (!vaat_stress(like) || dp == 31 || dp == 63)) return dp;
    return -1;
}

hipError_t vaat_dispatch(smcmc_vaat* h, const VaatParams& p) {
#ifdef SMCMC_USER_LIKELIHOOD
    if (h->large && h->likelihood == SMCMC_LIKE_USER) return launch_vaat_large_user(p, h->exact, h->stream);
#endif
    if (h->large) return launch_vaat_large(p, h->likelihood, h->exact, h->stream);
    switch (h->dp) {
#define SMCMC_DP_CASE(n) case n: return launch_vaat<n>(p, h->likelihood, h->exact, h->stream);
        SMCMC_FOR_EACH_DP(SMCMC_DP_CASE)
#undef SMCMC_DP_CASE
        default: return hipErrorInvalidValue;
    }
}

VaatParams vaat_params(smcmc_vaat* h, int nsteps) {
    VaatParams p;
    std::memset(&p, 0, sizeof(p));
    p.nchains = h->nchains; p.npad = h->npad; p.dim = h->dim; p.nsteps = nsteps;
    p.queue_len = h->queue_len;
    p.step0 = h->total_steps; p.chain_offset = h->chain_offset; p.seed = h->seed;
    p.like = h->d_like; p.ptype = h->d_ptype; p.param1 = h->d_param1; p.param2 = h->d_param2;
    p.acc_window = h->acc_window; p.rigidity = h->rigidity; p.target = h->target;
    p.step_rms_window = h->step_rms_window;
    p.x = h->d_x; p.lane_f64 = h->d_lane_f64; p.lane_i32 = h->d_lane_i32;
    p.sigma = h->d_sigma; p.acceptance = h->d_acceptance; p.acc_trials = h->d_acc_trials; p.queue = h->d_queue;
    p.save_stride = 1;
    return p;
}

int vaat_upload_settings(smcmc_vaat* h) {
    const size_t D = (size_t)h->dim;
    VAAT_TRY(h, hipMemcpyAsync(h->d_ptype, h->ptype.data(), D * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    VAAT_TRY(h, hipMemcpyAsync(h->d_param1, h->param1.data(), D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    VAAT_TRY(h, hipMemcpyAsync(h->d_param2, h->param2.data(), D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int vaat_upload_like(smcmc_vaat* h) {
    const int D = h->dim, DP = h->dp;
    std::vector<double> prm;
    switch (h->likelihood) {
        case SMCMC_LIKE_QUADFORM: {
            if ((int)h->like_params.size() != D * D)
                return vfail(h, SMCMC_ERR_INVALID, "QUADFORM needs dim*dim likelihood parameters (the Error matrix)");
            // the kernels walk Error(j,i) with j innermost (TDummyLogLikelihood.H:24-28): the transpose, zero padded
            // to the register-array size for dim <= 63, plain [dim][dim] above
            const int P = h->large ? D : DP;
            prm.assign((size_t)P * P, 0.0);
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) prm[(size_t)i * P + j] = h->like_params[(size_t)j * D + i];
            break;
        }
        case SMCMC_LIKE_ROSENBROCK:
            prm = {h->like_params.empty() ? 100.0 : h->like_params[0]};   // ROSEN_B, THardLogLikelihood.H:53
            break;
        case SMCMC_LIKE_ASYM:
            prm = {-1.0, 100.0};                                           // TAsymLogLikelihood.H:17-18
            if (h->like_params.size() == 2) prm = h->like_params;
            else if (!h->like_params.empty()) return vfail(h, SMCMC_ERR_INVALID, "ASYM takes {positiveSlope, negativeSlope}");
            break;
        case SMCMC_LIKE_CONSTRAINED:
            if ((int)h->like_params.size() != 2 + 2 * D)
                return vfail(h, SMCMC_ERR_INVALID,
                             "CONSTRAINED needs {SummedValues, SummedConstraint, ExpectedValues[dim], PriorConstraints[dim]}");
            prm = h->like_params;
            break;
        case SMCMC_LIKE_USER:
            if (h->like_params.size() > std::max((size_t)DP * DP, (size_t)(2 + 2 * D)))
                return vfail(h, SMCMC_ERR_INVALID, "a user likelihood takes at most max(dim_padded^2, 2 + 2 dim) parameters");
            prm = h->like_params;
            if (prm.empty()) prm = {0.0};
            break;
        default: prm = {0.0}; break;
    }
    VAAT_TRY(h, hipMemcpyAsync(h->d_like, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

template <typename T>
__global__ void vaat_fill_kernel(T* dst, size_t n, T v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

template <typename T>
int vaat_fill(smcmc_vaat* h, T* dst, size_t n, T v) {
    const int threads = 256;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(vaat_fill_kernel<T>), dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                       h->stream, dst, n, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vfail(h, SMCMC_ERR_HIP, std::string("fill launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

}  // namespace

extern "C" {

int smcmc_vaat_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset, int device,
                      smcmc_vaat** out) {
    if (!out) return SMCMC_ERR_INVALID;
    *out = nullptr;
    if (dim < 1 || nchains < 1) return SMCMC_ERR_INVALID;
    if (likelihood < SMCMC_LIKE_ISO_GAUSS || likelihood > SMCMC_LIKE_CONSTRAINED) return SMCMC_ERR_INVALID;
#ifndef SMCMC_USER_LIKELIHOOD
    if (likelihood == SMCMC_LIKE_USER) return SMCMC_ERR_UNSUPPORTED;   // this build carries no user likelihood
#endif
    if (likelihood == SMCMC_LIKE_ROSENBROCK && dim < 2) return SMCMC_ERR_INVALID;
    if (dim > smcmc_max_dim()) return SMCMC_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return SMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    int dp = vaat_pick_dp(dim, likelihood);
    const bool large = dp < 0;
#ifndef SMCMC_USER_LIKELIHOOD_ANY_DIM
    if (large && likelihood == SMCMC_LIKE_USER) return SMCMC_ERR_UNSUPPORTED;   // the header serves dim <= 63 only
#endif
    if (large) dp = dim;
    smcmc_vaat* h = new (std::nothrow) smcmc_vaat();
    if (!h) return SMCMC_ERR_RUNTIME;
    h->dim = dim; h->nchains = nchains; h->likelihood = likelihood; h->seed = seed; h->chain_offset = chain_offset;
    h->device = device; h->dp = dp; h->large = large;
    h->npad = (nchains + kWave - 1) / kWave * kWave;
    h->ptype.assign(dim, 0); h->param1.assign(dim, 0.0); h->param2.assign(dim, 0.0);   // SetDim :93-98
    *out = h;
    VAAT_ON_DEVICE(h);
    const size_t NP = (size_t)h->npad, D = (size_t)dim;
    const size_t like_doubles = std::max((size_t)dp * dp, (size_t)(2 + 2 * D));
    VAAT_TRY(h, hipMalloc(&h->d_x, sizeof(double) * NP * dp));
    VAAT_TRY(h, hipMalloc(&h->d_like, sizeof(double) * like_doubles));
    VAAT_TRY(h, hipMalloc(&h->d_lane_f64, sizeof(double) * NP * SMCMC_LANE_F64_COUNT_));
    VAAT_TRY(h, hipMalloc(&h->d_lane_i32, sizeof(int32_t) * NP * SMCMC_LANE_I32_COUNT_));
    VAAT_TRY(h, hipMalloc(&h->d_sigma, sizeof(double) * NP * D));
    VAAT_TRY(h, hipMalloc(&h->d_acceptance, sizeof(double) * NP * D));
    VAAT_TRY(h, hipMalloc(&h->d_acc_trials, sizeof(int32_t) * NP * D));
    VAAT_TRY(h, hipMalloc(&h->d_queue, sizeof(uint16_t) * NP * D));
    VAAT_TRY(h, hipMalloc(&h->d_ptype, sizeof(int32_t) * D));
    VAAT_TRY(h, hipMalloc(&h->d_param1, sizeof(double) * D));
    VAAT_TRY(h, hipMalloc(&h->d_param2, sizeof(double) * D));
    VAAT_TRY(h, hipMemset(h->d_x, 0, sizeof(double) * NP * dp));
    VAAT_TRY(h, hipMemset(h->d_like, 0, sizeof(double) * like_doubles));
    VAAT_TRY(h, hipMemset(h->d_lane_f64, 0, sizeof(double) * NP * SMCMC_LANE_F64_COUNT_));
    VAAT_TRY(h, hipMemset(h->d_lane_i32, 0, sizeof(int32_t) * NP * SMCMC_LANE_I32_COUNT_));
    VAAT_TRY(h, hipMemset(h->d_acceptance, 0, sizeof(double) * NP * D));
    VAAT_TRY(h, hipMemset(h->d_acc_trials, 0, sizeof(int32_t) * NP * D));
    VAAT_TRY(h, hipMemset(h->d_queue, 0, sizeof(uint16_t) * NP * D));
    int st = vaat_fill<double>(h, h->d_sigma, NP * D, 2.34);                                  // SetDim :98
    if (st) return st;
    st = vaat_fill<int32_t>(h, h->d_lane_i32 + (size_t)kVaatLaneLastIndex * NP, NP, -1);     // fLastIndex(-1) :27
    if (st) return st;
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int smcmc_vaat_destroy(smcmc_vaat* h) {
    if (!h) return SMCMC_OK;
    VAAT_ON_DEVICE(h);
    if (h->d_x) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_x); (void)hipFree(h->d_like); (void)hipFree(h->d_lane_f64); (void)hipFree(h->d_lane_i32);
    (void)hipFree(h->d_sigma); (void)hipFree(h->d_acceptance); (void)hipFree(h->d_acc_trials); (void)hipFree(h->d_queue);
    (void)hipFree(h->d_ptype); (void)hipFree(h->d_param1); (void)hipFree(h->d_param2);
    delete h;
    return SMCMC_OK;
}

const char* smcmc_vaat_last_error(const smcmc_vaat* h) { return h ? h->error.c_str() : "null engine"; }

int smcmc_vaat_set_stream(smcmc_vaat* h, void* hip_stream) {
    if (!h) return SMCMC_ERR_INVALID;
    h->stream = (hipStream_t)hip_stream;
    return SMCMC_OK;
}

int smcmc_vaat_set_likelihood_params(smcmc_vaat* h, const double* params, int count) {
    if (!h || count < 0 || (count > 0 && !params)) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    h->like_params.assign(params, params + count);
    return h->started ? vaat_upload_like(h) : SMCMC_OK;
}

int smcmc_vaat_set_exact_arithmetic(smcmc_vaat* h, int exact) {
    if (!h) return SMCMC_ERR_INVALID;
    if (h->started) return vfail(h, SMCMC_ERR_LOGIC, "choose the arithmetic before Start");
    h->exact = exact != 0;
    return SMCMC_OK;
}

int smcmc_vaat_set_uniform(smcmc_vaat* h, int dim, double minimum, double maximum) {   // :101-117
    if (!h) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    if (dim < 0 || dim >= h->dim) return vfail(h, SMCMC_ERR_INVALID, "Dimension is out of range.");   // an error message and no change in the reference
    h->ptype[dim] = 1; h->param1[dim] = minimum; h->param2[dim] = maximum;
    return h->started ? vaat_upload_settings(h) : SMCMC_OK;
}

int smcmc_vaat_set_gaussian(smcmc_vaat* h, int dim, double sigma) {                    // :123-133
    if (!h) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    if (dim < 0 || dim >= h->dim) return vfail(h, SMCMC_ERR_INVALID, "Dimension is out of range.");
    h->ptype[dim] = 0; h->param1[dim] = sigma;
    return h->started ? vaat_upload_settings(h) : SMCMC_OK;
}

int smcmc_vaat_set_acceptance_window(smcmc_vaat* h, double a) {                       // :137 (int member :277)
    if (!h) return SMCMC_ERR_INVALID;
    h->acc_window = (int)a;
    return SMCMC_OK;
}
int smcmc_vaat_get_acceptance_window(const smcmc_vaat* h, double* a) {
    if (!h || !a) return SMCMC_ERR_INVALID;
    *a = h->acc_window;
    return SMCMC_OK;
}
int smcmc_vaat_set_acceptance_rigidity(smcmc_vaat* h, double r) {                     // :148
    if (!h) return SMCMC_ERR_INVALID;
    h->rigidity = r;
    return SMCMC_OK;
}
int smcmc_vaat_get_acceptance_rigidity(const smcmc_vaat* h, double* r) {
    if (!h || !r) return SMCMC_ERR_INVALID;
    *r = h->rigidity;
    return SMCMC_OK;
}
int smcmc_vaat_set_step_rms_window(smcmc_vaat* h, int window) {                       // TSimpleMCMC.H:221
    if (!h) return SMCMC_ERR_INVALID;
    h->step_rms_window = window;
    return SMCMC_OK;
}

int smcmc_vaat_start(smcmc_vaat* h, const double* x0, int broadcast) {                // TSimpleMCMC.H:246-276
    if (!h || !x0) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    h->started = false;
    int st = vaat_upload_like(h);
    if (st) return st;
    st = vaat_upload_settings(h);
    if (st) return st;
    std::vector<double> x(NP * h->dp, 0.0);
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < N; ++c) x[(size_t)d * NP + c] = broadcast ? x0[d] : x0[(size_t)d * N + c];
    // padded chains start where chain 0 does, so that no lane of the last wavefront works on garbage
    for (int d = 0; d < D; ++d)
        for (size_t c = (size_t)N; c < NP; ++c) x[(size_t)d * NP + c] = x[(size_t)d * NP];
    VAAT_TRY(h, hipMemcpyAsync(h->d_x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // A second Start() (TSimpleMCMC.H:246-276) rewrites the point and the two likelihood values only: InitializeState
    // returns at once (TProposeVAATStep.H:197), so fLastValue, fStepRMS and every per-dimension width stay as they are.
    if (!h->initialized)
        VAAT_TRY(h, hipMemsetAsync(h->d_lane_f64, 0, sizeof(double) * NP * SMCMC_LANE_F64_COUNT_, h->stream));
    // Start's likelihood call (TSimpleMCMC.H:258) and InitializeState's fLastValue (:207)
    VaatParams p = vaat_params(h, 0);
    p.init_only = 1;
    p.restart = h->initialized ? 1 : 0;
    hipError_t e = vaat_dispatch(h, p);
    if (e != hipSuccess) return vfail(h, SMCMC_ERR_HIP, std::string("start kernel launch: ") + hipGetErrorString(e));
    std::vector<double> logl(NP);
    VAAT_TRY(h, hipMemcpyAsync(logl.data(), h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, NP * sizeof(double),
                               hipMemcpyDeviceToHost, h->stream));
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    for (int c = 0; c < N; ++c)
        if (!std::isfinite(logl[c]) || logl[c] < -0.999999E+10)                       // :265-268
            return vfail(h, SMCMC_ERR_BAD_START, "Invalid starting point");
    if (!h->initialized) {                                                            // InitializeState :198-214
        h->initialized = true;
        h->acc_window = 100;                                                          // :211
    }
    h->started = true;
    return SMCMC_OK;
}

int smcmc_vaat_update_proposal(smcmc_vaat* h) {                                       // UpdateProposal :177-195
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    if (h->queue_len != 0) return SMCMC_OK;                                           // :178
    VAAT_ON_DEVICE(h);
    VaatParams p = vaat_params(h, 0);
    p.shuffle_only = 1;
    hipError_t e = vaat_dispatch(h, p);
    if (e != hipSuccess) return vfail(h, SMCMC_ERR_HIP, std::string("shuffle launch: ") + hipGetErrorString(e));
    h->queue_len = h->dim;
    return SMCMC_OK;
}

int smcmc_vaat_step_save(smcmc_vaat* h, int nsteps, int stride, double* save_x_device, double* save_logl_device) {
    if (!h) return SMCMC_ERR_INVALID;
    if (!h->started) return vfail(h, SMCMC_ERR_INVALID, "Must initialize starting point");   // TSimpleMCMC.H:371-374
    if (nsteps <= 0) return SMCMC_OK;
    if (stride < 1) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    VaatParams p = vaat_params(h, nsteps);
    p.save_x = save_x_device; p.save_logl = save_logl_device; p.save_stride = stride;
    hipError_t e = vaat_dispatch(h, p);
    if (e != hipSuccess) return vfail(h, SMCMC_ERR_HIP, std::string("step launch: ") + hipGetErrorString(e));
    h->total_steps += (uint32_t)nsteps;
    // every step pops one index; an empty queue is refilled (dim entries) before the pop
    const int D = h->dim;
    h->queue_len = ((h->queue_len - nsteps) % D + D) % D;   // a full queue (D, after UpdateProposal) emptied by k D steps is 0
    return SMCMC_OK;
}

int smcmc_vaat_step(smcmc_vaat* h, int nsteps) { return smcmc_vaat_step_save(h, nsteps, 1, nullptr, nullptr); }

int smcmc_vaat_total_steps(const smcmc_vaat* h) { return h ? (int)h->total_steps : -1; }
int smcmc_vaat_queue_length(const smcmc_vaat* h) { return h ? h->queue_len : -1; }
int smcmc_vaat_nchains_padded(const smcmc_vaat* h) { return h ? h->npad : 0; }

int smcmc_vaat_read_state(smcmc_vaat* h, double* x, double* logl) {
    if (!h) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    const size_t NP = (size_t)h->npad, N = (size_t)h->nchains;
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    if (x) VAAT_TRY(h, hipMemcpy2D(x, N * sizeof(double), h->d_x, NP * sizeof(double), N * sizeof(double), (size_t)h->dim,
                                   hipMemcpyDeviceToHost));
    if (logl) VAAT_TRY(h, hipMemcpy(logl, h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, N * sizeof(double), hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_vaat_read_lane_f64(smcmc_vaat* h, int field, double* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_F64_COUNT_) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    VAAT_TRY(h, hipMemcpy(out, h->d_lane_f64 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(double), hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_vaat_read_lane_i32(smcmc_vaat* h, int field, int32_t* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_I32_COUNT_) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    VAAT_TRY(h, hipMemcpy(out, h->d_lane_i32 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

// per-dimension state as [dim][nchains]: SMCMC_VAAT_DIM_SIGMA, SMCMC_VAAT_DIM_ACCEPTANCE (f64)
int smcmc_vaat_read_dim_f64(smcmc_vaat* h, int field, double* out) {
    if (!h || !out || (field != SMCMC_VAAT_DIM_SIGMA && field != SMCMC_VAAT_DIM_ACCEPTANCE)) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    const size_t NP = (size_t)h->npad, N = (size_t)h->nchains;
    const double* src = field == SMCMC_VAAT_DIM_SIGMA ? h->d_sigma : h->d_acceptance;
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    VAAT_TRY(h, hipMemcpy2D(out, N * sizeof(double), src, NP * sizeof(double), N * sizeof(double), (size_t)h->dim, hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

// SMCMC_VAAT_DIM_ACCEPTANCE_TRIALS, SMCMC_VAAT_DIM_QUEUE (i32; queue slots >= smcmc_vaat_queue_length are stale)
int smcmc_vaat_read_dim_i32(smcmc_vaat* h, int field, int32_t* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    VAAT_ON_DEVICE(h);
    const size_t NP = (size_t)h->npad, N = (size_t)h->nchains, D = (size_t)h->dim;
    VAAT_TRY(h, hipStreamSynchronize(h->stream));
    if (field == SMCMC_VAAT_DIM_ACCEPTANCE_TRIALS) {
        VAAT_TRY(h, hipMemcpy2D(out, N * sizeof(int32_t), h->d_acc_trials, NP * sizeof(int32_t), N * sizeof(int32_t), D, hipMemcpyDeviceToHost));
        return SMCMC_OK;
    }
    if (field != SMCMC_VAAT_DIM_QUEUE) return SMCMC_ERR_INVALID;
    std::vector<uint16_t> q(NP * D);
    VAAT_TRY(h, hipMemcpy(q.data(), h->d_queue, q.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (size_t d = 0; d < D; ++d)
        for (size_t c = 0; c < N; ++c) out[d * N + c] = q[d * NP + c];
    return SMCMC_OK;
}

int smcmc_vaat_state_device_ptr(smcmc_vaat* h, double** x, double** logl) {
    if (!h) return SMCMC_ERR_INVALID;
    if (x) *x = h->d_x;
    if (logl) *logl = h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * h->npad;
    return SMCMC_OK;
}

}  // extern "C"
