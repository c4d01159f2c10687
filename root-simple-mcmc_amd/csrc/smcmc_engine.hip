// smcmc_engine.hip -- host engine behind the C ABI of include/smcmc.h.
//
// Owns the device buffers ([dim][chain] state, per-chain scalar columns, the
// per-wavefront moment tiles), the shared proposal (smcmc_proposal.hpp) and the
// launch logic.  There is no CPU execution path: every entry point that needs
// the device fails with SMCMC_ERR_NO_DEVICE / SMCMC_ERR_HIP when it is missing.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the functions are looked up at run time (rccl_api)

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smcmc.h"
#include "smcmc_kernels.hip.h"
#include "smcmc_panel_kernel.hip.h"
#include "smcmc_panel_mfma_kernel.hip.h"
#include "smcmc_pooled_update.hip.h"
#include "smcmc_fold_ring.hip.h"
#include "smcmc_perchain_kernel.hip.h"
#include "smcmc_perchain_wave.hip.h"
#include "smcmc_proposal.hpp"

using namespace smcmc;

namespace {

// register-resident kernel families, smallest first (SMCMC_FOR_EACH_DP)
#define SMCMC_DP_ENTRY(n) n,
constexpr int kDPList[] = {SMCMC_FOR_EACH_DP(SMCMC_DP_ENTRY)};
#undef SMCMC_DP_ENTRY
constexpr int kNumDP = sizeof(kDPList) / sizeof(kDPList[0]);

bool stress_likelihood(int like) {
    return like == SMCMC_LIKE_ASYM || like == SMCMC_LIKE_HORRIFIC || like == SMCMC_LIKE_CONSTRAINED;
}

int pick_dp(int dim, int like) {
    // the stress likelihoods are instantiated for the 31- and 63-wide families only (launch_step)
    for (int i = 0; i < kNumDP; ++i)
        if (dim <= kDPList[i] && (!stress_likelihood(like) || kDPList[i] == 31 || kDPList[i] == 63)) return kDPList[i];
    return -1;
}

hipError_t dispatch_step(int dp, const StepParams& p, int like, bool exact, bool fullu, bool moments,
                         hipStream_t s) {
    // the proposed point is stored by the SPECIAL instantiation in the reference order, by every kernel in the fused order
    const bool special = p.scan_dim >= 0 || p.uniform_mask != 0 || (p.proposed != nullptr && exact);
    switch (dp) {
#define SMCMC_DP_CASE(n) case n: return launch_step<n>(p, like, exact, fullu, moments, special, s);
        SMCMC_FOR_EACH_DP(SMCMC_DP_CASE)
#undef SMCMC_DP_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t dispatch_reduce(int dp, double* gacc, int ngroups, int D, double* chunks, double* moments,
                           hipStream_t s) {
    switch (dp) {
#define SMCMC_DP_CASE(n) case n: return launch_reduce<n>(gacc, ngroups, D, chunks, moments, s);
        SMCMC_FOR_EACH_DP(SMCMC_DP_CASE)
#undef SMCMC_DP_CASE
        default: return hipErrorInvalidValue;
    }
}

int tiles_for(int dp) {
    const int t = (dp + 1 + 15) / 16;
    return t * (t + 1) / 2;
}

}  // namespace

struct smcmc_engine {
    int dim = 0, nchains = 0, npad = 0, ngroups = 0, dp = 0, nt = 0;
    int moment_stride = 1;   // large-dimension path: fold the current point every moment_stride-th step
    int slice_chains = 0;    // large-dimension path: chains per moment group
    int fold_nslices = 0;    // ... and the number of groups (fold_slices(dim))
    int panel_w = 0;   // 0: register-resident kernels (dim <= 63); 4 / 8: wavefronts per chain group of the panel kernel
    int likelihood = 0, mode = SMCMC_MODE_POOLED, device = 0;
    bool exact = true, started = false;
    uint64_t seed = 0;
    uint32_t chain_offset = 0;
    uint32_t total_steps = 0;
    int step_rms_window = 1000;
    hipStream_t stream = nullptr;
    SharedProposal* prop = nullptr;
    std::vector<double> like_params;
    bool has_forced = false;
    // device
    double* d_x = nullptr;
    double* d_lane_f64 = nullptr;
    int32_t* d_lane_i32 = nullptr;
    double* d_U = nullptr;
    double* d_Uop = nullptr;       // large dimensions, fused order: U^T in matrix-operand order (smcmc_panel_mfma_kernel.hip.h)
    double* d_like = nullptr;
    // QUADFORM with a sparse Error matrix: the non-zero entries of Error^T, row by row (quadform_csr); nullptr = dense only
    int32_t* d_like_rowptr = nullptr;
    int32_t* d_like_cols = nullptr;
    int32_t* d_like_rows = nullptr;
    double* d_like_vals = nullptr;
    int like_nnz = 0;
    bool dense_quadform = false;   // SMCMC_P_DENSE_QUADFORM
    double* d_c0 = nullptr;
    double* d_gacc = nullptr;
    double* d_moments = nullptr;
    double* d_chunks = nullptr;
    double* h_moments = nullptr;   // pinned host copy of the packed moments (read back at every sync)
    double* d_forced = nullptr;
    double* d_scratch = nullptr;   // dim > 63, quadratic form in reference order: the proposal image its serial sum reads
    // dim > 63, pooled covariance fed every step: the accepted point after each step of a launch, for the folds that follow it
    double* d_ring = nullptr;      // [ring_steps][dim][npad]
    double* d_ring_logl = nullptr; // [ring_steps][npad] (the kernels save both)
    int ring_steps = -1;           // -1: not decided yet; 0: no ring (the state is too large), else steps per launch
    smcmc::FoldRing fold;          // dim > 63: the fold kernel's plan for this ensemble (smcmc_fold_ring.hip.h)
    ncclComm_t comm = nullptr;     // smcmc_comm_init
    int comm_ranks = 0;
    double* d_proposed = nullptr;  // [dp][npad], allocated by SMCMC_P_KEEP_PROPOSED
    bool keep_proposed = false;
    double* d_uniform = nullptr;   // [2][dp] bounds of the uniform dimensions
    int scan_dim = -1;             // fScanDimension
    // the pooled update on the device (smcmc_pooled_update.hip.h): the device copy of the shared proposal's numbers
    bool device_update = true;     // SMCMC_P_DEVICE_UPDATE
    bool overlap_update = false;   // SMCMC_P_OVERLAP_UPDATE
    double *d_centre = nullptr, *d_cov = nullptr, *d_decomp = nullptr, *d_scal = nullptr;
    double* h_scal = nullptr;      // pinned: the scalars (status word included) of the latest device update
    double* h_scal_dev = nullptr;  // ... as the device addresses it (hipHostGetDevicePointer), null if it cannot
    hipEvent_t status_event = nullptr;
    bool update_prepared = false;  // pooled_update_prepare() has run on this engine's device
    bool status_pending = false;   // a device update whose status the host has not looked at yet
    bool host_stale = false;       // the device holds newer centre / covariance / decomposition / trials than *prop
    bool device_stale = true;      // *prop was changed on the host since the device copy was written
    struct { int updateCount, nextUpdate, lastPath; double acceptanceTrials; bool decompFull; } before_update{};
    // SMCMC_MODE_PER_CHAIN (smcmc_perchain_kernel.hip.h): every chain's own adaptive state, [k][chain] columns
    double *d_pc_cov = nullptr, *d_pc_ut = nullptr, *d_pc_centre = nullptr, *d_pc_last = nullptr;
    double* d_pc_tmpl = nullptr;   // what the host hands to every chain at Start / Restore / ResetProposal: cov packed, then ut
    int* d_pc_flag = nullptr;      // chains that stopped for the host's fallback ladder in the latest launch
    bool pc_frozen = false;        // SMCMC_P_COVARIANCE_FROZEN
    bool pc_broken = false;        // a per-chain launch ended in an error with the chains part-way: Start / Restore again
    int pc_wave = -1;              // SMCMC_P_PERCHAIN_WAVE: -1 automatic, 0 / 1 one chain per lane / per wavefront
    smcmc::PerChainRecord pc_rec = {nullptr, 0, 0};   // the per-step record of the launch in progress (smcmc_step_recorded)
    double* d_pc_rec = nullptr;    // its device buffer
    double* d_pc_stage = nullptr;  // the fallback ladder's staging records (pc_host_ladder), allocated at the first ladder
    int32_t* d_pc_stage_chains = nullptr;
    // smcmc_snapshot / smcmc_rollback: a copy of the ensemble's state on the device (SMCMC_MODE_PER_CHAIN)
    void* snap[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t snap_total_steps = 0;
    bool snap_valid = false, snap_has_forced = false;
    size_t pc_rec_cap = 0;         // ... and capacity in doubles
    std::string error;
};

namespace {

int fail(smcmc_engine* h, int status, const std::string& msg) {
    if (h) h->error = msg;
    return status;
}

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail((h), SMCMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int status_of(smcmc_engine* h, UpdateStatus st) {
    switch (st) {
        case UpdateStatus::Ok: return SMCMC_OK;
        case UpdateStatus::InvalidTrace: return fail(h, SMCMC_ERR_RUNTIME, "Invalid trace");
        case UpdateStatus::IllegalProposalType: return fail(h, SMCMC_ERR_INVALID, "Illegal proposal type");
        case UpdateStatus::UserCorrelationsFailed:
            return fail(h, SMCMC_ERR_RUNTIME, "Decomposition of user correlations failed");
        case UpdateStatus::TargetNotSet: return fail(h, SMCMC_ERR_RUNTIME, "Target acceptance not initialized");
    }
    return SMCMC_ERR_RUNTIME;
}

// Every entry point that touches the device runs on the engine's own device and leaves the
// caller's current device as it found it (two engines on two devices may share a thread).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(h) DeviceGuard device_guard_((h)->device)

// What UpdateProposal does to the per-chain scalars (TSimpleMCMC.H:1042-1043, 1081-1086): sigma is rescaled
// by sqrt(old trace / new trace) and the acceptance trials are de-weighted.  Applied to every chain's
// column at the update itself, so that what is read or saved afterwards is what the reference would hold.
__global__ void adjust_lanes_kernel(double* lane_f64, int npad, int nchains, double sigma_scale, double acc_w,
                                    double acc_wW) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    double* sg = lane_f64 + (size_t)SMCMC_LANE_SIGMA * npad + c;
    *sg = *sg * sigma_scale;
    if (acc_w >= 0.0) {
        double* at = lane_f64 + (size_t)SMCMC_LANE_ACCEPTANCE_TRIALS * npad + c;
        double t = *at;
        t = dmax(1.0, acc_w * t);
        t = dmin(t, acc_wW);
        *at = t;
    }
}

// ResetProposal's effect on every chain's scalars (TSimpleMCMC.H:1405-1410, 1481-1482): counters cleared, the
// acceptance history erased, a collapsed sigma put back to sqrt(1/D); next_update as UpdateProposal(true) leaves it.
__global__ void reset_lanes_kernel(double* lane_f64, int32_t* lane_i32, int npad, int nchains, int next_update,
                                   double acceptance, double acceptance_trials, double sigma_floor, double sigma_reset) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    lane_i32[(size_t)SMCMC_LANE_TRIALS * npad + c] = 0;
    lane_i32[(size_t)SMCMC_LANE_SUCCESSES * npad + c] = 0;
    lane_i32[(size_t)SMCMC_LANE_NEXT_UPDATE * npad + c] = next_update;
    lane_f64[(size_t)SMCMC_LANE_ACCEPTANCE * npad + c] = acceptance;
    lane_f64[(size_t)SMCMC_LANE_ACCEPTANCE_TRIALS * npad + c] = acceptance_trials;
    double* sg = lane_f64 + (size_t)SMCMC_LANE_SIGMA * npad + c;
    if (*sg < sigma_floor) *sg = sigma_reset;
}

size_t npacked(const smcmc_engine* h) { return (size_t)(h->dim + 1) * (h->dim + 2) / 2; }

// doubles of the moment accumulators: per 64-chain group for the register kernels, per
// (slice, 16x16 tile) for the large-dimension fold kernel
size_t gacc_doubles(const smcmc_engine* h) {
    if (h->panel_w) {
        const size_t T = (size_t)(h->dim + 1 + 15) / 16;
        return (size_t)fold_slices(h->dim) * (T * (T + 1) / 2) * 4 * kWave;
    }
    return (size_t)h->ngroups * h->nt * 4 * kWave;
}

// decomposition (and, for QUADFORM, the Error matrix) zero-padded to [dp][dp]
int upload_padded(smcmc_engine* h, const double* src, double* dst_dev) {
    const int D = h->dim, DP = h->dp;
    std::vector<double> pad((size_t)DP * DP, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) pad[(size_t)i * DP + j] = src[(size_t)i * D + j];
    HIP_TRY(h, hipMemcpyAsync(dst_dev, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int upload_shared(smcmc_engine* h) {
    if (h->panel_w) {
        // Uperm[w][i][jl] = U(i, jl*W + w): every wavefront's columns contiguous per row
        const int D = h->dim, W = h->panel_w;
        std::vector<double> perm((size_t)W * D * kPanelCW, 0.0);
        // dimensions with a uniform proposal take no part in the Gaussian step (TSimpleMCMC.H:711-716, 721)
        std::vector<double> Uz(h->prop->decomp);
        std::vector<double> bounds((size_t)2 * D + 8, 0.0);
        uint64_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < D; ++i) {
            if (h->prop->ptype[i] != 1) continue;
            for (int j = 0; j < D; ++j) Uz[(size_t)i * D + j] = Uz[(size_t)j * D + i] = 0.0;
            bounds[i] = h->prop->param1[i];
            bounds[D + i] = h->prop->param2[i];
            mask[i >> 6] |= (uint64_t)1 << (i & 63);
        }
        std::memcpy(&bounds[(size_t)2 * D], mask, sizeof(mask));
        HIP_TRY(h, hipMemcpyAsync(h->d_uniform, bounds.data(), bounds.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        for (int w = 0; w < W; ++w)
            for (int i = 0; i < D; ++i)
                for (int jl = 0; jl < kPanelCW; ++jl) {
                    const int j = jl * W + w;
                    if (j < D) perm[((size_t)w * D + i) * kPanelCW + jl] = Uz[(size_t)i * D + j];
                }
        HIP_TRY(h, hipMemcpyAsync(h->d_U, perm.data(), perm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (!h->exact && !h->prop->decompFull) {
            // Uop[(tile * nkq + kq) * 64 + lane] = U(4 kq + (lane >> 4), 16 tile + (lane & 15))
            const int ntiles = (D + 15) / 16, nkq = (D + 3) / 4, nkqp = panel_mfma_nkq_padded(D);
            std::vector<double> uop(panel_mfma_uop_doubles(D), 0.0);
            for (int jt = 0; jt < ntiles; ++jt)
                for (int kq = 0; kq < nkq; ++kq)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 4 * kq + (l >> 4), j = 16 * jt + (l & 15);
                        if (i < D && j < D) uop[((size_t)jt * nkqp + kq) * 64 + l] = Uz[(size_t)i * D + j];
                    }
            HIP_TRY(h, hipMemcpyAsync(h->d_Uop, uop.data(), uop.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_c0, h->prop->centre.data(), (size_t)D * sizeof(double), hipMemcpyHostToDevice,
                                  h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return SMCMC_OK;
    }
    // Dimensions with a uniform proposal take no part in the Gaussian step
    // (TSimpleMCMC.H:711-716, 721): their rows and columns of the device copy are zero.
    std::vector<double> U(h->prop->decomp);
    std::vector<double> bounds((size_t)2 * h->dp, 0.0);
    for (int i = 0; i < h->dim; ++i) {
        if (h->prop->ptype[i] != 1) continue;
        for (int j = 0; j < h->dim; ++j) U[(size_t)i * h->dim + j] = U[(size_t)j * h->dim + i] = 0.0;
        bounds[i] = h->prop->param1[i];
        bounds[h->dp + i] = h->prop->param2[i];
    }
    int st = upload_padded(h, U.data(), h->d_U);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_uniform, bounds.data(), bounds.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::vector<double> c0(h->dp, 0.0);
    for (int d = 0; d < h->dim; ++d) c0[d] = h->prop->centre[d];
    HIP_TRY(h, hipMemcpyAsync(h->d_c0, c0.data(), c0.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

// The compressed form of Error^T for the serial (reference-order / one lane per chain) quadratic forms.  Taken when the
// matrix is finite, has a full diagonal (so that a non-finite coordinate always shows in the sparse sum, which then
// falls back on the dense one) and at most a quarter of its entries are non-zero.
int upload_like_csr(smcmc_engine* h) {
    (void)hipFree(h->d_like_rowptr); (void)hipFree(h->d_like_cols); (void)hipFree(h->d_like_vals); (void)hipFree(h->d_like_rows);
    h->d_like_rowptr = nullptr; h->d_like_cols = nullptr; h->d_like_vals = nullptr; h->d_like_rows = nullptr; h->like_nnz = 0;
    if (h->likelihood != SMCMC_LIKE_QUADFORM || h->dense_quadform) return SMCMC_OK;
    const int D = h->dim;
    std::vector<int32_t> rows, cols;
    std::vector<double> vals;
    for (int i = 0; i < D; ++i) {
        for (int j = 0; j < D; ++j) {
            const double e = h->like_params[(size_t)j * D + i];                  // Error(j, i) = Error^T(i, j)
            if (!std::isfinite(e)) return SMCMC_OK;
            if (i == j && e == 0.0) return SMCMC_OK;
            if (e != 0.0) { rows.push_back(i); cols.push_back(j); vals.push_back(e); }
        }
    }
    if (cols.size() * 4 > (size_t)D * D) return SMCMC_OK;
    h->like_nnz = (int)cols.size();
    while (cols.size() % smcmc::kQuadChunk != 0) { rows.push_back(0); cols.push_back(0); vals.push_back(0.0); }   // zero entries: skipped terms
    const int32_t padded = (int32_t)cols.size();
    HIP_TRY(h, hipMalloc(&h->d_like_rowptr, sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(&h->d_like_rows, sizeof(int32_t) * rows.size()));
    HIP_TRY(h, hipMalloc(&h->d_like_cols, sizeof(int32_t) * cols.size()));
    HIP_TRY(h, hipMalloc(&h->d_like_vals, sizeof(double) * vals.size()));
    HIP_TRY(h, hipMemcpyAsync(h->d_like_rowptr, &padded, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_like_rows, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_like_cols, cols.data(), sizeof(int32_t) * cols.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_like_vals, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int upload_like(smcmc_engine* h) {
    if (h->panel_w && !h->d_scratch && (h->likelihood == SMCMC_LIKE_USER || h->likelihood == SMCMC_LIKE_CONSTRAINED)) {
        // large dimensions: one lane per chain evaluates these from the proposal's [dim][chain] image
        HIP_TRY(h, hipMalloc(&h->d_scratch, sizeof(double) * (size_t)h->npad * h->dim));
        HIP_TRY(h, hipMemsetAsync(h->d_scratch, 0, sizeof(double) * (size_t)h->npad * h->dim, h->stream));
    }
    if (h->likelihood == SMCMC_LIKE_QUADFORM) {
        if ((int)h->like_params.size() != h->dim * h->dim)
            return fail(h, SMCMC_ERR_INVALID, "QUADFORM needs dim*dim likelihood parameters (the Error matrix)");
        const int D = h->dim;
        { int stc_ = upload_like_csr(h); if (stc_) return stc_; }
        if ((h->panel_w && h->exact) || h->mode == SMCMC_MODE_PER_CHAIN) {
            // large dimensions, reference order (and the per-chain kernel at any dimension): one lane per chain walks the
            // D^2-term sum of TDummyLogLikelihood.H:24-28 with j innermost; it reads row i of Error^T (scalar loads) and
            // the point from a [dim][chain] image
            std::vector<double> et((size_t)D * D);
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) et[(size_t)i * D + j] = h->like_params[(size_t)j * D + i];
            if (h->panel_w && !h->d_scratch) {
                HIP_TRY(h, hipMalloc(&h->d_scratch, sizeof(double) * (size_t)h->npad * D));
                HIP_TRY(h, hipMemsetAsync(h->d_scratch, 0, sizeof(double) * (size_t)h->npad * D, h->stream));
            }
            HIP_TRY(h, hipMemcpyAsync(h->d_like, et.data(), et.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return SMCMC_OK;
        }
        if (h->panel_w) {
            // large dimensions, fused order: Error in matrix-operand order for the row sums of panel_mfma_kernel
            const int ntiles = (D + 15) / 16, nkq = (D + 3) / 4, nkqp = panel_mfma_nkq_padded(D);
            std::vector<double> eop(panel_mfma_uop_doubles(D), 0.0);
            for (int it = 0; it < ntiles; ++it)
                for (int kq = 0; kq < nkq; ++kq)
                    for (int l = 0; l < 64; ++l) {
                        const int i = 16 * it + (l & 15), j = 4 * kq + (l >> 4);
                        if (i < D && j < D) eop[((size_t)it * nkqp + kq) * 64 + l] = h->like_params[(size_t)i * D + j];
                    }
            HIP_TRY(h, hipMemcpyAsync(h->d_like, eop.data(), eop.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return SMCMC_OK;
        }
        // the kernel walks Error(j,i) with j innermost (TDummyLogLikelihood.H:24-28): hand it
        // the transpose so that walk is contiguous
        std::vector<double> et((size_t)D * D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) et[(size_t)i * D + j] = h->like_params[(size_t)j * D + i];
        return upload_padded(h, et.data(), h->d_like);
    }
    if (stress_likelihood(h->likelihood)) {
        std::vector<double> prm;
        if (h->likelihood == SMCMC_LIKE_ASYM) {
            prm = {-1.0, 100.0};                                   // TAsymLogLikelihood.H:21-22
            if (h->like_params.size() == 2) prm = h->like_params;
            else if (!h->like_params.empty()) return fail(h, SMCMC_ERR_INVALID, "ASYM takes {positiveSlope, negativeSlope}");
        } else if (h->likelihood == SMCMC_LIKE_CONSTRAINED) {
            if ((int)h->like_params.size() != 2 + 2 * h->dim)
                return fail(h, SMCMC_ERR_INVALID,
                            "CONSTRAINED needs {SummedValues, SummedConstraint, ExpectedValues[dim], PriorConstraints[dim]}");
            prm = h->like_params;
        } else {
            prm = {0.0};
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_like, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return SMCMC_OK;
    }
    if (h->likelihood == SMCMC_LIKE_USER) {
        if (h->like_params.size() > (size_t)h->dp * h->dp)
            return fail(h, SMCMC_ERR_INVALID, "a user likelihood takes at most dim_padded^2 parameters");
        if (!h->like_params.empty())
            HIP_TRY(h, hipMemcpyAsync(h->d_like, h->like_params.data(), h->like_params.size() * sizeof(double),
                                      hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return SMCMC_OK;
    }
    double b = 100.0;   // ROSEN_B, THardLogLikelihood.H:53
    if (h->likelihood == SMCMC_LIKE_ROSENBROCK && !h->like_params.empty()) b = h->like_params[0];
    HIP_TRY(h, hipMemcpyAsync(h->d_like, &b, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

// sigma rescale + acceptance de-weighting of every chain after an UpdateProposal on the shared proposal
int adjust_lanes(smcmc_engine* h, double sigma_scale) {
    const SharedProposal& P = *h->prop;
    double acc_w = -1.0, acc_wW = 0.0;
    if (P.acceptanceDeweight > 0.0) {
        acc_w = 1.0 - std::min(P.acceptanceDeweight, 1.0);
        acc_wW = acc_w * P.acceptanceWindow;
    }
    const int threads = 256;
    hipLaunchKernelGGL(adjust_lanes_kernel, dim3((h->nchains + threads - 1) / threads), dim3(threads), 0, h->stream,
                       h->d_lane_f64, h->npad, h->nchains, sigma_scale, acc_w, acc_wW);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("adjust_lanes launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

// chain 0's current point: fLastPoint of the shared proposal when it is reset (TSimpleMCMC.H:1484-1485)
int read_chain0(smcmc_engine* h, std::vector<double>& x0) {
    x0.assign(h->dim, 0.0);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy2D(x0.data(), sizeof(double), h->d_x, (size_t)h->npad * sizeof(double), sizeof(double),
                           (size_t)h->dim, hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int reset_lanes(smcmc_engine* h) {
    const SharedProposal& P = *h->prop;
    const int threads = 256;
    const double sr = std::sqrt(1.0 / h->dim);
    hipLaunchKernelGGL(reset_lanes_kernel, dim3((h->nchains + threads - 1) / threads), dim3(threads), 0, h->stream,
                       h->d_lane_f64, h->d_lane_i32, h->npad, h->nchains, P.nextUpdate, P.acceptance,
                       P.acceptanceTrials, 0.01 * sr, sr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("reset_lanes launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}


// UpdateProposal() on the shared proposal and its consequences for the chains: every chain's sigma is
// rescaled by the factor the shared template was, sqrt(old trace / new trace) (TSimpleMCMC.H:1042), and its
// acceptance trials are de-weighted (:1081-1086); when the decomposition ladder ended in ResetProposal (:1389)
// the chains are reset with it, about chain 0's current point.
int update_shared(smcmc_engine* h) {
    SharedProposal& P = *h->prop;
    int st = status_of(h, P.update(false));
    if (st) return st;
    st = adjust_lanes(h, P.lastSigmaScale);
    if (st) return st;
    if (P.lastPath == 4) {
        std::vector<double> x0;
        st = read_chain0(h, x0);
        if (st) return st;
        P.lastPoint = x0;
        P.centre = x0;
        st = reset_lanes(h);
        if (st) return st;
        HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    }
    return SMCMC_OK;
}

// ---- the pooled update on the device (smcmc_pooled_update.hip.h) ---------------------------------
// *h->prop stays the owner of the settings and of everything the fallback ladder needs; the numbers the update
// changes every window (centre, covariance, decomposition, trials, sigma template, sigma trace) live on the device
// and come back to the host only when somebody asks (host_stale) or when a decomposition fails.

bool device_update_eligible(const smcmc_engine* h) {
    if (!h->device_update || h->mode != SMCMC_MODE_POOLED || !h->d_scal) return false;
    for (int d = 0; d < h->dim; ++d)
        if (h->prop->ptype[d] != 0) return false;   // uniform dimensions zero rows and columns of U on the host path
    return true;
}

// host copy -> device copy
int push_shared(smcmc_engine* h) {
    const SharedProposal& P = *h->prop;
    const size_t D = (size_t)h->dim;
    double sc[kPsCount] = {0, 0, 0, 0, 0, 0, 0, 0};
    sc[kPsCovTrials] = P.covTrials; sc[kPsCentreTrials] = P.centreTrials; sc[kPsSigma] = P.sigma;
    sc[kPsSigmaTrace] = P.sigmaTrace; sc[kPsLastScale] = P.lastSigmaScale; sc[kPsStatus] = kPooledOk;
    HIP_TRY(h, hipMemcpyAsync(h->d_cov, P.cov.data(), D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_decomp, P.decomp.data(), D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_centre, P.centre.data(), D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_scal, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // pageable sources
    h->device_stale = false;
    return SMCMC_OK;
}

// device copy -> host copy (synchronises the stream)
int pull_raw(smcmc_engine* h) {
    SharedProposal& P = *h->prop;
    const size_t D = (size_t)h->dim;
    double sc[kPsCount];
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(P.cov.data(), h->d_cov, D * D * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(P.centre.data(), h->d_centre, D * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(sc, h->d_scal, sizeof(sc), hipMemcpyDeviceToHost));
    if (sc[kPsStatus] == kPooledOk)
        HIP_TRY(h, hipMemcpy(P.decomp.data(), h->d_decomp, D * D * sizeof(double), hipMemcpyDeviceToHost));
    P.covTrials = sc[kPsCovTrials]; P.centreTrials = sc[kPsCentreTrials]; P.sigma = sc[kPsSigma];
    P.sigmaTrace = sc[kPsSigmaTrace]; P.lastSigmaScale = sc[kPsLastScale];
    h->host_stale = false;
    return SMCMC_OK;
}

int upload_shared(smcmc_engine* h);
int adjust_lanes(smcmc_engine* h, double sigma_scale);
int read_chain0(smcmc_engine* h, std::vector<double>& x0);
int reset_lanes(smcmc_engine* h);

// Looks at the status word of the latest device update.  A decomposition that failed there continues on the host
// exactly where SharedProposal::update would have: the fallback ladder, then the per-chain consequences.
int check_pending(smcmc_engine* h) {
    if (!h->status_pending) return SMCMC_OK;
    HIP_TRY(h, hipEventSynchronize(h->status_event));
    h->status_pending = false;
    SharedProposal& P = *h->prop;
    const int status = (int)h->h_scal[kPsStatus];
    if (status == kPooledOk) return SMCMC_OK;
    // the host-side bookkeeping of the update was done optimistically: take it back
    P.updateCount = h->before_update.updateCount; P.nextUpdate = h->before_update.nextUpdate;
    P.lastPath = h->before_update.lastPath; P.acceptanceTrials = h->before_update.acceptanceTrials;
    P.decompFull = h->before_update.decompFull;
    if (status == kPooledSkipped) return SMCMC_OK;     // no point was folded: no update (as the host path)
    int st = pull_raw(h);
    if (st) return st;
    h->device_stale = true;
    // what SharedProposal::update does around the decomposition (the device did the numbers)
    ++P.updateCount;
    if (status == kPooledInvalidTrace) return status_of(h, UpdateStatus::InvalidTrace);
    {
        const double maxUp = (double)h->dim * (double)h->dim;
        P.nextUpdate = (int)(P.acceptanceWindow + maxUp - maxUp / (0.5 * P.successes + 1.0));
        if (P.acceptanceDeweight > 0.0) {
            if (P.acceptanceDeweight > 1.0) P.acceptanceDeweight = 1.0;
            const double w = 1.0 - P.acceptanceDeweight;
            P.acceptanceTrials = std::max(1.0, w * P.acceptanceTrials);
            P.acceptanceTrials = std::min(P.acceptanceTrials, w * P.acceptanceWindow);
        }
        if (P.covDeweight > 1.0) P.covDeweight = 1.0;
    }
    st = status_of(h, P.finishUpdateOnHost(h->h_scal[kPsLastScale]));
    if (st) return st;
    st = adjust_lanes(h, P.lastSigmaScale);
    if (st) return st;
    if (P.lastPath == 4) {
        std::vector<double> x0;
        st = read_chain0(h, x0);
        if (st) return st;
        P.lastPoint = x0;
        P.centre = x0;
        st = reset_lanes(h);
        if (st) return st;
        HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    }
    return upload_shared(h);
}

// before the host reads (mutate = false) or changes (mutate = true) anything an update touches
int sync_shared_to_host(smcmc_engine* h, bool mutate) {
    int st = check_pending(h);
    if (st) return st;
    if (h->host_stale) {
        st = pull_raw(h);
        if (st) return st;
    }
    if (mutate) h->device_stale = true;
    return SMCMC_OK;
}

// smcmc_apply_moments without the host: absorb, scalar half, Cholesky, operand layouts, per-chain consequences
int device_apply(smcmc_engine* h) {
    int st = check_pending(h);
    if (st) return st;
    SharedProposal& P = *h->prop;
    if (h->device_stale) {
        if (h->host_stale) return fail(h, SMCMC_ERR_LOGIC, "shared proposal: host and device copies both changed");
        st = push_shared(h);
        if (st) return st;
    }
    if (!h->update_prepared) {   // per engine, i.e. per device: the panel kernel's dynamic LDS limit
        const hipError_t prepared = pooled_update_prepare();
        if (prepared != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("pooled update setup: ") + hipGetErrorString(prepared));
        h->update_prepared = true;
    }
    PooledUpdateParams u;
    u.D = h->dim; u.M = h->d_moments; u.centre = h->d_centre; u.cov = h->d_cov; u.decomp = h->d_decomp; u.scal = h->d_scal;
    u.cov_window = P.covWindow; u.cov_deweight = P.covDeweight;
    PooledPublishParams q;
    std::memset(&q, 0, sizeof(q));
    q.D = h->dim; q.decomp = h->d_decomp; q.centre = h->d_centre; q.scal = h->d_scal; q.U = h->d_U; q.c0 = h->d_c0;
    hipError_t e;
    if (h->panel_w) {
        q.W = h->panel_w; q.CW = kPanelCW;
        if (!h->exact) { q.Uop = h->d_Uop; q.nkq_padded = panel_mfma_nkq_padded(h->dim); }
        e = launch_pooled_update(u, h->stream);
        if (e == hipSuccess) e = launch_pooled_publish(q, h->stream);
    } else {
        q.DP = h->dp;
        e = launch_pooled_small_update(u, q, h->stream);   // dim <= 63: one single-workgroup kernel
    }
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("pooled update launch: ") + hipGetErrorString(e));
    // the host-only half of SharedProposal::update, optimistically (check_pending takes it back if the device says no)
    h->before_update = {P.updateCount, P.nextUpdate, P.lastPath, P.acceptanceTrials, P.decompFull};
    ++P.updateCount;
    const double maxUp = (double)h->dim * (double)h->dim;
    P.nextUpdate = (int)(P.acceptanceWindow + maxUp - maxUp / (0.5 * P.successes + 1.0));
    if (P.covDeweight > 1.0) P.covDeweight = 1.0;
    double acc_w = -1.0, acc_wW = 0.0;
    if (P.acceptanceDeweight > 0.0) {
        if (P.acceptanceDeweight > 1.0) P.acceptanceDeweight = 1.0;
        const double w = 1.0 - P.acceptanceDeweight;
        P.acceptanceTrials = std::max(1.0, w * P.acceptanceTrials);
        P.acceptanceTrials = std::min(P.acceptanceTrials, w * P.acceptanceWindow);
        acc_w = 1.0 - std::min(P.acceptanceDeweight, 1.0);
        acc_wW = acc_w * P.acceptanceWindow;
    }
    P.lastPath = 0;
    P.decompFull = false;
    // (the kernel also writes the update's scalars into h_scal, pinned and device-visible: no copy command behind it)
    e = launch_pooled_adjust_lanes(h->d_lane_f64, h->npad, h->nchains, h->d_scal, acc_w, acc_wW, SMCMC_LANE_SIGMA,
                                   SMCMC_LANE_ACCEPTANCE_TRIALS, h->h_scal_dev, h->stream);
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("adjust_lanes launch: ") + hipGetErrorString(e));
    if (!h->h_scal_dev)
        HIP_TRY(h, hipMemcpyAsync(h->h_scal, h->d_scal, sizeof(double) * kPsCount, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->status_event, h->stream));
    h->status_pending = true;
    h->host_stale = true;
    return SMCMC_OK;
}

// The ring of per-step states behind multi-step launches of the pooled large-dimension path (at most 8 steps, at
// most 2 GiB); 0 steps when even two states do not fit the budget.
int ensure_ring(smcmc_engine* h) {
    if (h->ring_steps >= 0) return SMCMC_OK;
    const size_t state = sizeof(double) * (size_t)h->npad * h->dim;
    size_t steps = ((size_t)4 << 30) / state;
    if (steps > (size_t)smcmc::kFoldMaxSrc) steps = smcmc::kFoldMaxSrc;   // one fold launch takes the whole ring
    if (steps < 2) { h->ring_steps = 0; return SMCMC_OK; }
    // no memory for the ring is no error: the one-step launches with a fold between them need none
    if (hipMalloc(&h->d_ring, state * steps) != hipSuccess ||
        hipMalloc(&h->d_ring_logl, sizeof(double) * (size_t)h->npad * steps) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(h->d_ring); (void)hipFree(h->d_ring_logl);
    (void)hipFree(h->d_pc_rec);
    for (int k = 0; k < 8; ++k) (void)hipFree(h->snap[k]);
    smcmc::fold_ring_release(h->fold);
        h->d_ring = nullptr; h->d_ring_logl = nullptr;
        h->ring_steps = 0;
        return SMCMC_OK;
    }
    h->ring_steps = (int)steps;
    return SMCMC_OK;
}

// Folds `n` points ([dim][npad] each, in this order) into the moment groups: one launch, the accumulators in registers
// from the first point to the last.
int fold_points(smcmc_engine* h, const double* const* pts, int n) {
    for (int done = 0; done < n; done += smcmc::kFoldMaxSrc) {
        smcmc::FoldRingParams fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.nsrc = std::min(n - done, (int)smcmc::kFoldMaxSrc);
        for (int k = 0; k < fp.nsrc; ++k) fp.src[k] = pts[done + k];
        fp.c0 = h->d_c0; fp.nchains = h->nchains; fp.npad = h->npad; fp.D = h->dim; fp.slice_chains = h->slice_chains;
        fp.gacc = h->d_gacc; fp.mask = nullptr;
        const hipError_t e = smcmc::launch_fold_ring(h->fold, fp, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("fold kernel launch: ") + hipGetErrorString(e));
    }
    return SMCMC_OK;
}

// ---- SMCMC_MODE_PER_CHAIN: every chain owns its adaptive state (smcmc_perchain_kernel.hip.h) --------------------
bool per_chain(const smcmc_engine* h) { return h->mode == SMCMC_MODE_PER_CHAIN; }

int pc_alloc(smcmc_engine* h) {
    if (h->d_pc_cov) return SMCMC_OK;
    const size_t NP = (size_t)h->npad, D = (size_t)h->dim, npk = D * (D + 1) / 2;
    // the kernel reads both streams in whole chunks and primes its buffers unconditionally: padding behind the last tile
    const size_t pad = (size_t)smcmc::kPcPad * smcmc::kWave;
    HIP_TRY(h, hipMalloc(&h->d_pc_cov, sizeof(double) * (npk * NP + pad)));
    HIP_TRY(h, hipMalloc(&h->d_pc_ut, sizeof(double) * (D * D * NP + pad)));
    HIP_TRY(h, hipMalloc(&h->d_pc_centre, sizeof(double) * D * NP));
    HIP_TRY(h, hipMalloc(&h->d_pc_last, sizeof(double) * D * NP));
    HIP_TRY(h, hipMalloc(&h->d_pc_tmpl, sizeof(double) * (npk + D * D + D)));
    HIP_TRY(h, hipMalloc(&h->d_pc_flag, sizeof(int)));
    HIP_TRY(h, hipMemset(h->d_pc_cov, 0, sizeof(double) * (npk * NP + pad)));
    HIP_TRY(h, hipMemset(h->d_pc_ut, 0, sizeof(double) * (D * D * NP + pad)));
    HIP_TRY(h, hipMemset(h->d_pc_centre, 0, sizeof(double) * D * NP));
    HIP_TRY(h, hipMemset(h->d_pc_last, 0, sizeof(double) * D * NP));
    HIP_TRY(h, hipMemset(h->d_pc_flag, 0, sizeof(int)));
    if (!h->d_proposed) {   // the proposal's image: fProposed is always kept in this mode
        HIP_TRY(h, hipMalloc(&h->d_proposed, sizeof(double) * NP * h->dp));
        HIP_TRY(h, hipMemset(h->d_proposed, 0, sizeof(double) * NP * h->dp));
    }
    h->keep_proposed = true;
    return SMCMC_OK;
}

// the engine's layout of a decomposition: kk = j (j + 1) / 2 + i holds U(i, j), i <= j; a full matrix keeps
// U(i, j), j < i, at npk + i (i - 1) / 2 + j
void pc_pack_decomp(const SharedProposal& P, std::vector<double>& ut) {
    const int D = P.D, npk = D * (D + 1) / 2;
    ut.assign((size_t)D * D, 0.0);
    for (int j = 0; j < D; ++j)
        for (int i = 0; i <= j; ++i) ut[(size_t)j * (j + 1) / 2 + i] = P.decomp[(size_t)i * D + j];
    if (P.decompFull)
        for (int i = 1; i < D; ++i)
            for (int j = 0; j < i; ++j) ut[(size_t)npk + (size_t)i * (i - 1) / 2 + j] = P.decomp[(size_t)i * D + j];
}

void pc_deweights(const SharedProposal& P, double& acc_w, double& acc_wW, double& cov_w, double& cov_wW) {
    acc_w = -1.0; acc_wW = 0.0; cov_w = -1.0; cov_wW = 0.0;
    if (P.acceptanceDeweight > 0.0) {
        acc_w = 1.0 - std::min(P.acceptanceDeweight, 1.0);
        acc_wW = acc_w * P.acceptanceWindow;
    }
    if (P.covDeweight > 0.0) {
        cov_w = 1.0 - std::min(P.covDeweight, 1.0);
        cov_wW = cov_w * P.covWindow;
    }
}

// Hands the template *h->prop (what InitializeState / RestoreState / ResetProposal computed once on the host) to every
// chain.  reset: the explicit ResetProposal() of running chains; centre: RestoreState's saved centre, or null for the
// chain's own point.
int pc_broadcast(smcmc_engine* h, bool reset, const double* centre) {
    const SharedProposal& P = *h->prop;
    const int D = h->dim, npk = D * (D + 1) / 2;
    std::vector<double> tmpl((size_t)npk + (size_t)D * D + D, 0.0), ut;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) tmpl[(size_t)i * (i + 1) / 2 + j] = P.cov[(size_t)i * D + j];
    pc_pack_decomp(P, ut);
    std::copy(ut.begin(), ut.end(), tmpl.begin() + npk);
    if (centre) std::copy(centre, centre + D, tmpl.begin() + npk + (size_t)D * D);
    HIP_TRY(h, hipMemcpyAsync(h->d_pc_tmpl, tmpl.data(), tmpl.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // pageable source
    PerChainBroadcast b;
    std::memset(&b, 0, sizeof(b));
    b.nchains = h->nchains; b.npad = h->npad; b.dim = D; b.reset = reset ? 1 : 0;
    b.cov_packed = h->d_pc_tmpl; b.ut = h->d_pc_tmpl + npk;
    b.centre = centre ? h->d_pc_tmpl + npk + (size_t)D * D : nullptr;
    b.decomp_full = P.decompFull ? 1 : 0; b.last_path = P.lastPath; b.update_count = P.updateCount;
    b.sigma = P.sigma; b.sigma_trace = P.sigmaTrace; b.centre_trials = P.centreTrials; b.cov_trials = P.covTrials;
    b.acceptance = P.acceptance; b.acceptance_trials = std::min(10.0, 0.5 * P.acceptanceWindow);   // :1482 (before the de-weighting)
    b.next_update = P.nextUpdate;
    const double sr = std::sqrt(1.0 / D);
    b.sigma_floor = 0.01 * sr; b.sigma_reset = sr;
    pc_deweights(P, b.acc_w, b.acc_wW, b.cov_w, b.cov_wW);
    b.x = h->d_x; b.last_point = h->d_pc_last; b.centre_out = h->d_pc_centre; b.cov = h->d_pc_cov; b.ut_out = h->d_pc_ut;
    b.lane_f64 = h->d_lane_f64; b.lane_i32 = h->d_lane_i32;
    const hipError_t e = launch_perchain_broadcast(b, h->stream);
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("per-chain broadcast launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

// one column of a [rows][npad] array
template <typename T>
hipError_t pc_get_column(const T* base, size_t npad, int chain, int rows, T* out) {
    return hipMemcpy2D(out, sizeof(T), base + chain, npad * sizeof(T), sizeof(T), (size_t)rows, hipMemcpyDeviceToHost);
}
template <typename T>
hipError_t pc_put_column(T* base, size_t npad, int chain, int rows, const T* in) {
    return hipMemcpy2D(base + chain, npad * sizeof(T), in, sizeof(T), sizeof(T), (size_t)rows, hipMemcpyHostToDevice);
}
// one chain's elements of a wavefront-tiled image (pc_tile_index): `count` of the `rows` elements of the tile
inline hipError_t pc_get_tiled(const double* base, int chain, int rows, int count, double* out) {
    return hipMemcpy2D(out, sizeof(double), base + smcmc::pc_tile_index(0, (size_t)chain, rows), smcmc::kWave * sizeof(double),
                       sizeof(double), (size_t)count, hipMemcpyDeviceToHost);
}
inline hipError_t pc_put_tiled(double* base, int chain, int rows, int count, const double* in) {
    return hipMemcpy2D(base + smcmc::pc_tile_index(0, (size_t)chain, rows), smcmc::kWave * sizeof(double), in, sizeof(double),
                       sizeof(double), (size_t)count, hipMemcpyHostToDevice);
}

// The chains the latest launch stopped (a Cholesky pivot failed inside their UpdateProposal, or their covariance has
// no trace): the fallback ladder of TSimpleMCMC.H:1134-1389 on the host, chain by chain, exactly where
// SharedProposal::update would go on; then the chain is marked to resume its step (or, for the explicit
// UpdateProposal() call, to be simply done).
int pc_host_ladder(smcmc_engine* h, bool explicit_update) {
    const int D = h->dim, npk = D * (D + 1) / 2;
    const size_t NP = (size_t)h->npad, stride = smcmc::pc_stage_stride(D);
    std::vector<int32_t> status(NP), flagged;
    HIP_TRY(h, hipMemcpy(status.data(), h->d_lane_i32 + (size_t)SMCMC_LANE_UPDATE_STATUS * NP, NP * sizeof(int32_t),
                         hipMemcpyDeviceToHost));
    for (int c = 0; c < h->nchains; ++c) {
        if (status[c] == kPcOk || status[c] == kPcResume) continue;
        if (status[c] == kPcInvalidTrace) return status_of(h, UpdateStatus::InvalidTrace);     // :1025-1028
        flagged.push_back(c);
    }
    // the flagged chains in batches through one staging buffer: one gather launch, one copy each way, one scatter launch
    // per batch (chain by chain this was ten strided copies of 8-byte rows per chain)
    const size_t batch_max = std::max<size_t>(1, std::min<size_t>(4096, ((size_t)256 << 20) / (stride * sizeof(double))));
    if (!h->d_pc_stage) {
        HIP_TRY(h, hipMalloc(&h->d_pc_stage, batch_max * stride * sizeof(double)));
        HIP_TRY(h, hipMalloc(&h->d_pc_stage_chains, batch_max * sizeof(int32_t)));
    }
    std::vector<double> stage, ut;
    for (size_t b0 = 0; b0 < flagged.size(); b0 += batch_max) {
        const int nb = (int)std::min(batch_max, flagged.size() - b0);
        stage.resize((size_t)nb * stride);
        HIP_TRY(h, hipMemcpyAsync(h->d_pc_stage_chains, flagged.data() + b0, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice,
                                  h->stream));
        smcmc::PerChainStage g;
        g.chains = h->d_pc_stage_chains; g.stage = h->d_pc_stage; g.npad = h->npad; g.dim = D;
        g.lane_f64 = h->d_lane_f64; g.lane_i32 = h->d_lane_i32;
        g.cov = h->d_pc_cov; g.ut = h->d_pc_ut; g.centre = h->d_pc_centre; g.last = h->d_pc_last;
        hipError_t e = smcmc::launch_perchain_stage(g, nb, false, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("per-chain ladder gather: ") + hipGetErrorString(e));
        HIP_TRY(h, hipMemcpyAsync(stage.data(), h->d_pc_stage, stage.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int q = 0; q < nb; ++q) {
            double* lf = stage.data() + (size_t)q * stride;
            double* li = lf + SMCMC_LANE_F64_COUNT_;          // (the 32-bit lane values, as doubles)
            double* packed = li + SMCMC_LANE_I32_COUNT_;
            double* utq = packed + npk;
            double* centre = utq + (size_t)D * D;
            double* last = centre + D;
            SharedProposal T(*h->prop);                      // the settings; the state comes from the chain
            std::copy(centre, centre + D, T.centre.begin());
            std::copy(last, last + D, T.lastPoint.begin());
            for (int i = 0; i < D; ++i)
                for (int j = 0; j <= i; ++j) T.C(i, j) = T.C(j, i) = packed[(size_t)i * (i + 1) / 2 + j];
            T.initialized = true;
            T.centreTrials = lf[SMCMC_LANE_CENTER_TRIALS]; T.covTrials = lf[SMCMC_LANE_COVARIANCE_TRIALS];
            T.sigma = lf[SMCMC_LANE_SIGMA]; T.sigmaTrace = lf[SMCMC_LANE_SIGMA_TRACE];
            T.acceptance = lf[SMCMC_LANE_ACCEPTANCE]; T.acceptanceTrials = lf[SMCMC_LANE_ACCEPTANCE_TRIALS];
            T.successes = (int)li[SMCMC_LANE_SUCCESSES]; T.nextUpdate = (int)li[SMCMC_LANE_NEXT_UPDATE];
            T.updateCount = (int)li[SMCMC_LANE_UPDATE_COUNT];
            const int st = status_of(h, T.finishUpdateOnHost(1.0));
            if (st) return st;
            for (int i = 0; i < D; ++i)
                for (int j = 0; j <= i; ++j) packed[(size_t)i * (i + 1) / 2 + j] = T.cov[(size_t)i * D + j];
            pc_pack_decomp(T, ut);
            std::copy(ut.begin(), ut.end(), utq);
            std::copy(T.centre.begin(), T.centre.begin() + D, centre);
            lf[SMCMC_LANE_CENTER_TRIALS] = T.centreTrials; lf[SMCMC_LANE_COVARIANCE_TRIALS] = T.covTrials;
            lf[SMCMC_LANE_SIGMA] = T.sigma; lf[SMCMC_LANE_SIGMA_TRACE] = T.sigmaTrace;
            lf[SMCMC_LANE_ACCEPTANCE] = T.acceptance; lf[SMCMC_LANE_ACCEPTANCE_TRIALS] = T.acceptanceTrials;
            li[SMCMC_LANE_SUCCESSES] = T.successes; li[SMCMC_LANE_NEXT_UPDATE] = T.nextUpdate;
            li[SMCMC_LANE_UPDATE_COUNT] = T.updateCount; li[SMCMC_LANE_LAST_UPDATE_PATH] = T.lastPath;
            li[SMCMC_LANE_DECOMP_FULL] = T.decompFull ? 1 : 0;
            if (T.lastPath == 4) li[SMCMC_LANE_TRIALS] = 0;    // the ladder ended in ResetProposal (:1389, 1405)
            li[SMCMC_LANE_UPDATE_STATUS] = explicit_update ? kPcOk : kPcResume;
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_pc_stage, stage.data(), stage.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        e = smcmc::launch_perchain_stage(g, nb, true, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("per-chain ladder scatter: ") + hipGetErrorString(e));
        HIP_TRY(h, hipStreamSynchronize(h->stream));     // pageable source
    }
    return SMCMC_OK;
}

PerChainParams pc_params(smcmc_engine* h, const StepParams& p) {
    const SharedProposal& P = *h->prop;
    PerChainParams q;
    std::memset(&q, 0, sizeof(q));
    q.nchains = p.nchains; q.npad = p.npad; q.dim = p.dim; q.metropolis = p.metropolis;
    q.step0 = h->total_steps; q.target_step = h->total_steps + (uint32_t)p.nsteps;
    q.chain_offset = p.chain_offset; q.seed = p.seed; q.like = h->d_like;
    q.like_csr = QuadCsr{h->d_like_rowptr, h->d_like_cols, h->d_like_vals, h->d_like_rows};
    q.target = p.target; q.acc_window = p.acc_window; q.asig = p.asig; q.max_up = p.max_up;
    pc_deweights(P, q.acc_w, q.acc_wW, q.cov_w, q.cov_wW);
    q.cov_window = P.covWindow; q.cov_frozen = h->pc_frozen ? 1 : 0;
    q.step_rms_window = p.step_rms_window;
    q.has_forced = p.has_forced; q.forced = p.forced;
    q.x = h->d_x; q.proposed = h->d_proposed; q.last_point = h->d_pc_last; q.centre = h->d_pc_centre;
    q.cov = h->d_pc_cov; q.ut = h->d_pc_ut; q.lane_f64 = h->d_lane_f64; q.lane_i32 = h->d_lane_i32;
    q.save_x = p.save_x; q.save_logl = p.save_logl; q.save_stride = p.save_stride;
    q.flag_count = h->d_pc_flag;
    return q;
}

// Launches until every chain has reached the target step: a chain whose decomposition failed waits for the host's
// ladder and catches up in the next launch (its draws are keyed on its own step count).
// One chain per wavefront (smcmc_perchain_wave.hip.h) where an ensemble is too small to fill the chip with one chain per
// lane; the two kernels share every image, so the choice can change from launch to launch.
bool pc_use_wave(const smcmc_engine* h) {
    if (!smcmc::perchain_wave_serves(h->likelihood)) return false;
    if (h->likelihood == SMCMC_LIKE_USER) return true;     // (the one-chain-per-lane kernel has no user instantiation)
    if (h->pc_wave >= 0) return h->pc_wave != 0;
    return true;     // measured faster at every ensemble size, 1 to 65 536 chains (profiles/r04_notes.md)
}

int pc_run(smcmc_engine* h, PerChainParams q) {
    const bool wave = pc_use_wave(h);
    if (h->pc_rec.rec && !wave)
        return fail(h, SMCMC_ERR_UNSUPPORTED, "a per-step record needs the one-chain-per-wavefront kernel (SMCMC_P_PERCHAIN_WAVE)");
    for (int round = 0; round < 1000; ++round) {
        const hipError_t e = wave ? smcmc::launch_perchain_wave(q, h->pc_rec, h->likelihood, h->stream)
                                  : launch_perchain(q, h->likelihood, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("per-chain kernel launch: ") + hipGetErrorString(e));
        int flagged = 0;
        HIP_TRY(h, hipMemcpyAsync(&flagged, h->d_pc_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (flagged == 0) return SMCMC_OK;
        HIP_TRY(h, hipMemset(h->d_pc_flag, 0, sizeof(int)));
        const int st = pc_host_ladder(h, q.update_only != 0);
        if (st) return st;
        if (q.update_only) return SMCMC_OK;
        q.has_forced = 0;
    }
    return fail(h, SMCMC_ERR_RUNTIME, "per-chain update: the fallback ladder does not converge");
}

int pc_check_supported(smcmc_engine* h) {
    if (h->panel_w || h->dim > kPcMaxDim)
        return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN serves dim <= 63");
    if (!h->exact)
        return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN runs in reference-order arithmetic (SMCMC_P_EXACT_ARITHMETIC = 1)");
    if (h->likelihood == SMCMC_LIKE_USER && !smcmc::perchain_wave_serves(SMCMC_LIKE_USER))   // (the one-chain-per-wavefront kernel only)
        return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN serves the built-in likelihoods");
    for (int d = 0; d < h->dim; ++d)
        if (h->prop->ptype[d] != 0)
            return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN: uniform proposals are served by the shared-proposal modes");
    if (h->scan_dim >= 0)
        return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN: the scan of a dimension is served by the shared-proposal modes");
    return SMCMC_OK;
}

StepParams make_params(smcmc_engine* h, int nsteps, int metropolis) {
    const SharedProposal& P = *h->prop;
    StepParams p;
    std::memset(&p, 0, sizeof(p));
    p.nchains = h->nchains; p.npad = h->npad; p.dim = h->dim;
    p.nsteps = nsteps; p.metropolis = metropolis;
    p.step0 = h->total_steps;
    p.chain_offset = h->chain_offset;
    p.seed = h->seed;
    p.U = h->d_U; p.like = h->d_like; p.c0 = h->d_c0;
    p.like_rowptr = h->d_like_rowptr; p.like_cols = h->d_like_cols; p.like_vals = h->d_like_vals; p.like_rows = h->d_like_rows;
    p.target = P.target;
    p.acc_window = P.acceptanceWindow;
    double asig = P.target * (1.0 - P.target);              // TSimpleMCMC.H:1746-1747
    asig = std::sqrt(asig / P.acceptanceWindow);
    p.asig = asig;
    p.max_up = (double)h->dim * (double)h->dim;             // :1050
    if (P.acceptanceDeweight > 0.0) {
        const double w = 1.0 - std::min(P.acceptanceDeweight, 1.0);
        p.acc_w = w;
        p.acc_wW = w * P.acceptanceWindow;
    } else {
        p.acc_w = -1.0;
        p.acc_wW = 0.0;
    }
    p.per_lane_update = (h->mode == SMCMC_MODE_FROZEN) ? 1 : 0;
    p.step_rms_window = h->step_rms_window;
    p.has_forced = h->has_forced ? 1 : 0;
    p.forced = h->d_forced;
    p.x = h->d_x; p.lane_f64 = h->d_lane_f64; p.lane_i32 = h->d_lane_i32;
    p.gacc = h->d_gacc;
    p.save_x = nullptr; p.save_logl = nullptr; p.save_stride = 1;
    p.proposed = h->keep_proposed ? h->d_proposed : nullptr;
    p.uniform = h->d_uniform;
    for (int d = 0; d < h->dim && d < 64; ++d)   // the register kernels (dim <= 63); larger dimensions carry theirs in d_uniform
        if (P.ptype[d] == 1) p.uniform_mask |= (uint64_t)1 << d;
    p.scan_dim = h->scan_dim;
    if (h->scan_dim >= 0) {
        const int sd = h->scan_dim;
        if (P.ptype[sd] == 1) {
            p.scan_uniform = 1; p.scan_a = P.param1[sd]; p.scan_b = P.param2[sd];
        } else {
            // Gaussian about the estimated centre (TSimpleMCMC.H:696-701); the centre is the
            // ensemble's shared estimate
            p.scan_a = P.centre[sd];
            p.scan_b = (P.param1[sd] > 0) ? std::sqrt(P.param1[sd]) : 1.0;
        }
    }
    return p;
}

// the reference-order kernel for dim > 63
hipError_t launch_panel_exact(smcmc_engine* h, const PanelParams& q) {
#ifdef SMCMC_USER_LIKELIHOOD
    if (h->likelihood == SMCMC_LIKE_USER)
        return (h->panel_w == 4) ? launch_panel_user<4, kPanelCW>(q, h->stream) : launch_panel_user<8, kPanelCW>(q, h->stream);
#endif
    return (h->panel_w == 4) ? launch_panel<4, kPanelCW>(q, h->likelihood, true, h->stream)
                             : launch_panel<8, kPanelCW>(q, h->likelihood, true, h->stream);
}

int launch(smcmc_engine* h, int nsteps, int metropolis, int stride, double* save_x, double* save_logl) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (!h->started) return fail(h, SMCMC_ERR_INVALID, "Uninitialized starting point");   // TSimpleMCMC.H:371-374
    if (nsteps <= 0) return SMCMC_OK;
    if (metropolis < 0 || metropolis > 2) return fail(h, SMCMC_ERR_INVALID, "metropolis must be 0, 1 or 2");
    if (!h->overlap_update) {
        // parity mode: a fallback of the latest pooled update takes effect before the next step
        int pst = check_pending(h);
        if (pst) return pst;
    }
    StepParams p = make_params(h, nsteps, metropolis);
    if (save_x) {
        if (stride <= 0) return fail(h, SMCMC_ERR_INVALID, "save stride must be positive");
        p.save_x = save_x; p.save_logl = save_logl; p.save_stride = stride;
    }
    if (per_chain(h)) {
        int st = pc_check_supported(h);
        if (st) return st;
        if (h->pc_broken)
            return fail(h, SMCMC_ERR_LOGIC, "an earlier launch of this per-chain ensemble ended in an error with its chains part-way "
                                            "through (the reference's Step() would have thrown, TSimpleMCMC.H:1025-1028): Start or Restore it");
        st = pc_run(h, pc_params(h, p));
        if (st) {
            // the chains that ran have moved their own step counts and state; the ensemble's count has not: terminal
            h->pc_broken = true;
            return st;
        }
        h->total_steps += (uint32_t)nsteps;
        h->has_forced = false;
        return SMCMC_OK;
    }
    if (h->panel_w) {
        PanelParams q;
        std::memset(&q, 0, sizeof(q));
        q.nchains = p.nchains; q.npad = p.npad; q.dim = p.dim; q.metropolis = p.metropolis;
        q.chain_offset = p.chain_offset; q.seed = p.seed;
        q.Uperm = h->d_U; q.like = h->d_like;
        q.like_csr = QuadCsr{h->d_like_rowptr, h->d_like_cols, h->d_like_vals, h->d_like_rows};
        q.target = p.target; q.acc_window = p.acc_window; q.asig = p.asig; q.max_up = p.max_up;
        q.acc_w = p.acc_w; q.acc_wW = p.acc_wW; q.per_lane_update = p.per_lane_update;
        q.step_rms_window = p.step_rms_window; q.full_u = h->prop->decompFull ? 1 : 0;
        q.x = p.x; q.lane_f64 = p.lane_f64; q.lane_i32 = p.lane_i32;
        q.save_stride = 1;
        q.has_forced = p.has_forced; q.forced = p.forced;
        q.proposed = p.proposed;
        q.scratch = h->d_scratch;
        q.uniform = p.uniform; q.scan_dim = p.scan_dim; q.scan_uniform = p.scan_uniform;
        q.scan_a = p.scan_a; q.scan_b = p.scan_b;
        for (int d = 0; d < h->dim; ++d)
            if (h->prop->ptype[d] == 1) q.special = 1;
        if (p.scan_dim >= 0) q.special = 1;
        const bool exact = h->exact || h->prop->decompFull;
        const bool special_proposal = q.special != 0;
        if (exact && q.proposed != nullptr) q.special = 1;   // the SPECIAL instantiation also stores the proposal
        if (special_proposal && !exact)
            return fail(h, SMCMC_ERR_UNSUPPORTED,
                        "uniform proposals and the scan of a dimension run in reference-order arithmetic only");
        if ((stress_likelihood(h->likelihood) || h->likelihood == SMCMC_LIKE_USER) && !exact)
            return fail(h, SMCMC_ERR_UNSUPPORTED,
                        "the stress likelihoods (ASYM, HORRIFIC, CONSTRAINED) and user likelihoods for dim > 63 run in "
                        "reference-order arithmetic only (SMCMC_P_EXACT_ARITHMETIC = 1)");
        if (h->likelihood == SMCMC_LIKE_QUADFORM && exact != h->exact)
            return fail(h, SMCMC_ERR_UNSUPPORTED,
                        "the quadratic-form likelihood for dim > 63 with a full (eigen) decomposition needs "
                        "reference-order arithmetic (SMCMC_P_EXACT_ARITHMETIC = 1)");
        const bool pooled = (h->mode == SMCMC_MODE_POOLED);
        if (pooled && save_x) return fail(h, SMCMC_ERR_UNSUPPORTED, "saving inside a pooled large-dimension launch");
        if (!pooled) { q.save_x = p.save_x; q.save_logl = p.save_logl; q.save_stride = p.save_stride; }
        // POOLED: the point UpdateState sees at the start of step t is folded into the moments
        // when (t - 1) % moment_stride == 0; the step launches are cut at those steps
        int done = 0;
        // Covariance fed every step: a launch of several steps leaves the point after each of them in a ring, and the
        // folds of those points follow the launch, in step order -- the order of a fold between every two one-step
        // launches, without their state round trips (255 us each at config 4 in the fused order).
        const bool ring_wanted = pooled && h->moment_stride == 1 && !q.has_forced && q.scan_dim < 0 && nsteps >= 2;
        if (ring_wanted) {
            int rst = ensure_ring(h);
            if (rst) return rst;
        }
        // x_folded: the point in d_x has already gone into the moments (as the last slot of the previous segment's ring)
        bool x_folded = false;
        while (done < nsteps) {
            int seg = nsteps - done;
            if (ring_wanted && h->ring_steps >= 2 && seg >= 2) {
                seg = std::min(seg, h->ring_steps);
                if (!x_folded) {
                    const double* px = h->d_x;
                    const int fst = fold_points(h, &px, 1);
                    if (fst) return fst;
                }
                q.nsteps = seg;
                q.step0 = h->total_steps;
                q.save_x = h->d_ring; q.save_logl = h->d_ring_logl; q.save_stride = 1;
                hipError_t e;
                if (!exact) {
                    q.Uperm = h->d_Uop;
                    e = launch_panel_mfma(q, h->likelihood, h->stream);
                } else {
                    q.Uperm = h->d_U;
                    e = launch_panel_exact(h, q);
                }
                if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("panel kernel launch: ") + hipGetErrorString(e));
                q.save_x = nullptr; q.save_logl = nullptr;
                // The point after step s of the launch is the one UpdateState sees at the start of step s + 1.  The one
                // after the LAST step (= d_x) goes in here too when this call steps on; otherwise the next call folds it.
                const bool more = done + seg < nsteps;
                const int npts = more ? seg : seg - 1;
                const double* pts[smcmc::kFoldMaxSrc];
                for (int s = 0; s < npts; ++s) pts[s] = h->d_ring + (size_t)s * h->dim * h->npad;
                if (npts > 0) {
                    const int fst = fold_points(h, pts, npts);
                    if (fst) return fst;
                }
                x_folded = more;
                h->total_steps += (uint32_t)seg;
                done += seg;
                continue;
            }
            if (pooled) {
                const int phase = (int)(h->total_steps % (uint32_t)h->moment_stride);
                // a forced step or a scan does not call UpdateState: nothing to fold
                if (phase == 0 && !q.has_forced && q.scan_dim < 0 && !x_folded) {
                    const double* px = h->d_x;
                    const int fst = fold_points(h, &px, 1);
                    if (fst) return fst;
                }
                x_folded = false;
                seg = std::min(seg, h->moment_stride - phase);
            }
            if (q.has_forced) seg = 1;   // the forced step is a launch of its own (FORCED instantiation of the fused kernel)
            q.nsteps = seg;
            q.step0 = h->total_steps;
            hipError_t e;
            if (!exact) {
                q.Uperm = h->d_Uop;   // fused order: the proposal on the matrix pipe
                e = launch_panel_mfma(q, h->likelihood, h->stream);
            } else {
                q.Uperm = h->d_U;
                e = launch_panel_exact(h, q);
            }
            if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("panel kernel launch: ") + hipGetErrorString(e));
            h->total_steps += (uint32_t)seg;
            done += seg;
            q.has_forced = 0;
        }
        h->has_forced = false;
        return SMCMC_OK;
    }
    const bool moments = (h->mode == SMCMC_MODE_POOLED);
    const bool fullu = h->prop->decompFull;
    const bool exact = h->exact || fullu;   // the full (eigen) decomposition only exists in reference order
    if ((p.scan_dim >= 0 || p.uniform_mask != 0) && !exact)
        return fail(h, SMCMC_ERR_UNSUPPORTED,
                    "uniform proposals and the scan of a dimension run in reference-order arithmetic only");
    hipError_t e = dispatch_step(h->dp, p, h->likelihood, exact, fullu, moments, h->stream);
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("step kernel launch: ") + hipGetErrorString(e));
    h->total_steps += (uint32_t)nsteps;
    h->has_forced = false;
    return SMCMC_OK;
}

}  // namespace

extern "C" {

int smcmc_version(void) { return 100; }
int smcmc_max_register_dim(void) { return kDPList[kNumDP - 1]; }
int smcmc_max_dim(void) { return 8 * kPanelCW; }

const char* smcmc_status_string(int status) {
    switch (status) {
        case SMCMC_OK: return "ok";
        case SMCMC_ERR_INVALID: return "invalid argument";
        case SMCMC_ERR_LOGIC: return "logic error";
        case SMCMC_ERR_RUNTIME: return "runtime error";
        case SMCMC_ERR_BAD_START: return "bad starting point";
        case SMCMC_ERR_UNSUPPORTED: return "unsupported on the HIP path";
        case SMCMC_ERR_HIP: return "HIP runtime error";
        case SMCMC_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

const char* smcmc_last_error(const smcmc_engine* h) { return h ? h->error.c_str() : "null engine"; }

int smcmc_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset, int device,
                 smcmc_engine** out) {
    if (!out) return SMCMC_ERR_INVALID;
    *out = nullptr;
    if (dim < 1 || nchains < 1) return SMCMC_ERR_INVALID;
    if (likelihood < SMCMC_LIKE_ISO_GAUSS || likelihood > SMCMC_LIKE_CONSTRAINED) return SMCMC_ERR_INVALID;
#ifndef SMCMC_USER_LIKELIHOOD
    if (likelihood == SMCMC_LIKE_USER) return SMCMC_ERR_UNSUPPORTED;   // this build carries no user likelihood
#endif
    if (likelihood == SMCMC_LIKE_ROSENBROCK && dim < 2) return SMCMC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return SMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return SMCMC_ERR_NO_DEVICE;
    int dp = pick_dp(dim, likelihood);
    int panel_w = 0;
    if (dp < 0) {
        // large dimensions: a workgroup of 4 or 8 wavefronts per 64-chain group (smcmc_panel_kernel.hip.h)
        if (dim <= 4 * kPanelCW) panel_w = 4;
        else if (dim <= 8 * kPanelCW) panel_w = 8;
        else return SMCMC_ERR_UNSUPPORTED;
#ifndef SMCMC_USER_LIKELIHOOD_ANY_DIM
        if (likelihood == SMCMC_LIKE_USER) return SMCMC_ERR_UNSUPPORTED;   // the user's header serves dim <= 63 only
#endif
        dp = dim;
    }
    smcmc_engine* h = new (std::nothrow) smcmc_engine();
    if (!h) return SMCMC_ERR_RUNTIME;
    h->dim = dim; h->nchains = nchains; h->likelihood = likelihood; h->seed = seed;
    h->chain_offset = chain_offset; h->device = device; h->dp = dp; h->panel_w = panel_w;
    h->npad = (nchains + kWave - 1) / kWave * kWave;
    h->ngroups = h->npad / kWave;
    h->nt = panel_w ? 1 : tiles_for(dp);
    h->fold_nslices = fold_slices(dim);
    h->slice_chains = ((h->ngroups + h->fold_nslices - 1) / h->fold_nslices) * kWave;
    h->prop = new SharedProposal(dim);
    *out = h;
    ON_DEVICE(h);
    if (panel_w) HIP_TRY(h, smcmc::fold_ring_prepare(h->fold, dim, nchains, h->npad, h->fold_nslices, h->slice_chains));
    const size_t np = (size_t)h->npad;
    const size_t u_doubles = panel_w ? (size_t)panel_w * dim * kPanelCW : (size_t)dp * dp;
    HIP_TRY(h, hipMalloc(&h->d_x, sizeof(double) * np * dp));        // rows >= dim stay zero
    HIP_TRY(h, hipMalloc(&h->d_forced, sizeof(double) * np * dp));
    HIP_TRY(h, hipMalloc(&h->d_lane_f64, sizeof(double) * np * SMCMC_LANE_F64_COUNT_));
    HIP_TRY(h, hipMalloc(&h->d_lane_i32, sizeof(int32_t) * np * SMCMC_LANE_I32_COUNT_));
    HIP_TRY(h, hipMalloc(&h->d_U, sizeof(double) * u_doubles));
    HIP_TRY(h, hipMemset(h->d_U, 0, sizeof(double) * u_doubles));
    if (panel_w) {
        HIP_TRY(h, hipMalloc(&h->d_Uop, sizeof(double) * panel_mfma_uop_doubles(dim)));
        HIP_TRY(h, hipMemset(h->d_Uop, 0, sizeof(double) * panel_mfma_uop_doubles(dim)));
    }
    const size_t like_doubles = std::max((size_t)dp * dp, panel_w ? panel_mfma_uop_doubles(dim) : (size_t)0);   // >= 2 + 2 dim
    HIP_TRY(h, hipMalloc(&h->d_like, sizeof(double) * like_doubles));
    HIP_TRY(h, hipMalloc(&h->d_c0, sizeof(double) * dp));
    HIP_TRY(h, hipMalloc(&h->d_gacc, sizeof(double) * gacc_doubles(h)));
    HIP_TRY(h, hipMalloc(&h->d_moments, sizeof(double) * npacked(h)));
    HIP_TRY(h, hipHostMalloc((void**)&h->h_moments, sizeof(double) * npacked(h), hipHostMallocDefault));
    HIP_TRY(h, hipMalloc(&h->d_chunks, sizeof(double) * npacked(h) * ((h->ngroups + kReduceChunk - 1) / kReduceChunk)));
    HIP_TRY(h, hipMemset(h->d_x, 0, sizeof(double) * np * dp));
    HIP_TRY(h, hipMemset(h->d_forced, 0, sizeof(double) * np * dp));
    HIP_TRY(h, hipMalloc(&h->d_uniform, sizeof(double) * (2 * dp + 8)));
    HIP_TRY(h, hipMemset(h->d_uniform, 0, sizeof(double) * (2 * dp + 8)));
    HIP_TRY(h, hipMemset(h->d_lane_f64, 0, sizeof(double) * np * SMCMC_LANE_F64_COUNT_));
    HIP_TRY(h, hipMemset(h->d_lane_i32, 0, sizeof(int32_t) * np * SMCMC_LANE_I32_COUNT_));
    HIP_TRY(h, hipMemset(h->d_like, 0, sizeof(double) * like_doubles));
    HIP_TRY(h, hipMemset(h->d_c0, 0, sizeof(double) * dp));
    HIP_TRY(h, hipMemset(h->d_gacc, 0, sizeof(double) * gacc_doubles(h)));
    HIP_TRY(h, hipMemset(h->d_moments, 0, sizeof(double) * npacked(h)));
    HIP_TRY(h, hipMalloc(&h->d_centre, sizeof(double) * dim));
    HIP_TRY(h, hipMalloc(&h->d_cov, sizeof(double) * (size_t)dim * dim));
    HIP_TRY(h, hipMalloc(&h->d_decomp, sizeof(double) * (size_t)dim * dim));
    HIP_TRY(h, hipMalloc(&h->d_scal, sizeof(double) * kPsCount));
    HIP_TRY(h, hipMemset(h->d_scal, 0, sizeof(double) * kPsCount));
    HIP_TRY(h, hipHostMalloc((void**)&h->h_scal, sizeof(double) * kPsCount, hipHostMallocDefault));
    if (hipHostGetDevicePointer((void**)&h->h_scal_dev, h->h_scal, 0) != hipSuccess) {
        (void)hipGetLastError();
        h->h_scal_dev = nullptr;
    }
    HIP_TRY(h, hipEventCreateWithFlags(&h->status_event, hipEventDisableTiming));
    return SMCMC_OK;
}

int smcmc_destroy(smcmc_engine* h) {
    if (!h) return SMCMC_OK;
    (void)smcmc_comm_destroy(h);
    ON_DEVICE(h);
    if (h->d_x) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_x); (void)hipFree(h->d_forced); (void)hipFree(h->d_proposed); (void)hipFree(h->d_scratch); (void)hipFree(h->d_uniform); (void)hipFree(h->d_lane_f64); (void)hipFree(h->d_lane_i32);
    (void)hipFree(h->d_U); (void)hipFree(h->d_Uop); (void)hipFree(h->d_like); (void)hipFree(h->d_like_rowptr); (void)hipFree(h->d_like_cols); (void)hipFree(h->d_like_vals); (void)hipFree(h->d_like_rows); (void)hipFree(h->d_c0); (void)hipFree(h->d_gacc);
    (void)hipFree(h->d_moments); (void)hipFree(h->d_chunks); (void)hipHostFree(h->h_moments);
    (void)hipFree(h->d_centre); (void)hipFree(h->d_cov); (void)hipFree(h->d_decomp); (void)hipFree(h->d_scal);
    (void)hipFree(h->d_ring); (void)hipFree(h->d_ring_logl);
    (void)hipFree(h->d_pc_cov); (void)hipFree(h->d_pc_ut); (void)hipFree(h->d_pc_centre); (void)hipFree(h->d_pc_last);
    (void)hipFree(h->d_pc_tmpl); (void)hipFree(h->d_pc_flag); (void)hipFree(h->d_pc_stage); (void)hipFree(h->d_pc_stage_chains);
    (void)hipHostFree(h->h_scal);
    if (h->status_event) (void)hipEventDestroy(h->status_event);
    delete h->prop;
    delete h;
    return SMCMC_OK;
}

int smcmc_set_stream(smcmc_engine* h, void* hip_stream) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    // a pooled update (and the copy of its status block) may still be queued on the old stream: resolve it and drain
    // the stream before anything is launched on the new one
    { int sst_ = check_pending(h); if (sst_) return sst_; }
    if (h->d_x) HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)hip_stream;
    return SMCMC_OK;
}

int smcmc_set_likelihood_params(smcmc_engine* h, const double* params, int count) {
    if (!h || count < 0 || (count > 0 && !params)) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    h->like_params.assign(params, params + count);
    if (h->started) return upload_like(h);
    return SMCMC_OK;
}

int smcmc_set_mode(smcmc_engine* h, int mode) {
    if (!h || (mode != SMCMC_MODE_FROZEN && mode != SMCMC_MODE_POOLED && mode != SMCMC_MODE_PER_CHAIN)) return SMCMC_ERR_INVALID;
    if (h->started && mode != h->mode && (mode == SMCMC_MODE_PER_CHAIN || h->mode == SMCMC_MODE_PER_CHAIN))
        return fail(h, SMCMC_ERR_LOGIC, "SMCMC_MODE_PER_CHAIN is chosen before Start");
    if (mode == SMCMC_MODE_PER_CHAIN && (h->panel_w || h->dim > kPcMaxDim))
        return fail(h, SMCMC_ERR_UNSUPPORTED, "SMCMC_MODE_PER_CHAIN serves dim <= 63");
    h->mode = mode;
    h->prop->covFrozen = (mode == SMCMC_MODE_FROZEN);
    return SMCMC_OK;
}

int smcmc_set_gaussian(smcmc_engine* h, int dim, double sigma) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    if (dim < 0 || dim >= h->dim) return fail(h, SMCMC_ERR_INVALID, "Dimension is out of range.");   // :856-860
    const bool was_uniform = h->prop->ptype[dim] == 1;
    h->prop->ptype[dim] = 0;
    h->prop->param1[dim] = sigma * sigma;
    return (h->started && was_uniform) ? upload_shared(h) : SMCMC_OK;
}

int smcmc_set_uniform(smcmc_engine* h, int dim, double minimum, double maximum) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    if (dim < 0 || dim >= h->dim) return fail(h, SMCMC_ERR_INVALID, "Dimension is out of range.");
    h->prop->ptype[dim] = 1;                                                                        // :845-847
    h->prop->param1[dim] = minimum;
    h->prop->param2[dim] = maximum;
    return h->started ? upload_shared(h) : SMCMC_OK;
}

int smcmc_set_scan_dimension(smcmc_engine* h, int dim) {
    if (!h) return SMCMC_ERR_INVALID;
    h->scan_dim = (dim < 0 || dim >= h->dim) ? -1 : dim;                                            // :827-829
    return SMCMC_OK;
}

int smcmc_set_correlation(smcmc_engine* h, int d1, int d2, double c) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    if (d1 < 0 || d2 < 0 || d1 >= h->dim || d2 >= h->dim) return fail(h, SMCMC_ERR_INVALID, "Dimension is out of range.");
    if (d1 == d2) return fail(h, SMCMC_ERR_INVALID, "Dimensions must be different for correlations");   // :884-890
    const double mc = h->prop->maxCorrelation;
    if (c < -mc) c = -mc;                                                                            // :891-902
    if (c > mc) c = mc;
    h->prop->correlations.push_back({d1, d2, c});
    return SMCMC_OK;
}

int smcmc_reset_correlations(smcmc_engine* h) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    h->prop->correlations.clear();
    return SMCMC_OK;
}

static int broadcast_lane_f64(smcmc_engine* h, int field, double v) {
    std::vector<double> col(h->npad, 0.0);
    for (int c = 0; c < h->nchains; ++c) col[c] = v;
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_f64 + (size_t)field * h->npad, col.data(), col.size() * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

static int broadcast_lane_i32(smcmc_engine* h, int field, int32_t v) {
    std::vector<int32_t> col(h->npad, 0);
    for (int c = 0; c < h->nchains; ++c) col[c] = v;
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_i32 + (size_t)field * h->npad, col.data(), col.size() * sizeof(int32_t),
                              hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int smcmc_set_param(smcmc_engine* h, int which, double v) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (which == SMCMC_P_SIGMA || which == SMCMC_P_COVARIANCE_TRIALS || which == SMCMC_P_CENTER_TRIALS ||
        which == SMCMC_P_NEXT_UPDATE || (which == SMCMC_P_EXACT_ARITHMETIC && h->started)) {
        int sst_ = sync_shared_to_host(h, true);   // the rest are settings the host alone keeps
        if (sst_) return sst_;
    }
    SharedProposal& P = *h->prop;
    switch (which) {
        case SMCMC_P_COVARIANCE_WINDOW: P.covWindow = v; return SMCMC_OK;
        case SMCMC_P_COVARIANCE_DEWEIGHT: P.covDeweight = v; return SMCMC_OK;
        case SMCMC_P_ACCEPTANCE_WINDOW: P.acceptanceWindow = v; return SMCMC_OK;
        case SMCMC_P_ACCEPTANCE_DEWEIGHT: P.acceptanceDeweight = v; return SMCMC_OK;
        case SMCMC_P_ACCEPTANCE_RIGIDITY:
            P.rigidity = v;
            return h->started ? broadcast_lane_f64(h, SMCMC_LANE_RIGIDITY, v) : SMCMC_OK;
        case SMCMC_P_TARGET_ACCEPTANCE: P.target = v; return SMCMC_OK;
        case SMCMC_P_SIGMA:
            P.sigma = v;
            return h->started ? broadcast_lane_f64(h, SMCMC_LANE_SIGMA, v) : SMCMC_OK;
        case SMCMC_P_MAXIMUM_CORRELATION: P.maxCorrelation = v; return SMCMC_OK;
        case SMCMC_P_STEP_RMS_WINDOW: h->step_rms_window = (int)v; return SMCMC_OK;
        case SMCMC_P_NEXT_UPDATE:
            P.nextUpdate = (int)v;
            return h->started ? broadcast_lane_i32(h, SMCMC_LANE_NEXT_UPDATE, (int32_t)v) : SMCMC_OK;
        case SMCMC_P_COVARIANCE_TRIALS:
            P.covTrials = v;
            return (per_chain(h) && h->started) ? broadcast_lane_f64(h, SMCMC_LANE_COVARIANCE_TRIALS, v) : SMCMC_OK;
        case SMCMC_P_CENTER_TRIALS:
            P.centreTrials = v;
            return (per_chain(h) && h->started) ? broadcast_lane_f64(h, SMCMC_LANE_CENTER_TRIALS, v) : SMCMC_OK;
        case SMCMC_P_COVARIANCE_FROZEN: h->pc_frozen = (v != 0.0); return SMCMC_OK;
        case SMCMC_P_PERCHAIN_WAVE: h->pc_wave = (v < 0.0) ? -1 : (v != 0.0 ? 1 : 0); return SMCMC_OK;
        case SMCMC_P_DENSE_QUADFORM:
            h->dense_quadform = (v != 0.0);
            return (h->started && h->likelihood == SMCMC_LIKE_QUADFORM) ? upload_like_csr(h) : SMCMC_OK;
        case SMCMC_P_EXACT_ARITHMETIC:
            h->exact = (v != 0.0);
            if (h->started) {                                  // the fused order keeps its own operand images
                int st = upload_like(h);
                if (st) return st;
                return upload_shared(h);
            }
            return SMCMC_OK;
        case SMCMC_P_KEEP_PROPOSED:
            if (v != 0.0 && !h->d_proposed) {
                const size_t bytes = sizeof(double) * (size_t)h->npad * h->dp;
                HIP_TRY(h, hipMalloc(&h->d_proposed, bytes));
                // before the first step the proposed point is the start point (TSimpleMCMC.H:250-253)
                HIP_TRY(h, hipMemcpyAsync(h->d_proposed, h->d_x, bytes, hipMemcpyDeviceToDevice, h->stream));
            }
            h->keep_proposed = (v != 0.0);
            return SMCMC_OK;
        case SMCMC_P_DEVICE_UPDATE: h->device_update = (v != 0.0); return SMCMC_OK;
        case SMCMC_P_OVERLAP_UPDATE: h->overlap_update = (v != 0.0); return SMCMC_OK;
        case SMCMC_P_MOMENT_STRIDE:
            if (v < 1.0) return fail(h, SMCMC_ERR_INVALID, "moment stride must be >= 1");
            if (!h->panel_w && v != 1.0)
                return fail(h, SMCMC_ERR_UNSUPPORTED, "the register-resident kernels fold every step");
            h->moment_stride = (int)v;
            return SMCMC_OK;
        default: return fail(h, SMCMC_ERR_INVALID, "parameter is read only or unknown");
    }
}

int smcmc_get_param(smcmc_engine* h, int which, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    // only what a pooled update changes needs the device's copy (a stream synchronisation and a download)
    if (which == SMCMC_P_COVARIANCE_TRIALS || which == SMCMC_P_CENTER_TRIALS || which == SMCMC_P_COVARIANCE_TRACE ||
        which == SMCMC_P_SIGMA_TRACE) {
        int sst_ = sync_shared_to_host(h, false);
        if (sst_) return sst_;
    } else if (which == SMCMC_P_UPDATE_COUNT || which == SMCMC_P_LAST_UPDATE_PATH || which == SMCMC_P_NEXT_UPDATE ||
               which == SMCMC_P_SIGMA) {
        int sst_ = check_pending(h);
        if (sst_) return sst_;
    }
    const SharedProposal& P = *h->prop;
    if (per_chain(h) && h->started) {
        // the members every chain keeps for itself: chain 0 answers
        const size_t NP = (size_t)h->npad;
        int lane_f = -1, lane_i = -1;
        switch (which) {
            case SMCMC_P_COVARIANCE_TRIALS: lane_f = SMCMC_LANE_COVARIANCE_TRIALS; break;
            case SMCMC_P_CENTER_TRIALS: lane_f = SMCMC_LANE_CENTER_TRIALS; break;
            case SMCMC_P_SIGMA_TRACE: lane_f = SMCMC_LANE_SIGMA_TRACE; break;
            case SMCMC_P_NEXT_UPDATE: lane_i = SMCMC_LANE_NEXT_UPDATE; break;
            case SMCMC_P_UPDATE_COUNT: lane_i = SMCMC_LANE_UPDATE_COUNT; break;
            case SMCMC_P_LAST_UPDATE_PATH: lane_i = SMCMC_LANE_LAST_UPDATE_PATH; break;
            default: break;
        }
        if (lane_f >= 0) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipMemcpy(out, h->d_lane_f64 + (size_t)lane_f * NP, sizeof(double), hipMemcpyDeviceToHost));
            return SMCMC_OK;
        }
        if (lane_i >= 0) {
            int32_t v = 0;
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipMemcpy(&v, h->d_lane_i32 + (size_t)lane_i * NP, sizeof(int32_t), hipMemcpyDeviceToHost));
            *out = v;
            return SMCMC_OK;
        }
        if (which == SMCMC_P_COVARIANCE_TRACE) {
            std::vector<double> cov((size_t)h->dim * h->dim);
            int st = smcmc_read_chain_proposal(h, 0, nullptr, cov.data(), nullptr);
            if (st) return st;
            double t = 0.0;
            for (int i = 0; i < h->dim; ++i) t += cov[(size_t)i * h->dim + i];
            *out = t;
            return SMCMC_OK;
        }
    }
    switch (which) {
        case SMCMC_P_COVARIANCE_WINDOW: *out = P.covWindow; break;
        case SMCMC_P_COVARIANCE_DEWEIGHT: *out = P.covDeweight; break;
        case SMCMC_P_ACCEPTANCE_WINDOW: *out = P.acceptanceWindow; break;
        case SMCMC_P_ACCEPTANCE_DEWEIGHT: *out = P.acceptanceDeweight; break;
        case SMCMC_P_ACCEPTANCE_RIGIDITY: *out = P.rigidity; break;
        case SMCMC_P_TARGET_ACCEPTANCE: *out = P.target; break;
        case SMCMC_P_SIGMA: {
            if (!h->started) { *out = P.sigma; break; }
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipMemcpy(out, h->d_lane_f64 + (size_t)SMCMC_LANE_SIGMA * h->npad, sizeof(double),
                                 hipMemcpyDeviceToHost));
            break;
        }
        case SMCMC_P_MAXIMUM_CORRELATION: *out = P.maxCorrelation; break;
        case SMCMC_P_STEP_RMS_WINDOW: *out = h->step_rms_window; break;
        case SMCMC_P_NEXT_UPDATE: *out = P.nextUpdate; break;
        case SMCMC_P_COVARIANCE_TRIALS: *out = P.covTrials; break;
        case SMCMC_P_CENTER_TRIALS: *out = P.centreTrials; break;
        case SMCMC_P_COVARIANCE_TRACE: *out = P.trace(); break;
        case SMCMC_P_TOTAL_STEPS: *out = h->total_steps; break;
        case SMCMC_P_SIGMA_TRACE: *out = P.sigmaTrace; break;
        case SMCMC_P_UPDATE_COUNT: *out = P.updateCount; break;
        case SMCMC_P_LAST_UPDATE_PATH: *out = P.lastPath; break;
        case SMCMC_P_EXACT_ARITHMETIC: *out = h->exact ? 1.0 : 0.0; break;
        case SMCMC_P_MOMENT_STRIDE: *out = h->moment_stride; break;
        case SMCMC_P_MOMENT_GROUP: *out = h->panel_w ? h->slice_chains : kWave; break;
        case SMCMC_P_KEEP_PROPOSED: *out = h->keep_proposed ? 1.0 : 0.0; break;
        case SMCMC_P_DEVICE_UPDATE: *out = h->device_update ? 1.0 : 0.0; break;
        case SMCMC_P_OVERLAP_UPDATE: *out = h->overlap_update ? 1.0 : 0.0; break;
        case SMCMC_P_COVARIANCE_FROZEN: *out = (h->pc_frozen || h->mode == SMCMC_MODE_FROZEN) ? 1.0 : 0.0; break;
        case SMCMC_P_DENSE_QUADFORM: *out = (h->dense_quadform || h->d_like_rowptr == nullptr) ? 1.0 : 0.0; break;
        case SMCMC_P_PERCHAIN_WAVE: *out = (per_chain(h) && h->dim <= kPcMaxDim && pc_use_wave(h)) ? 1.0 : 0.0; break;
        default: return fail(h, SMCMC_ERR_INVALID, "unknown parameter");
    }
    return SMCMC_OK;
}

// Puts every chain at x0 and evaluates the likelihood there on the device (the
// GetLogLikelihoodValue call of Start, TSimpleMCMC.H:258, and of Restore, :335).  All per-chain
// columns are zero afterwards except the likelihood; x and logl come back for the caller.
static int place_chains(smcmc_engine* h, const double* x0, int broadcast, std::vector<double>& x,
                        std::vector<double>& logl) {
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    int st = upload_like(h);
    if (st) return st;
    x.assign(NP * h->dp, 0.0);
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < N; ++c) x[(size_t)d * NP + c] = broadcast ? x0[d] : x0[(size_t)d * N + c];
    HIP_TRY(h, hipMemcpyAsync(h->d_x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->d_proposed)   // fProposed = start (TSimpleMCMC.H:250-253, 326-329)
        HIP_TRY(h, hipMemcpyAsync(h->d_proposed, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));

    // Start's likelihood call (TSimpleMCMC.H:258): a scan step (metropolis == 2) that is
    // forced to the start point evaluates and stores logL(start) for every chain
    // without touching the proposal state.
    HIP_TRY(h, hipMemcpyAsync(h->d_forced, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_lane_f64, 0, sizeof(double) * NP * SMCMC_LANE_F64_COUNT_, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_lane_i32, 0, sizeof(int32_t) * NP * SMCMC_LANE_I32_COUNT_, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    if (h->panel_w && h->likelihood == SMCMC_LIKE_QUADFORM && !h->exact) {
        PanelParams q;
        std::memset(&q, 0, sizeof(q));
        q.nchains = N; q.npad = h->npad; q.dim = D; q.init_only = 1;
        q.like = h->d_like; q.x = h->d_x; q.lane_f64 = h->d_lane_f64; q.lane_i32 = h->d_lane_i32;
        q.save_stride = 1;
        hipError_t e = launch_panel_mfma(q, h->likelihood, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("start kernel launch: ") + hipGetErrorString(e));
    } else if (h->panel_w || h->mode == SMCMC_MODE_PER_CHAIN) {
        hipError_t e;
#ifdef SMCMC_USER_LIKELIHOOD
        if (h->likelihood == SMCMC_LIKE_USER)
            e = launch_start_loglike_user(h->d_x, N, NP, D, h->d_like, h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, h->stream);
        else
#endif
        e = launch_start_loglike(h->d_x, N, NP, D, h->d_like, h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP,
                                 h->likelihood, h->exact, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("start kernel launch: ") + hipGetErrorString(e));
    } else {
        StepParams p;
        std::memset(&p, 0, sizeof(p));
        p.nchains = N; p.npad = h->npad; p.dim = D; p.nsteps = 1; p.metropolis = 2;
        p.seed = h->seed; p.chain_offset = h->chain_offset;
        p.U = h->d_U; p.like = h->d_like; p.c0 = h->d_c0;
        p.target = 0.234; p.acc_window = 1.0; p.asig = 1.0; p.max_up = 1.0; p.acc_w = -1.0;
        p.step_rms_window = 0;
        p.has_forced = 1; p.forced = h->d_forced;
        p.x = h->d_x; p.lane_f64 = h->d_lane_f64; p.lane_i32 = h->d_lane_i32; p.gacc = h->d_gacc;
        p.save_stride = 1;
        p.scan_dim = -1;
        hipError_t e = dispatch_step(h->dp, p, h->likelihood, h->exact, false, false, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("start kernel launch: ") + hipGetErrorString(e));
    }
    logl.assign(NP, 0.0);
    HIP_TRY(h, hipMemcpyAsync(logl.data(), h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, NP * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int smcmc_start(smcmc_engine* h, const double* x0, int broadcast) {
    if (!h || !x0) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    std::vector<double> x, logl;
    int st = SMCMC_OK;
    if (per_chain(h)) {
        st = pc_check_supported(h);
        if (st) return st;
        st = pc_alloc(h);
        if (st) return st;
    }
    st = place_chains(h, x0, broadcast, x, logl);
    if (st) return st;
    for (int c = 0; c < N; ++c)
        if (!std::isfinite(logl[c]) || logl[c] < -0.999999E+10)                      // :265-268
            return fail(h, SMCMC_ERR_BAD_START, "start likelihood is not finite or < -0.999999E+10");

    // InitializeState on chain 0's start (TSimpleMCMC.H:272, 1679-1714)
    std::vector<double> p0(D);
    for (int d = 0; d < D; ++d) p0[d] = x[(size_t)d * NP];
    SharedProposal& P = *h->prop;
    P.covFrozen = (h->mode == SMCMC_MODE_FROZEN);
    st = status_of(h, P.initialize(p0.data()));
    if (st) return st;

    std::vector<double> lf(NP * SMCMC_LANE_F64_COUNT_, 0.0);
    std::vector<int32_t> li(NP * SMCMC_LANE_I32_COUNT_, 0);
    for (int c = 0; c < N; ++c) {
        lf[(size_t)SMCMC_LANE_LOGL * NP + c] = logl[c];
        lf[(size_t)SMCMC_LANE_SIGMA * NP + c] = P.sigma;
        lf[(size_t)SMCMC_LANE_ACCEPTANCE * NP + c] = P.acceptance;
        lf[(size_t)SMCMC_LANE_ACCEPTANCE_TRIALS * NP + c] = P.acceptanceTrials;
        lf[(size_t)SMCMC_LANE_RIGIDITY * NP + c] = P.rigidity;
        lf[(size_t)SMCMC_LANE_LAST_VALUE * NP + c] = logl[c];
        lf[(size_t)SMCMC_LANE_LAST_X0 * NP + c] = x[c];
        lf[(size_t)SMCMC_LANE_LOGL_PROPOSED * NP + c] = logl[c];
        li[(size_t)SMCMC_LANE_NEXT_UPDATE * NP + c] = P.nextUpdate;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_f64, lf.data(), lf.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_i32, li.data(), li.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    st = upload_shared(h);
    if (st) return st;
    if (per_chain(h)) {
        // InitializeState (:1679-1714) is the same computation for every chain but for the point it is centred on
        st = pc_broadcast(h, false, nullptr);
        if (st) return st;
    }
    h->total_steps = 0;
    h->has_forced = false;
    h->started = true;
    h->pc_broken = false;
    h->snap_valid = false;
    return SMCMC_OK;
}

// Restore (TSimpleMCMC.H:282-352, randomize = false) followed by
// TProposeAdaptiveStep::RestoreState (:1501-1612): every chain resumes from `accepted` with the
// saved scalar state of the tree entry; the shared proposal takes the saved centre and covariance
// and is updated once (:1612).
int smcmc_restore(smcmc_engine* h, const double* accepted, int broadcast, const smcmc_saved_state* s) {
    if (!h || !accepted || !s || !s->central_point || !s->covariance) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    if (!h->started)
        return fail(h, SMCMC_ERR_INVALID, "Restore needs a started chain (Start first, SimpleMCMC.C:151-154)");
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    std::vector<double> x, logl;
    int st = place_chains(h, accepted, broadcast, x, logl);
    if (st) return st;
    // the saved likelihood stands unless the recomputed one differs by more than 1E-4 (:336-345)
    for (int c = 0; c < N; ++c)
        if (!(std::fabs(logl[c] - s->log_likelihood) > 1E-4)) logl[c] = s->log_likelihood;

    SharedProposal& P = *h->prop;
    P.initialized = true;                                                            // :1503
    for (int d = 0; d < D; ++d) P.lastPoint[d] = x[(size_t)d * NP];                  // chain 0, :1515
    P.successes = s->successes;                                                      // :1563-1570
    P.nextUpdate = s->next_update;
    P.acceptance = s->acceptance;
    P.acceptanceTrials = s->acceptance_trials;
    P.sigma = s->sigma;
    for (int d = 0; d < D; ++d) P.centre[d] = s->central_point[d];
    P.centreTrials = s->central_point_trials;
    const double* cov = s->covariance;                                               // :1574-1586
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < i + 1; ++j) P.C(i, j) = P.C(j, i) = *cov++;
    P.sigmaTrace = P.trace();                                                        // :1587
    P.covTrials = s->covariance_trials;
    st = status_of(h, P.update(false));                                              // :1612
    if (st) return st;

    std::vector<double> lf(NP * SMCMC_LANE_F64_COUNT_, 0.0);
    std::vector<int32_t> li(NP * SMCMC_LANE_I32_COUNT_, 0);
    for (int c = 0; c < N; ++c) {
        lf[(size_t)SMCMC_LANE_LOGL * NP + c] = logl[c];
        lf[(size_t)SMCMC_LANE_SIGMA * NP + c] = P.sigma;
        lf[(size_t)SMCMC_LANE_ACCEPTANCE * NP + c] = P.acceptance;
        lf[(size_t)SMCMC_LANE_ACCEPTANCE_TRIALS * NP + c] = P.acceptanceTrials;
        lf[(size_t)SMCMC_LANE_RIGIDITY * NP + c] = P.rigidity;
        lf[(size_t)SMCMC_LANE_LAST_VALUE * NP + c] = logl[c];
        lf[(size_t)SMCMC_LANE_LAST_X0 * NP + c] = x[c];
        lf[(size_t)SMCMC_LANE_STEP_RMS * NP + c] = s->step_rms;
        lf[(size_t)SMCMC_LANE_LOGL_PROPOSED * NP + c] = logl[c];
        li[(size_t)SMCMC_LANE_TRIALS * NP + c] = s->trials;
        li[(size_t)SMCMC_LANE_SUCCESSES * NP + c] = s->successes;
        li[(size_t)SMCMC_LANE_NEXT_UPDATE * NP + c] = P.nextUpdate;
        li[(size_t)SMCMC_LANE_CHAIN_STEPS * NP + c] = s->total_steps;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_f64, lf.data(), lf.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_lane_i32, li.data(), li.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->total_steps = (uint32_t)s->total_steps;
    h->has_forced = false;
    st = upload_shared(h);
    if (st) return st;
    if (per_chain(h)) return pc_broadcast(h, false, s->central_point);   // RestoreState for every chain (:1501-1612)
    return SMCMC_OK;
}

int smcmc_step(smcmc_engine* h, int nsteps, int metropolis) {
    return launch(h, nsteps, metropolis, 1, nullptr, nullptr);
}

int smcmc_step_save(smcmc_engine* h, int nsteps, int metropolis, int stride, double* save_x, double* save_logl) {
    if (!save_x || !save_logl) return fail(h, SMCMC_ERR_INVALID, "save buffers must be device pointers");
    return launch(h, nsteps, metropolis, stride, save_x, save_logl);
}

namespace {
// the arrays that make up the state of a per-chain ensemble, and their sizes in bytes
int snap_arrays(smcmc_engine* h, void* (&arr)[8], size_t (&bytes)[8]) {
    const size_t NP = (size_t)h->npad, D = (size_t)h->dim, npk = D * (D + 1) / 2;
    const size_t pad = (size_t)smcmc::kPcPad * smcmc::kWave;
    arr[0] = h->d_x; bytes[0] = sizeof(double) * NP * h->dp;
    arr[1] = h->d_lane_f64; bytes[1] = sizeof(double) * NP * SMCMC_LANE_F64_COUNT_;
    arr[2] = h->d_lane_i32; bytes[2] = sizeof(int32_t) * NP * SMCMC_LANE_I32_COUNT_;
    arr[3] = h->d_proposed; bytes[3] = sizeof(double) * NP * h->dp;
    arr[4] = h->d_pc_cov; bytes[4] = sizeof(double) * (npk * NP + pad);
    arr[5] = h->d_pc_ut; bytes[5] = sizeof(double) * (D * D * NP + pad);
    arr[6] = h->d_pc_centre; bytes[6] = sizeof(double) * D * NP;
    arr[7] = h->d_pc_last; bytes[7] = sizeof(double) * D * NP;
    for (int k = 0; k < 8; ++k)
        if (!arr[k]) return fail(h, SMCMC_ERR_LOGIC, "the ensemble has not been started in SMCMC_MODE_PER_CHAIN");
    return SMCMC_OK;
}
}  // namespace

int smcmc_snapshot(smcmc_engine* h) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (!per_chain(h) || !h->started) return fail(h, SMCMC_ERR_UNSUPPORTED, "smcmc_snapshot serves a started SMCMC_MODE_PER_CHAIN ensemble");
    void* arr[8]; size_t bytes[8];
    int st = snap_arrays(h, arr, bytes);
    if (st) return st;
    for (int k = 0; k < 8; ++k) {
        if (!h->snap[k]) HIP_TRY(h, hipMalloc(&h->snap[k], bytes[k]));
        HIP_TRY(h, hipMemcpyAsync(h->snap[k], arr[k], bytes[k], hipMemcpyDeviceToDevice, h->stream));
    }
    h->snap_total_steps = h->total_steps;
    h->snap_has_forced = h->has_forced;
    h->snap_valid = true;
    return SMCMC_OK;
}

int smcmc_rollback(smcmc_engine* h) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (!h->snap_valid || !per_chain(h)) return fail(h, SMCMC_ERR_LOGIC, "no snapshot to return to");
    void* arr[8]; size_t bytes[8];
    int st = snap_arrays(h, arr, bytes);
    if (st) return st;
    for (int k = 0; k < 8; ++k)
        HIP_TRY(h, hipMemcpyAsync(arr[k], h->snap[k], bytes[k], hipMemcpyDeviceToDevice, h->stream));
    h->total_steps = h->snap_total_steps;
    h->has_forced = h->snap_has_forced;
    return SMCMC_OK;
}

int smcmc_record_stride(const smcmc_engine* h) { return h ? 3 * h->dim + smcmc::kPcRecScalars : 0; }

int smcmc_step_recorded(smcmc_engine* h, int nsteps, int metropolis, int chain, double* records) {
    if (!h || !records) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (nsteps <= 0) return SMCMC_OK;
    if (!per_chain(h)) return fail(h, SMCMC_ERR_UNSUPPORTED, "smcmc_step_recorded serves SMCMC_MODE_PER_CHAIN");
    if (chain < 0 || chain >= h->nchains) return fail(h, SMCMC_ERR_INVALID, "no such chain");
    const int stride = smcmc_record_stride(h);
    const size_t need = (size_t)nsteps * stride;
    if (need > h->pc_rec_cap) {
        (void)hipFree(h->d_pc_rec);
        h->d_pc_rec = nullptr; h->pc_rec_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_pc_rec, need * sizeof(double)));
        h->pc_rec_cap = need;
    }
    h->pc_rec = smcmc::PerChainRecord{h->d_pc_rec, chain, stride};
    const int st = launch(h, nsteps, metropolis, 1, nullptr, nullptr);
    h->pc_rec = smcmc::PerChainRecord{nullptr, 0, 0};
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(records, h->d_pc_rec, need * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int smcmc_force_step(smcmc_engine* h, const double* point, int broadcast) {
    if (!h || !point) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = check_pending(h); if (sst_) return sst_; }
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    std::vector<double> x(NP * h->dp, 0.0);
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < N; ++c) x[(size_t)d * NP + c] = broadcast ? point[d] : point[(size_t)d * N + c];
    HIP_TRY(h, hipMemcpyAsync(h->d_forced, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->has_forced = true;
    return SMCMC_OK;
}

int smcmc_moments_size(const smcmc_engine* h) { return h ? (int)npacked(h) : 0; }

int smcmc_reduce_moments(smcmc_engine* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (h->panel_w) {
        hipError_t e = launch_fold_reduce(h->d_gacc, h->dim, h->fold_nslices, h->d_moments, h->stream);
        if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("fold reduce launch: ") + hipGetErrorString(e));
        HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
        return SMCMC_OK;
    }
    // (the first level leaves zero in the accumulators it read: no memset of the 21 MB per window)
    hipError_t e = dispatch_reduce(h->dp, h->d_gacc, h->ngroups, h->dim, h->d_chunks, h->d_moments, h->stream);
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("reduce kernel launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

int smcmc_export_moments(smcmc_engine* h, double* dst) {
    if (!h || !dst) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(dst, h->d_moments, npacked(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return SMCMC_OK;
}

int smcmc_import_moments(smcmc_engine* h, const double* src) {
    if (!h || !src) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(h->d_moments, src, npacked(h) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return SMCMC_OK;
}

int smcmc_read_moments(smcmc_engine* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(out, h->d_moments, npacked(h) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SMCMC_OK;
}

int smcmc_apply_moments(smcmc_engine* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    if (device_update_eligible(h)) return device_apply(h);
    int st = sync_shared_to_host(h, true);
    if (st) return st;
    const double* M = h->h_moments;
    HIP_TRY(h, hipMemcpyAsync(h->h_moments, h->d_moments, npacked(h) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    SharedProposal& P = *h->prop;
    if (!(M[npacked(h) - 1] > 0.0)) return SMCMC_OK;
    P.absorbMoments(M, h->mode == SMCMC_MODE_POOLED);
    st = update_shared(h);
    if (st) return st;
    return upload_shared(h);
}

int smcmc_sync(smcmc_engine* h) {
    int st = smcmc_reduce_moments(h);
    if (st) return st;
    if (h->comm) {
        st = smcmc_allreduce_moments(h);
        if (st) return st;
    }
    return smcmc_apply_moments(h);
}

// ---- RCCL, looked up at run time ---------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // a copy the process has already loaded (PyTorch ships one) wins over a second one
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char* n : names)
            if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) a.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
        return a;
    }();
    return api;
}
}  // namespace

int smcmc_comm_unique_id(void* id_out) {
    if (!id_out) return SMCMC_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == SMCMC_COMM_ID_BYTES, "SMCMC_COMM_ID_BYTES must be the size of an ncclUniqueId");
    RcclApi& api = rccl_api();
    if (!api.ok) return SMCMC_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return SMCMC_ERR_RUNTIME;
    std::memcpy(id_out, &id, sizeof(id));
    return SMCMC_OK;
}

int smcmc_comm_init(smcmc_engine* h, const void* id, int rank, int nranks) {
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return SMCMC_ERR_INVALID;
    if (h->comm) return fail(h, SMCMC_ERR_LOGIC, "the engine already has a communicator");
    RcclApi& api = rccl_api();
    if (!api.ok) return fail(h, SMCMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    ON_DEVICE(h);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = api.CommInitRank(&h->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        h->comm = nullptr;
        return fail(h, SMCMC_ERR_RUNTIME, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    }
    h->comm_ranks = nranks;
    return SMCMC_OK;
}

int smcmc_comm_destroy(smcmc_engine* h) {
    if (!h) return SMCMC_ERR_INVALID;
    if (!h->comm) return SMCMC_OK;
    ON_DEVICE(h);
    (void)hipStreamSynchronize(h->stream);
    rccl_api().CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_ranks = 0;
    return SMCMC_OK;
}

int smcmc_comm_ranks(smcmc_engine* h) {
    if (!h) return -1;
    if (!h->comm) return 0;
    RcclApi& api = rccl_api();
    int n = -1;
    if (!api.CommCount || api.CommCount(h->comm, &n) != ncclSuccess) return -1;
    return n;
}

// sum of the packed moment vector M over the ranks, in place, on the engine's stream
int smcmc_allreduce_moments(smcmc_engine* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    if (!h->comm) return fail(h, SMCMC_ERR_LOGIC, "no communicator: smcmc_comm_init first");
    ON_DEVICE(h);
    RcclApi& api = rccl_api();
    const ncclResult_t r = api.AllReduce(h->d_moments, h->d_moments, npacked(h), ncclDouble, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) return fail(h, SMCMC_ERR_RUNTIME, std::string("ncclAllReduce: ") + api.GetErrorString(r));
    return SMCMC_OK;
}

int smcmc_update_proposal(smcmc_engine* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    SharedProposal& P = *h->prop;
    if (per_chain(h)) {
        // every chain's own UpdateProposal() (SimpleMCMC.C:254): the step kernel's update code without a step
        int pst = pc_check_supported(h);
        if (pst) return pst;
        PerChainParams q = pc_params(h, make_params(h, 0, 0));
        q.update_only = 1;
        return pc_run(h, q);
    }
    int st = update_shared(h);
    if (st) return st;
    if (h->mode == SMCMC_MODE_FROZEN && P.lastPath != 4) {
        // every chain reschedules its own next update from its own successes (:1050-1052)
        const size_t NP = (size_t)h->npad;
        std::vector<int32_t> succ(NP), next(NP);
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemcpy(succ.data(), h->d_lane_i32 + (size_t)SMCMC_LANE_SUCCESSES * NP, NP * sizeof(int32_t),
                             hipMemcpyDeviceToHost));
        const double maxUp = (double)h->dim * (double)h->dim;
        for (size_t c = 0; c < NP; ++c)
            next[c] = (int32_t)(P.acceptanceWindow + maxUp - maxUp / (0.5 * succ[c] + 1.0));
        HIP_TRY(h, hipMemcpy(h->d_lane_i32 + (size_t)SMCMC_LANE_NEXT_UPDATE * NP, next.data(), NP * sizeof(int32_t),
                             hipMemcpyHostToDevice));
    }
    return upload_shared(h);
}

int smcmc_reset_proposal(smcmc_engine* h) {
    if (!h || !h->started) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    SharedProposal& P = *h->prop;
    // fLastPoint of the shared proposal := chain 0's current point
    std::vector<double> x0;
    int st = read_chain0(h, x0);
    if (st) return st;
    P.lastPoint = x0;
    st = status_of(h, P.reset());
    if (st) return st;
    if (per_chain(h)) return pc_broadcast(h, true, nullptr);   // the covariance template is the same for every chain; the rest per lane
    st = reset_lanes(h);
    if (st) return st;
    HIP_TRY(h, hipMemsetAsync(h->d_gacc, 0, sizeof(double) * gacc_doubles(h), h->stream));
    return upload_shared(h);
}

int smcmc_nchains_padded(const smcmc_engine* h) { return h ? h->npad : 0; }
int smcmc_dim_padded(const smcmc_engine* h) { return h ? h->dp : 0; }

int smcmc_read_state(smcmc_engine* h, double* x, double* logl) {
    if (!h) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    const int D = h->dim, N = h->nchains;
    const size_t NP = (size_t)h->npad;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (x) {
        HIP_TRY(h, hipMemcpy2D(x, (size_t)N * sizeof(double), h->d_x, NP * sizeof(double), (size_t)N * sizeof(double),
                               (size_t)D, hipMemcpyDeviceToHost));
    }
    if (logl) HIP_TRY(h, hipMemcpy(logl, h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * NP, (size_t)N * sizeof(double),
                                   hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_read_proposed(smcmc_engine* h, double* x) {
    if (!h || !x) return SMCMC_ERR_INVALID;
    if (!h->keep_proposed || !h->d_proposed)
        return fail(h, SMCMC_ERR_LOGIC, "the proposed point is not kept: set SMCMC_P_KEEP_PROPOSED first");
    ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy2D(x, (size_t)h->nchains * sizeof(double), h->d_proposed, (size_t)h->npad * sizeof(double),
                           (size_t)h->nchains * sizeof(double), (size_t)h->dim, hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_read_lane_f64(smcmc_engine* h, int field, double* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_F64_COUNT_) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = check_pending(h); if (sst_) return sst_; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->d_lane_f64 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(double),
                         hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_read_lane_i32(smcmc_engine* h, int field, int32_t* out) {
    if (!h || !out || field < 0 || field >= SMCMC_LANE_I32_COUNT_) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = check_pending(h); if (sst_) return sst_; }   // a fallback that ends in ResetProposal clears the counters
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->d_lane_i32 + (size_t)field * h->npad, (size_t)h->nchains * sizeof(int32_t),
                         hipMemcpyDeviceToHost));
    return SMCMC_OK;
}

int smcmc_read_chain(smcmc_engine* h, int chain, double* x, double* proposed, double* lanes_f64, int32_t* lanes_i32) {
    if (!h || chain < 0 || chain >= h->nchains) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = check_pending(h); if (sst_) return sst_; }
    if (proposed && (!h->keep_proposed || !h->d_proposed))
        return fail(h, SMCMC_ERR_LOGIC, "the proposed point is not kept: set SMCMC_P_KEEP_PROPOSED first");
    const size_t NP = (size_t)h->npad;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (x) HIP_TRY(h, pc_get_column(h->d_x, NP, chain, h->dim, x));
    if (proposed) HIP_TRY(h, pc_get_column(h->d_proposed, NP, chain, h->dim, proposed));
    if (lanes_f64) HIP_TRY(h, pc_get_column(h->d_lane_f64, NP, chain, SMCMC_LANE_F64_COUNT_, lanes_f64));
    if (lanes_i32) HIP_TRY(h, pc_get_column(h->d_lane_i32, NP, chain, SMCMC_LANE_I32_COUNT_, lanes_i32));
    return SMCMC_OK;
}

int smcmc_read_chain_proposal(smcmc_engine* h, int chain, double* centre, double* covariance, double* decomposition) {
    if (!h || chain < 0 || chain >= h->nchains) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    const int D = h->dim;
    if (!per_chain(h) || !h->d_pc_cov) {
        int sst_ = sync_shared_to_host(h, false);
        if (sst_) return sst_;
        if (centre) std::copy(h->prop->centre.begin(), h->prop->centre.end(), centre);
        if (covariance) std::copy(h->prop->cov.begin(), h->prop->cov.end(), covariance);
        if (decomposition) std::copy(h->prop->decomp.begin(), h->prop->decomp.end(), decomposition);
        return SMCMC_OK;
    }
    const size_t NP = (size_t)h->npad;
    const int npk = D * (D + 1) / 2;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (centre) HIP_TRY(h, pc_get_column(h->d_pc_centre, NP, chain, D, centre));
    if (covariance) {
        std::vector<double> packed(npk);
        HIP_TRY(h, pc_get_tiled(h->d_pc_cov, chain, npk, npk, packed.data()));
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j)
                covariance[(size_t)i * D + j] = covariance[(size_t)j * D + i] = packed[(size_t)i * (i + 1) / 2 + j];
    }
    if (decomposition) {
        std::vector<double> ut((size_t)D * D);
        int32_t full = 0;
        HIP_TRY(h, pc_get_tiled(h->d_pc_ut, chain, D * D, D * D, ut.data()));
        HIP_TRY(h, hipMemcpy(&full, h->d_lane_i32 + (size_t)SMCMC_LANE_DECOMP_FULL * NP + chain, sizeof(int32_t), hipMemcpyDeviceToHost));
        std::fill(decomposition, decomposition + (size_t)D * D, 0.0);
        for (int j = 0; j < D; ++j)
            for (int i = 0; i <= j; ++i) decomposition[(size_t)i * D + j] = ut[(size_t)j * (j + 1) / 2 + i];
        if (full)
            for (int i = 1; i < D; ++i)
                for (int j = 0; j < i; ++j) decomposition[(size_t)i * D + j] = ut[(size_t)npk + (size_t)i * (i - 1) / 2 + j];
    }
    return SMCMC_OK;
}

// SMCMC_MODE_PER_CHAIN: a [rows] vector handed to every chain's column, on the device (the vector goes up once; a host
// image of the whole array would be 0.67 GB for a covariance at D = 50 x 65 536 chains)
static int pc_broadcast_rows(smcmc_engine* h, double* dst, const double* values, int rows, bool tiled) {
    HIP_TRY(h, hipMemcpyAsync(h->d_pc_tmpl, values, (size_t)rows * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // pageable source
    const hipError_t e = launch_perchain_broadcast_rows(dst, h->d_pc_tmpl, rows, h->nchains, (size_t)h->npad, tiled, h->stream);
    if (e != hipSuccess) return fail(h, SMCMC_ERR_HIP, std::string("per-chain broadcast launch: ") + hipGetErrorString(e));
    return SMCMC_OK;
}

int smcmc_get_center(smcmc_engine* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    if (per_chain(h) && h->d_pc_cov) return smcmc_read_chain_proposal(h, 0, out, nullptr, nullptr);
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, false); if (sst_) return sst_; }
    std::copy(h->prop->centre.begin(), h->prop->centre.end(), out);
    return SMCMC_OK;
}

int smcmc_set_center(smcmc_engine* h, const double* in) {
    if (!h || !in) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    std::copy(in, in + h->dim, h->prop->centre.begin());
    if (per_chain(h) && h->started) return pc_broadcast_rows(h, h->d_pc_centre, in, h->dim, false);
    return h->started ? upload_shared(h) : SMCMC_OK;
}

int smcmc_get_covariance(smcmc_engine* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    if (per_chain(h) && h->d_pc_cov) return smcmc_read_chain_proposal(h, 0, nullptr, out, nullptr);
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, false); if (sst_) return sst_; }
    std::copy(h->prop->cov.begin(), h->prop->cov.end(), out);
    return SMCMC_OK;
}

int smcmc_set_covariance(smcmc_engine* h, const double* in) {
    if (!h || !in) return SMCMC_ERR_INVALID;
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, true); if (sst_) return sst_; }
    std::copy(in, in + (size_t)h->dim * h->dim, h->prop->cov.begin());
    if (per_chain(h) && h->started) {
        const int D = h->dim;
        std::vector<double> packed((size_t)D * (D + 1) / 2);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j) packed[(size_t)i * (i + 1) / 2 + j] = in[(size_t)i * D + j];
        return pc_broadcast_rows(h, h->d_pc_cov, packed.data(), (int)packed.size(), true);
    }
    return SMCMC_OK;
}

int smcmc_get_decomposition(smcmc_engine* h, double* out) {
    if (!h || !out) return SMCMC_ERR_INVALID;
    if (per_chain(h) && h->d_pc_cov) return smcmc_read_chain_proposal(h, 0, nullptr, nullptr, out);
    ON_DEVICE(h);
    { int sst_ = sync_shared_to_host(h, false); if (sst_) return sst_; }
    std::copy(h->prop->decomp.begin(), h->prop->decomp.end(), out);
    return SMCMC_OK;
}

int smcmc_state_device_ptr(smcmc_engine* h, double** x, double** logl) {
    if (!h) return SMCMC_ERR_INVALID;
    if (x) *x = h->d_x;
    if (logl) *logl = h->d_lane_f64 + (size_t)SMCMC_LANE_LOGL * h->npad;
    return SMCMC_OK;
}

}  // extern "C"
