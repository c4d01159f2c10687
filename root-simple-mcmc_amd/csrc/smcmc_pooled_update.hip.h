// smcmc_pooled_update.hip.h -- the pooled UpdateProposal on the device: the batch form of the running centre /
// covariance (reference TSimpleMCMC.H:1780-1820 per point; SharedProposal::absorbMoments in smcmc_proposal.hpp for a
// batch), the scalar half of UpdateProposal (:1030-1086) and the Cholesky decomposition (:1097-1106, ROOT's
// TDecompChol restated as a row-ordered A = U^T U in SharedProposal::cholesky), in the operation order of the host
// code they replace so that not a bit changes.  The host keeps the ladder of fallbacks (:1134-1389): a failed pivot
// or a non-positive trace raises a status word and nothing downstream of it runs on the device.
//
// Kernels, all on the engine's stream, no host synchronisation between them:
//   pooled_absorb_kernel     one thread per (i, j <= i): centre and covariance from the packed moments
//   pooled_scalars_kernel    one wavefront: trials, trace (summed in index order), sigma rescale, de-weighting
//   chol_panel_kernel        one workgroup per 32-row panel, the panel in LDS (32 x 512 doubles = 128 KB)
//   chol_trailing_kernel     the rank-32 update of everything below the panel, one 32 x 32 tile per workgroup
//   chol_finish_kernel       zeros below the diagonal
//   pooled_publish_kernel    U into the layouts the step kernels read; the moment centre
//   pooled_adjust_lanes_kernel  every chain's sigma and acceptance trials (TSimpleMCMC.H:1042-1043, 1081-1086)
// Every element of U sees its subtractions v -= U(r,j) * U(r,c) for r = 0, 1, 2, ... in that order, un-fused, whether
// a row r belongs to an earlier panel (applied by that panel's trailing update) or to its own (applied inside the
// panel kernel): the same sequence of roundings as the host loop.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smcmc {

// device-resident scalars of the shared proposal (doubles)
enum {
    kPsCovTrials = 0,      // fCovarianceTrials
    kPsCentreTrials,       // fCentralPointTrials
    kPsSigma,              // the shared template of fSigma
    kPsSigmaTrace,         // fSigmaTrace
    kPsLastScale,          // sqrt(fSigmaTrace / trace) of the latest update
    kPsStatus,             // kPooledOk ...
    kPsCount = 8
};
enum { kPooledOk = 0, kPooledSkipped = 1, kPooledInvalidTrace = 2, kPooledCholeskyFailed = 3 };

constexpr int kCholPanel = 32;

struct PooledUpdateParams {
    int D;
    const double* M;       // packed moments about the centre: row i <= D, column j <= i; row D = {sum y_j, n}
    double* centre;        // [D]
    double* cov;           // [D][D]
    double* decomp;        // [D][D]
    double* scal;          // [kPsCount]
    double cov_window, cov_deweight;
};

struct PooledPublishParams {
    int D, DP;             // DP: row pitch of the register kernels' U (dim <= 63), 0 for the large-dimension layouts
    int W, CW;             // large dimensions: wavefronts per group, columns per wavefront
    int nkq_padded;        // large dimensions, fused order: k-quads per tile of the operand layout (0 = not wanted)
    const double* decomp;
    const double* centre;
    const double* scal;
    double* U;             // dim <= 63: [DP][DP]; else Uperm[w][i][jl] = U(i, jl*W + w)
    double* Uop;           // Uop[(tile * nkq + kq) * 64 + lane] = U(4 kq + (lane >> 4), 16 tile + (lane & 15))
    double* c0;            // [DP or D] the centre the moments are taken about
};

hipError_t launch_pooled_update(const PooledUpdateParams& p, hipStream_t stream);
hipError_t launch_pooled_publish(const PooledPublishParams& p, hipStream_t stream);
// dim <= 64 with the register kernels' layout of U (q.DP > 0): update and publish in one single-workgroup kernel
hipError_t launch_pooled_small_update(const PooledUpdateParams& p, const PooledPublishParams& q, hipStream_t stream);
// host_scal: the pinned host copy of the kPsCount scalars as the device sees it (or null: the caller copies them itself)
hipError_t launch_pooled_adjust_lanes(double* lane_f64, int npad, int nchains, const double* scal, double acc_w,
                                      double acc_wW, int sigma_lane, int trials_lane, double* host_scal, hipStream_t stream);
hipError_t pooled_update_prepare();   // once per process, before the first launch_pooled_update

}  // namespace smcmc
