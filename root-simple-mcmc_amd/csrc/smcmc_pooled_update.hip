// smcmc_pooled_update.hip -- kernels of the device-resident pooled UpdateProposal (see smcmc_pooled_update.hip.h).
// Compiled, like every other unit, with -ffp-contract=off: each multiply and each subtract below rounds on its own,
// as in SharedProposal::absorbMoments / update / cholesky (smcmc_proposal.hpp), which this file restates for the GPU.
#include "smcmc_pooled_update.hip.h"

namespace smcmc {

namespace {

__device__ __forceinline__ double pu_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double pu_max(double a, double b) { return __builtin_fmax(a, b); }

// SharedProposal::absorbMoments: one thread per (i, j <= i)
__global__ void __launch_bounds__(256) pooled_absorb_kernel(const PooledUpdateParams p) {
    const int D = p.D;
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= D || j > i) return;
    const double* S1 = p.M + (size_t)D * (D + 1) / 2;
    const double n = S1[D];
    if (!(n > 0.0)) return;
    const double centre_trials = p.scal[kPsCentreTrials], cov_trials = p.scal[kPsCovTrials];
    const double di = S1[i] / (centre_trials + n);
    const double dj = S1[j] / (centre_trials + n);
    double b = p.M[(size_t)i * (i + 1) / 2 + j];
    b -= S1[i] * dj;
    b -= di * S1[j];
    b += (n * di) * dj;
    double v = p.cov[(size_t)i * D + j];
    v *= cov_trials;
    v += b;
    v /= cov_trials + n;
    p.cov[(size_t)i * D + j] = v;
    p.cov[(size_t)j * D + i] = v;
    if (j == 0) p.centre[i] = p.centre[i] + di;
}

// the trials, and the scalar half of SharedProposal::update (TSimpleMCMC.H:1030-1075): one wavefront
__global__ void __launch_bounds__(64) pooled_scalars_kernel(const PooledUpdateParams p) {
    __shared__ double diag[512];
    const int D = p.D;
    const int lane = threadIdx.x;
    for (int d = lane; d < D; d += 64) diag[d] = p.cov[(size_t)d * D + d];
    __syncthreads();
    if (lane != 0) return;
    const double* S1 = p.M + (size_t)D * (D + 1) / 2;
    const double n = S1[D];
    if (!(n > 0.0)) {
        p.scal[kPsStatus] = kPooledSkipped;
        return;
    }
    double centre_trials = pu_min(p.cov_window, p.scal[kPsCentreTrials] + n);
    double cov_trials = pu_min(p.cov_window, p.scal[kPsCovTrials] + n);
    double trace = 0.0;
    for (int d = 0; d < D; ++d) trace += diag[d];
    if (trace <= 0) {
        p.scal[kPsCentreTrials] = centre_trials;
        p.scal[kPsCovTrials] = cov_trials;
        p.scal[kPsStatus] = kPooledInvalidTrace;
        return;
    }
    const double scale = __builtin_sqrt(p.scal[kPsSigmaTrace] / trace);
    p.scal[kPsSigma] = p.scal[kPsSigma] * scale;
    p.scal[kPsSigmaTrace] = trace;
    p.scal[kPsLastScale] = scale;
    if (p.cov_deweight > 0.0) {
        const double w = 1.0 - pu_min(p.cov_deweight, 1.0);
        cov_trials = pu_max(1.0, w * cov_trials);
        cov_trials = pu_min(cov_trials, w * p.cov_window);
        centre_trials = pu_max(1.0, w * centre_trials);
        centre_trials = pu_min(centre_trials, w * p.cov_window);
    }
    p.scal[kPsCentreTrials] = centre_trials;
    p.scal[kPsCovTrials] = cov_trials;
    p.scal[kPsStatus] = kPooledOk;
}

__global__ void __launch_bounds__(256) chol_copy_kernel(const double* cov, double* decomp, size_t n, const double* scal) {
    if (scal[kPsStatus] != kPooledOk) return;
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) decomp[k] = cov[k];
}

// rows r0 .. r0+31 of U: one workgroup, the panel (its rows, columns r0 .. D-1) in LDS
__global__ void __launch_bounds__(512) chol_panel_kernel(double* R, int D, int r0, double* scal) {
    extern __shared__ double panel[];                   // [nrows][ncols]
    __shared__ double s_piv;
    __shared__ int s_fail;
    if (scal[kPsStatus] != kPooledOk) return;
    const int nrows = (D - r0 < kCholPanel) ? D - r0 : kCholPanel;
    const int ncols = D - r0;
    const int jj = threadIdx.x;                         // local column (ncols <= 512)
    for (int rr = 0; rr < nrows; ++rr)
        if (jj < ncols) panel[rr * ncols + jj] = R[(size_t)(r0 + rr) * D + r0 + jj];
    if (jj == 0) s_fail = 0;
    __syncthreads();
    for (int c = 0; c < nrows; ++c) {
        double v = 0.0;
        if (jj >= c && jj < ncols) {
            v = panel[c * ncols + jj];
            for (int rr = 0; rr < c; ++rr) v -= panel[rr * ncols + jj] * panel[rr * ncols + c];
            if (jj == c) {
                if (!(v > 0.0) || !__builtin_isfinite(v)) {
                    s_fail = 1;
                } else {
                    const double piv = __builtin_sqrt(v);
                    panel[c * ncols + c] = piv;
                    s_piv = piv;
                }
            }
        }
        __syncthreads();
        if (s_fail) {
            if (jj == 0) scal[kPsStatus] = kPooledCholeskyFailed;
            return;
        }
        if (jj > c && jj < ncols) panel[c * ncols + jj] = v / s_piv;
        __syncthreads();
    }
    for (int rr = 0; rr < nrows; ++rr)
        if (jj < ncols) R[(size_t)(r0 + rr) * D + r0 + jj] = panel[rr * ncols + jj];
}

// v(c, j) -= U(r, j) * U(r, c) for the panel's rows r, in order, for every element below the panel (c >= r1, j >= c)
__global__ void __launch_bounds__(256) chol_trailing_kernel(double* R, int D, int r0, int nrows, const double* scal) {
    __shared__ double A[kCholPanel][kCholPanel + 1];    // U(r0 + r, c-tile)
    __shared__ double B[kCholPanel][kCholPanel + 1];    // U(r0 + r, j-tile)
    if (blockIdx.x < blockIdx.y) return;                // the tile lies below the diagonal
    if (scal[kPsStatus] != kPooledOk) return;
    const int r1 = r0 + nrows;
    const int c0 = r1 + blockIdx.y * kCholPanel, j0 = r1 + blockIdx.x * kCholPanel;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int rr = ty; rr < nrows; rr += 8) {
        A[rr][tx] = (c0 + tx < D) ? R[(size_t)(r0 + rr) * D + c0 + tx] : 0.0;
        B[rr][tx] = (j0 + tx < D) ? R[(size_t)(r0 + rr) * D + j0 + tx] : 0.0;
    }
    __syncthreads();
    const int j = j0 + tx;
    for (int cc = ty; cc < kCholPanel; cc += 8) {
        const int c = c0 + cc;
        if (c >= D || j >= D || j < c) continue;
        double v = R[(size_t)c * D + j];
        for (int rr = 0; rr < nrows; ++rr) v -= B[rr][tx] * A[rr][cc];
        R[(size_t)c * D + j] = v;
    }
}

__global__ void __launch_bounds__(256) chol_finish_kernel(double* R, int D, const double* scal) {
    if (scal[kPsStatus] != kPooledOk) return;
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t)D * D) return;
    const int r = (int)(k / D), c = (int)(k % D);
    if (c < r) R[k] = 0.0;
}

__global__ void __launch_bounds__(256) pooled_publish_kernel(const PooledPublishParams p) {
    const int D = p.D;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * 256;
    // the moments of the next window are taken about the centre the running average now holds, whatever became of U
    for (size_t d = tid; d < (size_t)D; d += nthreads) p.c0[d] = p.centre[d];
    if (p.scal[kPsStatus] != kPooledOk) return;
    if (p.DP > 0) {
        const int DP = p.DP;
        for (size_t k = tid; k < (size_t)DP * DP; k += nthreads) {
            const int i = (int)(k / DP), j = (int)(k % DP);
            p.U[k] = (i < D && j < D) ? p.decomp[(size_t)i * D + j] : 0.0;
        }
    } else {
        const int W = p.W, CW = p.CW;
        for (size_t k = tid; k < (size_t)W * D * CW; k += nthreads) {
            const int jl = (int)(k % CW);
            const int i = (int)((k / CW) % D);
            const int w = (int)(k / ((size_t)CW * D));
            const int j = jl * W + w;
            p.U[k] = (j < D) ? p.decomp[(size_t)i * D + j] : 0.0;
        }
        if (p.nkq_padded > 0) {
            const int ntiles = (D + 15) / 16, nkq = (D + 3) / 4, nkqp = p.nkq_padded;
            for (size_t k = tid; k < (size_t)ntiles * nkqp * 64; k += nthreads) {
                const int l = (int)(k % 64);
                const int kq = (int)((k / 64) % nkqp);
                const int jt = (int)(k / ((size_t)64 * nkqp));
                const int i = 4 * kq + (l >> 4), j = 16 * jt + (l & 15);
                p.Uop[k] = (kq < nkq && i < D && j < D) ? p.decomp[(size_t)i * D + j] : 0.0;
            }
        }
    }
}

__global__ void __launch_bounds__(256) pooled_adjust_lanes_kernel(double* lane_f64, int npad, int nchains, const double* scal,
                                                                   double acc_w, double acc_wW, int sigma_lane, int trials_lane,
                                                                   double* host_scal) {
    // the update's scalars (status word included) for the host: written straight into its pinned, device-visible copy --
    // there when the launch has completed -- instead of a copy command behind this kernel
    if (host_scal != nullptr && blockIdx.x == 0 && threadIdx.x < kPsCount) host_scal[threadIdx.x] = scal[threadIdx.x];
    if (scal[kPsStatus] != kPooledOk) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nchains) return;
    const double scale = scal[kPsLastScale];
    double* sg = lane_f64 + (size_t)sigma_lane * npad + c;
    *sg = *sg * scale;
    if (acc_w >= 0.0) {
        double* at = lane_f64 + (size_t)trials_lane * npad + c;
        double t = *at;
        t = pu_max(1.0, acc_w * t);
        t = pu_min(t, acc_wW);
        *at = t;
    }
}

// dim <= 64: the whole update in one workgroup -- absorb, scalar half, Cholesky (the matrix in LDS), decomposition, the
// register kernels' [DP][DP] copy of U and the moment centre.  Element by element the arithmetic of the kernels above
// (which are the host's): a 64-row panel sees the same subtractions in the same order as two 32-row panels and a
// trailing update.  Twelve launches per window become one at the headline's D = 50.
constexpr int kSmallDim = 64;
__global__ void __launch_bounds__(512) pooled_small_update_kernel(const PooledUpdateParams p, const PooledPublishParams q) {
    __shared__ double R[kSmallDim][kSmallDim + 1];
    __shared__ double s_piv;
    __shared__ int s_state;    // kPooledOk ...
    const int D = p.D;
    const int tid = threadIdx.x;
    const double* S1 = p.M + (size_t)D * (D + 1) / 2;
    const double n = S1[D];
    if (!(n > 0.0)) {          // no point was folded: no update (uniform: every thread reads the same word)
        if (tid == 0) p.scal[kPsStatus] = kPooledSkipped;
        return;
    }
    // ---- SharedProposal::absorbMoments ----
    const double centre_trials0 = p.scal[kPsCentreTrials], cov_trials0 = p.scal[kPsCovTrials];
    for (int k = tid; k < D * (D + 1) / 2; k += 512) {
        int i = (int)((__builtin_sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= k) ++i;
        while (i * (i + 1) / 2 > k) --i;
        const int j = k - i * (i + 1) / 2;
        const double di = S1[i] / (centre_trials0 + n);
        const double dj = S1[j] / (centre_trials0 + n);
        double b = p.M[k];
        b -= S1[i] * dj;
        b -= di * S1[j];
        b += (n * di) * dj;
        double v = p.cov[(size_t)i * D + j];
        v *= cov_trials0;
        v += b;
        v /= cov_trials0 + n;
        p.cov[(size_t)i * D + j] = v;
        p.cov[(size_t)j * D + i] = v;
        R[i][j] = v;
        R[j][i] = v;
        if (j == 0) p.centre[i] = p.centre[i] + di;
    }
    __syncthreads();
    // ---- trials and the scalar half of SharedProposal::update ----
    if (tid == 0) {
        double centre_trials = pu_min(p.cov_window, centre_trials0 + n);
        double cov_trials = pu_min(p.cov_window, cov_trials0 + n);
        double trace = 0.0;
        for (int d = 0; d < D; ++d) trace += R[d][d];
        int state = kPooledOk;
        if (trace <= 0) {
            state = kPooledInvalidTrace;
        } else {
            const double scale = __builtin_sqrt(p.scal[kPsSigmaTrace] / trace);
            p.scal[kPsSigma] = p.scal[kPsSigma] * scale;
            p.scal[kPsSigmaTrace] = trace;
            p.scal[kPsLastScale] = scale;
            if (p.cov_deweight > 0.0) {
                const double w = 1.0 - pu_min(p.cov_deweight, 1.0);
                cov_trials = pu_max(1.0, w * cov_trials);
                cov_trials = pu_min(cov_trials, w * p.cov_window);
                centre_trials = pu_max(1.0, w * centre_trials);
                centre_trials = pu_min(centre_trials, w * p.cov_window);
            }
        }
        p.scal[kPsCentreTrials] = centre_trials;
        p.scal[kPsCovTrials] = cov_trials;
        s_state = state;
    }
    __syncthreads();
    // the moments of the next window are taken about the centre the running average now holds, whatever becomes of U
    for (int d = tid; d < D; d += 512) q.c0[d] = p.centre[d];
    if (s_state != kPooledOk) {
        if (tid == 0) p.scal[kPsStatus] = s_state;
        return;
    }
    // ---- SharedProposal::cholesky ----
    // Row by row as the reference's decomposition goes, but RIGHT-LOOKING: once row c is final, every element (i, j),
    // c < i <= j, of the rows below takes ITS subtraction of that row at once -- v -= U(c, i) U(c, j), un-fused -- spread
    // over all 512 threads; by the time row i is the pivot row its elements have seen r = 0 .. i - 1 in ascending order,
    // the same subtractions in the same order as a dot product over the rows above (a product's operands swapped:
    // commutative).  Two workgroup barriers per row instead of a chain of c dependent LDS round trips: the one-wavefront
    // left-looking version was 38 of this kernel's 44 us at D = 50.
    // tab: the elements (i, j), i <= j, with the rows in DESCENDING order, so that the block below row c is the first
    // (D - 1 - c)(D - c) / 2 entries.
    __shared__ unsigned short tab[kSmallDim * (kSmallDim + 1) / 2];
    for (int e = tid; e < D * (D + 1) / 2; e += 512) {
        int m = (int)((__builtin_sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);   // rows below the top: m (m + 1) / 2 <= e
        while ((m + 1) * (m + 2) / 2 <= e) ++m;
        while (m * (m + 1) / 2 > e) --m;
        const int i = D - 1 - m;
        tab[e] = (unsigned short)((i << 8) | (i + (e - m * (m + 1) / 2)));
    }
    for (int c = 0; c < D; ++c) {
        __syncthreads();                       // the subtractions of row c - 1 are in (and tab, the first time)
        if (tid < 64) {
            const double piv = R[c][c];
            if (!(piv > 0.0) || !__builtin_isfinite(piv)) {
                if (tid == 0) s_state = kPooledCholeskyFailed;
            } else {
                const double root = __builtin_sqrt(piv);
                if (tid == c) R[c][c] = root;
                else if (tid > c && tid < D) R[c][tid] = R[c][tid] / root;
            }
        }
        __syncthreads();                       // row c is final
        if (s_state != kPooledOk) break;       // (uniform: read behind the barrier)
        const int below = (D - 1 - c) * (D - c) / 2;
        for (int e = tid; e < below; e += 512) {
            const int ij = tab[e], i = ij >> 8, j = ij & 255;
            double v = R[i][j];
            v -= R[c][i] * R[c][j];
            R[i][j] = v;
        }
    }
    __syncthreads();
    if (s_state != kPooledOk) {
        if (tid == 0) p.scal[kPsStatus] = s_state;
        return;
    }
    // ---- the decomposition (zeros below the diagonal) and the step kernels' padded copy ----
    for (int k = tid; k < D * D; k += 512) {
        const int r = k / D, c = k % D;
        p.decomp[k] = (c < r) ? 0.0 : R[r][c];
    }
    const int DP = q.DP;
    for (int k = tid; k < DP * DP; k += 512) {
        const int i = k / DP, j = k % DP;
        q.U[k] = (i < D && j < D && j >= i) ? R[i][j] : 0.0;
    }
    if (tid == 0) p.scal[kPsStatus] = kPooledOk;
}

}  // namespace

hipError_t launch_pooled_small_update(const PooledUpdateParams& p, const PooledPublishParams& q, hipStream_t s) {
    if (p.D < 1 || p.D > kSmallDim || q.DP < p.D) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pooled_small_update_kernel, dim3(1), dim3(512), 0, s, p, q);
    return hipGetLastError();
}

hipError_t launch_pooled_update(const PooledUpdateParams& p, hipStream_t s) {
    const int D = p.D;
    if (D < 1 || D > 512) return hipErrorInvalidValue;
    const int t16 = (D + 15) / 16;
    hipLaunchKernelGGL(pooled_absorb_kernel, dim3(t16, t16), dim3(256), 0, s, p);
    hipLaunchKernelGGL(pooled_scalars_kernel, dim3(1), dim3(64), 0, s, p);
    const size_t nn = (size_t)D * D;
    const unsigned nb = (unsigned)((nn + 255) / 256);
    hipLaunchKernelGGL(chol_copy_kernel, dim3(nb), dim3(256), 0, s, (const double*)p.cov, p.decomp, nn, (const double*)p.scal);
    for (int r0 = 0; r0 < D; r0 += kCholPanel) {
        const int nrows = (D - r0 < kCholPanel) ? D - r0 : kCholPanel;
        const size_t lds = sizeof(double) * (size_t)nrows * (size_t)(D - r0);
        hipLaunchKernelGGL(chol_panel_kernel, dim3(1), dim3(512), lds, s, p.decomp, D, r0, p.scal);
        const int rest = D - (r0 + nrows);
        if (rest > 0) {
            const int nt = (rest + kCholPanel - 1) / kCholPanel;
            hipLaunchKernelGGL(chol_trailing_kernel, dim3(nt, nt), dim3(256), 0, s, p.decomp, D, r0, nrows, (const double*)p.scal);
        }
    }
    hipLaunchKernelGGL(chol_finish_kernel, dim3(nb), dim3(256), 0, s, p.decomp, D, (const double*)p.scal);
    return hipGetLastError();
}

// one-time: the panel kernel's dynamic LDS goes up to 128 KB
hipError_t pooled_update_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(chol_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * kCholPanel * 512));
}

hipError_t launch_pooled_publish(const PooledPublishParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pooled_publish_kernel, dim3(256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_pooled_adjust_lanes(double* lane_f64, int npad, int nchains, const double* scal, double acc_w, double acc_wW,
                                      int sigma_lane, int trials_lane, double* host_scal, hipStream_t s) {
    static_assert(kPsCount <= 256, "one thread per scalar");
    hipLaunchKernelGGL(pooled_adjust_lanes_kernel, dim3((nchains + 255) / 256), dim3(256), 0, s, lane_f64, npad, nchains, scal,
                       acc_w, acc_wW, sigma_lane, trials_lane, host_scal);
    return hipGetLastError();
}

}  // namespace smcmc
