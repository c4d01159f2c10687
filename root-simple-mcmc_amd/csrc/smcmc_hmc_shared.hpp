// smcmc_hmc_shared.hpp -- host side of the HMC engine's pooled tuning state.
//
// TSimpleHMC derives its step length and leapfrog count from a running covariance of the
// points the chain visits (reference TSimpleHMC.H: UpdateCovariance :665-695, UpdateErrorMatrix
// :703-858).  A D x D covariance per chain is impossible for an ensemble (8 192 x 500^2 doubles), so
// the ensemble pools it exactly as the Metropolis engine pools its proposal covariance: the device
// folds every contributing chain's accepted point into packed moment sums (smcmc_fold_kernel.hip.h),
// and this class feeds the reference's running averages with the batch (n points at once; n = 1 is
// the reference's own arithmetic) and runs UpdateErrorMatrix once per sync.  What the chains take
// from an update (step length, leapfrog count, :833-847) is applied per chain on the device
// (hmc_retune_kernel).  The O(D^3) pieces the reference gets from ROOT (eigenvalues of the covariance
// :766, TMatrixD::Invert :850) are a Householder tridiagonalisation + QL iteration and a Gauss-Jordan
// elimination here.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace smcmc {

class HmcShared {
public:
    explicit HmcShared(int dim)
        : D(dim), average(dim, 0.0), exxt((size_t)dim * dim, 0.0), cov((size_t)dim * dim, 0.0),
          error((size_t)dim * dim, 0.0) {}

    const int D;
    double covWindow = 1000000;                  // fCovarianceWindow, TSimpleHMC.H:134
    std::vector<double> average;                 // fAveragePoint
    double averageTrials = 0.0;
    std::vector<double> exxt, cov, error;        // fEXXT, fEstimatedCovariance, fEstimatedError
    double covTrials = 0.0;
    double estTrace = 0.0, curTrace = 0.0, orbitLength = 0.0;
    int stepsRemaining = 0, stepsSinceUpdate = 0, stepCount = 0, updateCount = 0;
    double maxScale = 0.0, minScale = 0.0;
    bool leapfrogZero = false;                   // SetLeapFrog(0): UpdateErrorMatrix returns at once (:704)

    // the covariance part of Start (:236-266)
    void start(const double* x0) {
        for (int i = 0; i < D; ++i) average[i] = x0[i];
        averageTrials = 0.0;
        covTrials = 0.0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                cov[(size_t)i * D + j] = (i == j) ? 1.0 : 0.0;
                exxt[(size_t)i * D + j] = 0.0;
            }
        for (int i = 0; i < D; ++i)                                          // :261-262: the inverse of the identity
            for (int j = 0; j < D; ++j) error[(size_t)i * D + j] = (i == j) ? 1.0 : 0.0;
        estTrace = D;
        curTrace = 0.0;
        orbitLength = 0.0;
        stepsRemaining = 0;
        stepsSinceUpdate = 0;
        stepCount = 0;
        updateCount = 0;
    }

    // UpdateCovariance (:665-695) fed with a batch: M is the packed moment vector (row i <= D, column j <= i,
    // row D = {sum x_j, n}) of the n accepted points of `steps` ensemble steps.
    void absorb(const double* M, int steps) {
        const double* S1 = M + (size_t)D * (D + 1) / 2;
        const double n = S1[D];
        stepsSinceUpdate += steps;                                           // :667-668
        stepsRemaining -= steps;
        for (int i = 0; i < D; ++i) {                                        // :671-677
            double v = average[i];
            v *= averageTrials;
            v += S1[i];
            v /= averageTrials + n;
            average[i] = v;
        }
        averageTrials = std::min(covWindow, averageTrials + n);              // :678-679
        for (int i = 0; i < D; ++i) {                                        // :681-691
            for (int j = 0; j < i + 1; ++j) {
                double v = exxt[(size_t)i * D + j];
                v *= covTrials;
                v += M[(size_t)i * (i + 1) / 2 + j];
                v /= covTrials + n;
                exxt[(size_t)i * D + j] = exxt[(size_t)j * D + i] = v;
                cov[(size_t)i * D + j] = cov[(size_t)j * D + i] = exxt[(size_t)i * D + j] - average[i] * average[j];
            }
        }
        covTrials = std::min(covWindow, covTrials + n);                      // :692-693
    }

    // The step counters of UpdateCovariance (:667-668) alone: the running averages of this batch were formed on the
    // device (smcmc_hmc_engine.hip: hmc_absorb_* kernels, the arithmetic of absorb()), which hands back the trial
    // counts it left.
    void absorbedOnDevice(int steps, double newAverageTrials, double newCovTrials) {
        stepsSinceUpdate += steps;
        stepsRemaining -= steps;
        averageTrials = newAverageTrials;
        covTrials = newCovTrials;
    }

    // UpdateErrorMatrix's decision (:704-719) from the trace of the current covariance (sum of |diagonal|, index order)
    bool wantsUpdate(double trace) {
        if (leapfrogZero) return false;                                      // :704
        if (covTrials < 2 * D) return false;                                 // :705
        curTrace = trace;                                                    // :708-711
        const double change = std::fabs(curTrace - estTrace);
        bool doIt = false;                                                   // :715-719
        if (stepsRemaining < 0) doIt = true;
        if (stepsSinceUpdate > 2.0 * D && change > 0.01 * estTrace) doIt = true;
        return doIt;
    }

    // UpdateErrorMatrix (:703-858) without the central-point bookkeeping of :733-744 (outputs only: left to the
    // host mirror of the class).  True when the update went through.
    bool updateErrorMatrix() {
        double trace = 0.0;
        if (!leapfrogZero && !(covTrials < 2 * D))
            for (int i = 0; i < D; ++i) trace += std::fabs(cov[(size_t)i * D + i]);
        if (!wantsUpdate(trace)) return false;
        finishUpdate();
        return true;
    }

    // :760-858 on the covariance in `cov`
    void finishUpdate() {
        stepsRemaining = 2 * D + stepCount;                                  // :760
        stepsSinceUpdate = 0;
        std::vector<double> eig((size_t)D);
        double maxS = 0.0, minS = 1E+20;                                     // :764-765
        for (;;) {                                                           // :766-809
            eigenvalues(cov, eig);
            bool positive = true;
            for (int i = 0; i < D; ++i) {
                const double e = eig[i];
                if (maxS < std::fabs(e)) maxS = std::fabs(e);
                if (minS > std::fabs(e)) minS = std::fabs(e);
                if (e < 0) positive = false;
            }
            if (positive) break;
            for (int i = 0; i < D; ++i) {
                double r = estTrace * 1E-6;
                r /= D;
                r = std::fabs(r);
                if (cov[(size_t)i * D + i] < r) cov[(size_t)i * D + i] = r;
                for (int j = i + 1; j < D; ++j) {
                    cov[(size_t)i * D + j] = 0.0;
                    cov[(size_t)j * D + i] = cov[(size_t)i * D + j];
                }
            }
        }
        curTrace = 0.0;                                                      // :815-819
        for (int i = 0; i < D; ++i) curTrace += std::fabs(cov[(size_t)i * D + i]);
        estTrace = curTrace;
        maxS = std::sqrt(maxS);                                              // :822-827
        if (maxS < 0.1) maxS = 0.1;
        minS = std::sqrt(minS);
        if (minS < 0.01) minS = 0.01;
        orbitLength = 2.0 * 3.14 * maxS;                                     // :830
        maxScale = maxS;
        minScale = minS;
        invert(cov, error);                                                  // :849-850
        ++updateCount;
    }

private:
    // eigenvalues of the symmetric matrix A (natural order of the algorithm): Householder reduction to
    // tridiagonal form, then QL iterations with implicit shifts
    void eigenvalues(const std::vector<double>& A, std::vector<double>& d) const {
        const int n = D;
        std::vector<double> a(A), e((size_t)n + 1, 0.0);
        for (int i = n - 1; i >= 1; --i) {
            const int l = i - 1;
            double h = 0.0, scale = 0.0;
            if (l > 0) {
                for (int k = 0; k <= l; ++k) scale += std::fabs(a[(size_t)i * n + k]);
                if (scale == 0.0) {
                    e[i] = a[(size_t)i * n + l];
                } else {
                    for (int k = 0; k <= l; ++k) {
                        a[(size_t)i * n + k] /= scale;
                        h += a[(size_t)i * n + k] * a[(size_t)i * n + k];
                    }
                    double f = a[(size_t)i * n + l];
                    double g = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
                    e[i] = scale * g;
                    h -= f * g;
                    a[(size_t)i * n + l] = f - g;
                    f = 0.0;
                    for (int j = 0; j <= l; ++j) {
                        g = 0.0;
                        for (int k = 0; k <= j; ++k) g += a[(size_t)j * n + k] * a[(size_t)i * n + k];
                        for (int k = j + 1; k <= l; ++k) g += a[(size_t)k * n + j] * a[(size_t)i * n + k];
                        e[j] = g / h;
                        f += e[j] * a[(size_t)i * n + j];
                    }
                    const double hh = f / (h + h);
                    for (int j = 0; j <= l; ++j) {
                        f = a[(size_t)i * n + j];
                        g = e[j] - hh * f;
                        e[j] = g;
                        for (int k = 0; k <= j; ++k) a[(size_t)j * n + k] -= (f * e[k] + g * a[(size_t)i * n + k]);
                    }
                }
            } else {
                e[i] = a[(size_t)i * n + l];
            }
        }
        for (int i = 0; i < n; ++i) d[i] = a[(size_t)i * n + i];
        for (int i = 1; i < n; ++i) e[i - 1] = e[i];
        e[n - 1] = 0.0;
        for (int l = 0; l < n; ++l) {
            int iter = 0, m;
            do {
                for (m = l; m < n - 1; ++m) {
                    const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                    if (std::fabs(e[m]) + dd == dd) break;
                }
                if (m != l) {
                    if (iter++ == 60) break;
                    double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                    double r = std::sqrt(g * g + 1.0);
                    g = d[m] - d[l] + e[l] / (g + ((g >= 0.0) ? std::fabs(r) : -std::fabs(r)));
                    double s = 1.0, c = 1.0, p = 0.0;
                    int i;
                    for (i = m - 1; i >= l; --i) {
                        double f = s * e[i];
                        const double b = c * e[i];
                        r = std::sqrt(f * f + g * g);
                        e[i + 1] = r;
                        if (r == 0.0) {
                            d[i + 1] -= p;
                            e[m] = 0.0;
                            break;
                        }
                        s = f / r;
                        c = g / r;
                        g = d[i + 1] - p;
                        r = (d[i] - g) * s + 2.0 * c * b;
                        p = s * r;
                        d[i + 1] = g + p;
                        g = c * r - b;
                    }
                    if (r == 0.0 && i >= l) continue;
                    d[l] -= p;
                    e[l] = g;
                    e[m] = 0.0;
                }
            } while (m != l);
        }
    }

    // Gauss-Jordan elimination with partial pivoting
    bool invert(const std::vector<double>& A, std::vector<double>& out) const {
        const int n = D;
        std::vector<double> a((size_t)n * 2 * n, 0.0);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) a[(size_t)i * 2 * n + j] = A[(size_t)i * n + j];
            a[(size_t)i * 2 * n + n + i] = 1.0;
        }
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int r = col + 1; r < n; ++r)
                if (std::fabs(a[(size_t)r * 2 * n + col]) > std::fabs(a[(size_t)piv * 2 * n + col])) piv = r;
            if (a[(size_t)piv * 2 * n + col] == 0.0) return false;
            if (piv != col)
                for (int k = 0; k < 2 * n; ++k) std::swap(a[(size_t)col * 2 * n + k], a[(size_t)piv * 2 * n + k]);
            const double d = a[(size_t)col * 2 * n + col];
            for (int k = 0; k < 2 * n; ++k) a[(size_t)col * 2 * n + k] /= d;
            for (int r = 0; r < n; ++r) {
                if (r == col) continue;
                const double f = a[(size_t)r * 2 * n + col];
                if (f == 0.0) continue;
                for (int k = 0; k < 2 * n; ++k) a[(size_t)r * 2 * n + k] -= f * a[(size_t)col * 2 * n + k];
            }
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = a[(size_t)i * 2 * n + n + j];
        return true;
    }
};

}  // namespace smcmc
