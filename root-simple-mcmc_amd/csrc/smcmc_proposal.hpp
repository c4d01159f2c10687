// smcmc_proposal.hpp -- host side of the shared adaptive proposal.
//
// The ensemble shares one centre / covariance / decomposition (the
// TProposeAdaptiveStep members fCentralPoint, fCurrentCov, fDecomposition of
// reference TSimpleMCMC.H:1841-1893); the per-chain scalars live on the device.
// This class owns the shared half and runs, on the host, the rare O(D^3) part of
// the reference algorithm: ResetProposal (TSimpleMCMC.H:1396-1494),
// UpdateProposal with its decomposition ladder (:1009-1390) and the running
// centre/covariance averages (:1780-1820) fed with a batch of pooled moments.
// ROOT's TDecompChol / TMatrixDSymEigen are replaced by the row-ordered Cholesky
// and a cyclic Jacobi solver below.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <vector>

namespace smcmc {

struct CorrelationHint { int d1, d2; double c; };

enum class UpdateStatus { Ok, InvalidTrace, IllegalProposalType, UserCorrelationsFailed, TargetNotSet };

class SharedProposal {
public:
    explicit SharedProposal(int dim)
        : D(dim), lastPoint(dim, 0.0), centre(dim, 0.0), cov((size_t)dim * dim, 0.0),
          decomp((size_t)dim * dim, 0.0), ptype(dim, 0), param1(dim, 0.0), param2(dim, 0.0) {
        maxCorrelation = 1.0 - std::sqrt(DBL_EPSILON);   // TSimpleMCMC.H:652-653
    }

    const int D;
    std::vector<double> lastPoint, centre, cov, decomp;
    std::vector<int> ptype;
    std::vector<double> param1, param2;
    std::vector<CorrelationHint> correlations;
    // constructor defaults TSimpleMCMC.H:642-651
    double centreTrials = 0.0, covTrials = 0.0, covDeweight = 0.5, covWindow = -1;
    bool covFrozen = false;
    int successes = 0, nextUpdate = -1;
    double acceptance = 0.0, acceptanceTrials = 0.0, acceptanceDeweight = 0.5;
    double acceptanceWindow = -1, rigidity = 2.0, target = -1, sigma = 0.0, sigmaTrace = 0.0;
    double maxCorrelation;
    bool initialized = false, decompFull = false;
    int updateCount = 0, lastPath = 0;
    double lastSigmaScale = 1.0;   // sqrt(old trace / new trace) of the latest update(): what every chain's sigma takes

    double& C(int i, int j) { return cov[(size_t)i * D + j]; }
    double& U(int i, int j) { return decomp[(size_t)i * D + j]; }

    double trace() const {                                   // GetCovarianceTrace :961-967
        double t = 0.0;
        for (int i = 0; i < D; ++i) t += cov[(size_t)i * D + i];
        return t;
    }

    // InitializeState :1679-1714
    UpdateStatus initialize(const double* start) {
        if (initialized) return UpdateStatus::Ok;
        initialized = true;
        std::copy(start, start + D, lastPoint.begin());
        if (acceptanceWindow < 0) acceptanceWindow = std::pow(1.0 * D, 1.5) + 1000;
        nextUpdate = (int)acceptanceWindow;
        if (target < 1E-4) target = (D > 4) ? 0.234 : 0.44;
        return reset();
    }

    // ResetProposal :1396-1494
    UpdateStatus reset() {
        successes = 0;
        if (sigma < 0.01 * std::sqrt(1.0 / D)) sigma = std::sqrt(1.0 / D);
        for (int i = 0; i < D; ++i) {
            for (int j = i; j < D; ++j) {
                if (i != j) { C(i, j) = C(j, i) = 0.0; continue; }
                if (ptype[i] == 0 && param1[i] > 0) C(i, i) = param1[i];
                else if (ptype[i] == 1) {
                    double delta = param1[i];
                    delta -= param2[i];
                    C(i, i) = delta * delta / 12.0;
                } else C(i, i) = 1.0;
            }
        }
        for (const CorrelationHint& h : correlations) {
            if (h.d1 == h.d2) continue;
            const double v1 = C(h.d1, h.d1), v2 = C(h.d2, h.d2);
            C(h.d1, h.d2) = C(h.d2, h.d1) = h.c * std::sqrt(v1) * std::sqrt(v2);
        }
        sigmaTrace = trace();
        const int minWindow = 100 + 4 * D;
        if (covWindow < minWindow) {
            covWindow = D;
            covWindow *= D;
            covWindow *= D;
            covWindow += minWindow;
            covWindow = std::min(covWindow, std::sqrt(1.0 / DBL_EPSILON));
        }
        if (target < 0.0) return UpdateStatus::TargetNotSet;
        acceptance = target;
        acceptanceTrials = std::min(10.0, 0.5 * acceptanceWindow);
        centre = lastPoint;
        centreTrials = std::max(centreTrials, 1.0);
        return update(true);
    }

    // UpdateProposal :1009-1390.  `sigma`, `acceptanceTrials` here are the shared
    // template values; the engine applies the same rescale / de-weighting per chain.
    UpdateStatus update(bool fromReset) {
        ++updateCount;
        const double currentTrace = trace();
        if (currentTrace <= 0) return UpdateStatus::InvalidTrace;
        const double sigmaScale = std::sqrt(sigmaTrace / currentTrace);
        sigma = sigma * sigmaScale;
        sigmaTrace = currentTrace;
        const double maxUp = (double)D * (double)D;
        const double up = 0.5 * successes;
        nextUpdate = (int)(acceptanceWindow + maxUp - maxUp / (up + 1.0));
        if (covDeweight > 0.0) {
            if (covDeweight > 1.0) covDeweight = 1.0;
            const double w = 1.0 - covDeweight;
            covTrials = std::max(1.0, w * covTrials);
            covTrials = std::min(covTrials, w * covWindow);
            centreTrials = std::max(1.0, w * centreTrials);
            centreTrials = std::min(centreTrials, w * covWindow);
        }
        if (acceptanceDeweight > 0.0) {
            if (acceptanceDeweight > 1.0) acceptanceDeweight = 1.0;
            const double w = 1.0 - acceptanceDeweight;
            acceptanceTrials = std::max(1.0, w * acceptanceTrials);
            acceptanceTrials = std::min(acceptanceTrials, w * acceptanceWindow);
        }
        const UpdateStatus st = decompose(fromReset);
        lastSigmaScale = sigmaScale;   // after the ladder: a reset on its last rung runs update() again
        return st;
    }

    // The rest of update() after the device has done the running averages, the scalar half and found that the plain
    // Cholesky decomposition fails (smcmc_pooled_update.hip.h): the ladder of :1134-1389 on the host.
    UpdateStatus finishUpdateOnHost(double sigmaScale) {
        const UpdateStatus st = decompose(false);
        lastSigmaScale = sigmaScale;
        return st;
    }

    // Batch form of the running averages :1780-1820.  M is the packed moment
    // vector about c0 == centre: row i <= D, column j <= i, row D = {sum y_j, n}.
    void absorbMoments(const double* M, bool updateCovariance) {
        const double* S1 = M + (size_t)D * (D + 1) / 2;
        const double n = S1[D];
        if (!(n > 0.0)) return;
        std::vector<double> delta(D);
        for (int d = 0; d < D; ++d) {
            delta[d] = S1[d] / (centreTrials + n);
            centre[d] = centre[d] + delta[d];
        }
        centreTrials = std::min(covWindow, centreTrials + n);
        if (!updateCovariance) return;
        for (int i = 0; i < D; ++i) {
            for (int j = 0; j <= i; ++j) {
                double b = M[(size_t)i * (i + 1) / 2 + j];
                b -= S1[i] * delta[j];
                b -= delta[i] * S1[j];
                b += (n * delta[i]) * delta[j];
                double v = C(i, j);
                v *= covTrials;
                v += b;
                v /= covTrials + n;
                C(i, j) = v;
                C(j, i) = v;
            }
        }
        covTrials = std::min(covWindow, covTrials + n);
    }

private:
    bool cholesky() {
        // row-ordered A = U^T U; a pivot that is not a positive finite number fails
        std::vector<double>& R = decomp;
        R = cov;
        for (int c = 0; c < D; ++c) {
            double piv = R[(size_t)c * D + c];
            for (int r = 0; r < c; ++r) {
                const double v = R[(size_t)r * D + c];
                piv -= v * v;
            }
            if (!(piv > 0.0) || !std::isfinite(piv)) return false;
            piv = std::sqrt(piv);
            R[(size_t)c * D + c] = piv;
            for (int j = c + 1; j < D; ++j) {
                double v = R[(size_t)c * D + j];
                for (int r = 0; r < c; ++r) v -= R[(size_t)r * D + j] * R[(size_t)r * D + c];
                R[(size_t)c * D + j] = v / piv;
            }
        }
        for (int r = 1; r < D; ++r)
            for (int c = 0; c < r; ++c) R[(size_t)r * D + c] = 0.0;
        decompFull = false;
        return true;
    }

    // cyclic Jacobi; columns of vec are eigenvectors, val sorted descending
    void symmetricEigen(std::vector<double>& vec, std::vector<double>& val) const {
        std::vector<double> A = cov;
        vec.assign((size_t)D * D, 0.0);
        for (int i = 0; i < D; ++i) vec[(size_t)i * D + i] = 1.0;
        for (int sweep = 0; sweep < 100; ++sweep) {
            double off = 0.0;
            for (int p = 0; p < D; ++p)
                for (int q = p + 1; q < D; ++q) off += A[(size_t)p * D + q] * A[(size_t)p * D + q];
            if (!(off > 0.0)) break;
            for (int p = 0; p < D; ++p) {
                for (int q = p + 1; q < D; ++q) {
                    const double apq = A[(size_t)p * D + q];
                    if (apq == 0.0 || !std::isfinite(apq)) continue;
                    const double theta = (A[(size_t)q * D + q] - A[(size_t)p * D + p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                    for (int k = 0; k < D; ++k) {
                        const double akp = A[(size_t)k * D + p], akq = A[(size_t)k * D + q];
                        A[(size_t)k * D + p] = c * akp - s * akq;
                        A[(size_t)k * D + q] = s * akp + c * akq;
                    }
                    for (int k = 0; k < D; ++k) {
                        const double apk = A[(size_t)p * D + k], aqk = A[(size_t)q * D + k];
                        A[(size_t)p * D + k] = c * apk - s * aqk;
                        A[(size_t)q * D + k] = s * apk + c * aqk;
                    }
                    for (int k = 0; k < D; ++k) {
                        const double vkp = vec[(size_t)k * D + p], vkq = vec[(size_t)k * D + q];
                        vec[(size_t)k * D + p] = c * vkp - s * vkq;
                        vec[(size_t)k * D + q] = s * vkp + c * vkq;
                    }
                }
            }
        }
        val.resize(D);
        for (int i = 0; i < D; ++i) val[i] = A[(size_t)i * D + i];
        for (int i = 0; i < D; ++i) {
            int best = i;
            for (int j = i + 1; j < D; ++j) if (val[j] > val[best]) best = j;
            if (best == i) continue;
            std::swap(val[i], val[best]);
            for (int k = 0; k < D; ++k) std::swap(vec[(size_t)k * D + i], vec[(size_t)k * D + best]);
        }
    }

    // the ladder of :1097-1389
    UpdateStatus decompose(bool fromReset) {
        const double minVar = DBL_EPSILON;
        lastPath = 0;
        if (cholesky()) return UpdateStatus::Ok;

        for (int i = 0; i < D; ++i) {                        // :1134-1183
            double expected = 1.0;
            if (ptype[i] == 0) { if (param1[i] > 0) expected = param1[i]; }
            else if (ptype[i] == 1) {
                expected = param2[i];
                expected -= param1[i];
                expected = expected * expected / 12.0;
            } else return UpdateStatus::IllegalProposalType;
            double& v = C(i, i);
            if (!std::isfinite(v)) v = expected;
            if (v < 0.0) v = minVar * expected;
            if (v < minVar * expected) v = minVar * expected;
            if (v < minVar) v = minVar;
        }
        for (int i = 0; i < D; ++i) {                        // :1187-1217
            for (int j = i + 1; j < D; ++j) {
                double corr = C(i, j);
                corr /= std::sqrt(C(i, i));
                corr /= std::sqrt(C(j, j));
                if (!std::isfinite(corr)) corr = 0.0;
                if (std::fabs(corr) > maxCorrelation) corr = (corr > 0.0) ? maxCorrelation : -maxCorrelation;
                double v = corr;
                v *= std::sqrt(C(i, i));
                v *= std::sqrt(C(j, j));
                C(i, j) = v;
                C(j, i) = v;
            }
        }
        lastPath = 1;
        if (cholesky()) return UpdateStatus::Ok;

        {                                                    // :1252-1321
            std::vector<double> vec, val;
            symmetricEigen(vec, val);
            double eigenSum = 0.0;
            for (int i = 0; i < D; ++i) if (!(val[i] < 0.0)) eigenSum += val[i];
            double minAxis = 1.0 - maxCorrelation;
            if (minAxis < minVar) minAxis = minVar;
            minAxis = minAxis * val[0];
            for (int i = 0; i < D; ++i) {
                const double rms = std::sqrt(std::max(minAxis, val[i]));
                for (int j = 0; j < D; ++j) U(i, j) = rms * vec[(size_t)j * D + i];
            }
            lastPath = 2;
            decompFull = true;
            if (eigenSum > 1E-6) return UpdateStatus::Ok;
        }

        double step = DBL_EPSILON;                           // :1335-1377
        for (int i = 0; i < D; ++i) step = std::max(step, C(i, i));
        step *= 1E-4;
        double dec = 1.0;
        lastPath = 3;
        for (int trial = 0; trial < 10; ++trial) {
            dec *= 0.84;
            for (int i = 0; i < D; ++i) {
                C(i, i) += step;
                for (int j = i + 1; j < D; ++j) {
                    const double v = dec * C(i, j);
                    C(i, j) = v;
                    C(j, i) = v;
                }
            }
            if (cholesky()) return UpdateStatus::Ok;
        }
        if (fromReset) return UpdateStatus::UserCorrelationsFailed;   // :1383-1386
        lastPath = 4;
        const UpdateStatus st = reset();                     // :1389
        lastPath = 4;
        return st;
    }
};

}  // namespace smcmc
