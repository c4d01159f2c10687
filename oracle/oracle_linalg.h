/* oracle_linalg.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Dense helpers standing in for the slice of ROOT linear algebra the reference
 * hot path calls.  ROOT is an un-vendored, version-unpinned dependency of the
 * reference ("5.34+ and 6.06+", /root/reference/README.md:3-4) and is absent
 * from this image, so these restate the published algorithms:
 *   - TDecompChol::Decompose / GetU  (call sites TSimpleMCMC.H:1103-1106,
 *     1220-1225, 1357-1362): row-ordered Cholesky A = U^T U, fails on a
 *     pivot <= 0.
 *   - TMatrixDSymEigen (TSimpleMCMC.H:1261-1263): here cyclic Jacobi, eigenvalues
 *     sorted descending as the reference assumes (TSimpleMCMC.H:1287).
 * PARITY UNPINNED against ROOT for last-bit results of both (no golden vectors
 * exist in the reference).
 */
#ifndef ORACLE_LINALG_H_SEEN
#define ORACLE_LINALG_H_SEEN

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* A (n x n, row-major, symmetric) -> U upper-triangular with A = U^T U.
 * Returns 1 on success, 0 when a pivot is not a positive finite number. */
static int oracle_cholesky_upper(int n, const double* A, double* U) {
    memcpy(U, A, sizeof(double) * (size_t)n * (size_t)n);
    for (int icol = 0; icol < n; ++icol) {
        double ujj = U[icol * n + icol];
        for (int irow = 0; irow < icol; ++irow) {
            double v = U[irow * n + icol];
            ujj -= v * v;
        }
        if (!(ujj > 0.0) || !isfinite(ujj)) return 0;
        ujj = sqrt(ujj);
        U[icol * n + icol] = ujj;
        for (int j = icol + 1; j < n; ++j) {
            double v = U[icol * n + j];
            for (int i = 0; i < icol; ++i) v -= U[i * n + j] * U[i * n + icol];
            U[icol * n + j] = v / ujj;
        }
    }
    for (int irow = 0; irow < n; ++irow)
        for (int icol = 0; icol < irow; ++icol) U[irow * n + icol] = 0.0;
    return 1;
}

/* Cyclic Jacobi for a symmetric matrix.  vec(:,k) (column k) is the k-th
 * eigenvector, val[k] descending. */
static void oracle_sym_eigen(int n, const double* Ain, double* vec, double* val) {
    double* A = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    memcpy(A, Ain, sizeof(double) * (size_t)n * (size_t)n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) vec[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (!(off > 0.0)) break;
        for (int p = 0; p < n; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0 || !isfinite(apq)) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0);
                double s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = vec[k * n + p], vkq = vec[k * n + q];
                    vec[k * n + p] = c * vkp - s * vkq;
                    vec[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) val[i] = A[i * n + i];
    /* selection sort, descending, carrying columns */
    for (int i = 0; i < n; ++i) {
        int best = i;
        for (int j = i + 1; j < n; ++j) if (val[j] > val[best]) best = j;
        if (best != i) {
            double tv = val[i]; val[i] = val[best]; val[best] = tv;
            for (int k = 0; k < n; ++k) {
                double t = vec[k * n + i]; vec[k * n + i] = vec[k * n + best]; vec[k * n + best] = t;
            }
        }
    }
    free(A);
}

#endif
