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

/* Eigenvalues of a symmetric matrix (n x n, row-major; natural order of the algorithm, not sorted): Householder
 * reduction to tridiagonal form followed by QL iterations with implicit shifts.  Stands in for the values
 * TMatrixD::EigenVectors(TVectorD&) hands TSimpleHMC::UpdateErrorMatrix (TSimpleHMC.H:766), which only looks at
 * their signs and extremes.  PARITY UNPINNED against ROOT like the rest of this file. */
static void oracle_sym_eigenvalues(int n, const double* Ain, double* d) {
    double* a = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double* e = (double*)calloc((size_t)n + 1, sizeof(double));
    memcpy(a, Ain, sizeof(double) * (size_t)n * (size_t)n);
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += fabs(a[i * n + k]);
            if (scale == 0.0) {
                e[i] = a[i * n + l];
            } else {
                for (int k = 0; k <= l; ++k) {
                    a[i * n + k] /= scale;
                    h += a[i * n + k] * a[i * n + k];
                }
                double f = a[i * n + l];
                double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                a[i * n + l] = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    g = 0.0;
                    for (int k = 0; k <= j; ++k) g += a[j * n + k] * a[i * n + k];
                    for (int k = j + 1; k <= l; ++k) g += a[k * n + j] * a[i * n + k];
                    e[j] = g / h;
                    f += e[j] * a[i * n + j];
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = a[i * n + j];
                    g = e[j] - hh * f;
                    e[j] = g;
                    for (int k = 0; k <= j; ++k) a[j * n + k] -= (f * e[k] + g * a[i * n + k]);
                }
            }
        } else {
            e[i] = a[i * n + l];
        }
    }
    for (int i = 0; i < n; ++i) d[i] = a[i * n + i];
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) + dd == dd) break;
            }
            if (m != l) {
                if (iter++ == 60) break;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = sqrt(g * g + 1.0);
                g = d[m] - d[l] + e[l] / (g + ((g >= 0.0) ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = sqrt(f * f + g * g);
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
    free(a); free(e);
}

/* A^-1 by Gauss-Jordan elimination with partial pivoting (TMatrixD::Invert at TSimpleHMC.H:262, 850 and
 * TDummyLogLikelihood.H:141; ROOT's own factorisation is unpinned).  Returns 0 for a singular matrix. */
static int oracle_invert(int n, const double* A, double* out) {
    double* a = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n * 2);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * 2 * n + j] = A[i * n + j];
            a[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabs(a[r * 2 * n + col]) > fabs(a[piv * 2 * n + col])) piv = r;
        if (a[piv * 2 * n + col] == 0.0) { free(a); return 0; }
        if (piv != col)
            for (int k = 0; k < 2 * n; ++k) { double t = a[col * 2 * n + k]; a[col * 2 * n + k] = a[piv * 2 * n + k]; a[piv * 2 * n + k] = t; }
        double d = a[col * 2 * n + col];
        for (int k = 0; k < 2 * n; ++k) a[col * 2 * n + k] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            double f = a[r * 2 * n + col];
            if (f == 0.0) continue;
            for (int k = 0; k < 2 * n; ++k) a[r * 2 * n + k] -= f * a[col * 2 * n + k];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) out[i * n + j] = a[i * 2 * n + n + j];
    free(a);
    return 1;
}

#endif
