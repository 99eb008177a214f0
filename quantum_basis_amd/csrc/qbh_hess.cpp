// qbh_hess.cpp -- host eigen-solver of the Lanczos tridiagonal (replaces hess_eigen,
// src/lanczos.cc:355-390, which calls LAPACKE_dstedc('I')).  Stays on the host: the matrix
// is m x m with m <= maxit (default 1000), solved once per Lanczos step (K9 in SURVEY §2.3).
//
// Implicit-shift QL with Wilkinson shifts.  Two flavours:
//   * full:    all eigenvalues + the full eigenvector matrix (the public qbh_hess_eigen)
//   * lastrow: all eigenvalues + only the LAST component of every eigenvector, which is all
//              the per-step stop test needs (accuracy = |b_m * s[m-1]|, src/lanczos.cc:231);
//              O(m^2) per call instead of O(m^3).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "qbh_internal.hpp"

namespace qbh {

namespace {

// d[0..n) diagonal, e[0..n) with e[i] coupling (i, i+1) and e[n-1] unused.
// rows: number of rows of Z that are tracked; z is rows x n column-major (ld = rows).
int ql_implicit(int64_t n, double *d, double *e, double *z, int64_t rows)
{
    const double eps = 2.220446049250313e-16;
    if (n <= 1) return 0;
    e[n - 1] = 0.0;
    for (int64_t l = 0; l < n; ++l) {
        int iter = 0;
        int64_t m;
        for (;;) {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m == l) break;
            if (++iter > 200) return 1;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
            double s = 1.0, c = 1.0, p = 0.0;
            int64_t i;
            bool underflow = false;
            for (i = m - 1; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = std::hypot(f, g);
                e[i + 1] = r;
                if (r == 0.0) {
                    d[i + 1] -= p;
                    e[m] = 0.0;
                    underflow = true;
                    break;
                }
                s = f / r;
                c = g / r;
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                double *zi = z + (size_t)i * rows;
                double *zi1 = z + (size_t)(i + 1) * rows;
                for (int64_t k = 0; k < rows; ++k) {
                    f = zi1[k];
                    zi1[k] = s * zi[k] + c * f;
                    zi[k] = c * zi[k] - s * f;
                }
            }
            if (underflow) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    return 0;
}

}  // namespace

// a[0..m): diagonal, b1[0..m-1): off-diagonal (b1[i] couples i,i+1).
// w[m] eigenvalues (unsorted), z[m*m] column-major eigenvectors.
int tridiag_eigen_full(int64_t m, const double *a, const double *b1, double *w, double *z)
{
    std::vector<double> e((size_t)m, 0.0);
    for (int64_t j = 0; j < m; ++j) w[j] = a[j];
    for (int64_t j = 0; j + 1 < m; ++j) e[j] = b1[j];
    for (int64_t j = 0; j < m * m; ++j) z[j] = 0.0;
    for (int64_t j = 0; j < m; ++j) z[j * m + j] = 1.0;
    return ql_implicit(m, w, e.data(), z, m) ? QBH_ENOCONV : QBH_OK;
}

// zlast[j] = last component of eigenvector j (row m-1 of Z).
int tridiag_eigen_lastrow(int64_t m, const double *a, const double *b1, double *w, double *zlast)
{
    std::vector<double> e((size_t)m, 0.0);
    for (int64_t j = 0; j < m; ++j) w[j] = a[j];
    for (int64_t j = 0; j + 1 < m; ++j) e[j] = b1[j];
    for (int64_t j = 0; j < m; ++j) zlast[j] = 0.0;
    zlast[m - 1] = 1.0;
    return ql_implicit(m, w, e.data(), zlast, 1) ? QBH_ENOCONV : QBH_OK;
}


// Cyclic Jacobi for the small dense real symmetric projected matrix of the thick-restart Lanczos
// (m <= 32: tridiagonal plus the arrowhead row left by a restart).  a: m*m column-major, destroyed;
// w[m] ascending eigenvalues; z[m*m] column-major eigenvectors.
int symmetric_eigen_jacobi(int m, double *a, double *w, double *z)
{
    for (int i = 0; i < m * m; ++i) z[i] = 0.0;
    for (int i = 0; i < m; ++i) z[i * m + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < m; ++p) {
            diag += a[p * m + p] * a[p * m + p];
            for (int q = p + 1; q < m; ++q) off += a[q * m + p] * a[q * m + p];
        }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                const double apq = a[q * m + p];
                if (apq == 0.0) continue;
                const double theta = (a[q * m + q] - a[p * m + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < m; ++k) {                 // rotate columns p, q of A
                    const double akp = a[p * m + k], akq = a[q * m + k];
                    a[p * m + k] = c * akp - s * akq;
                    a[q * m + k] = s * akp + c * akq;
                }
                for (int k = 0; k < m; ++k) {                 // rotate rows p, q of A
                    const double apk = a[k * m + p], aqk = a[k * m + q];
                    a[k * m + p] = c * apk - s * aqk;
                    a[k * m + q] = s * apk + c * aqk;
                }
                for (int k = 0; k < m; ++k) {                 // accumulate eigenvectors
                    const double zkp = z[p * m + k], zkq = z[q * m + k];
                    z[p * m + k] = c * zkp - s * zkq;
                    z[q * m + k] = s * zkp + c * zkq;
                }
            }
    }
    // sort ascending (selection sort, m is tiny)
    for (int i = 0; i < m; ++i) w[i] = a[i * m + i];
    for (int i = 0; i < m - 1; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (w[j] < w[k]) k = j;
        if (k != i) {
            std::swap(w[i], w[k]);
            for (int r = 0; r < m; ++r) std::swap(z[i * m + r], z[k * m + r]);
        }
    }
    return QBH_OK;
}

}  // namespace qbh
