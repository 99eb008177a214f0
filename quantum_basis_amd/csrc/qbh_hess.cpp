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


// The per-step stop test of lanczos() (src/lanczos.cc:228-247) needs only the few lowest Ritz values and the LAST
// component of the lowest Ritz vector (accuracy = |b_m s_{m-1}|).  The reference solves the whole m x m problem with
// dstedc every step (K9 in SURVEY 2.3); even the O(m^2) QL above costs 4 ms at m = 300 and 40 ms at m = 1000 per
// step on the host -- as much as the device work of a step.  Here: bisection on the Sturm count for the nev lowest
// eigenvalues (O(m) per evaluation, ~55 evaluations each, to ~1 ulp of the spectrum's scale) and the eigenvector's
// last component from the twisted factorisation (top-down and bottom-up pivots meeting at the smallest |gamma|,
// as LAPACK's dlar1v does), O(m).  ritz[0..nev) ascending; *zlast0 = |last component| of the unit eigenvector of ritz[0].
int tridiag_lowest(int64_t m, const double *a, const double *b1, int nev, double *ritz, double *zlast0)
{
    if (m <= 0 || nev <= 0) return QBH_EINVAL;
    if (nev > m) nev = (int)m;
    double gl = a[0], gu = a[0], bmax = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        const double r = (i > 0 ? std::fabs(b1[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(b1[i]) : 0.0);
        gl = std::min(gl, a[i] - r);
        gu = std::max(gu, a[i] + r);
        if (i + 1 < m) bmax = std::max(bmax, b1[i] * b1[i]);
    }
    const double eps = 2.220446049250313e-16;
    const double scale = std::max(std::fabs(gl), std::fabs(gu));
    const double pivmin = std::max(2.2250738585072014e-308 * std::max(1.0, bmax) * 4.0, 1e-300);
    gl -= 2.0 * eps * scale * (double)m + 2.0 * pivmin;
    gu += 2.0 * eps * scale * (double)m + 2.0 * pivmin;
    auto count_below = [&](double x) -> int64_t {          // number of eigenvalues < x
        int64_t cnt = 0;
        double q = a[0] - x;
        if (std::fabs(q) < pivmin) q = -pivmin;
        if (q < 0.0) cnt++;
        for (int64_t k = 1; k < m; ++k) {
            q = a[k] - x - b1[k - 1] * b1[k - 1] / q;
            if (std::fabs(q) < pivmin) q = -pivmin;
            if (q < 0.0) cnt++;
        }
        return cnt;
    };
    // eight Sturm sequences per sweep (independent division chains: the loop is latency-bound, so eight cost what
    // one does): each sweep cuts the bracket ninefold
    constexpr int NS = 8;
    auto count_below8 = [&](const double *x, int64_t *cnt) {
        double q[NS];
        for (int s2 = 0; s2 < NS; ++s2) {
            q[s2] = a[0] - x[s2];
            if (std::fabs(q[s2]) < pivmin) q[s2] = -pivmin;
            cnt[s2] = q[s2] < 0.0 ? 1 : 0;
        }
        for (int64_t k = 1; k < m; ++k) {
            const double ak = a[k], bb = b1[k - 1] * b1[k - 1];
            for (int s2 = 0; s2 < NS; ++s2) {
                double t = ak - x[s2] - bb / q[s2];
                if (std::fabs(t) < pivmin) t = -pivmin;
                q[s2] = t;
                cnt[s2] += t < 0.0 ? 1 : 0;
            }
        }
    };
    double lo_prev = gl;
    for (int j = 0; j < nev; ++j) {
        double lo = lo_prev, hi = gu;
        for (int it = 0; it < 64; ++it) {
            if (hi - lo <= eps * std::max(std::fabs(lo), std::fabs(hi))) break;
            double x[NS];
            int64_t cnt[NS];
            const double h = (hi - lo) / (NS + 1);
            for (int s2 = 0; s2 < NS; ++s2) x[s2] = lo + h * (s2 + 1);
            if (!(x[0] > lo && x[NS - 1] < hi)) {          // bracket only a few ulps wide: finish by plain bisection
                const double mid = 0.5 * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                if (count_below(mid) >= j + 1) hi = mid;
                else lo = mid;
                continue;
            }
            count_below8(x, cnt);
            // eigenvalue j lies between the last point with count <= j and the first with count >= j + 1
            int first = NS;
            for (int s2 = 0; s2 < NS; ++s2)
                if (cnt[s2] >= j + 1) {
                    first = s2;
                    break;
                }
            if (first < NS) hi = x[first];
            if (first > 0) lo = x[first - 1];
        }
        ritz[j] = 0.5 * (lo + hi);
        lo_prev = lo;                                      // the next eigenvalue is not below this one
    }
    if (zlast0) {
        const double th = ritz[0];
        if (m == 1) {
            *zlast0 = 1.0;
            return QBH_OK;
        }
        std::vector<double> qp((size_t)m), qm((size_t)m);
        qp[0] = a[0] - th;
        for (int64_t k = 1; k < m; ++k) {
            double d = qp[(size_t)k - 1];
            if (std::fabs(d) < pivmin) d = d < 0.0 ? -pivmin : pivmin;
            qp[(size_t)k] = (a[k] - th) - b1[k - 1] * b1[k - 1] / d;
        }
        qm[(size_t)m - 1] = a[m - 1] - th;
        for (int64_t k = m - 2; k >= 0; --k) {
            double d = qm[(size_t)k + 1];
            if (std::fabs(d) < pivmin) d = d < 0.0 ? -pivmin : pivmin;
            qm[(size_t)k] = (a[k] - th) - b1[k] * b1[k] / d;
        }
        int64_t r = 0;
        double gbest = std::fabs(qp[0] + qm[0] - (a[0] - th));
        for (int64_t k = 1; k < m; ++k) {
            const double g = std::fabs(qp[(size_t)k] + qm[(size_t)k] - (a[k] - th));
            if (g < gbest) {
                gbest = g;
                r = k;
            }
        }
        // z_r = 1; upwards with the top-down pivots, downwards with the bottom-up pivots; rescale against overflow
        double nrm2 = 1.0, z = 1.0, zm = (r == m - 1) ? 1.0 : 0.0;
        for (int64_t k = r - 1; k >= 0; --k) {
            double d = qp[(size_t)k];
            if (std::fabs(d) < pivmin) d = d < 0.0 ? -pivmin : pivmin;
            z = -b1[k] * z / d;
            nrm2 += z * z;
            if (!(nrm2 < 1e280)) return QBH_ENOCONV;
        }
        z = 1.0;
        for (int64_t k = r + 1; k < m; ++k) {
            double d = qm[(size_t)k];
            if (std::fabs(d) < pivmin) d = d < 0.0 ? -pivmin : pivmin;
            z = -b1[k - 1] * z / d;
            nrm2 += z * z;
            if (!(nrm2 < 1e280)) return QBH_ENOCONV;       // same guard as the upward sweep: the caller falls back to QL
            if (k == m - 1) zm = z;
        }
        *zlast0 = std::fabs(zm) / std::sqrt(nrm2);
        if (!std::isfinite(*zlast0)) return QBH_ENOCONV;
    }
    return QBH_OK;
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
