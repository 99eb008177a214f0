/*
 * qb_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See qb_oracle.h.
 *
 * CPU restatement of the reference's CSR x vector / Lanczos / CG path.
 * Written from the behaviour of /root/reference (cited per function); the
 * third-party arithmetic the reference delegates to Intel MKL (unpinned
 * version, closed source: mkl_sparse_z_mv, cblas_z*, LAPACKE_dstedc) is
 * restated from the published definitions of those routines (BLAS level-1
 * semantics; Hermitian-upper CSR product; symmetric tridiagonal eigenproblem
 * solved here by implicit-shift QL).
 */
#include "qb_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OMP_MIN_N 100000

int qbo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Parallel copy whose only purpose is NUMA placement for the timed CPU baseline: dst must be freshly
 * allocated (untouched pages); every thread copies -- and thereby first-touches -- one contiguous slice, the
 * same static partition the SpMV and BLAS-1 loops below use, so pages end up spread over the memory
 * controllers of all sockets instead of on the node of the one thread that filled the array. */
void qbo_first_touch_copy(int64_t nbytes, const void *src, void *dst)
{
    const int64_t page = 4096, npages = (nbytes + page - 1) / page;
    int64_t i;
    #pragma omp parallel for schedule(static)
    for (i = 0; i < npages; ++i) {
        const int64_t b = i * page, e = (b + page < nbytes) ? b + page : nbytes;
        memcpy((char *)dst + b, (const char *)src + b, (size_t)(e - b));
    }
}

/* ------------------------------------------------------------------------- */
/* BLAS-1 (src/lanczos.cc:10-53 wrap cblas_zaxpy/zcopy/dznrm2/zscal/zdotc)    */
/* ------------------------------------------------------------------------- */
double qbo_nrm2(int64_t n, const double *x)
{
    double s = 0.0;
    int64_t j;
    #pragma omp parallel for reduction(+:s) schedule(static) if (n > OMP_MIN_N)
    for (j = 0; j < 2 * n; j++) s += x[j] * x[j];
    return sqrt(s);
}

/* res = conj(x) . y */
void qbo_dotc(int64_t n, const double *x, const double *y, double *res)
{
    double sr = 0.0, si = 0.0;
    int64_t j;
    #pragma omp parallel for reduction(+:sr,si) schedule(static) if (n > OMP_MIN_N)
    for (j = 0; j < n; j++) {
        double xr = x[2*j], xi = x[2*j+1], yr = y[2*j], yi = y[2*j+1];
        sr += xr * yr + xi * yi;
        si += xr * yi - xi * yr;
    }
    res[0] = sr;
    res[1] = si;
}

/* y += alpha * x, alpha complex */
static void zaxpy(int64_t n, double ar, double ai, const double *x, double *y)
{
    int64_t j;
    #pragma omp parallel for schedule(static) if (n > OMP_MIN_N)
    for (j = 0; j < n; j++) {
        double xr = x[2*j], xi = x[2*j+1];
        y[2*j]   += ar * xr - ai * xi;
        y[2*j+1] += ar * xi + ai * xr;
    }
}

/* x *= alpha, alpha complex */
static void zscal(int64_t n, double ar, double ai, double *x)
{
    int64_t j;
    #pragma omp parallel for schedule(static) if (n > OMP_MIN_N)
    for (j = 0; j < n; j++) {
        double xr = x[2*j], xi = x[2*j+1];
        x[2*j]   = ar * xr - ai * xi;
        x[2*j+1] = ar * xi + ai * xr;
    }
}

static void zcopy(int64_t n, const double *x, double *y)
{
    memcpy(y, x, (size_t)n * 2 * sizeof(double));
}

static void zzero(int64_t n, double *x)
{
    memset(x, 0, (size_t)n * 2 * sizeof(double));
}

/* ------------------------------------------------------------------------- */
/* vec_randomize, src/miscellaneous.cc:371-386                                */
/* std::minstd_rand0 = Lehmer generator x <- 16807 x mod (2^31 - 1)           */
/* ------------------------------------------------------------------------- */
void qbo_vec_randomize(int64_t n, double *x, uint32_t seed)
{
    int64_t j;
    if (seed == 0) {
        double ele = sqrt(1.0 / (double)n);
        for (j = 0; j < n; j++) { x[2*j] = ele; x[2*j+1] = 0.0; }
        return;
    }
    {
        /* minstd_rand0 seeding: state = seed mod m, and 0 is mapped to 1 */
        uint64_t state = (uint64_t)seed % 2147483647ULL;
        const double pref = 1.0 / 2147483647.0;
        double rnorm;
        if (state == 0) state = 1;
        for (j = 0; j < n; j++) {
            state = (state * 16807ULL) % 2147483647ULL;
            x[2*j]   = (double)state * pref - 0.5;   /* multiply, not divide */
            x[2*j+1] = 0.0;
        }
        rnorm = qbo_nrm2(n, x);
        zscal(n, 1.0 / rnorm, 0.0, x);
    }
}

/* ------------------------------------------------------------------------- */
/* csr_mat::MultMv2 / MultMv, src/sparse.cc:262-297                           */
/* sym  -> descr = {HERMITIAN, UPPER, NON_UNIT}: stored entry (i,j,v), j>=i   */
/*         contributes v*x[j] to y[i] and, if j != i, conj(v)*x[i] to y[j]    */
/* !sym -> descr = GENERAL                                                    */
/* ------------------------------------------------------------------------- */
void qbo_multmv2(const qbo_csr *A, const double *x, double *y)
{
    const int64_t dim = A->dim;
    const int64_t *ia = A->ia, *ja = A->ja;
    const double *val = A->val;
    int64_t i;
    if (!A->sym) {
        #pragma omp parallel for schedule(static) if (dim > 20000)
        for (i = 0; i < dim; i++) {
            double sr = 0.0, si = 0.0;
            int64_t p;
            for (p = ia[i]; p < ia[i+1]; p++) {
                int64_t c = ja[p];
                double vr = val[2*p], vi = val[2*p+1];
                double xr = x[2*c], xi = x[2*c+1];
                sr += vr * xr - vi * xi;
                si += vr * xi + vi * xr;
            }
            y[2*i]   += sr;
            y[2*i+1] += si;
        }
    } else {
        for (i = 0; i < dim; i++) {
            double sr = 0.0, si = 0.0;
            double xir = x[2*i], xii = x[2*i+1];
            int64_t p;
            for (p = ia[i]; p < ia[i+1]; p++) {
                int64_t c = ja[p];
                double vr = val[2*p], vi = val[2*p+1];
                double xr = x[2*c], xi = x[2*c+1];
                sr += vr * xr - vi * xi;
                si += vr * xi + vi * xr;
                if (c != i) {                       /* conj(v) * x[i] -> y[c] */
                    y[2*c]   += vr * xir + vi * xii;
                    y[2*c+1] += vr * xii - vi * xir;
                }
            }
            y[2*i]   += sr;
            y[2*i+1] += si;
        }
    }
}

void qbo_multmv(const qbo_csr *A, const double *x, double *y)
{
    zzero(A->dim, y);                               /* src/sparse.cc:294-295 */
    qbo_multmv2(A, x, y);
}

int64_t qbo_expand_upper(const qbo_csr *A, int64_t *ia_full, int64_t *ja_full, double *val_full)
{
    const int64_t dim = A->dim;
    int64_t i, p;
    int64_t *cnt = (int64_t *)calloc((size_t)dim + 1, sizeof(int64_t));
    for (i = 0; i < dim; i++) {
        for (p = A->ia[i]; p < A->ia[i+1]; p++) {
            int64_t c = A->ja[p];
            cnt[i]++;
            if (A->sym && c != i) cnt[c]++;
        }
    }
    ia_full[0] = 0;
    for (i = 0; i < dim; i++) ia_full[i+1] = ia_full[i] + cnt[i];
    if (ja_full == NULL) { free(cnt); return ia_full[dim]; }
    /* The mirrored entries (c,i) with i < c must precede row c's own upper
     * part and arrive in ascending i because rows are visited in order; so a
     * two-pass fill keeps every row's columns ascending. */
    for (i = 0; i < dim; i++) cnt[i] = ia_full[i];
    if (A->sym) {
        for (i = 0; i < dim; i++) {                 /* lower part first */
            for (p = A->ia[i]; p < A->ia[i+1]; p++) {
                int64_t c = A->ja[p];
                if (c != i) {
                    int64_t q = cnt[c]++;
                    ja_full[q] = i;
                    val_full[2*q]   =  A->val[2*p];
                    val_full[2*q+1] = -A->val[2*p+1];
                }
            }
        }
    }
    for (i = 0; i < dim; i++) {
        for (p = A->ia[i]; p < A->ia[i+1]; p++) {
            int64_t q = cnt[i]++;
            ja_full[q] = A->ja[p];
            val_full[2*q]   = A->val[2*p];
            val_full[2*q+1] = A->val[2*p+1];
        }
    }
    free(cnt);
    return ia_full[dim];
}

/* src/sparse.cc:299-315 */
void qbo_to_dense(const qbo_csr *A, double *dense)
{
    const int64_t dim = A->dim;
    int64_t row, p;
    memset(dense, 0, (size_t)dim * (size_t)dim * 2 * sizeof(double));
    for (row = 0; row < dim; row++) {
        for (p = A->ia[row]; p < A->ia[row+1]; p++) {
            int64_t col = A->ja[p];
            dense[2*(row + col*dim)]   = A->val[2*p];
            dense[2*(row + col*dim)+1] = A->val[2*p+1];
            if (A->sym && row != col) {
                dense[2*(col + row*dim)]   =  A->val[2*p];
                dense[2*(col + row*dim)+1] = -A->val[2*p+1];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* hess_eigen, src/lanczos.cc:355-390                                         */
/* The reference calls LAPACKE_dstedc('I') on diag a[0..m) and off-diagonal   */
/* b[1..m), then sorts eigenpairs by `order`.  Restated with the implicit     */
/* QL algorithm with Wilkinson shifts (eigenvectors accumulated).             */
/* ------------------------------------------------------------------------- */
static int tridiag_ql(int64_t n, double *d, double *e, double *z)
{
    /* d[0..n): diagonal; e[0..n-1): e[i] couples i,i+1; e[n-1] scratch.
     * z: n*n column-major, identity on entry, eigenvectors in columns on exit */
    int64_t l, m, i, k;
    const double eps = 2.220446049250313e-16;
    if (n == 1) return 0;
    e[n-1] = 0.0;
    for (l = 0; l < n; l++) {
        int iter = 0;
        do {
            for (m = l; m < n - 1; m++) {
                double dd = fabs(d[m]) + fabs(d[m+1]);
                if (fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                double g, r, s, c, p, f, b;
                if (iter++ == 200) return 1;
                g = (d[l+1] - d[l]) / (2.0 * e[l]);
                r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                s = c = 1.0;
                p = 0.0;
                for (i = m - 1; i >= l; i--) {
                    f = s * e[i];
                    b = c * e[i];
                    e[i+1] = r = hypot(f, g);
                    if (r == 0.0) {
                        d[i+1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i+1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i+1] = g + p;
                    g = c * r - b;
                    for (k = 0; k < n; k++) {
                        double *zi  = z + (size_t)i * n;
                        double *zi1 = z + (size_t)(i+1) * n;
                        f = zi1[k];
                        zi1[k] = s * zi[k] + c * f;
                        zi[k]  = c * zi[k] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return 0;
}

typedef struct { double key; double val; int64_t idx; } sort_item;

static int cmp_item(const void *pa, const void *pb)
{
    const sort_item *a = (const sort_item *)pa, *b = (const sort_item *)pb;
    if (a->key < b->key) return -1;
    if (a->key > b->key) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

int qbo_hess_eigen(const double *hessenberg, int64_t maxit, int64_t m,
                   const char *order, double *ritz, double *s)
{
    int64_t j;
    int info;
    char o0 = (char)tolower((unsigned char)order[0]);
    char o1 = (char)tolower((unsigned char)order[1]);
    double *d = (double *)malloc((size_t)m * sizeof(double));
    double *e = (double *)malloc((size_t)m * sizeof(double));
    double *z = (double *)calloc((size_t)m * (size_t)m, sizeof(double));
    sort_item *items = (sort_item *)malloc((size_t)m * sizeof(sort_item));
    for (j = 0; j < m; j++) d[j] = hessenberg[maxit + j];          /* a[j]   */
    for (j = 0; j + 1 < m; j++) e[j] = hessenberg[j + 1];          /* b[j+1] */
    for (j = 0; j < m; j++) z[j * m + j] = 1.0;
    info = tridiag_ql(m, d, e, z);
    for (j = 0; j < m; j++) {
        items[j].val = d[j];
        items[j].idx = j;
        if (o1 == 'm')      items[j].key = (o0 == 's') ? fabs(d[j]) : -fabs(d[j]);
        else                items[j].key = (o0 == 's') ? d[j] : -d[j];
    }
    qsort(items, (size_t)m, sizeof(sort_item), cmp_item);
    for (j = 0; j < m; j++) {
        ritz[j] = items[j].val;
        memcpy(s + (size_t)j * m, z + (size_t)items[j].idx * m, (size_t)m * sizeof(double));
    }
    free(d); free(e); free(z); free(items);
    return info;
}

/* ------------------------------------------------------------------------- */
/* lanczos, src/lanczos.cc:134-266 (live purposes only)                       */
/* ------------------------------------------------------------------------- */
int qbo_lanczos(int64_t k, int64_t np, int64_t maxit, int64_t *m_out, int64_t dim,
                const qbo_csr *A, double *v, double *hessenberg, const char *purpose,
                qbo_lanczos_log *log, int64_t *nlog, int64_t *n_reorth)
{
    const double prec = QBO_LANCZOS_PRECISION;
    const int is_val  = strstr(purpose, "val")  != NULL;
    const int is_val1 = strstr(purpose, "val1") != NULL;
    const int is_dn   = strcmp(purpose, "dnmcs") == 0;
    int64_t mm = k + np;
    int64_t m;
    double theta0_prev = 0.0, theta1_prev = 0.0;
    int cnt_accuE0 = 0;
    double accuracy = 0.0;
    double *phipt = v + 2 * 2 * dim;                               /* :154   */
    double *ritz, *s;
    double dot[2];
    int64_t l;

    if (nlog) *nlog = 0;
    if (n_reorth) *n_reorth = 0;
    if (!(is_val || is_dn)) return -1;
    m = k;                                                         /* :145   */
    *m_out = m;
    if (!(mm < maxit && k >= 0 && np >= 0)) return -1;             /* :147   */
    if (np == 0) return 0;                                         /* :150   */

    ritz = (double *)malloc((size_t)(mm + 2) * sizeof(double));
    s    = (double *)malloc((size_t)(mm + 2) * (size_t)(mm + 2) * sizeof(double));

#define VPT(j) (v + (size_t)(((j) % 2) * dim) * 2)                 /* :160   */

    if (fabs(qbo_nrm2(dim, VPT(k)) - 1.0) >= prec) {               /* :166   */
        free(ritz); free(s);
        return -2;
    }
    if (k == 0) {                                                  /* :167   */
        hessenberg[0] = 0.0;
        zzero(dim, VPT(1));
        qbo_multmv2(A, VPT(0), VPT(1));
        qbo_dotc(dim, VPT(0), VPT(1), dot);
        hessenberg[maxit] = dot[0];                                /* a[0]   */
        zaxpy(dim, -hessenberg[maxit], 0.0, VPT(0), VPT(1));
        hessenberg[1] = qbo_nrm2(dim, VPT(1));                     /* b[1]   */
        zscal(dim, 1.0 / hessenberg[1], 0.0, VPT(1));
        m = ++k;
        --np;
    }

    do {                                                           /* :193   */
        double *vm, *vm1;
        m++;
        vm = VPT(m); vm1 = VPT(m - 1);
        {
            double nb = -hessenberg[m-1];
            #pragma omp parallel for schedule(static) if (dim > OMP_MIN_N)
            for (l = 0; l < 2 * dim; l++) vm[l] = nb * vm[l];      /* :195 (v[m-2] aliases v[m]) */
        }
        qbo_multmv2(A, vm1, vm);                                   /* :197   */
        qbo_dotc(dim, vm1, vm, dot);
        hessenberg[maxit + m - 1] = dot[0];                        /* :200   */
        zaxpy(dim, -hessenberg[maxit + m - 1], 0.0, vm1, vm);      /* :206   */
        hessenberg[m] = qbo_nrm2(dim, vm);                         /* :208   */
        zscal(dim, 1.0 / hessenberg[m], 0.0, vm);                  /* :214   */

        if (fabs(hessenberg[m]) < prec) break;                     /* :216   */

        if (is_val1) {                                             /* :218   */
            qbo_dotc(dim, phipt, vm, dot);
            if (hypot(dot[0], dot[1]) > prec) {
                double rnorm;
                zaxpy(dim, -dot[0], -dot[1], phipt, vm);
                rnorm = qbo_nrm2(dim, vm);
                zscal(dim, 1.0 / rnorm, 0.0, vm);
                if (n_reorth) (*n_reorth)++;
            }
        }

        if (is_val) {                                              /* :228   */
            qbo_hess_eigen(hessenberg, maxit, m, "sr", ritz, s);
            if (m > 3) {
                double accu_E0, accu_E1;
                accuracy = fabs(hessenberg[m] * s[m-1]);
                accu_E0  = fabs((ritz[0] - theta0_prev) / ritz[0]);
                accu_E1  = fabs((ritz[1] - theta1_prev) / ritz[1]);
                if (log && nlog) {
                    qbo_lanczos_log *r = &log[(*nlog)++];
                    r->k = m;
                    r->ritz[0] = ritz[0]; r->ritz[1] = ritz[1];
                    r->ritz[2] = ritz[2]; r->ritz[3] = ritz[3];
                    r->a_km1 = hessenberg[maxit + m - 1];
                    r->b_k = hessenberg[m];
                    r->accuracy = accuracy;
                    r->accu_E0 = accu_E0; r->accu_E1 = accu_E1;
                }
                if (accu_E0 < prec) cnt_accuE0++; else cnt_accuE0 = 0;
                if (cnt_accuE0 > 15 && accuracy < prec) break;     /* :240   */
            }
            theta0_prev = ritz[0];
            theta1_prev = ritz[1];
        }
    } while (m < mm);
#undef VPT
    *m_out = m;
    free(ritz); free(s);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* eigenvec_CG, src/lanczos.cc:281-341                                        */
/* ------------------------------------------------------------------------- */
int qbo_eigenvec_cg(int64_t dim, int64_t maxit, int64_t *m_io, const qbo_csr *A,
                    double E0, double *accu_out, double *v, double *r, double *p, double *pp,
                    double *resid_log)
{
    const double prec = QBO_LANCZOS_PRECISION;
    const double machine_prec = 2.220446049250313e-16;
    int64_t m = *m_io;
    double accu;
    if (!(m >= 0 && m < maxit)) return -1;
    accu = (m == 0) ? 0.0 : qbo_nrm2(dim, r);                      /* :290   */

    while (m < maxit) {
        if (accu < prec) {
            double rnorm = qbo_nrm2(dim, v);
            if (m == 0 || fabs(rnorm - 1.0) > prec) {              /* :297   */
                zscal(dim, 1.0 / rnorm, 0.0, v);
                zzero(dim, r);
                qbo_multmv2(A, v, r);
                zscal(dim, -1.0, 0.0, r);
                zaxpy(dim, E0, 0.0, v, r);                         /* r = (E0-H) v */
                zcopy(dim, r, p);
                accu = qbo_nrm2(dim, r);
                m++;
                if (resid_log) resid_log[m] = accu;
                if (accu < prec) break;
            } else {
                break;
            }
        } else {
            double delta[2], ar, ai, den, beta;
            zcopy(dim, p, pp);
            zscal(dim, machine_prec - E0, 0.0, pp);                /* :321   */
            qbo_multmv2(A, p, pp);                                 /* pp = (H-E0) p */
            qbo_dotc(dim, p, pp, delta);
            den = delta[0] * delta[0] + delta[1] * delta[1];       /* alpha = accu^2 / delta */
            ar =  accu * accu * delta[0] / den;
            ai = -accu * accu * delta[1] / den;
            zaxpy(dim,  ar,  ai, p,  v);
            zaxpy(dim, -ar, -ai, pp, r);
            beta = qbo_nrm2(dim, r) / accu;
            zscal(dim, beta * beta, 0.0, p);
            zaxpy(dim, 1.0, 0.0, r, p);
            accu *= beta;
            m++;
            if (resid_log) resid_log[m] = accu;
        }
    }
    *m_io = m;
    *accu_out = accu;
    return 0;
}
