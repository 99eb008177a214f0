/*
 * qb_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference hot path (wztzjhn/quantum_basis):
 * CSR Hamiltonian x vector, the Lanczos three-term recurrence, the CG
 * eigenvector refiner, the tridiagonal Ritz solve and the deterministic start
 * vector.  Each function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  Nothing under
 * quantum_basis_amd/ links, imports or calls it.
 *
 * Parity pin: the reference itself is unbuildable in this image (it needs
 * mkl.h, Boost and ARPACK-NG headers that are absent), so the
 * oracle is pinned against (a) the known answers asserted by the reference's
 * own tests/examples (E0, correlators, 1e-8) and (b) the 17-digit vectors the
 * survey captured from the reference.  See
 * tests/test_oracle_golden.py.
 *
 * All complex arrays are interleaved (re, im) doubles == std::complex<double>.
 * All integers are 64-bit (the reference builds with -DMKL_ILP64).
 */
#ifndef QB_ORACLE_H
#define QB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tolerances, src/miscellaneous.cc:44-47 */
#define QBO_LANCZOS_PRECISION 2e-12
#define QBO_SPARSE_PRECISION  1e-14

typedef struct {
    int64_t dim;
    int64_t nnz;
    int     sym;          /* 1: only the upper triangle (col >= row) is stored */
    const int64_t *ia;    /* [dim+1] zero-based row pointers                  */
    const int64_t *ja;    /* [nnz]   zero-based columns                       */
    const double  *val;   /* [2*nnz] interleaved complex                      */
} qbo_csr;

/* src/miscellaneous.cc:371-386 */
void qbo_vec_randomize(int64_t n, double *x, uint32_t seed);

/* src/sparse.cc:262-289 (y += H x) and :291-297 (y = H x) */
void qbo_multmv2(const qbo_csr *A, const double *x, double *y);
void qbo_multmv(const qbo_csr *A, const double *x, double *y);

/* upper-triangle storage -> full storage (both triangles), columns ascending.
 * ia_full[dim+1] must be allocated by the caller; call once with ja_full ==
 * NULL to obtain nnz_full (= ia_full[dim]), then again to fill. */
int64_t qbo_expand_upper(const qbo_csr *A, int64_t *ia_full, int64_t *ja_full, double *val_full);

/* src/sparse.cc:299-315; dense is column-major dim x dim complex */
void qbo_to_dense(const qbo_csr *A, double *dense);

/* src/lanczos.cc:355-390.  order: "sr","lr","sm","lm" (case-insensitive,
 * "sa"/"la" accepted).  ritz[m], s[m*m] column-major.  returns 0 or >0 if the
 * QL iteration failed to converge. */
int qbo_hess_eigen(const double *hessenberg, int64_t maxit, int64_t m,
                   const char *order, double *ritz, double *s);

/* per-step record written by qbo_lanczos when log != NULL: the columns of
 * log_Lanczos_srval (src/lanczos.cc:102-128) */
typedef struct {
    int64_t k;
    double ritz[4];
    double a_km1, b_k, accuracy, accu_E0, accu_E1;
} qbo_lanczos_log;

/* src/lanczos.cc:134-266, purposes "sr_val0", "sr_val1", "dnmcs".
 * v: 2*dim complex (3*dim for sr_val1: phi0 at v+2*dim).
 * log may be NULL; otherwise room for maxit records, *nlog receives the count.
 * *n_reorth (may be NULL) counts the phi0 re-orthogonalisations.
 * returns 0, or -1 on invalid arguments. */
int qbo_lanczos(int64_t k, int64_t np, int64_t maxit, int64_t *m, int64_t dim,
                const qbo_csr *A, double *v, double *hessenberg, const char *purpose,
                qbo_lanczos_log *log, int64_t *nlog, int64_t *n_reorth);

/* src/lanczos.cc:281-341.  E0 is passed as a real number (the reference
 * passes static_cast<T>(E0)).  resid_log may be NULL, else room for maxit+1
 * doubles: the accu value after each step (log_CG.txt). */
int qbo_eigenvec_cg(int64_t dim, int64_t maxit, int64_t *m, const qbo_csr *A,
                    double E0, double *accu, double *v, double *r, double *p, double *pp,
                    double *resid_log);

/* BLAS-1 restatements (src/lanczos.cc:10-53) exposed for tests */
double qbo_nrm2(int64_t n, const double *x);
void   qbo_dotc(int64_t n, const double *x, const double *y, double *res /*[2]*/);

int qbo_num_threads(void);
/* NUMA first-touch placement helper of the timed CPU baseline (bench.py) */
void qbo_first_touch_copy(int64_t nbytes, const void *src, void *dst);

#ifdef __cplusplus
}
#endif
#endif
