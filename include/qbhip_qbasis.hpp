// qbhip_qbasis.hpp -- header-only C++ adaptor: the reference's operator concept on top of the
// C ABI of libqbhip.so.
//
// The reference's solvers are templated on a duck-typed MAT (src/qbasis.h:1065-1093):
//     void MultMv2(const T *x, T *y) const;   // y += H x   (src/lanczos.cc:170,197,301,322)
//     void MultMv (const T *x, T *y) const;   // y  = H x   (src/lanczos.cc:426,476; ARPACK workd slices)
//     std::vector<T> to_dense() const;        // dim <= 30  (src/lanczos.cc:509)
// and its only sparse model is csr_mat<T> (src/qbasis.h:976-1021).  qbhip::csr_mat below has
// the same public surface and error behaviour (std::runtime_error where the reference throws,
// src/sparse.cc:130,259,288), holds the host CSR arrays exactly like the reference does, and
// keeps a device-resident copy behind the `handle` member -- the slot where the reference
// stores MKL's sparse_matrix_t.
//
// qbhip::lanczos / eigenvec_CG / hess_eigen have the reference's signatures
// (src/qbasis.h:1065-1082) and forward to the fused device solvers, so a maintainer can
// replace the bodies of the csr_mat instantiations in src/lanczos.cc by one call each (see
// INTEGRATION.md).
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "qbhip.h"

namespace qbhip {

using cplx = std::complex<double>;
using qint = long long;   // MKL_INT under -DMKL_ILP64

inline void check(int rc, const char *where)
{
    if (rc == QBH_OK) return;
    const std::string msg = std::string(where) + ": " + qbh_strerror(rc) + " (" + qbh_last_error() + ")";
    if (rc == QBH_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// Tell the library once what the row index of the matrices the host is about to hand over means: the reference's own order of
// a two-species fermion basis (Lin tables, src/basis.cc:1144-1190; operators ordered by site).  csr_mat's constructor passes no
// options, so this sets the process-wide defaults (qbh_opts_set_default): operators created afterwards are held species-major
// inside the library (Kronecker split), callers keep seeing the reference's order.  A hint that does not describe a matrix
// changes nothing for that matrix.  n_sites = 0 restores the built-in defaults.
// OPTIONAL since ABI 500: without any declaration the library looks for the basis itself (qbh_opts.basis_detect: every
// (n_sites, n_up, n_dn) whose two-species dimension equals dim is tried through a one-pass structure check); naming it only
// saves that search (C3: 3.8 s, once per operator).
inline void declare_reference_fermion_basis(int n_sites, int n_up, int n_dn)
{
    if (n_sites <= 0) {
        qbh_opts_set_default(nullptr);
        return;
    }
    qbh_opts o;
    qbh_opts_set_default(nullptr);
    qbh_opts_default(&o);
    o.basis_kind = QBH_BASIS_REF_FERMION2;
    o.n_sites = n_sites;
    o.n_up = n_up;
    o.n_dn = n_dn;
    qbh_opts_set_default(&o);
}

class csr_mat;
inline void vec_randomize(const csr_mat &mat, cplx *x, const uint32_t &seed);

class csr_mat {
public:
    qint dim = 0;
    qint nnz = 0;
    bool sym = false;          // only the upper triangle stored (reference default)
    cplx *val = nullptr;       // host arrays, owned, same layout as src/qbasis.h:979-985
    qint *ja = nullptr;
    qint *ia = nullptr;
    qbh_csr *handle = nullptr; // device operator (the reference keeps sparse_matrix_t here)

    csr_mat() = default;

    // adopt host CSR arrays allocated with new[] (what csr_mat(lil_mat&) produces, src/sparse.cc:202-260)
    csr_mat(qint dim_, qint nnz_, bool sym_, cplx *val_, qint *ja_, qint *ia_, const qbh_opts *opts = nullptr)
        : dim(dim_), nnz(nnz_), sym(sym_), val(val_), ja(ja_), ia(ia_)
    {
        static_assert(sizeof(qint) == sizeof(int64_t) && sizeof(cplx) == sizeof(qbh_z), "ILP64 / complex128 layout");
        check(qbh_csr_create(&handle, dim, nnz, sym ? 1 : 0, reinterpret_cast<const int64_t *>(ia),
                             reinterpret_cast<const int64_t *>(ja), reinterpret_cast<const qbh_z *>(val), opts),
              "create_handle failed");                       // src/sparse.cc:259
    }

    // wrap an operator that was assembled on the device (qbh_gen_*): there are no host arrays; dim = its row count
    static csr_mat from_device(qbh_csr *h)
    {
        csr_mat m;
        qbh_csr_info info;
        check(qbh_csr_get_info(h, &info), "qbh_csr_get_info failed");
        m.dim = info.nrows;
        m.nnz = info.nnz;
        m.handle = h;
        return m;
    }

    csr_mat(const csr_mat &old) : dim(old.dim), nnz(old.nnz), sym(old.sym)   // deep copy, src/sparse.cc:114-138
    {
        if (nnz > 0) {
            val = new cplx[nnz];
            ja = new qint[nnz];
            ia = new qint[dim + 1];
            for (qint j = 0; j < nnz; j++) { val[j] = old.val[j]; ja[j] = old.ja[j]; }
            for (qint j = 0; j <= dim; j++) ia[j] = old.ia[j];
            check(qbh_csr_create(&handle, dim, nnz, sym ? 1 : 0, reinterpret_cast<const int64_t *>(ia),
                                 reinterpret_cast<const int64_t *>(ja), reinterpret_cast<const qbh_z *>(val), nullptr),
                  "create_handle failed");
        }
    }

    csr_mat(csr_mat &&old) noexcept
        : dim(old.dim), nnz(old.nnz), sym(old.sym), val(old.val), ja(old.ja), ia(old.ia), handle(old.handle)
    {
        old.val = nullptr; old.ja = nullptr; old.ia = nullptr; old.handle = nullptr;
    }

    csr_mat &operator=(csr_mat old) { swap(*this, old); return *this; }

    friend void swap(csr_mat &l, csr_mat &r) noexcept
    {
        using std::swap;
        swap(l.dim, r.dim); swap(l.nnz, r.nnz); swap(l.sym, r.sym);
        swap(l.val, r.val); swap(l.ja, r.ja); swap(l.ia, r.ia); swap(l.handle, r.handle);
    }

    void destroy()                                           // src/sparse.cc:150-167
    {
        delete[] val; val = nullptr;
        delete[] ja; ja = nullptr;
        delete[] ia; ia = nullptr;
        qbh_csr_destroy(handle); handle = nullptr;
    }

    ~csr_mat() { destroy(); }

    qint dimension() const { return dim; }

    void MultMv2(const cplx *x, cplx *y) const               // y = H * x + y, src/sparse.cc:262-289
    {
        check(qbh_multmv2(handle, reinterpret_cast<const qbh_z *>(x), reinterpret_cast<qbh_z *>(y)),
              "matrix-vector product failed.");
    }

    void MultMv(const cplx *x, cplx *y) const                // y = H * x, src/sparse.cc:291-297
    {
        check(qbh_multmv(handle, reinterpret_cast<const qbh_z *>(x), reinterpret_cast<qbh_z *>(y)),
              "matrix-vector product failed.");
    }

    std::vector<cplx> to_dense() const                       // src/sparse.cc:299-315 (host arrays)
    {
        std::vector<cplx> res(dim * dim, cplx(0.0));
        for (qint row = 0; row < dim; row++)
            for (qint pt = ia[row]; pt < ia[row + 1]; pt++) {
                const qint col = ja[pt];
                res[row + col * dim] = val[pt];
                if (sym && row != col) res[col + row * dim] = std::conj(val[pt]);
            }
        return res;
    }
};

// lanczos<T, csr_mat<T>> (src/lanczos.cc:134-266): vectors stay in HBM for the whole call.
inline void lanczos(qint k, qint np, const qint &maxit, qint &m, const qint &dim, const csr_mat &mat, cplx v[],
                    double hessenberg[], const std::string &purpose)
{
    if (dim != mat.dim) throw std::invalid_argument("lanczos: dim mismatch");
    int64_t mm = 0;
    check(qbh_lanczos(mat.handle, k, np, maxit, &mm, reinterpret_cast<qbh_z *>(v), hessenberg, purpose.c_str(), nullptr),
          "lanczos");
    m = mm;
}

// eigenvec_CG<T, csr_mat<T>> (src/lanczos.cc:281-341)
inline void eigenvec_CG(const qint &dim, const qint &maxit, qint &m, const csr_mat &mat, const cplx &E0, double &accu,
                        cplx v[], cplx r[], cplx p[], cplx pp[])
{
    if (dim != mat.dim) throw std::invalid_argument("eigenvec_CG: dim mismatch");
    int64_t mm = m;
    check(qbh_eigenvec_cg(mat.handle, maxit, &mm, E0.real(), &accu, reinterpret_cast<qbh_z *>(v),
                          reinterpret_cast<qbh_z *>(r), reinterpret_cast<qbh_z *>(p), reinterpret_cast<qbh_z *>(pp), nullptr),
          "eigenvec_CG");
    m = mm;
}

// hess_eigen (src/lanczos.cc:355-390)
inline void hess_eigen(const double hessenberg[], const qint &maxit, const qint &m, const std::string &order,
                       std::vector<double> &ritz, std::vector<double> &s)
{
    ritz.resize(m);
    s.resize(m * m);
    check(qbh_hess_eigen(hessenberg, maxit, m, order.c_str(), ritz.data(), s.data()), "hess_eigen");
}

// iram<T, csr_mat<T>> (src/lanczos.cc:497-603) with the Krylov basis resident in HBM (qbh_iram).  Same
// argument checks and outputs; v0 is ignored exactly as in the reference (ARPACK info = 0, :470).
inline void iram(const qint &dim, csr_mat &mat, cplx v0[], const qint &nev, const qint &ncv, const qint &maxit,
                 const std::string &order, qint &nconv, double eigenvals[], cplx eigenvecs[])
{
    (void)v0;
    if (dim != mat.dim) throw std::invalid_argument("iram: dim mismatch");
    if (nev <= 0 || nev >= dim - 1) throw std::invalid_argument("0 < nev < N-1 should be satisfied.");
    if (maxit < 20) throw std::invalid_argument("maxit should not be smaller than 20!");
    if (dim <= 30) throw std::invalid_argument("iram: dim <= 30 uses the dense fall-back of the host code (to_dense)");
    int64_t nc = 0;
    check(qbh_iram(mat.handle, nev, ncv, maxit, order.c_str(), 0.0, 1u, &nc, eigenvals,
                   reinterpret_cast<qbh_z *>(eigenvecs), nullptr), "iram");
    if (nc <= 0) throw std::runtime_error("nconv == 0...");           // src/lanczos.cc:566
    nconv = nc;
}

// energy_scale<T, csr_mat<T>> (src/kpm.cc:45-88): spectral bounds from iters - 1 Lanczos steps without a stop rule
inline void energy_scale(const qint &dim, const csr_mat &mat, cplx v[], double &lo, double &hi, const double &extend = 0.1,
                         const qint &iters = 128)
{
    const qint mm = iters - 1;
    std::vector<double> hessenberg(2 * iters, 0.0), ritz, s;
    vec_randomize(mat, v, 1);
    qint m = 0;
    lanczos(0, mm, iters, m, dim, mat, v, hessenberg.data(), "dnmcs");
    hess_eigen(hessenberg.data(), iters, m, "sr", ritz, s);
    lo = ritz[0];
    hi = ritz[m - 1];
    const double slack = extend * (hi - lo);
    lo -= slack;
    hi += slack;
}

// lanczos() with the reference's checkpoints enabled (enable_ckpt, src/lanczos.cc:144,190,242,263 -> src/ckpt.cc): resumes
// from `dir` when it holds a usable step, commits every `every` steps; returns true when the stop rule fired.
inline bool lanczos_ckpt(const qint &maxit, qint &m, const csr_mat &mat, cplx v[], double hessenberg[], const std::string &purpose,
                         qint every = 50, const std::string &dir = "out_Qckpt", qint max_steps = 0)
{
    int64_t mm = 0;
    int conv = 0;
    check(qbh_lanczos_ckpt(mat.handle, maxit, &mm, reinterpret_cast<qbh_z *>(v), hessenberg, purpose.c_str(), every, max_steps,
                           dir.c_str(), &conv, nullptr), "lanczos_ckpt");
    m = mm;
    return conv != 0;
}

// vec_disk_write / vec_disk_read (src/miscellaneous.cc:391-469)
template <typename T> inline int vec_disk_write(const std::string &filename, qint n, T *x)
{
    check(qbh_vec_disk_write(filename.c_str(), n, (int)sizeof(T), x), "vec_disk_write");
    return 0;
}
template <typename T> inline int vec_disk_read(const std::string &filename, qint n, T *x)
{
    return qbh_vec_disk_read(filename.c_str(), n, (int)sizeof(T), x);           // 0, or 1 where the reference returns 1
}

// One rank of a row-sharded run: this rank's rows of the full operator from the host CSR every rank holds, on
// the native RCCL communicator.  `uid` is the 128-byte id of qbh_rccl_unique_id() from rank 0.  The returned handle is a
// plain qbh_csr*: the device-vector entry points (qbh_lanczos_dev, qbh_eigenvec_cg_dev, qbh_iram, ...) take it as is,
// with vectors of the shard-local length cuts[rank+1] - cuts[rank].
inline qbh_csr *create_row_shard(qint dim, qint nnz, bool sym, const cplx *val, const qint *ja, const qint *ia, int rank, int nranks,
                                 const void *uid, std::vector<qint> *cuts_out = nullptr, const qbh_opts *opts = nullptr)
{
    std::vector<int64_t> cuts((size_t)nranks + 1);
    check(qbh_balanced_row_cuts(dim, nnz, sym ? 1 : 0, reinterpret_cast<const int64_t *>(ia), reinterpret_cast<const int64_t *>(ja),
                                nranks, cuts.data()), "balanced_row_cuts");
    qbh_csr *h = nullptr;
    check(qbh_csr_create_rows(&h, dim, nnz, sym ? 1 : 0, reinterpret_cast<const int64_t *>(ia), reinterpret_cast<const int64_t *>(ja),
                              reinterpret_cast<const qbh_z *>(val), cuts[(size_t)rank], cuts[(size_t)rank + 1], opts), "create_row_shard");
    const int rc = qbh_comm_create_rccl(h, uid, rank, nranks, cuts.data());
    if (rc != QBH_OK) {
        qbh_csr_destroy(h);
        check(rc, "comm_create_rccl");
    }
    if (cuts_out) cuts_out->assign(cuts.begin(), cuts.end());
    return h;
}

// vec_randomize (src/miscellaneous.cc:371-386), produced on the device
inline void vec_randomize(const csr_mat &mat, cplx *x, const uint32_t &seed)
{
    qbh_z *d = nullptr;
    check(qbh_vec_alloc(&d, mat.dim), "vec_randomize");
    int rc = qbh_vec_randomize(mat.handle, d, seed);
    if (rc == QBH_OK) rc = qbh_vec_download(mat.handle, reinterpret_cast<qbh_z *>(x), d, mat.dim);
    qbh_vec_free(d);
    check(rc, "vec_randomize");
}

// A translation-symmetric sector of the Hubbard family assembled on the device: what
// model::enumerate_basis_repr + generate_Ham_sparse_repr produce for
// examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc (see qbh_gen_hubbard_repr in qbhip.h).  bonds: (i, j) pairs,
// each giving -t (c+_i c_j + h.c.) for both species; perms[g * n_sites + s] = image of site s under translation g
// (lattice::translation_plan), chars[g] = exp(-i k.t_g).
// matrix_free = true: the same operator without the matrix (qbh_mf_hubbard_repr: block tables + a small stored remainder).  Its
// host vectors are the sector's vectors as for the stored form; its DEVICE vectors are kept in the handle's own row order
// (qbh_opts.sector_orbit; qbh_vec_to_internal / qbh_vec_from_internal convert, qbh_csr_info.basis_internal tells).
inline csr_mat hubbard_sector(int n_sites, int n_up, int n_dn, const std::vector<std::pair<int, int>> &bonds, double t, double U,
                              const std::vector<int32_t> &perms, const std::vector<cplx> &chars, const qbh_opts *opts = nullptr,
                              bool matrix_free = false)
{
    std::vector<int32_t> sites;
    std::vector<cplx> amp;
    for (const auto &b : bonds) {
        sites.insert(sites.end(), {b.first, b.second, b.second, b.first});
        amp.push_back(cplx(-t));
        amp.push_back(cplx(-t));
    }
    qbh_csr *h = nullptr;
    int64_t dim = 0;
    if (matrix_free) {
        check(qbh_mf_hubbard_repr(&h, n_sites, n_up, n_dn, (int)amp.size(), sites.data(), reinterpret_cast<const qbh_z *>(amp.data()),
                                  reinterpret_cast<const qbh_z *>(amp.data()), U, 0, nullptr, nullptr, (int)chars.size(), perms.data(),
                                  reinterpret_cast<const double *>(chars.data()), 100.0, &dim, opts),
              "qbh_mf_hubbard_repr failed");
        return csr_mat::from_device(h);
    }
    check(qbh_gen_hubbard_repr(&h, n_sites, n_up, n_dn, (int)amp.size(), sites.data(), reinterpret_cast<const qbh_z *>(amp.data()),
                               reinterpret_cast<const qbh_z *>(amp.data()), U, 0, nullptr, nullptr, 0, nullptr, nullptr, 0,
                               (int)chars.size(), perms.data(), reinterpret_cast<const double *>(chars.data()), 100.0, 0, 1, &dim,
                               opts),
          "qbh_gen_hubbard_repr failed");
    return csr_mat::from_device(h);
}

// The four stages of model<T>::locate_E0_lanczos (src/model.cc:1123-1316) for a CSR operator, as a free
// function with the fields the model object exposes afterwards.
struct E0_result {
    std::vector<double> eigenvals;
    std::vector<cplx> eigenvecs;
    double E0 = 0.0, E1 = 0.0, gap = 0.0;
    qint nconv = 0, steps_E0 = 0, steps_V0 = 0, steps_E1 = 0, steps_V1 = 0;
};

inline E0_result locate_E0_lanczos(const csr_mat &HamMat, const qint &nev = 1, const qint &ncv = 1, qint maxit = 1000)
{
    if (!(nev > 0 && nev <= 2 && ncv >= nev - 1 && ncv <= nev)) throw std::invalid_argument("locate_E0_lanczos: nev/ncv");
    const qint dim = HamMat.dim;
    const uint32_t seed = 1;
    E0_result res;
    const double lanczos_precision = QBH_LANCZOS_PRECISION;
    std::vector<double> hessenberg(2 * maxit, 0.0), ritz, s;
    std::vector<cplx> v(ncv > 0 ? dim * 4 : dim * 2);
    vec_randomize(HamMat, v.data(), seed);                                                  // :1165
    qint m = 0;
    lanczos(0, maxit - 1, maxit, m, dim, HamMat, v.data(), hessenberg.data(), "sr_val0");   // :1180
    hess_eigen(hessenberg.data(), maxit, m, "sr", ritz, s);
    res.eigenvals = {ritz[0]};
    res.E0 = ritz[0];
    res.steps_E0 = m;
    if (ncv == 0) return res;
    vec_randomize(HamMat, v.data() + 2 * dim, seed);                                        // :1209
    double accuracy = 0.0;
    m = 0;
    eigenvec_CG(dim, maxit, m, HamMat, cplx(res.E0), accuracy, v.data() + 2 * dim, v.data(), v.data() + dim,
                v.data() + 3 * dim);
    if (!(accuracy < lanczos_precision)) throw std::runtime_error("eigenvec_CG did not converge");   // assert :1221
    res.steps_V0 = m;
    res.nconv = 1;
    if (nev == 2) {                                                                         // :1233-1265
        vec_randomize(HamMat, v.data(), seed);
        cplx alpha(0.0);
        for (qint j = 0; j < dim; j++) alpha += std::conj(v[2 * dim + j]) * v[j];
        double rnorm = 0.0;
        for (qint j = 0; j < dim; j++) { v[j] -= alpha * v[2 * dim + j]; rnorm += std::norm(v[j]); }
        rnorm = std::sqrt(rnorm);
        for (qint j = 0; j < dim; j++) v[j] /= rnorm;
        m = 0;
        lanczos(0, maxit - 1, maxit, m, dim, HamMat, v.data(), hessenberg.data(), "sr_val1");
        hess_eigen(hessenberg.data(), maxit, m, "sr", ritz, s);
        res.eigenvals.push_back(ritz[0]);
        res.E1 = ritz[0];
        res.gap = res.E1 - res.E0;
        res.steps_E1 = m;
    }
    if (ncv == 1) {                                                                         // :1267-1273
        res.eigenvecs.assign(v.begin() + 2 * dim, v.begin() + 3 * dim);
        return res;
    }
    v.resize(5 * dim);                                                                      // :1279
    vec_randomize(HamMat, v.data() + 3 * dim, seed + 7);
    m = 0;
    eigenvec_CG(dim, maxit, m, HamMat, cplx(res.E1), accuracy, v.data() + 3 * dim, v.data(), v.data() + dim,
                v.data() + 4 * dim);
    res.steps_V1 = m;
    res.nconv = 2;
    res.eigenvecs.assign(v.begin() + 2 * dim, v.begin() + 4 * dim);
    if (res.gap < lanczos_precision) {                                                      // :1301-1310
        cplx alpha(0.0);
        for (qint j = 0; j < dim; j++) alpha += std::conj(res.eigenvecs[j]) * res.eigenvecs[dim + j];
        double rnorm = 0.0;
        for (qint j = 0; j < dim; j++) { res.eigenvecs[dim + j] -= alpha * res.eigenvecs[j]; rnorm += std::norm(res.eigenvecs[dim + j]); }
        rnorm = std::sqrt(rnorm);
        for (qint j = 0; j < dim; j++) res.eigenvecs[dim + j] /= rnorm;
    }
    return res;
}

}  // namespace qbhip
