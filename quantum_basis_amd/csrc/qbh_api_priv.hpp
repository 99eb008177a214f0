// qbh_api_priv.hpp -- what the translation units behind the extern "C" surface share (round 5: qbh_api.cpp was one 3,900-line file):
//   qbh_api.cpp     errors, options, operator lifetime (create / adopt / destroy / info / set_basis), geometry of the unsplit
//                   forms, vectors, download
//   qbh_split.cpp   the Kronecker split built in place (complex128) and its coded sibling: kron_build, kron_short_cols,
//                   kron_restore, kronc_build
//   qbh_commattach.cpp   qbh_csr_set_comm: the collective agreement on the exchange form, the gather in parts
//   qbh_spmv.cpp    SpMV dispatch (spmv_run, spmv_kron), reductions, the real wire format, BLAS-1 runs, device building blocks,
//                   the host-vector seam (qbh_multmv / qbh_multmv2)
//   qbh_solvers.cpp lanczos_core, cg_core, qbh_iram, qbh_hess_eigen
#pragma once

#include <initializer_list>
#include <vector>

#include "qbh_internal.hpp"

namespace qbhapi {
using qbh::d2;

// ---- qbh_api.cpp ----
double now_ms();
int require_device(const qbh_opts *opts, int *dev_out);
int setup_geometry(qbh_csr *A, const int64_t *d_ia, int64_t nnz, int dict_mode, int *npb_o, int *tpr_o, int *unroll_o, int64_t *window_o,
                   int64_t *n_blocks_o, int32_t **d_rb_o, int64_t **d_bp_o, int *grid_o);
int split_shard(qbh_csr *A);
int setup_wave_geometry(qbh_csr *A, const int64_t *d_ia, int64_t nnz, qbh::WaveDesc **d_wd_o, int64_t *n_wb_o, int *tpr_o, int *grid_o);
int build_geometry(qbh_csr *A);
int autotune_kernel(qbh_csr *A);
int finalize(qbh_csr *A);
int new_handle(qbh_csr **out, const qbh_opts *opts, bool host_arrays = false);
int try_value_dict(qbh_csr *A);
int vec_h2d(qbh_csr *A, d2 *d_dst, const void *h_src, int64_t n);
int vec_d2h(qbh_csr *A, void *h_dst, const d2 *d_src, int64_t n);
// ---- qbh_split.cpp ----
void kron_free_aux(qbh_csr *A);
qbh::KronParts kron_parts(const qbh_csr *A);
qbh::KronCols kron_cols_one(int64_t S, int64_t NUg, int B);
int wave_geometry_for(qbh_csr *A, const int64_t *ia, int64_t nr, int64_t nnz, double avg, bool slots, int ops, qbh::WaveDesc **wd_io, int64_t *nwb_o,
                      int *tpr_o, int *grid_o, int64_t shift = 0);
int kron_geometry(qbh_csr *A);
int kron_short_cols(qbh_csr *A);
int kron_build(qbh_csr *A);
int kron_restore(qbh_csr *A);
void kronc_release(qbh_csr *A);
int kronc_build_sliced(qbh_csr *A, int64_t S, int64_t NU);
int kronc_table_route(qbh_csr *A);
int kronc_build(qbh_csr *A);
// ---- qbh_commattach.cpp ----
int kron_parts_wanted(const qbh_csr *A, const qbh_comm *comm);
int kron_gather_parts(qbh_csr *A, const qbh_comm *comm, int64_t want);
int kron_needed_majors(qbh_csr *A, int nranks, std::vector<uint8_t> *bits_out);
int kron_sparse_setup(qbh_csr *A, const qbh_comm *comm, const std::vector<uint8_t> &bits);
// ---- qbh_spmv.cpp ----
int finish_reduction(qbh_csr *A, int nparts, int ncomp, double *host_out);
void harvest_events(qbh_csr *A);
int next_event_set(qbh_csr *A);
int deferred_reduction(qbh_csr *A, int nparts);
int spmv_kron(qbh_csr *A, const d2 *x, d2 *y, double alpha, double beta, double gamma, double *red);
int spmv_run(qbh_csr *A, const d2 *x, d2 *y, double alpha, double beta, double gamma, double *red);
int enable_real_wire(qbh_csr *A, std::initializer_list<const d2 *> vecs);
int finish_real_wire(qbh_csr *A);
int dotc_run(qbh_csr *A, const d2 *x, const d2 *y, double *res2);
int axpy_norm_run(qbh_csr *A, d2 alpha, const d2 *x, d2 *y, double *nrm2sq);
int axpy_norm_deferred(qbh_csr *A, double scale, const d2 *x, d2 *y, double *dot_out, double *nrm2sq);
int nrm2_run(qbh_csr *A, const d2 *x, double *nrm);
int multmv_host(qbh_csr *A, const qbh_z *x_host, qbh_z *y_host, double beta);

struct Bind {   // make the operator's device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit Bind(const qbh_csr *A)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != A->device) ok = (hipSetDevice(A->device) == hipSuccess);
    }
    ~Bind()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline double *scal_buf(qbh_csr *A) { return A->has_comm ? A->comm.d_scal : A->d_scal; }

// y <- alpha*H x + beta*y + gamma*x_local ; red (host, 3 doubles) optional.
// Without a communicator x is the full-length vector (ncols) and the shard-local part is
// x + row_offset; with one, x is shard-local and is gathered through the hooks first.
// A split shard runs two launches: the locally-owned columns (which only need this rank's block of x
// and therefore overlap with the all-gather), then the remote columns, accumulating into y; the fused
// reductions are produced by the last launch, on the final y.
// does the next complex SpMV of this handle take the passes of the Kronecker split?  (An operator split in place has no
// other form: spmv_run refuses whatever would need its CSR.)
inline bool kron_path(const qbh_csr *A)
{
    return A->use_wave && A->kron.active && !(A->debug & 1) && A->ovr_yr == nullptr && !(A->real_mode && A->kernel == QBH_KERNEL_ROWS) &&
           (!A->has_comm || A->kron.comm_tiled);
}

// where the pass that writes the next SpMV's x also writes its tiled copy (nullptr: the SpMV makes the copy itself): the
// handle's own buffer, or -- under a communicator -- the send buffer of the exchange (the rank's block travels tiled)
inline d2 *tiled_target(const qbh_csr *A)
{
    if (!A->opts.tile_fold || !A->kron.fold || !kron_path(A) || A->kron.map.nc != 1 || A->kron.t.B != 8 || A->kron.t.S < 8) return nullptr;
    if (A->has_comm) return reinterpret_cast<d2 *>(A->comm.d_xsend);      // (as packed real parts under the real wire: tiled_real)
    return (A->nrows == A->ncols && A->kron.xt_cap >= A->nrows) ? A->kron.d_xt : nullptr;
}

// does the tiled copy of a split shard's block travel as packed real parts (qbh_opts.real_wire, enabled per solve)?
inline int tiled_real(const qbh_csr *A) { return (A->has_comm && A->real_wire && A->kron.active && A->kron.comm_tiled) ? 1 : 0; }

// Only a driver knows that nothing else writes its vectors between the pass that produces x and the SpMV that reads it
// (a caller of the building-block entry points may scale or overwrite a vector in between): the drivers hold this guard.
// the coded split's tiled x (packed doubles): written by the all-real Lanczos step's axpy when the operator runs that form
inline double *kronc_tiled_target(const qbh_csr *A)
{
    return (A->kronc.active && A->kronc.sl.active && !A->kronc.table_route && !A->has_comm && !A->has_rem && A->opts.tile_fold) ? A->kronc.d_xt : nullptr;
}

struct FoldGuard {
    qbh_csr *A;
    explicit FoldGuard(qbh_csr *a) : A(a)
    {
        A->kron.xt_of = nullptr;
        A->kronc.xt_of = nullptr;
        A->kron.fold = true;
    }
    ~FoldGuard()
    {
        A->kron.xt_of = nullptr;
        A->kron.fold = false;
        A->kronc.xt_of = nullptr;
    }
};

struct WireGuard {            // whatever path a driver leaves by, the next call starts with the complex wire
    qbh_csr *A;
    ~WireGuard() { A->real_wire = false; A->real_mode = false; A->xr_of = nullptr; }
};

// where the packed real parts of the next gather source go in the real fast path (nullptr: not active)
inline double *packed_target(qbh_csr *A)
{
    if (!A->real_mode || A->kernel != QBH_KERNEL_ROWS) return nullptr;
    if (A->has_comm) return A->real_wire ? reinterpret_cast<double *>(A->comm.d_xsend) : nullptr;
    return (A->nrows == A->ncols) ? A->d_xr : nullptr;
}

}  // namespace qbhapi
