// qbh_solvers.cpp -- the device-resident drivers: lanczos (src/lanczos.cc:134-266), eigenvec_CG (src/lanczos.cc:281-341),
// the restarted Lanczos behind qbh_iram (src/lanczos.cc:497-603) and qbh_hess_eigen (src/lanczos.cc:355-390).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "qbh_api_priv.hpp"

using qbh::d2;
using namespace qbhapi;


// -------------------------------------------------------------- hess_eigen ------
extern "C" int qbh_hess_eigen(const double *hessenberg, int64_t maxit, int64_t m, const char *order,
                              double *ritz, double *s)
{
    if (!hessenberg || !order || !ritz || !s || m <= 0 || m >= maxit || strlen(order) < 2) {
        qbh::set_error("qbh_hess_eigen: invalid argument (need 0 < m < maxit)");
        return QBH_EINVAL;
    }
    std::vector<double> w((size_t)m), z((size_t)m * (size_t)m);
    QBH_TRY(qbh::tridiag_eigen_full(m, hessenberg + maxit, hessenberg + 1, w.data(), z.data()));
    const char o0 = (char)std::tolower((unsigned char)order[0]);
    const char o1 = (char)std::tolower((unsigned char)order[1]);
    if (!((o0 == 's' || o0 == 'l') && (o1 == 'r' || o1 == 'a' || o1 == 'm'))) {
        qbh::set_error("qbh_hess_eigen: order must be sr/lr/sm/lm");
        return QBH_EINVAL;
    }
    std::vector<int64_t> idx((size_t)m);
    for (int64_t j = 0; j < m; ++j) idx[j] = j;
    auto key = [&](int64_t j) {
        const double v = (o1 == 'm') ? std::fabs(w[j]) : w[j];
        return (o0 == 's') ? v : -v;
    };
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return key(a) < key(b); });
    for (int64_t j = 0; j < m; ++j) {
        ritz[j] = w[idx[j]];
        memcpy(s + (size_t)j * m, z.data() + (size_t)idx[j] * m, (size_t)m * sizeof(double));
    }
    return QBH_OK;
}

// ----------------------------------------------------------------- Lanczos ------
// ext_rv != nullptr: the caller's vectors ARE packed doubles (two slots of nrows doubles; qbh_lanczos_real_dev) -- the
// all-real path runs in place, nothing complex is ever allocated.  Otherwise d_v holds the reference's complex slots.
static void host_delay(const qbh_csr *A)
{
    const int us = A->dbg.host_delay_us;
    if (us <= 0) return;
    const double t0 = now_ms();
    while ((now_ms() - t0) * 1e3 < us) { }
}

static int lanczos_core(qbh_csr *A, int64_t k, int64_t np, int64_t maxit, int64_t *m_out, qbh_z *d_v, double *ext_rv,
                        double *hess, const char *purpose, qbh_solver_info *info)
{
    if (!A || !m_out || (!d_v && !ext_rv) || !hess || !purpose) return QBH_EINVAL;
    if (!A->has_comm && A->nrows != A->ncols) {
        qbh::set_error("qbh_lanczos: a row shard needs a communicator");
        return QBH_EINVAL;
    }
    const std::string pur(purpose);
    const bool is_val = pur.find("val") != std::string::npos;
    const bool is_val1 = pur.find("val1") != std::string::npos;
    const bool is_dn = pur == "dnmcs";
    if (!(is_val || is_dn)) {
        // "iram" and "*vec*" are dead branches in the reference (no caller): not provided
        qbh::set_error("qbh_lanczos: purpose '%s' not supported (sr_val0, sr_val1, dnmcs)", purpose);
        return QBH_EUNSUPP;
    }
    Bind bind(A);
    const double t_start = now_ms();
    const double prec = QBH_LANCZOS_PRECISION;
    const int64_t n = A->nrows;
    const int64_t mm = k + np;
    int64_t m = k;
    *m_out = m;
    if (!(mm < maxit && k >= 0 && np >= 0)) {              // assert at src/lanczos.cc:147
        qbh::set_error("qbh_lanczos: need k >= 0, np >= 0, k + np < maxit");
        return QBH_EINVAL;
    }
    if (info) {
        info->log_len = 0;
        info->n_matvec = 0;
        info->n_reorth = 0;
        info->ms_total = 0.0;
        info->ms_spmv = 0.0;
    }
    if (np == 0) return QBH_OK;                           // :150
    const int64_t spmv0 = A->stats.n_spmv;
    const double ms_spmv0 = A->stats.ms_spmv;

    d2 *v = reinterpret_cast<d2 *>(d_v);
    auto vpt = [&](int64_t j) { return v + (size_t)(j % 2) * (size_t)n; };   // :160
    d2 *phi = v + 2 * (size_t)n;                                             // :154
    double *a = hess + maxit, *b = hess;

    double nrm = 0.0;
    if (ext_rv) {
        if (is_val1 || A->has_comm || !A->values_real || A->kernel != QBH_KERNEL_ROWS || A->nrows != A->ncols || A->kron.active) {
            qbh::set_error("qbh_lanczos_real: needs a real operator on one GPU (row kernel / matrix-free), purpose sr_val0 or dnmcs");
            return QBH_EINVAL;
        }
        double sq0 = 0.0;
        QBH_TRY(qbh::launch_nrm2sq_re(ext_rv + (size_t)(k % 2) * (size_t)n, n, A->d_partials, A->stream));
        QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &sq0));
        nrm = std::sqrt(sq0);
    } else {
        QBH_TRY(nrm2_run(A, vpt(k), &nrm));               // assert at :166
    }
    if (!(std::fabs(nrm - 1.0) < prec)) {
        qbh::set_error("qbh_lanczos: |v[k]| - 1 = %.3e", nrm - 1.0);
        return QBH_ENOTNORM;
    }

    WireGuard wire_guard{A};
    FoldGuard fold_guard{A};
    if (ext_rv) {
        A->real_wire = false;
        A->real_mode = true;                               // no packed side buffer is needed: the vectors are the packed form
        A->xr_of = nullptr;
        QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream));
    }
    else if (is_val1) QBH_TRY(enable_real_wire(A, {vpt(k), phi}));
    else if (k > 0)  QBH_TRY(enable_real_wire(A, {vpt(k), vpt(k + 1)}));
    else             QBH_TRY(enable_real_wire(A, {vpt(k)}));
    if (is_val1 && k > 0 && A->real_wire) {       // the second live vector must be real as well
        double sq0 = 0.0;
        QBH_TRY(qbh::launch_imag_norm(vpt(k + 1), n, A->d_partials, A->stream));
        QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &sq0));
        if (sq0 != 0.0) A->real_wire = false;
    }

    // All-real vectors (one GPU, real operator, real Lanczos vectors and phi0): the slots (and phi0) live as packed doubles
    // for the whole solve -- the SpMV gathers from, reads and writes 8-byte elements and the axpy pass moves half the
    // bytes; (a+0i)(b+0i) = ab+0i exactly, so the coefficients are the same numbers.  Expanded back into v on exit.
    double *rv = nullptr;
    const bool rv_external = ext_rv != nullptr;
    struct RvGuard {
        double **p;
        const bool *ext;
        ~RvGuard() { if (*p && !*ext) (void)hipFree(*p); }
    } rv_guard{&rv, &rv_external};
    if (rv_external) rv = ext_rv;
    {
        const bool no_realvec = !(A->opts.real_forms & 4);
        if (!rv_external && !A->has_comm && A->real_mode && A->kernel == QBH_KERNEL_ROWS && A->nrows == A->ncols && !no_realvec) {
            if (qbh::dev_alloc(&rv, (size_t)(is_val1 ? 3 : 2) * (size_t)n * sizeof(double)) != hipSuccess) {
                (void)hipGetLastError();
                rv = nullptr;                         // no room: stay on the complex vectors
            } else {
                // live on entry: v_k, and v_{k-1} when the run continues (k > 0); the other slot is not read before
                // it is written (the bootstrap step runs with beta = 0)
                for (int j = 0; j < 2; ++j)
                    if (k > 0 || j == 0)
                        QBH_TRY(qbh::launch_pack_real(v + (size_t)j * (size_t)n, rv + (size_t)j * (size_t)n, n, A->d_flag, A->stream));
                if (is_val1) QBH_TRY(qbh::launch_pack_real(phi, rv + 2 * (size_t)n, n, A->d_flag, A->stream));   // phi0
                A->xr_of = nullptr;
            }
        }
    }
    auto rpt = [&](int64_t j) { return rv + (size_t)(j % 2) * (size_t)n; };
    double red[3], sq;
    // The 1/b normalisation (K7, src/lanczos.cc:214) is never a pass of its own: slot j%2 holds an
    // unnormalised u_j with v_j = sc[j%2] * u_j, and the scale is folded into the coefficients of the
    // next SpMV / axpy.  Both slots are scaled to unit norm once, on exit.
    double sc[2] = {1.0, 1.0};
    // one three-term step into slot mcur%2 given x = v[mcur-1]; bprev = b[mcur-1] (0 at bootstrap)
    auto step = [&](int64_t mcur, double bprev) -> int {
        const int sx = (int)((mcur - 1) % 2), sy = (int)(mcur % 2);
        // w = H v_{m-1} - b_{m-1} v_{m-2}  and  <u_{m-1}, w>                               K3+K1+K4
        if (rv != nullptr) {
            A->defer_red = true;
            A->ovr_xr = rpt(mcur - 1);
            A->ovr_yr = rpt(mcur);
            const int rc1 = spmv_run(A, nullptr, nullptr, sc[sx], -bprev * sc[sy], 0.0, red);
            A->defer_red = false;
            A->ovr_xr = nullptr;
            A->ovr_yr = nullptr;
            QBH_TRY(rc1);
            // the result is the next SpMV's x: a coded Kronecker split gets its tiled copy written here (as tiled_target does for
            // the complex128 form)
            double *yt = kronc_tiled_target(A);
            QBH_TRY(qbh::launch_axpy_norm_re(-sc[sx] * sc[sx], A->d_scal, rpt(mcur - 1), rpt(mcur), n, A->d_partials, A->stream, yt, A->kronc.t));
            A->kronc.xt_of = yt ? (const void *)rpt(mcur) : nullptr;
            QBH_TRY(qbh::launch_reduce_partials(A->d_partials, qbh::blas_grid(n), 1, A->d_scal + 4, A->stream));
            QBH_HIP(hipMemcpyAsync(A->h_scal, A->d_scal, 5 * sizeof(double), hipMemcpyDeviceToHost, A->stream));
            QBH_HIP(hipStreamSynchronize(A->stream));
            if (A->opts.profile) harvest_events(A);
            a[mcur - 1] = sc[sx] * A->h_scal[0];
            b[mcur] = std::sqrt(A->h_scal[4]);
            sc[sy] = 1.0 / b[mcur];
            return QBH_OK;
        }
        const bool no_defer = A->dbg.no_defer != 0;                   // A/B switch
        if (!A->has_comm && !no_defer) {
            // one GPU: <u, w> stays on the device and feeds the axpy directly; one host synchronisation per step
            A->defer_red = true;
            const int rc1 = spmv_run(A, vpt(mcur - 1), vpt(mcur), sc[sx], -bprev * sc[sy], 0.0, red);
            A->defer_red = false;
            QBH_TRY(rc1);
            double dot = 0.0;
            QBH_TRY(axpy_norm_deferred(A, -sc[sx] * sc[sx], vpt(mcur - 1), vpt(mcur), &dot, &sq));
            host_delay(A);
            a[mcur - 1] = sc[sx] * dot;
            b[mcur] = std::sqrt(sq);
            sc[sy] = 1.0 / b[mcur];
            return QBH_OK;
        }
        QBH_TRY(spmv_run(A, vpt(mcur - 1), vpt(mcur), sc[sx], -bprev * sc[sy], 0.0, red));
        a[mcur - 1] = sc[sx] * red[0];
        // w -= a v_{m-1} ; b = |w|                                                         K5+K6
        // (the coefficient in the association every form of the step uses: (-sc^2) * <u, w> -- the deferred and the pipelined loops
        // multiply on the device in this order, so a_j, b_j do not depend on which loop ran)
        QBH_TRY(axpy_norm_run(A, d2{(-sc[sx] * sc[sx]) * red[0], 0.0}, vpt(mcur - 1), vpt(mcur), &sq));
        b[mcur] = std::sqrt(sq);
        sc[sy] = 1.0 / b[mcur];
        return QBH_OK;
    };
    auto normalise_slots = [&]() -> int {
        A->kronc.xt_of = nullptr;                          // whatever happens to the slots below, no tiled copy describes them
        if (rv != nullptr && rv_external) {                // the caller's vectors are the packed doubles themselves
            for (int j = 0; j < 2; ++j)
                if (sc[j] != 1.0) {
                    QBH_TRY(qbh::launch_scal_re(sc[j], rv + (size_t)j * (size_t)n, n, A->stream));
                    sc[j] = 1.0;
                }
            return QBH_OK;
        }
        if (rv != nullptr) {                               // back to the caller's complex vectors
            for (int j = 0; j < 2; ++j)
                QBH_TRY(qbh::launch_unpack_real(rv + (size_t)j * (size_t)n, v + (size_t)j * (size_t)n, n, A->stream));
            QBH_HIP(hipStreamSynchronize(A->stream));
            (void)hipFree(rv);
            rv = nullptr;
        }
        for (int j = 0; j < 2; ++j)
            if (sc[j] != 1.0) {
                QBH_TRY(qbh::launch_scal(sc[j], v + (size_t)j * (size_t)n, n, A->stream));
                sc[j] = 1.0;
                A->xr_of = nullptr;
                A->kron.xt_of = nullptr;
            }
        return QBH_OK;
    };

    // convergence bookkeeping; restored from / returned in info->state so that a run can be resumed
    // exactly where a checkpoint left it (what ckpt_lanczos_init restores, src/ckpt.cc:38-176)
    double theta0_prev = 0.0, theta1_prev = 0.0, accuracy = 0.0;
    int cnt_accuE0 = 0;
    if (info && info->resume) {
        cnt_accuE0 = (int)info->cnt_accuE0;
        accuracy = info->accuracy;
        theta0_prev = info->theta0_prev;
        theta1_prev = info->theta1_prev;
        if (cnt_accuE0 > 15 && accuracy < prec) {          // already converged (src/lanczos.cc:149)
            QBH_TRY(normalise_slots());
            *m_out = m;
            return QBH_OK;
        }
    }
    std::vector<double> w((size_t)mm + 8), zl((size_t)mm + 8), ws((size_t)mm + 8);     // ws always holds four Ritz values
    int rc = QBH_OK;
    // the convergence test of src/lanczos.cc:228-247 after step m: 1 = converged (stop), 0 = go on, -1 = error (rc set)
    auto ritz_test = [&]() -> int {
        // the four lowest Ritz values and the last component of the lowest Ritz vector: all the test below
        // uses of hess_eigen's full decomposition (src/lanczos.cc:229-231), in O(m) instead of O(m^2..m^3)
        double zl0 = 0.0;
        const int nsm = (int)std::min<int64_t>(4, m);
        for (int q = 0; q < 4; ++q) ws[(size_t)q] = 0.0;
        rc = qbh::tridiag_lowest(m, a, b + 1, nsm, ws.data(), &zl0);
        if (rc == QBH_ENOCONV) {           // overflow guard of the twisted factorisation: fall back to QL
            rc = qbh::tridiag_eigen_lastrow(m, a, b + 1, w.data(), zl.data());
            if (rc != QBH_OK) return -1;
            int64_t imin = 0;
            for (int64_t j = 1; j < m; ++j)
                if (w[j] < w[imin]) imin = j;
            std::copy(w.begin(), w.begin() + m, ws.begin());
            std::partial_sort(ws.begin(), ws.begin() + nsm, ws.begin() + m);
            zl0 = zl[(size_t)imin];
        }
        if (rc != QBH_OK) return -1;
        const double ritz0 = ws[0], ritz1 = m > 1 ? ws[1] : 0.0;
        if (m > 3) {
            accuracy = std::fabs(b[m] * zl0);
            const double accu_E0 = std::fabs((ritz0 - theta0_prev) / ritz0);
            const double accu_E1 = std::fabs((ritz1 - theta1_prev) / ritz1);
            if (info && info->log && info->log_len < info->log_cap) {
                qbh_lanczos_row &r = info->log[info->log_len];
                r.k = m;
                for (int q = 0; q < 4; ++q) r.ritz[q] = ws[q];
                r.a_km1 = a[m - 1];
                r.b_k = b[m];
                r.accuracy = accuracy;
                r.accu_E0 = accu_E0;
                r.accu_E1 = accu_E1;
            }
            if (info) info->log_len++;
            if (accu_E0 < prec) cnt_accuE0++;
            else cnt_accuE0 = 0;
            if (cnt_accuE0 > 15 && accuracy < prec) return 1;   // :240
        }
        theta0_prev = ritz0;
        theta1_prev = ritz1;
        return 0;
    };

    // ---- the pipelined loop (qbh_opts.lanczos_pipeline): one GPU, complex vectors, no phi0 re-orthogonalisation ----
    // Step j turns v_{j-1} (and v_{j-2}) into u_j in three launches that never need the host: the SpMV reads alpha_j = 1/b_{j-1}
    // and beta_j = -b_{j-1}/b_{j-2} from device memory, the axpy its scale -1/b_{j-1}^2 likewise, and k_lanczos_tail turns the two
    // reduced sums into a_{j-1}, b_j and the coefficients of step j + 1.  The host enqueues step j + 1 BEFORE it waits for step j's
    // two numbers, so the Ritz test, the log row and every host-side latency run under the next SpMV.  u_j is written where
    // v_{j-3} was (three buffers in rotation: the caller's two slots and one the handle keeps): a speculative step that the test
    // then rules out -- convergence at :240, breakdown at :216 -- has destroyed nothing and is simply not counted.
    const bool no_defer_sw = A->dbg.no_defer != 0;
    // (under a communicator the two sums of a step are all-reduced in stream order between the same launches: still no host)
    bool pipe = A->opts.lanczos_pipeline != 0 && !is_val1 && rv == nullptr && !no_defer_sw && A->kind == 0 &&
                (A->has_comm || (A->nrows == A->ncols && !A->has_rem)) && (A->kron.active ? kron_path(A) : true);
    if (pipe) {
        qbh_csr::LzPipe &P = A->lz;
        if (P.cap < n) {
            if (P.d_buf) (void)hipFree(P.d_buf);
            P.d_buf = nullptr;
            P.cap = 0;
            if (qbh::dev_alloc(&P.d_buf, (size_t)n * sizeof(d2)) != hipSuccess) {
                (void)hipGetLastError();
                P.d_buf = nullptr;
                pipe = false;                              // no room for the third vector: the unpipelined loop
            } else {
                P.cap = n;
            }
        }
        if (pipe && !P.d_state) {
            QBH_HIP(qbh::dev_alloc(&P.d_state, 8 * sizeof(double)));
            QBH_HIP(hipHostMalloc(&P.h_log, (size_t)kLzRing * 4 * sizeof(double), hipHostMallocMapped));
            QBH_HIP(hipHostGetDevicePointer((void **)&P.d_log, P.h_log, 0));
            for (hipEvent_t &e : P.ev) QBH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    if (pipe) {
        qbh_csr::LzPipe &P = A->lz;
        struct KeepGuard {
            qbh_csr *A;
            ~KeepGuard() { A->ev_keep = false; A->ovr_yin = nullptr; A->ovr_coef = nullptr; A->defer_red = false; }
        } keep_guard{A};
        A->ev_keep = true;
        d2 *B[3] = {v, v + (size_t)n, P.d_buf};
        const int64_t k_in = k;
        const int c0 = (int)(k_in % 2);
        const int cyc[3] = {1 - c0, c0, 2};                // where v_{k-1}, v_k and u_{k+1} live; period 3 from there
        auto buf = [&](int64_t j) { return B[cyc[(j - (k_in - 1)) % 3]]; };
        int64_t enq = k_in;                                // last step enqueued
        auto enqueue = [&](int64_t j) -> int {
            const bool first = j == k_in + 1;              // its coefficients are the host's: sc = 1 for both vectors handed in
            d2 *x = buf(j - 1), *y = buf(j);
            A->defer_red = true;
            A->ovr_yin = buf(j - 2);
            A->ovr_coef = first ? nullptr : P.d_state;
            const int rc1 = spmv_run(A, x, y, 1.0, -b[k_in], 0.0, red);     // (alpha, beta ignored when the device names them)
            A->defer_red = false;
            A->ovr_yin = nullptr;
            A->ovr_coef = nullptr;
            QBH_TRY(rc1);
            const double *scale_dev = first ? nullptr : P.d_state + 2;
            double *ds = scal_buf(A);                      // <u, w> of this step (summed over the ranks), left there by the deferred SpMV
            double *yr = packed_target(A);
            if (d2 *yt = tiled_target(A)) {
                QBH_TRY(qbh::launch_axpy_norm_tile(d2{-1.0, 0.0}, ds, x, y, yt, n, A->kron.t, A->d_partials, A->stream, scale_dev, tiled_real(A), A->d_flag));
                A->kron.xt_of = y;
                A->xr_of = nullptr;
            } else {
                QBH_TRY(qbh::launch_axpy_norm(d2{-1.0, 0.0}, ds, x, y, n, A->d_partials, yr, A->d_flag, A->stream, scale_dev));
                A->xr_of = yr ? y : nullptr;
                A->kron.xt_of = nullptr;
            }
            const int slot = (int)(j % kLzRing);
            if (A->has_comm) {                             // |w'|^2: local sum, sum over the ranks, then the scalars -- all in stream order
                QBH_TRY(qbh::launch_reduce_partials(A->d_partials, qbh::blas_grid(n), 1, ds + 4, A->stream));
                if (A->comm.allreduce_sum(A->comm.ctx, 4, 1) != 0) {
                    qbh::set_error("allreduce_sum hook failed");
                    return QBH_ECOMM;
                }
                QBH_TRY(qbh::launch_lanczos_tail(A->d_partials, 0, ds, P.d_state, P.d_log + 4 * slot, 1.0, first ? 1 : 0, A->stream, ds + 4));
            } else
            QBH_TRY(qbh::launch_lanczos_tail(A->d_partials, qbh::blas_grid(n), A->d_scal, P.d_state, P.d_log + 4 * slot, 1.0, first ? 1 : 0, A->stream));
            QBH_HIP(hipEventRecord(P.ev[slot], A->stream));
            enq = j;
            return QBH_OK;
        };
        // a_{j-1}, b_j of step j; with `speculate` the step after it goes out first
        auto accept = [&](int64_t j, bool speculate) -> int {
            if (speculate && enq == j && !A->dbg.pipe_nospec) QBH_TRY(enqueue(j + 1));
            const int slot = (int)(j % kLzRing);
            QBH_HIP(hipEventSynchronize(P.ev[slot]));
            host_delay(A);
            const volatile double *L = P.h_log + 4 * slot;
            a[j - 1] = L[2];
            b[j] = L[3];
            return QBH_OK;
        };
        if (k == 0) {                                      // :167-191 -- the bootstrap step is never tested, the do-while always follows
            b[0] = 0.0;
            QBH_TRY(enqueue(1));
            QBH_TRY(accept(1, true));
            m = ++k;
            --np;
        }
        do {                                               // :193
            m++;
            if (enq < m) rc = enqueue(m);
            if (rc == QBH_OK) rc = accept(m, m < mm);
            if (rc != QBH_OK) break;
            if (std::fabs(b[m]) < prec) break;             // :216
            if (is_val && ritz_test() != 0) break;         // :228-247
        } while (m < mm);
        if (enq > m) {                                     // the speculative step: not a step of this run
            A->stats.n_spmv--;
            A->ev_drop = true;
            A->kron.xt_of = nullptr;
            A->xr_of = nullptr;
        }
        if (rc == QBH_OK) {
            // v_m and v_{m-1} = u / b into the caller's slots m % 2 and (m - 1) % 2 (src/qbasis.h:1056-1058): the normalisation
            // pass the unpipelined loop makes on exit, out of place where the rotation left a vector in another buffer
            auto scale_of = [&](int64_t j) { return j > k_in ? 1.0 / b[j] : 1.0; };
            struct Mv { d2 *src, *dst; double sc; } mv[2] = {{buf(m), B[m % 2], scale_of(m)}, {buf(m - 1), B[(m - 1) % 2], scale_of(m - 1)}};
            auto move = [&](const Mv &q) -> int {
                if (q.src == q.dst) return q.sc != 1.0 ? qbh::launch_scal(q.sc, q.dst, n, A->stream) : QBH_OK;
                return qbh::launch_scal_to(q.sc, q.src, q.dst, n, A->stream);
            };
            if (mv[0].dst == mv[1].src && mv[1].dst == mv[0].src) {        // exchanged: through the free buffer
                d2 *spare = B[2];                          // both sources are the caller's slots, so the handle's buffer is the free one
                rc = qbh::launch_scal_to(mv[0].sc, mv[0].src, spare, n, A->stream);
                if (rc == QBH_OK) rc = move(mv[1]);
                if (rc == QBH_OK) rc = qbh::launch_scal_to(1.0, spare, mv[0].dst, n, A->stream);
            } else if (mv[0].dst == mv[1].src) {
                rc = move(mv[1]);
                if (rc == QBH_OK) rc = move(mv[0]);
            } else {
                rc = move(mv[0]);
                if (rc == QBH_OK) rc = move(mv[1]);
            }
            A->kron.xt_of = nullptr;
            A->xr_of = nullptr;
        }
    } else {
    if (k == 0) {                                          // :167-191
        b[0] = 0.0;
        QBH_TRY(step(1, 0.0));
        m = ++k;
        --np;
    }
    do {                                                   // :193
        m++;
        rc = step(m, b[m - 1]);
        if (rc != QBH_OK) break;
        if (std::fabs(b[m]) < prec) break;                 // :216

        if (is_val1) {                                     // :218-226
            double t[2];
            const int sy = (int)(m % 2);
            if (rv != nullptr) {
                t[1] = 0.0;
                rc = qbh::launch_dot_re(rv + 2 * (size_t)n, rpt(m), n, A->d_partials, A->stream);
                if (rc == QBH_OK) rc = finish_reduction(A, qbh::blas_grid(n), 1, t);
            } else {
                rc = dotc_run(A, phi, vpt(m), t);              // <phi0, u_m>; <phi0, v_m> = sc * that
            }
            if (rc != QBH_OK) break;
            if (sc[sy] * std::hypot(t[0], t[1]) > prec) {
                if (rv != nullptr) {
                    double *yt = kronc_tiled_target(A);
                    rc = qbh::launch_axpy_norm_re(-t[0], nullptr, rv + 2 * (size_t)n, rpt(m), n, A->d_partials, A->stream, yt, A->kronc.t);
                    A->kronc.xt_of = yt ? (const void *)rpt(m) : nullptr;
                    if (rc == QBH_OK) rc = finish_reduction(A, qbh::blas_grid(n), 1, &sq);
                } else {
                    rc = axpy_norm_run(A, d2{-t[0], -t[1]}, phi, vpt(m), &sq);   // u_m -= <phi0,u_m> phi0
                }
                if (rc != QBH_OK) break;
                sc[sy] = 1.0 / std::sqrt(sq);                  // renormalise
                if (info) info->n_reorth++;
            }
        }

        if (is_val) {                                      // :228-247
            const int t = ritz_test();
            if (t != 0) break;
        }
    } while (m < mm);
    }
    if (rc == QBH_OK) rc = finish_real_wire(A);
    if (rc == QBH_OK) rc = normalise_slots();
    if (rc == QBH_OK) {
        hipError_t e = hipStreamSynchronize(A->stream);
        if (e != hipSuccess) {
            qbh::set_error("stream sync failed: %s", hipGetErrorString(e));
            rc = QBH_EHIP;
        }
    }
    harvest_events(A);
    *m_out = m;
    if (info) {
        info->cnt_accuE0 = cnt_accuE0;
        info->accuracy = accuracy;
        info->theta0_prev = theta0_prev;
        info->theta1_prev = theta1_prev;
        if (info->log && info->log_len > info->log_cap) info->log_len = info->log_cap;
        info->n_matvec = A->stats.n_spmv - spmv0;
        info->ms_spmv = A->stats.ms_spmv - ms_spmv0;
        info->ms_total = now_ms() - t_start;
    }
    return rc;
}

extern "C" int qbh_lanczos_dev(const qbh_csr *Ac, int64_t k, int64_t np, int64_t maxit, int64_t *m_out,
                               qbh_z *d_v, double *hess, const char *purpose, qbh_solver_info *info)
{
    if (!d_v) return QBH_EINVAL;
    return lanczos_core(const_cast<qbh_csr *>(Ac), k, np, maxit, m_out, d_v, nullptr, hess, purpose, info);
}

extern "C" int qbh_lanczos_real_dev(const qbh_csr *Ac, int64_t k, int64_t np, int64_t maxit, int64_t *m_out,
                                    double *d_v, double *hess, const char *purpose, qbh_solver_info *info)
{
    if (!d_v) return QBH_EINVAL;
    return lanczos_core(const_cast<qbh_csr *>(Ac), k, np, maxit, m_out, nullptr, d_v, hess, purpose, info);
}

extern "C" int qbh_vec_randomize_real(const qbh_csr *Ac, double *d_x, uint32_t seed)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x || seed == 0) return QBH_EINVAL;
    Bind bind(A);
    const int64_t nruns = (A->nrows + 15) / 16;
    double *xr = d_x;                            // the Lehmer stream is indexed by the CALLER's element number (as qbh_vec_randomize)
    if (A->basis.kind != 0) {
        if (!A->basis.d_stage) QBH_HIP(qbh::dev_alloc(&A->basis.d_stage, (size_t)A->nrows * sizeof(d2)));
        xr = reinterpret_cast<double *>(A->basis.d_stage);
    }
    QBH_TRY(qbh::launch_randomize(nullptr, xr, A->nrows, A->has_comm ? A->row_offset : 0, seed, A->d_partials, A->stream));
    double sq = 0.0;
    QBH_TRY(finish_reduction(A, qbh::blas_grid(nruns), 1, &sq));
    if (xr != d_x) QBH_TRY(qbh::launch_basis_scatter_re(A->basis.d_map, xr, d_x, A->nrows, A->stream));
    return qbh::launch_scal_re(1.0 / std::sqrt(sq), d_x, A->nrows, A->stream);
}

extern "C" int qbh_lanczos(const qbh_csr *A, int64_t k, int64_t np, int64_t maxit, int64_t *m, qbh_z *v_host,
                           double *hessenberg, const char *purpose, qbh_solver_info *info)
{
    if (!A || !v_host || !purpose) return QBH_EINVAL;
    Bind bind(A);
    const bool val1 = std::string(purpose).find("val1") != std::string::npos;
    const int64_t nvec = val1 ? 3 : 2;
    const size_t bytes = (size_t)nvec * (size_t)A->nrows * sizeof(qbh_z);
    qbh_z *d_v = nullptr;
    QBH_HIP(qbh::dev_alloc((void **)&d_v, bytes));
    int rc = QBH_OK;
    rc = vec_h2d(const_cast<qbh_csr *>(A), reinterpret_cast<d2 *>(d_v), v_host, nvec * A->nrows);
    if (rc == QBH_OK) rc = qbh_lanczos_dev(A, k, np, maxit, m, d_v, hessenberg, purpose, info);
    // on exit the last two Lanczos vectors are returned (src/qbasis.h:1056-1058); phi0 is read-only
    if (rc == QBH_OK) rc = vec_d2h(const_cast<qbh_csr *>(A), v_host, reinterpret_cast<const d2 *>(d_v), 2 * A->nrows);
    (void)hipFree(d_v);
    return rc;
}

// ---------------------------------------------------------------------- CG ------
// ext != nullptr: the caller's four vectors are packed doubles (qbh_eigenvec_cg_real_dev): the all-real loop runs in place
static int cg_core(qbh_csr *A, int64_t maxit, int64_t *m_io, double E0, double *accu_out, qbh_z *d_v, qbh_z *d_r, qbh_z *d_p,
                   qbh_z *d_pp, double *const *ext, qbh_solver_info *info)
{
    if (!A || !m_io || !accu_out || (!ext && (!d_v || !d_r || !d_p || !d_pp))) return QBH_EINVAL;
    if (!A->has_comm && A->nrows != A->ncols) return QBH_EINVAL;
    Bind bind(A);
    const double t_start = now_ms();
    const double prec = QBH_LANCZOS_PRECISION;
    const double machine_prec = std::numeric_limits<double>::epsilon();
    const int64_t n = A->nrows;
    d2 *v = reinterpret_cast<d2 *>(d_v), *r = reinterpret_cast<d2 *>(d_r);
    d2 *p = reinterpret_cast<d2 *>(d_p), *pp = reinterpret_cast<d2 *>(d_pp);
    int64_t m = *m_io;
    if (!(m >= 0 && m < maxit)) {                           // assert at src/lanczos.cc:287
        qbh::set_error("qbh_eigenvec_cg: need 0 <= m < maxit");
        return QBH_EINVAL;
    }
    const int64_t spmv0 = A->stats.n_spmv;
    const double ms_spmv0 = A->stats.ms_spmv;
    WireGuard wire_guard{A};
    FoldGuard fold_guard{A};
    if (ext) {
        if (A->has_comm || !A->values_real || A->kernel != QBH_KERNEL_ROWS || A->nrows != A->ncols || A->kron.active) {
            qbh::set_error("qbh_eigenvec_cg_real: needs a real operator on one GPU (row kernel / matrix-free)");
            return QBH_EINVAL;
        }
        A->real_wire = false;
        A->real_mode = true;
        A->xr_of = nullptr;
        QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream));
    }
    else if (m != 0) QBH_TRY(enable_real_wire(A, {v, r, p}));
    else             QBH_TRY(enable_real_wire(A, {v}));
    double accu = 0.0;
    double red[3], sq;
    // All-real vectors, as in qbh_lanczos_dev: one GPU, real operator, real v (and r, p when the run continues): the
    // four CG vectors live as packed doubles for the whole solve and are expanded back on exit.
    double *rv = nullptr;
    struct RvGuard {
        double **p;
        ~RvGuard() { if (*p) (void)hipFree(*p); }
    } rv_guard{&rv};
    {
        const bool no_realvec = !(A->opts.real_forms & 4);
        if (!ext && !A->has_comm && A->real_mode && A->kernel == QBH_KERNEL_ROWS && A->nrows == A->ncols && !no_realvec) {
            if (qbh::dev_alloc(&rv, (size_t)4 * (size_t)n * sizeof(double)) != hipSuccess) {
                (void)hipGetLastError();
                rv = nullptr;
            }
        }
    }
    if (rv != nullptr || ext) {
        double *vr = ext ? ext[0] : rv, *rr = ext ? ext[1] : rv + (size_t)n, *pr = ext ? ext[2] : rv + 2 * (size_t)n,
               *ppr = ext ? ext[3] : rv + 3 * (size_t)n;
        if (!ext) {
            QBH_TRY(qbh::launch_pack_real(v, vr, n, A->d_flag, A->stream));
            if (m != 0) {
                QBH_TRY(qbh::launch_pack_real(r, rr, n, A->d_flag, A->stream));
                QBH_TRY(qbh::launch_pack_real(p, pr, n, A->d_flag, A->stream));
            }
        }
        A->xr_of = nullptr;
        auto nrm2_re = [&](const double *x, double *out) -> int {
            double s2 = 0.0;
            QBH_TRY(qbh::launch_nrm2sq_re(x, n, A->d_partials, A->stream));
            QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &s2));
            *out = std::sqrt(s2);
            return QBH_OK;
        };
        auto spmv_re = [&](const double *x, double *y, double al, double be, double ga) -> int {
            A->ovr_xr = x;
            A->ovr_yr = y;
            const int rc1 = spmv_run(A, nullptr, nullptr, al, be, ga, red);
            A->ovr_xr = nullptr;
            A->ovr_yr = nullptr;
            return rc1;
        };
        if (m != 0) QBH_TRY(nrm2_re(rr, &accu));           // :290
        while (m < maxit) {
            if (accu < prec) {
                double rnorm = 0.0;
                QBH_TRY(nrm2_re(vr, &rnorm));
                if (m == 0 || std::fabs(rnorm - 1.0) > prec) {  // re-normalise and restart, :297-317
                    QBH_TRY(qbh::launch_scal_re(1.0 / rnorm, vr, n, A->stream));
                    QBH_TRY(spmv_re(vr, rr, -1.0, 0.0, E0));                 // r = (E0 - H) v
                    QBH_HIP(hipMemcpyAsync(pr, rr, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, A->stream));
                    accu = std::sqrt(red[2]);
                    m++;
                    if (info && info->cg_resid) info->cg_resid[m] = accu;
                    if (accu < prec) break;
                } else {
                    break;
                }
            } else {
                QBH_TRY(spmv_re(pr, ppr, 1.0, 0.0, machine_prec - E0));      // pp = (H - E0) p, delta = <p,pp>  :319-323
                const double alpha = accu * accu / red[0];
                QBH_TRY(qbh::launch_cg_update_re(alpha, pr, ppr, vr, rr, n, A->d_partials, A->stream));   // :324-325
                QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &sq));
                const double beta = std::sqrt(sq) / accu;                    // :326
                QBH_TRY(qbh::launch_xpby_re(rr, beta * beta, pr, n, A->stream));   // :327-328
                accu *= beta;
                m++;
                if (info && info->cg_resid) info->cg_resid[m] = accu;
            }
        }
        if (!ext) {
            QBH_TRY(qbh::launch_unpack_real(vr, v, n, A->stream));
            QBH_TRY(qbh::launch_unpack_real(rr, r, n, A->stream));
            QBH_TRY(qbh::launch_unpack_real(pr, p, n, A->stream));
            QBH_TRY(qbh::launch_unpack_real(ppr, pp, n, A->stream));   // the reference leaves pp = (H - E0) p there (:322)
            QBH_HIP(hipStreamSynchronize(A->stream));
            (void)hipFree(rv);
            rv = nullptr;
        }
        m = -m - 1;                                         // done: skip the complex loop below
    }
    if (m >= 0 && m != 0) QBH_TRY(nrm2_run(A, r, &accu));  // :290
    const bool did_real = m < 0;
    if (did_real) m = -m - 1;
    while (!did_real && m < maxit) {
        if (accu < prec) {
            double rnorm = 0.0;
            QBH_TRY(nrm2_run(A, v, &rnorm));
            if (m == 0 || std::fabs(rnorm - 1.0) > prec) {  // re-normalise and restart, :297-317
                QBH_TRY(qbh::launch_scal(1.0 / rnorm, v, n, A->stream));
                QBH_TRY(spmv_run(A, v, r, -1.0, 0.0, E0, red));            // r = (E0 - H) v
                QBH_HIP(hipMemcpyAsync(p, r, (size_t)n * sizeof(d2), hipMemcpyDeviceToDevice, A->stream));
                accu = std::sqrt(red[2]);
                m++;
                if (info && info->cg_resid) info->cg_resid[m] = accu;
                if (accu < prec) break;
            } else {
                break;
            }
        } else {
            // pp = (H - E0) p with the reference's (machine_prec - E0) shift, delta = <p,pp>  :319-323
            if (A->opts.lanczos_pipeline != 0 && !A->dbg.no_defer) {
                // delta stays on the device (summed over the ranks in stream order) and the update pass forms alpha itself: one host
                // synchronisation per CG step instead of two.  (The stop test of :293 cannot run a step behind without out-of-place
                // copies of v, r, p: the remaining one stays.)
                A->defer_red = true;
                const int rc1 = spmv_run(A, p, pp, 1.0, 0.0, machine_prec - E0, red);
                A->defer_red = false;
                QBH_TRY(rc1);
                QBH_TRY(qbh::launch_cg_update(d2{0.0, 0.0}, p, pp, v, r, n, A->d_partials, A->stream, scal_buf(A), accu * accu));   // :324-325
            } else {
                QBH_TRY(spmv_run(A, p, pp, 1.0, 0.0, machine_prec - E0, red));
                const double den = red[0] * red[0] + red[1] * red[1];
                const d2 alpha = {accu * accu * red[0] / den, -accu * accu * red[1] / den};
                QBH_TRY(qbh::launch_cg_update(alpha, p, pp, v, r, n, A->d_partials, A->stream));   // :324-325
            }
            QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &sq));
            const double beta = std::sqrt(sq) / accu;                                        // :326
            {
                double *pr = packed_target(A);      // p is the next SpMV's x: emit its packed real copy in the same pass
                if (d2 *pt = tiled_target(A)) {     // ... or its tiled copy (Kronecker split)
                    QBH_TRY(qbh::launch_xpby_tile(r, beta * beta, p, pt, n, A->kron.t, A->stream, tiled_real(A), A->d_flag));
                    A->kron.xt_of = p;
                    pr = nullptr;
                } else {
                    QBH_TRY(qbh::launch_xpby(r, beta * beta, p, n, pr, A->d_flag, A->stream));   // :327-328
                    A->kron.xt_of = nullptr;
                }
                A->xr_of = pr ? p : nullptr;
            }
            accu *= beta;
            m++;
            if (info && info->cg_resid) info->cg_resid[m] = accu;
        }
    }
    QBH_TRY(finish_real_wire(A));
    QBH_HIP(hipStreamSynchronize(A->stream));
    harvest_events(A);
    *m_io = m;
    *accu_out = accu;
    if (info) {
        info->n_matvec = A->stats.n_spmv - spmv0;
        info->ms_spmv = A->stats.ms_spmv - ms_spmv0;
        info->ms_total = now_ms() - t_start;
    }
    return QBH_OK;
}

extern "C" int qbh_eigenvec_cg_dev(const qbh_csr *Ac, int64_t maxit, int64_t *m_io, double E0, double *accu_out,
                                   qbh_z *d_v, qbh_z *d_r, qbh_z *d_p, qbh_z *d_pp, qbh_solver_info *info)
{
    return cg_core(const_cast<qbh_csr *>(Ac), maxit, m_io, E0, accu_out, d_v, d_r, d_p, d_pp, nullptr, info);
}

extern "C" int qbh_eigenvec_cg_real_dev(const qbh_csr *Ac, int64_t maxit, int64_t *m_io, double E0, double *accu_out,
                                        double *d_v, double *d_r, double *d_p, double *d_pp, qbh_solver_info *info)
{
    if (!d_v || !d_r || !d_p || !d_pp) return QBH_EINVAL;
    double *ext[4] = {d_v, d_r, d_p, d_pp};
    return cg_core(const_cast<qbh_csr *>(Ac), maxit, m_io, E0, accu_out, nullptr, nullptr, nullptr, nullptr, ext, info);
}

extern "C" int qbh_eigenvec_cg(const qbh_csr *A, int64_t maxit, int64_t *m, double E0, double *accu,
                               qbh_z *v_host, qbh_z *r_host, qbh_z *p_host, qbh_z *pp_host,
                               qbh_solver_info *info)
{
    if (!A || !v_host || !r_host || !p_host || !pp_host) return QBH_EINVAL;
    Bind bind(A);
    const size_t n = (size_t)A->nrows, bytes = n * sizeof(qbh_z);
    qbh_z *d = nullptr;
    QBH_HIP(qbh::dev_alloc((void **)&d, 4 * bytes));
    int rc = QBH_OK;
    qbh_z *hv[4] = {v_host, r_host, p_host, pp_host};
    qbh_csr *Am = const_cast<qbh_csr *>(A);
    for (int i = 0; i < 3 && rc == QBH_OK; ++i)      // pp is scratch on entry
        rc = vec_h2d(Am, reinterpret_cast<d2 *>(d + i * n), hv[i], (int64_t)n);
    if (rc == QBH_OK) rc = qbh_eigenvec_cg_dev(A, maxit, m, E0, accu, d, d + n, d + 2 * n, d + 3 * n, info);
    for (int i = 0; i < 4 && rc == QBH_OK; ++i) rc = vec_d2h(Am, hv[i], reinterpret_cast<const d2 *>(d + i * n), (int64_t)n);
    (void)hipFree(d);
    return rc;
}

// -------------------------------------------------------------------- IRAM -------
// Device-resident replacement of iram<T,csr_mat<T>> -> call_arpack (src/lanczos.cc:438-603).
// ARPACK's implicitly restarted Arnoldi process applied to a Hermitian operator is a restarted
// Lanczos process; here it is run as thick-restart Lanczos (Wu & Simon) with the Krylov basis
// V[ncv+1][n] resident in HBM, two passes of classical Gram-Schmidt against the whole basis
// (k_multi_dot / k_multi_axpy), the ncv x ncv projected problem on the host (Jacobi), and the
// restart rotation V <- V S on the device.  Same convergence rule as ARPACK's dsconv/znconv:
// |beta * s_last,i| <= tol * max(eps^(2/3), |theta_i|), tol <= 0 meaning machine epsilon
// (src/lanczos.cc:452).  The start vector is random (ARPACK info = 0, src/lanczos.cc:470).
extern "C" int qbh_iram(const qbh_csr *Ac, int64_t nev, int64_t ncv, int64_t maxit, const char *order, double tol,
                        uint32_t seed, int64_t *nconv_out, double *eigenvals, qbh_z *eigenvecs_host,
                        qbh_solver_info *info)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !order || !nconv_out || !eigenvals || strlen(order) < 2) return QBH_EINVAL;
    if (!A->has_comm && A->nrows != A->ncols) return QBH_EINVAL;
    const int64_t dim = A->ncols, n = A->nrows;
    if (nev <= 0 || nev >= dim - 1) {                       // src/lanczos.cc:502
        qbh::set_error("0 < nev < N-1 should be satisfied.");
        return QBH_EINVAL;
    }
    if (ncv < nev + 2 || ncv > dim) {
        qbh::set_error("qbh_iram: need nev + 2 <= ncv <= dim");
        return QBH_EINVAL;
    }
    if (ncv > 64) {
        qbh::set_error("qbh_iram: ncv > 64 not supported on the device path");
        return QBH_EUNSUPP;
    }
    if (maxit < 1) return QBH_EINVAL;
    const char o0 = (char)std::tolower((unsigned char)order[0]), o1 = (char)std::tolower((unsigned char)order[1]);
    if (!((o0 == 's' || o0 == 'l') && (o1 == 'r' || o1 == 'a'))) {
        qbh::set_error("qbh_iram: order '%s' not supported on the device path (sr, lr)", order);
        return QBH_EUNSUPP;
    }
    const double sign = (o0 == 's') ? 1.0 : -1.0;            // largest of H = smallest of -H
    Bind bind(A);
    const double t_start = now_ms();
    const int64_t spmv0 = A->stats.n_spmv;
    const double ms_spmv0 = A->stats.ms_spmv;
    const int m = (int)ncv;
    const double eps = std::numeric_limits<double>::epsilon();
    const double eps23 = std::pow(eps, 2.0 / 3.0);
    const double tol_eff = tol > 0.0 ? tol : eps;

    // The start vector is real; when the operator is real too (one GPU, row kernel) the whole Krylov basis is kept
    // as packed doubles: a vector of n doubles (padded to an even count) IS a complex vector of n/2 elements for every
    // BLAS-1 kernel below (real inner products = real parts, real coefficients), the SpMV runs all-real, and the
    // orthogonalisation -- the dominant cost at ncv = 32 -- moves half the bytes.
    d2 *V = nullptr;
    double *d_S = nullptr;
    WireGuard wire_guard{A};
    int rc = QBH_OK;
    bool all_real = false;
    int64_t nc = n, ldr = 0;                  // complex length / leading dimension the BLAS-1 kernels see
    {
        d2 *v0 = nullptr;
        QBH_HIP(qbh::dev_alloc(&v0, (size_t)n * sizeof(d2)));
        rc = qbh_vec_randomize(A, reinterpret_cast<qbh_z *>(v0), seed ? seed : 1u);
        if (rc == QBH_OK) rc = enable_real_wire(A, {v0});      // the random start vector is real
        const bool no_realvec = !(A->opts.real_forms & 4);
        all_real = rc == QBH_OK && !A->has_comm && A->real_mode && A->kernel == QBH_KERNEL_ROWS && A->nrows == A->ncols && !no_realvec;
        hipError_t e0 = hipSuccess;
        if (all_real) {
            ldr = n + (n & 1);
            nc = ldr / 2;
            e0 = qbh::dev_alloc(&V, (size_t)(m + 1) * (size_t)ldr * sizeof(double));
            // every vector is written in full by the SpMV (beta = 0) before it is read; only the padding element
            // of an odd dimension has to be zero
            for (int j = 0; e0 == hipSuccess && (n & 1) && j <= m; ++j)
                e0 = hipMemsetAsync(reinterpret_cast<double *>(V) + (size_t)j * (size_t)ldr + n, 0, sizeof(double), A->stream);
            if (e0 == hipSuccess && rc == QBH_OK)
                rc = qbh::launch_pack_real(v0, reinterpret_cast<double *>(V), n, A->d_flag, A->stream);
        } else {
            e0 = qbh::dev_alloc(&V, (size_t)(m + 1) * (size_t)n * sizeof(d2));
            if (e0 == hipSuccess)
                e0 = hipMemcpyAsync(V, v0, (size_t)n * sizeof(d2), hipMemcpyDeviceToDevice, A->stream);
        }
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(A->stream);
        (void)hipFree(v0);
        if (e0 != hipSuccess) {
            if (V) (void)hipFree(V);
            qbh::set_error("qbh_iram: Krylov basis allocation failed: %s", hipGetErrorString(e0));
            return e0 == hipErrorOutOfMemory ? QBH_ENOMEM : QBH_EHIP;
        }
    }
    hipError_t e = qbh::dev_alloc(&d_S, 64 * 64 * sizeof(double));
    if (e != hipSuccess) {
        (void)hipFree(V);
        return QBH_ENOMEM;
    }
    const int64_t ldc = all_real ? nc : n;     // in complex elements
    auto vec = [&](int j) { return V + (size_t)j * (size_t)ldc; };
    auto rvec = [&](int j) { return reinterpret_cast<double *>(V) + (size_t)j * (size_t)ldr; };

    std::vector<double> T((size_t)m * m, 0.0), Tw((size_t)m * m), theta((size_t)m), S((size_t)m * m);
    int k = 0;                          // vectors kept from the previous restart
    d2 *d_fresh = nullptr;              // breakdown: a fresh random vector (allocated when the first one is needed)
    int n_fresh = 0;
    int64_t restarts = 0, nconv = 0;
    double beta_last = 0.0;
    double red[16];
    // One classical Gram-Schmidt pass of w against V_0..V_{nv-1}: h = V^H w (8 inner products per sweep over
    // w), w -= V h, and |w|^2 of the result from the last sweep.  hj receives Re h_{nv-1}.
    auto cgs_pass = [&](d2 *w, int nv, double *hj, double *nrm2sq) -> int {
        std::vector<double> h((size_t)2 * nv);
        for (int i0 = 0; i0 < nv; i0 += 8) {
            const int cnt = std::min(8, nv - i0);
            QBH_TRY(qbh::launch_multi_dot8(vec(i0), ldc, w, nc, cnt, A->d_partials, A->stream));
            QBH_TRY(finish_reduction(A, qbh::blas_grid(nc), 16, red));
            for (int i = 0; i < 2 * cnt; ++i) h[(size_t)2 * i0 + i] = (all_real && (i & 1)) ? 0.0 : red[i];
        }
        for (int i0 = 0; i0 < nv; i0 += 8) {
            const int cnt = std::min(8, nv - i0);
            const bool last = i0 + 8 >= nv;
            qbh::Coef8 c{};
            for (int i = 0; i < 2 * cnt; ++i) c.v[i] = h[(size_t)2 * i0 + i];
            QBH_TRY(qbh::launch_multi_axpy8(vec(i0), ldc, c, cnt, w, nc, last ? A->d_partials : nullptr, A->stream));
        }
        QBH_TRY(finish_reduction(A, qbh::blas_grid(nc), 1, nrm2sq));
        *hj = h[(size_t)2 * (nv - 1)];
        return QBH_OK;
    };

    while (rc == QBH_OK) {
        for (int j = k; j < m && rc == QBH_OK; ++j) {
            d2 *w = vec(j + 1);
            // w = (+-H) v_j.  The three-term recurrence is not applied separately: h = V^H w contains alpha_j,
            // beta_{j-1} (or the arrowhead couplings right after a restart) and the rounding-level
            // components along the older vectors, and all of them are removed in one pass (ARPACK does the
            // same in its Arnoldi step, followed by one DGKS correction when cancellation was severe).
            if (all_real) {
                A->ovr_xr = rvec(j);
                A->ovr_yr = rvec(j + 1);
                rc = spmv_run(A, nullptr, nullptr, sign, 0.0, 0.0, red);
                A->ovr_xr = nullptr;
                A->ovr_yr = nullptr;
            } else {
                rc = spmv_run(A, vec(j), w, sign, 0.0, 0.0, red);
            }
            if (rc != QBH_OK) break;
            const double wnorm2 = red[2];
            double alpha = 0.0, b2 = 0.0;
            rc = cgs_pass(w, j + 1, &alpha, &b2);
            if (rc != QBH_OK) break;
            // DGKS correction ("twice is enough"): a second pass whenever the first one removed more than half of |w|^2 -- ARPACK's own
            // criterion (dsaitr step 4: rnorm1 <= 0.717 rnorm).  Until round 6 the second pass ran only below 0.02 |w|^2 ("a Hamiltonian with a
            // large diagonal would otherwise trigger it on every step"): tools/r6/fuzz_solvers.py found operators whose spectrum lies to one side
            // of zero (more than half filling: every state has doubly occupied sites) for which the basis then LOST its orthogonality over the
            // restarts -- Ritz values far outside the spectrum, returned as converged.  A numpy emulation of this loop reproduces both the
            // failure at 0.02 and its absence at 0.5.  The cost: for such operators every step pays the second pass.
            bool broke = false;
            if (b2 < 0.5 * wnorm2) {
                const double b2_first = b2;
                double corr = 0.0;
                rc = cgs_pass(w, j + 1, &corr, &b2);
                if (rc != QBH_OK) break;
                alpha += corr;
                broke = b2 < 0.5 * b2_first;                 // the second pass took most of what was left (ARPACK: rnorm1 <= 0.717 rnorm, dsaitr step 5): w lies in span(V)
            }
            // BREAKDOWN: H v_j lies in span(V) to rounding -- the Krylov space of the start vector is an invariant subspace of fewer than ncv
            // dimensions (a small or highly degenerate operator; found by tools/r6/fuzz_solvers.py: dividing the rounding noise by its norm gave
            // a "basis vector" inside span(V) and Ritz values far outside the spectrum).  As ARPACK does (dsaitr -> dgetv0): beta_j = 0 and
            // the basis continues with a fresh random vector orthogonalised against V; the Ritz pairs of the closed block are exact.
            if (!(b2 > 1e-20 * wnorm2) || !(b2 > 0.0)) broke = true;
            double beta = broke ? 0.0 : std::sqrt(b2);
            if (broke) {
                if (!d_fresh && qbh::dev_alloc(&d_fresh, (size_t)n * sizeof(d2)) != hipSuccess) {
                    (void)hipGetLastError();
                    qbh::set_error("qbh_iram: no room for the fresh vector of a breakdown");
                    rc = QBH_ENOMEM;               // (left through the common exit: the basis is released there)
                    break;
                }
                rc = qbh_vec_randomize(A, reinterpret_cast<qbh_z *>(d_fresh), (seed ? seed : 1u) + 7919u * (uint32_t)(++n_fresh));
                if (rc != QBH_OK) break;
                if (all_real) rc = qbh::launch_pack_real(d_fresh, rvec(j + 1), n, A->d_flag, A->stream);
                else if (hipMemcpyAsync(w, d_fresh, (size_t)n * sizeof(d2), hipMemcpyDeviceToDevice, A->stream) != hipSuccess) rc = QBH_EHIP;
                double dummy = 0.0, f2 = 0.0;
                if (rc == QBH_OK) rc = cgs_pass(w, j + 1, &dummy, &f2);
                if (rc == QBH_OK) rc = cgs_pass(w, j + 1, &dummy, &f2);
                if (rc != QBH_OK) break;
                if (f2 > 1e-24) rc = qbh::launch_scal(1.0 / std::sqrt(f2), w, nc, A->stream);
                else if (hipMemsetAsync(w, 0, (size_t)ldc * sizeof(d2), A->stream) != hipSuccess) rc = QBH_EHIP;      // the basis spans the whole space: nothing left
                if (rc != QBH_OK) break;
            }
            T[(size_t)j * m + j] = alpha;
            beta_last = beta;
            if (j + 1 < m) T[(size_t)j * m + (j + 1)] = T[(size_t)(j + 1) * m + j] = beta;
            if (beta > 0.0) rc = qbh::launch_scal(1.0 / beta, w, nc, A->stream);
        }
        if (rc != QBH_OK) break;
        Tw = T;
        qbh::symmetric_eigen_jacobi(m, Tw.data(), theta.data(), S.data());
        nconv = 0;
        for (int i = 0; i < (int)nev; ++i) {
            const double resid = std::fabs(beta_last * S[(size_t)i * m + (m - 1)]);
            if (resid <= tol_eff * std::max(eps23, std::fabs(theta[i]))) nconv++;
            else break;
        }
        restarts++;
        const bool done = nconv >= nev || restarts >= maxit;
        const int keep = done ? (int)nev : (int)std::min<int64_t>(m - 1, nev + std::max<int64_t>(1, (m - nev) / 2));
        // V[:, 0..keep) <- V[:, 0..m) S[:, 0..keep)
        e = hipMemcpyAsync(d_S, S.data(), (size_t)m * keep * sizeof(double), hipMemcpyHostToDevice, A->stream);
        if (e != hipSuccess) { rc = QBH_EHIP; break; }
        e = hipStreamSynchronize(A->stream);                 // S.data() is pageable host memory
        if (e != hipSuccess) { rc = QBH_EHIP; break; }
        rc = qbh::launch_basis_rotate(V, ldc, nc, m, keep, d_S, A->stream);
        if (rc != QBH_OK || done) break;
        e = hipMemcpyAsync(vec(keep), vec(m), (size_t)ldc * sizeof(d2), hipMemcpyDeviceToDevice, A->stream);
        if (e != hipSuccess) { rc = QBH_EHIP; break; }
        std::fill(T.begin(), T.end(), 0.0);
        for (int i = 0; i < keep; ++i) {
            T[(size_t)i * m + i] = theta[i];
            const double s_i = beta_last * S[(size_t)i * m + (m - 1)];
            T[(size_t)keep * m + i] = T[(size_t)i * m + keep] = s_i;
        }
        k = keep;
    }
    if (rc == QBH_OK) rc = finish_real_wire(A);
    if (rc == QBH_OK) {
        for (int i = 0; i < (int)nev; ++i) eigenvals[i] = sign * theta[i];
        *nconv_out = nconv;
        if (eigenvecs_host && all_real) {
            d2 *tmp = nullptr;
            if (qbh::dev_alloc(&tmp, (size_t)n * sizeof(d2)) != hipSuccess) rc = QBH_ENOMEM;
            for (int i = 0; rc == QBH_OK && i < (int)nev; ++i) {
                rc = qbh::launch_unpack_real(rvec(i), tmp, n, A->stream);
                if (rc == QBH_OK) rc = vec_d2h(A, eigenvecs_host + (size_t)i * (size_t)n, tmp, n);
            }
            if (tmp) (void)hipFree(tmp);
        } else if (eigenvecs_host) {
            for (int i = 0; rc == QBH_OK && i < (int)nev; ++i) rc = vec_d2h(A, eigenvecs_host + (size_t)i * (size_t)n, vec(i), n);
        }
        e = hipStreamSynchronize(A->stream);
        if (e != hipSuccess) rc = QBH_EHIP;
    }
    harvest_events(A);
    (void)hipFree(V);
    (void)hipFree(d_S);
    if (d_fresh) (void)hipFree(d_fresh);
    if (info) {
        info->n_matvec = A->stats.n_spmv - spmv0;
        info->ms_spmv = A->stats.ms_spmv - ms_spmv0;
        info->ms_total = now_ms() - t_start;
        info->n_reorth = restarts;                           // number of restarts (ARPACK's niter)
    }
    return rc;
}
