// qbh_spmv.cpp -- SpMV dispatch (spmv_run / spmv_kron), reductions, the real wire format, the BLAS-1 runs, the device
// building blocks (qbh_spmv_dev ...) and the host-vector seam qbh_multmv / qbh_multmv2 (src/sparse.cc:262-297).
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

// ------------------------------------------------- reductions / scalars --------
namespace qbhapi {


// partials[nparts*ncomp] -> host_out[ncomp], summed over ranks under a communicator.
int finish_reduction(qbh_csr *A, int nparts, int ncomp, double *host_out)
{
    double *ds = scal_buf(A);
    QBH_TRY(qbh::launch_reduce_partials(A->d_partials, nparts, ncomp, ds, A->stream));
    if (A->has_comm) {
        if (A->comm.allreduce_sum(A->comm.ctx, 0, ncomp) != 0) {
            qbh::set_error("allreduce_sum hook failed");
            return QBH_ECOMM;
        }
    }
    QBH_HIP(hipMemcpyAsync(A->h_scal, ds, (size_t)ncomp * sizeof(double), hipMemcpyDeviceToHost, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    for (int c = 0; c < ncomp; ++c) host_out[c] = A->h_scal[c];
    return QBH_OK;
}

static void harvest_set(qbh_csr *A, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3, bool p, bool p2, bool drop)
{
    float ms = 0.f, total = 0.f;
    bool any = false;
    if (p && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
        total += ms;
        any = true;
    }
    if (p2 && hipEventSynchronize(e3) == hipSuccess && hipEventElapsedTime(&ms, e2, e3) == hipSuccess) {
        total += ms;
        any = true;
    }
    if (any && !drop) {
        A->stats.ms_spmv += total;
        if (total < A->stats.ms_spmv_min) A->stats.ms_spmv_min = total;
    }
}

// everything that is pending: the sets queued behind the current one (oldest first), then the current one
// defer_red: the three fused sums stay on the device -- in the scalar buffer the communicator's all-reduce works on, summed
// over the ranks in stream order (no copy, no synchronisation)
int deferred_reduction(qbh_csr *A, int nparts)
{
    double *ds = scal_buf(A);
    QBH_TRY(qbh::launch_reduce_partials(A->d_partials, nparts, 3, ds, A->stream));
    if (A->has_comm && A->comm.allreduce_sum(A->comm.ctx, 0, 3) != 0) {
        qbh::set_error("allreduce_sum hook failed");
        return QBH_ECOMM;
    }
    return QBH_OK;
}

void harvest_events(qbh_csr *A)
{
    for (int i = 0; i < A->n_ev_old; ++i) {
        qbh_csr::EvSet &o = A->ev_old[i];
        harvest_set(A, o.e[0], o.e[1], o.e[2], o.e[3], o.p, o.p2, o.drop);
        o.p = o.p2 = o.drop = false;
    }
    // the queued sets keep their events (reused by next_event_set); they are simply no longer pending
    harvest_set(A, A->ev0, A->ev1, A->ev2, A->ev3, A->ev_pending, A->ev_pending2, A->ev_drop);
    A->ev_pending = A->ev_pending2 = A->ev_drop = false;
}

// In front of an SpMV's first timing event.  Ordinarily: wait for the previous SpMV's events (as ever).  Under a pipelined driver
// (ev_keep) the previous SpMV may not even have started: its set is queued and a free one becomes current; only a set three
// SpMVs old is ever waited for.
int next_event_set(qbh_csr *A)
{
    if (!A->ev_keep) {
        harvest_events(A);
        return QBH_OK;
    }
    if (!A->ev_pending && !A->ev_pending2) return QBH_OK;            // the current set is free
    constexpr int NQ = (int)(sizeof(A->ev_old) / sizeof(A->ev_old[0]));
    // a slot whose set is no longer pending (or was never created) takes the current events; its own become current
    int slot = -1;
    for (int i = 0; i < NQ && slot < 0; ++i)
        if (!A->ev_old[i].p && !A->ev_old[i].p2) slot = i;
    if (slot < 0) {                                                   // all in flight: the oldest is three SpMVs old
        qbh_csr::EvSet &o = A->ev_old[0];
        harvest_set(A, o.e[0], o.e[1], o.e[2], o.e[3], o.p, o.p2, o.drop);
        o.p = o.p2 = o.drop = false;
        qbh_csr::EvSet first = A->ev_old[0];
        for (int i = 0; i + 1 < NQ; ++i) A->ev_old[i] = A->ev_old[i + 1];
        A->ev_old[NQ - 1] = first;
        slot = NQ - 1;
    }
    qbh_csr::EvSet &q = A->ev_old[slot];
    for (int k = 0; k < 4; ++k)
        if (!q.e[k]) QBH_HIP(hipEventCreate(&q.e[k]));
    std::swap(q.e[0], A->ev0);
    std::swap(q.e[1], A->ev1);
    std::swap(q.e[2], A->ev2);
    std::swap(q.e[3], A->ev3);
    q.p = A->ev_pending;
    q.p2 = A->ev_pending2;
    q.drop = A->ev_drop;
    A->ev_pending = A->ev_pending2 = A->ev_drop = false;
    if (slot + 1 > A->n_ev_old) A->n_ev_old = slot + 1;
    return QBH_OK;
}



// The SpMV of an operator split in place (kron_build).  One GPU (or a shard driven with the full-length x): tiled copy of x
// unless the pass that produced x wrote it, far pass (row sums in tiled order), near pass (+ far result, fused epilogue and
// reductions).  Under a communicator every rank sends the TILED copy of its own block; the near pass needs only the rank's own
// x and runs while the all-gather is in flight (epilogue without the far addend), then the far pass reads the gathered blocks
// and a light third pass adds its result and produces the reductions on the finished y.
int spmv_kron(qbh_csr *A, const d2 *x, d2 *y, double alpha, double beta, double gamma, double *red)
{
    qbh_csr::KronSplit &K = A->kron;
    if (!kron_path(A)) {
        qbh::set_error("this operator is stored split in place (qbh_opts.kron_split): the requested form of the SpMV (%s) needs its CSR",
                       A->has_comm ? "a communicator whose ranks do not all exchange tiled blocks" : (A->debug & 1) ? "QBH_DEBUG column mask"
                                                                                                             : "packed-real vectors");
        return QBH_EUNSUPP;
    }
    hipStream_t s = A->stream;
    const bool prof = A->opts.profile != 0;
    const bool comm = A->has_comm;
    // the dynamic ordered walk per XCD -- also for a caller who wants bit-reproducible results: y does not depend on which wavefront
    // computes a row, and the fused reductions are collected per CHUNK of blocks and added in a fixed order (k_spmv_wave2's
    // CHUNKRED, k_reduce_chunks) since round 5; qbh_opts.wave_walk = 2 still names the static walk
    int kron_swz = 3;
    if (A->opts.wave_walk >= 0) kron_swz = A->opts.wave_walk;
    if (kron_swz == 3) {
        if (!A->d_wctr) kron_swz = 2;
        else QBH_HIP(hipMemsetAsync(A->d_wctr, 0, (size_t)(comm && (K.n_parts > 1 || K.sparse) ? qbh::kWctrRegions : 3) * 128 * sizeof(unsigned long long), s));
#ifdef QBH_XCD_TIMING
        if (A->d_wctr) {                                     // slot 2 of every XCD collects a minimum
            unsigned long long h[3 * 128] = {0};
            for (int p = 0; p < 3; ++p)
                for (int k = 0; k < 8; ++k) h[p * 128 + k * 16 + 2] = ~0ull;
            QBH_HIP(hipMemcpyAsync(A->d_wctr, h, sizeof(h), hipMemcpyHostToDevice, s));
            QBH_HIP(hipStreamSynchronize(s));
        }
#endif
    }
    const d2 *xl = comm ? x : x + A->row_offset;            // the rank's own block of x
    const d2 *xt = nullptr;                                  // what the far pass gathers from: the tiled copy of the WHOLE x
    bool async_gather = false;
    if (K.xt_cap < A->ncols) {                               // first use: the tiled copy of the full-length x
        if (K.d_xt) (void)hipFree(K.d_xt);
        K.d_xt = nullptr;
        K.xt_cap = 0;
        QBH_HIP(qbh::dev_alloc(&K.d_xt, (size_t)A->ncols * sizeof(d2)));
        K.xt_cap = A->ncols;
        if (A->dbg.print_ptrs) fprintf(stderr, "qbhip kron xt %p x %p y %p\n", (void *)K.d_xt, (const void *)x, (void *)y);
        K.xt_of = nullptr;
    }
    // under a communicator: the wire format of this solve (8-byte real parts when the drivers agreed on it), and whether the
    // gather travels in parts (with real parts only through the hook that names the format)
    const int realw = tiled_real(A);
    const bool sparse = comm && K.sparse && A->comm.exchange_v != nullptr;      // personalised exchange: only what each peer reads travels
    const int np_used = sparse ? std::max(1, K.n_parts) : (comm && K.n_parts > 1 && (!realw || A->comm.allgather_part_begin_w)) ? K.n_parts : 1;
    const int64_t nfb_all = K.t.S / K.t.B, w_edge = K.t.S - nfb_all * K.t.B;
    // part k of the exchange = bands [pb0, pb1) of every piece, the last part takes the narrow edge band along
    auto part_bands = [&](int k, int64_t &pb0, int64_t &pb1, bool &with_edge) {
        pb0 = np_used > 1 ? K.part_band[k] : 0;
        pb1 = np_used > 1 ? K.part_band[k + 1] : nfb_all;
        with_edge = k == np_used - 1 && w_edge > 0;
    };
    if (comm) {
        d2 *send = reinterpret_cast<d2 *>(A->comm.d_xsend);
        if (K.xt_of != (const void *)x) QBH_TRY(qbh::launch_kron_tile(x, send, A->nrows, K.t, s, realw, A->d_flag));
        K.xt_of = nullptr;
        async_gather = A->comm.allgather_begin && A->comm.allgather_wait;
        int hrc = 0;
        if (sparse) {
            const int npk = A->comm.nranks, me = A->comm.rank;
            // (1) the rank's own needed major indices straight from its tiled block into the tiled x
            {
                qbh::KronPlace po{};
                po.real = realw;
                po.src = send;
                po.dst = K.d_xt;
                po.nr = npk;
                po.B = K.t.B;
                po.S = K.t.S;
                po.NUg = K.NUg;
                po.nfb = nfb_all;
                for (int q = 0; q <= npk; ++q) po.cu[q] = K.rank_cu[q];
                po.list = K.d_need;
                for (int q = 0; q <= npk; ++q) po.lo[q] = q <= me ? K.need_lo[me] : K.need_lo[me + 1];     // only the own rank has entries
                po.band0 = 0;
                po.band1 = nfb_all + (w_edge > 0 ? 1 : 0);
                QBH_TRY(qbh::launch_kron_place(po, s));
            }
            // (2) per destination: the major indices it reads, packed band-major
            qbh::KronPack pk{};
            pk.src = send;
            pk.dst = K.d_vsend;
            pk.real = realw;
            pk.nr = npk;
            pk.B = K.t.B;
            pk.S = K.t.S;
            pk.NUq = K.t.NU;
            pk.nfb = nfb_all;
            pk.list = K.d_send_list;
            for (int q = 0; q <= npk; ++q) pk.lo[q] = K.send_lo[q];
            for (int q = 0; q < npk; ++q) pk.base[q] = K.send_lo[q] * K.t.S;
            QBH_TRY(qbh::launch_kron_pack(pk, s));
            // (3) the pieces, band range after band range
            std::vector<int64_t> so((size_t)2 * npk), ro((size_t)2 * npk);
            int64_t rbase = 0;
            std::vector<int64_t> rb((size_t)npk, 0);
            for (int q = 0; q < npk; ++q) {
                rb[(size_t)q] = rbase;
                if (q != me) rbase += (K.need_lo[q + 1] - K.need_lo[q]) * K.t.S;
            }
            for (int k = 0; k < np_used && hrc == 0; ++k) {
                int64_t pb0, pb1;
                bool with_edge;
                part_bands(k, pb0, pb1, with_edge);
                for (int q = 0; q < npk; ++q) {
                    const int64_t ns = q == me ? 0 : K.send_lo[q + 1] - K.send_lo[q], nr_ = q == me ? 0 : K.need_lo[q + 1] - K.need_lo[q];
                    so[(size_t)2 * q] = K.send_lo[q] * K.t.S + pb0 * K.t.B * ns;
                    so[(size_t)2 * q + 1] = (pb1 - pb0) * K.t.B * ns + (with_edge ? ns * w_edge : 0);
                    ro[(size_t)2 * q] = rb[(size_t)q] + pb0 * K.t.B * nr_;
                    ro[(size_t)2 * q + 1] = (pb1 - pb0) * K.t.B * nr_ + (with_edge ? nr_ * w_edge : 0);
                }
                hrc = A->comm.exchange_v(A->comm.ctx, k, np_used, so.data(), ro.data(), realw ? 1 : 2, K.d_vsend, K.d_vrecv);
            }
        } else if (np_used > 1) {                            // band ranges one after another: the far pass follows them (below)
            for (int k = 0; k < np_used && hrc == 0; ++k) {
                const int64_t *ol = K.part_off_len.data() + (size_t)k * 2 * (size_t)A->comm.nranks;
                hrc = realw ? A->comm.allgather_part_begin_w(A->comm.ctx, k, np_used, ol, 1) : A->comm.allgather_part_begin(A->comm.ctx, k, np_used, ol);
            }
        } else {
            hrc = async_gather ? A->comm.allgather_begin(A->comm.ctx, realw) : A->comm.allgather_x(A->comm.ctx, realw);
        }
        if (hrc != 0) {
            qbh::set_error("allgather hook failed");
            return QBH_ECOMM;
        }
        A->stats.n_gather++;
        A->wire_bytes_last = realw ? 8 : 16;
        xt = K.d_xt;
    } else {
        if (prof) {
            QBH_TRY(next_event_set(A));
            QBH_HIP(hipEventRecord(A->ev0, s));
        }
        if (K.xt_of != (const void *)x) {
            if (K.map.nc == 1) {
                QBH_TRY(qbh::launch_kron_tile(x, K.d_xt, A->ncols, qbh::KronTile{K.t.S, K.NUg, K.t.B}, s));
            } else {                         // every class is a product basis of its own: tiled class by class
                for (int c = 0; c < K.map.nc; ++c)
                    QBH_TRY(qbh::launch_kron_tile(x + K.map.rbase[c], K.d_xt + K.map.rbase[c], K.map.rbase[c + 1] - K.map.rbase[c],
                                                  qbh::KronTile{K.map.S[c], K.map.NU[c], K.map.B}, s));
            }
        }
        K.xt_of = nullptr;
        xt = K.d_xt;
    }
    // the pieces of the gathered blocks -> their place in the tiled x (part k of np_used, or everything)
    auto place = [&](int k) -> int {
        qbh::KronPlace pa{};
        pa.real = realw;
        if (sparse) {                                        // the peers' packed pieces: only the listed major indices, compact
            const int npk = A->comm.nranks, me = A->comm.rank;
            pa.src = K.d_vrecv;
            pa.dst = K.d_xt;
            pa.nr = npk;
            pa.B = K.t.B;
            pa.S = K.t.S;
            pa.NUg = K.NUg;
            pa.nfb = nfb_all;
            pa.compact = 1;
            pa.list = K.d_need;
            for (int q = 0; q <= npk; ++q) pa.cu[q] = K.rank_cu[q];
            // the list keeps the own rank's entries in the middle: the peers' ranges are named one by one, the own range is empty
            int64_t rbase = 0;
            for (int q = 0; q < npk; ++q) {
                pa.base[q] = rbase;
                if (q != me) rbase += (K.need_lo[q + 1] - K.need_lo[q]) * K.t.S;
            }
            for (int q = 0; q <= npk; ++q) pa.lo[q] = K.need_lo[q];
            pa.skip = me;
            int64_t pb0, pb1;
            bool with_edge;
            part_bands(k, pb0, pb1, with_edge);
            pa.band0 = pb0;
            pa.band1 = pb1 + (with_edge ? 1 : 0);
            return qbh::launch_kron_place(pa, s);
        }
        pa.src = realw ? reinterpret_cast<const d2 *>(A->comm.d_xfull_r) : reinterpret_cast<const d2 *>(A->comm.d_xfull);
        pa.dst = K.d_xt;
        pa.nr = A->comm.nranks;
        pa.B = K.t.B;
        pa.S = K.t.S;
        pa.NUg = K.NUg;
        pa.nfb = K.t.S / K.t.B;
        for (int q = 0; q <= pa.nr; ++q) pa.cu[q] = K.rank_cu[q];
        pa.list = K.d_need;                                  // only the major indices this shard reads
        for (int q = 0; q <= pa.nr; ++q) pa.lo[q] = K.need_lo[q];
        const int64_t edge = (K.t.S % K.t.B) ? 1 : 0;         // the narrow last band travels with the last piece
        pa.band0 = np_used > 1 ? K.part_band[k] : 0;
        pa.band1 = np_used > 1 ? (k == np_used - 1 ? pa.nfb + edge : K.part_band[k + 1]) : pa.nfb + edge;
        for (int q = 0; q < pa.nr; ++q) {
            pa.base[q] = A->comm.row_cuts ? A->comm.row_cuts[q] : (int64_t)q * A->comm.nblk;
            if (np_used > 1) {
                pa.off[q] = K.part_off_len[((size_t)k * (size_t)pa.nr + (size_t)q) * 2];
                pa.len[q] = K.part_off_len[((size_t)k * (size_t)pa.nr + (size_t)q) * 2 + 1];
            } else {
                pa.off[q] = 0;
                pa.len[q] = (K.rank_cu[q + 1] - K.rank_cu[q]) * K.t.S;
            }
        }
        return qbh::launch_kron_place(pa, s);
    };
    qbh::SpmvArgs f{};                                       // far pass
    f.ia = K.ia_f;
    f.ja = K.ja_f;
    f.ja16 = K.c16_f;
    f.kS = K.t.S;
    f.kNU = K.NUg;
    f.kB = K.t.B;
    f.val = K.val_f;
    f.wd = K.wd_f;
    f.n_wb = K.nwb_f;
    f.nrows = K.map.nfar_rows();                             // sliced: whole groups of the full bands
    f.xg = xt;
    f.xl = xl;
    f.y = K.d_far;
    f.alpha = 1.0;
    f.colmask = -1;
    f.chunk_mult = A->chunk_mult;
    f.swizzle = kron_swz == 3 ? 3 : 1;      // one contiguous eighth of the bands per XCD: a band of x stays in that XCD's L2
    f.wctr = A->d_wctr ? A->d_wctr + 128 : nullptr;
    qbh::SpmvArgs nr{};                                      // near pass
    nr.ia = K.ia_n;
    nr.ja = K.ja_n;
    nr.ja16 = K.c16_n;
    nr.val = K.val_n;
    nr.wd = K.wd_n;
    nr.n_wb = K.nwb_n;
    nr.nrows = A->nrows;
    nr.xg = K.c16_n ? xl : xl - A->row_offset;               // near columns are global indices of locally-owned elements (2-byte: relative to the shard)
    nr.xl = xl;
    nr.y = y;
    nr.alpha = alpha;
    nr.beta = beta;
    nr.gamma = gamma;
    nr.colmask = -1;
    nr.chunk_mult = A->chunk_mult;
    nr.kS = K.t.S;
    nr.kNU = K.t.NU;
    nr.kB = K.t.B;
    nr.swizzle = kron_swz;
    nr.wctr = A->d_wctr ? A->d_wctr + 256 : nullptr;
    nr.chunk_red = K.d_chunk_red;                            // dynamic walk: the near pass's reduction partials, one slot per chunk
    nr.yin = A->ovr_yin;                                     // pipelined three-term step (lanczos_core): out of place, coefficients on the device
    nr.coef = A->ovr_coef;
    nr.coef_mode = A->ovr_coef ? 1 : 0;
    const bool chunk_red = kron_swz == 3 && K.d_chunk_red != nullptr;
    if (kron_swz == 3 && !K.d_chunk_red) {
        qbh::set_error("Kronecker split: the chunk partials of the near pass were not allocated");
        return QBH_EHIP;
    }
    int nparts = K.grid_n;
    if (!comm) {
        if (K.sliced) QBH_TRY(qbh::launch_zero_cut_groups(K.wd_f, K.nwb_f, f.nrows, K.d_far, s));
        QBH_TRY(qbh::launch_spmv_wave2(f, K.tpr_f, K.sliced ? 3 : 0, K.grid_f, s));
        nr.far = K.d_far;
        if (K.map.nc == 1) {
            // the far entries of the rows that do not fill a band: their sums go into those rows' slots of the far buffer
            QBH_TRY(qbh::launch_kron_cross_rows(K.ia_x, K.xrow, K.n_xrows, K.ja_x, K.val_x, xt, K.t, K.d_far, s));
            nr.partials = red ? A->d_partials : nullptr;
            QBH_TRY(qbh::launch_spmv_wave2(nr, K.tpr_n, 2, K.grid_n, s));
            if (red && chunk_red) QBH_TRY(qbh::launch_reduce_chunks(K.d_chunk_red, K.n_chunk_slots, A->d_partials, &nparts, s));
        } else {
            nr.kcls = K.d_cls;
            nr.partials = (red && K.nnz_x == 0) ? A->d_partials : nullptr;
            QBH_TRY(qbh::launch_spmv_wave2(nr, K.tpr_n, 4, K.grid_n, s));
            if (red && K.nnz_x == 0 && chunk_red) QBH_TRY(qbh::launch_reduce_chunks(K.d_chunk_red, K.n_chunk_slots, A->d_partials, &nparts, s));
            if (K.nnz_x > 0) {               // third pass: the unstructured part, accumulating onto y; reductions on the finished y
                qbh::SpmvArgs cr{};
                cr.ia = K.ia_x;
                cr.ja = K.ja_x;
                cr.val = K.val_x;
                cr.wd = K.wd_x;
                cr.n_wb = K.nwb_x;
                cr.nrows = A->nrows;
                cr.xg = xt;                  // columns are stored in the tiled order of x
                cr.xl = xl;
                cr.y = y;
                cr.alpha = alpha;
                cr.beta = 1.0;
                cr.gamma = 0.0;
                cr.coef = A->ovr_coef;
                cr.coef_mode = A->ovr_coef ? 2 : 0;
                cr.colmask = -1;
                cr.chunk_mult = A->chunk_mult;
                cr.swizzle = (A->opts.xcd_swizzle == 3 && !A->opts.deterministic && A->d_wctr) ? 3 : (A->opts.xcd_swizzle == 3 ? 2 : A->opts.xcd_swizzle);
                cr.wctr = A->d_wctr;
                cr.partials = red ? A->d_partials : nullptr;
                QBH_TRY(qbh::launch_spmv_wave(cr, K.tpr_x, K.grid_x, s));
                nparts = K.grid_x;
            }
        }
#ifdef QBH_XCD_TIMING
        {   // debug build: last and first wavefront of every XCD to run out of blocks, relative to the earliest of the pass (100 MHz ticks -> us)
            unsigned long long h[3 * 128];
            QBH_HIP(hipStreamSynchronize(s));
            QBH_HIP(hipMemcpy(h, A->d_wctr, sizeof(h), hipMemcpyDeviceToHost));
            for (int pass = 1; pass <= 2; ++pass) {
                const unsigned long long *d = h + pass * 128;
                unsigned long long lo = ~0ull, hi = 0;
                for (int k = 0; k < 8; ++k) {
                    lo = std::min(lo, d[k * 16 + 2]);
                    hi = std::max(hi, d[k * 16 + 1]);
                }
                fprintf(stderr, "[xcd timing] %s pass: last wavefront of each XCD done at (us before the pass ends):", pass == 1 ? "far" : "near");
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %.0f", (double)(hi - d[k * 16 + 1]) * 0.01);
                fprintf(stderr, " | first wavefront anywhere idle %.0f us before the end\n", (double)(hi - lo) * 0.01);
            }
        }
#endif
#ifdef QBH_WAVE_TIMING
        {   // debug build: where the wavefronts of the two passes spend their cycles (s_memtime ticks, 100 MHz)
            unsigned long long h[3 * 128];
            QBH_HIP(hipStreamSynchronize(s));
            QBH_HIP(hipMemcpy(h, A->d_wctr, sizeof(h), hipMemcpyDeviceToHost));
            for (int pass = 1; pass <= 2; ++pass) {
                const unsigned long long *d = h + (pass + 1) * 128 - 8;
                const double nb = d[4] ? (double)d[4] : 1.0;
                fprintf(stderr, "[wave timing] %s pass: blocks %llu, ticks per block: issue+column wait %.1f, gather wait %.1f, reduce %.1f, stream rest %.1f\n",
                        pass == 1 ? "far" : "near", d[4], d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb);
            }
        }
#endif
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev1, s));
            A->ev_pending = true;
        }
    } else {
        if (prof) {
            QBH_TRY(next_event_set(A));
            QBH_HIP(hipEventRecord(A->ev0, s));
        }
        nr.far = nullptr;
        nr.partials = nullptr;
        // qbh_opts.comm_reserve: room for RCCL's own kernels beside the persistent passes -- workgroups left out of the grids (multiples
        // of 8: one per XCD) and at most two far workgroups per CU (QBH_DEBUG=comm_reserve=W / comm_far_cap=C override both for A/B runs)
        int reserve = 0, far_cap = K.grid_f;
        if (A->comm.nranks > 1 && A->opts.comm_reserve >= 0) {
            reserve = A->opts.comm_reserve > 0 ? (A->opts.comm_reserve / 8) * 8 : 64;
            if (A->ncu <= 0) {
                hipDeviceProp_t prop;
                A->ncu = (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            }
            far_cap = std::min(K.grid_f, 2 * A->ncu);
        }
        if (A->dbg.comm_reserve != 0) reserve = A->dbg.comm_reserve > 0 ? (A->dbg.comm_reserve / 8) * 8 : 0;
        if (A->dbg.comm_far_cap != 0) far_cap = A->dbg.comm_far_cap > 0 ? std::min(K.grid_f, A->dbg.comm_far_cap * std::max(A->ncu, 256)) : K.grid_f;
        // (a small shard keeps at least half of either grid; grids stay multiples of 8 -- the static walks count slots per XCD)
        auto mult8 = [](int g) { return std::max(8, (g / 8) * 8); };
        const int grid_n = mult8(std::max(K.grid_n / 2, K.grid_n - reserve)), grid_f = mult8(std::max(std::min(far_cap, K.grid_f) / 2, far_cap - reserve));
        QBH_TRY(qbh::launch_spmv_wave2(nr, K.tpr_n, 1, grid_n, s));          // y = alpha H_near x + beta y + gamma x
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev1, s));
            A->ev_pending = true;
        }
        if (np_used > 1 || sparse) {
            // every band range of the far part as soon as its piece of the gathered x is there and has been moved to its place; a
            // block that straddles a range boundary belongs to the later range (the pieces complete in order), its cut groups add
            // up through the atomics
            QBH_TRY(qbh::launch_zero_cut_groups(K.wd_f, K.nwb_f, f.nrows, K.d_far, s));
            for (int k = 0; k < np_used; ++k) {
                if (A->comm.allgather_part_wait(A->comm.ctx, k) != 0) {
                    qbh::set_error("allgather_part_wait hook failed");
                    return QBH_ECOMM;
                }
                if (prof && k == 0) QBH_HIP(hipEventRecord(A->ev2, s));
                QBH_TRY(place(k));
                qbh::SpmvArgs fk = f;
                fk.wd = K.wd_f + (np_used > 1 ? K.part_blk[k] : 0);
                fk.n_wb = np_used > 1 ? K.part_blk[k + 1] - K.part_blk[k] : K.nwb_f;
                fk.wctr = A->d_wctr ? A->d_wctr + (3 + k) * 128 : nullptr;
                if (fk.n_wb > 0) QBH_TRY(qbh::launch_spmv_wave2(fk, K.tpr_f, 3, (int)std::min<int64_t>(grid_f, std::max<int64_t>(fk.n_wb, 8)), s));
            }
        } else {
            if (async_gather && A->comm.allgather_wait(A->comm.ctx) != 0) {
                qbh::set_error("allgather_wait hook failed");
                return QBH_ECOMM;
            }
            if (prof) QBH_HIP(hipEventRecord(A->ev2, s));
            QBH_TRY(place(0));
            if (K.sliced) QBH_TRY(qbh::launch_zero_cut_groups(K.wd_f, K.nwb_f, f.nrows, K.d_far, s));
            QBH_TRY(qbh::launch_spmv_wave2(f, K.tpr_f, K.sliced ? 3 : 0, grid_f, s));
        }
        QBH_TRY(qbh::launch_kron_cross_rows(K.ia_x, K.xrow, K.n_xrows, K.ja_x, K.val_x, xt, K.t, K.d_far, s));
        QBH_TRY(qbh::launch_kron_combine(K.d_far, K.t, xl, y, A->nrows, alpha, red ? A->d_partials : nullptr, &nparts, s, A->ovr_coef));
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev3, s));
            A->ev_pending2 = true;
        }
    }
    A->xr_of = nullptr;
    A->stats.n_spmv++;
    if (red && A->defer_red) {
        QBH_TRY(deferred_reduction(A, nparts));
    } else if (red) {
        QBH_TRY(finish_reduction(A, nparts, 3, red));
        if (prof) harvest_events(A);
    }
    return QBH_OK;
}

int spmv_run(qbh_csr *A, const d2 *x, d2 *y, double alpha, double beta, double gamma, double *red)
{
    if (A->broken) {
        qbh::set_error("the operator was left inconsistent by an earlier failed call (qbh_csr_set_comm / creation): destroy it");
        return QBH_EINVAL;
    }
    if (A->kron.active) return spmv_kron(A, x, y, alpha, beta, gamma, red);
    if ((A->ovr_yin || A->ovr_coef) && (A->kind != 0 || A->ovr_yr != nullptr)) {
        qbh::set_error("internal: the pipelined three-term step needs a stored operator and complex vectors");
        return QBH_EUNSUPP;
    }
    const d2 *xg, *xl;
    bool async_gather = false;
    const bool packed = A->has_comm && A->real_wire;
    const bool realm = A->real_mode && A->kernel == QBH_KERNEL_ROWS && (packed || !A->has_comm);
    auto expand_packed = [&]() -> int {          // d_xfull_r (doubles) -> d_xfull (complex, zero imaginary part)
        return qbh::launch_unpack_real(A->comm.d_xfull_r, reinterpret_cast<d2 *>(A->comm.d_xfull), A->comm_full, A->stream);
    };
    if (A->has_comm) {
        if (packed) {
            if (!(realm && A->xr_of == x))
                QBH_TRY(qbh::launch_pack_real(x, reinterpret_cast<double *>(A->comm.d_xsend), A->nrows, A->d_flag, A->stream));
        } else
            QBH_HIP(hipMemcpyAsync(A->comm.d_xsend, x, (size_t)A->nrows * sizeof(d2), hipMemcpyDeviceToDevice,
                                   A->stream));
        async_gather = A->comm.allgather_begin && A->comm.allgather_wait;
        const int hrc = async_gather ? A->comm.allgather_begin(A->comm.ctx, packed ? 1 : 0)
                                     : A->comm.allgather_x(A->comm.ctx, packed ? 1 : 0);
        if (hrc != 0) {
            qbh::set_error("allgather hook failed");
            return QBH_ECOMM;
        }
        if (!async_gather && packed && !realm) QBH_TRY(expand_packed());
        A->stats.n_gather++;
        A->wire_bytes_last = packed ? 8 : 16;
        xg = reinterpret_cast<const d2 *>(A->comm.d_xfull);
        xl = x;
    } else if (A->ovr_yr != nullptr) {           // all-real operation on packed vectors (driver-internal)
        if (!realm || A->nrows != A->ncols) {
            qbh::set_error("all-real SpMV needs the real fast path on an unsharded operator");
            return QBH_EINVAL;
        }
        xg = xl = nullptr;
    } else {
        xg = x;
        xl = x + A->row_offset;
        if (realm && A->xr_of != x) QBH_TRY(qbh::launch_pack_real(x, A->d_xr, A->ncols, A->d_flag, A->stream));
    }
    const double *xr_nocomm = A->ovr_yr != nullptr ? A->ovr_xr : A->d_xr;
    if (A->kind != 0) {                          // matrix-free operator: one launch, needs the whole gathered x
        if (async_gather) {
            if (A->comm.allgather_wait(A->comm.ctx) != 0) {
                qbh::set_error("allgather_wait hook failed");
                return QBH_ECOMM;
            }
            if (packed && !realm) QBH_TRY(expand_packed());
        }
        qbh::MfArgs m{};
        m.t = A->mf;
        m.row_begin = A->row_offset;
        m.nrows = A->nrows;
        m.xg = xg;
        m.xl = xl;
        m.xr = realm ? (A->has_comm ? A->comm.d_xfull_r : xr_nocomm) : nullptr;
        m.y = y;
        m.y_re = A->has_comm ? nullptr : A->ovr_yr;
        m.alpha = alpha;
        m.beta = beta;
        m.gamma = gamma;
        m.partials = red ? A->d_partials : nullptr;
        const bool profm = A->opts.profile != 0;
        if (profm) {
            QBH_TRY(next_event_set(A));
            QBH_HIP(hipEventRecord(A->ev0, A->stream));
        }
        int mf_parts = A->grid;
        if (A->kind == 3) {
            qbh::MfSecArgs ms{};
            ms.t = A->d_mfsec;
            ms.n_items = A->mfsec->n_items;
            ms.dim = A->nrows;
            ms.n_rrows = A->mfsec->n_rrows;
            ms.rrow = A->mfsec->rrow;
            ms.ria = A->mfsec->ria;
            ms.rja = A->mfsec->rja;
            ms.rval = A->mfsec->rval;
            ms.xg = m.xg;
            ms.xl = m.xl;
            ms.xr = m.xr;
            ms.xl_re = m.y_re ? xr_nocomm : nullptr;
            ms.y = m.y;
            ms.y_re = m.y_re;
            ms.alpha = alpha;
            ms.beta = beta;
            ms.gamma = gamma;
            ms.partials = m.partials;
            ms.orbit = A->mfsec->orbit;
            ms.ucfg = A->mfsec->ucfg;
            ms.oid = A->mfsec->oid;
            ms.utab = A->mfsec->utab;
            ms.oek = A->mfsec->oek;
            ms.uext = A->mfsec->uext;
            ms.tpar = A->mfsec->tpar;
            ms.usgn = A->mfsec->usgn;
            ms.n_orb = A->mfsec->n_orb;
            ms.w_orb = A->mfsec->w_orb;
            ms.tile = A->mfsec->tile;
            // items drawn from per-XCD counters: the orbit-order kernel by default (its workgroups finish far apart under a
            // static assignment: 87 -> 61 ms on 4x5 with 8+8), the rank-table kernel only on request (it got slower: 223 -> 233 ms)
            const int sec_walk = A->dbg.sec_walk < 0 ? (ms.orbit ? 1 : 0) : A->dbg.sec_walk;
            if (sec_walk) {
                if (!A->d_wctr) QBH_HIP(qbh::dev_alloc(&A->d_wctr, qbh::kWctrRegions * 128 * sizeof(unsigned long long)));
                QBH_HIP(hipMemsetAsync(A->d_wctr, 0, 3 * 128 * sizeof(unsigned long long), A->stream));
                ms.ctr = reinterpret_cast<unsigned int *>(A->d_wctr);
            }
            QBH_TRY(qbh::launch_mf_sector(ms, A->stream, &mf_parts));
        } else if (A->kind == 2) {
            qbh::MfHeisArgs h{};
            h.t = A->mfh;
            h.row_begin = m.row_begin;
            h.nrows = m.nrows;
            h.xg = m.xg;
            h.xl = m.xl;
            h.xr = m.xr;
            h.y = m.y;
            h.y_re = m.y_re;
            h.alpha = alpha;
            h.beta = beta;
            h.gamma = gamma;
            h.partials = m.partials;
            QBH_TRY(qbh::launch_mf_heis(h, A->stream, &mf_parts));
        } else {
            QBH_TRY(qbh::launch_mf_hubbard(m, A->grid, A->stream, &mf_parts));
        }
        if (profm) {
            QBH_HIP(hipEventRecord(A->ev1, A->stream));
            A->ev_pending = true;
        }
        A->xr_of = nullptr;
        A->stats.n_spmv++;
        if (realm) A->stats.n_spmv_real++;
        if (red && A->defer_red) {
            QBH_TRY(deferred_reduction(A, mf_parts));
        } else if (red) {
            QBH_TRY(finish_reduction(A, mf_parts, 3, red));
            if (profm) harvest_events(A);
        }
        return QBH_OK;
    }
    qbh::SpmvArgs a{};
    a.ia = A->d_ia;
    a.ja = A->d_ja;
    a.val = A->d_val;
    a.code = A->d_code;
    a.dict = A->d_dict;
    a.dict_mode = A->dict_mode;
    a.rb = A->d_rb;
    a.bp = A->d_bp;
    a.n_blocks = A->n_blocks;
    a.nrows = A->nrows;
    // the local part indexes x by GLOBAL column but only touches [row_offset, row_offset + nrows):
    // serve it from the local block so it does not depend on the gather
    a.xg = (A->has_rem && A->has_comm) ? xl - A->row_offset : xg;
    a.xr = nullptr;
    a.y_re = A->has_comm ? nullptr : A->ovr_yr;
    a.xl_re = a.y_re ? xr_nocomm : nullptr;
    if (realm) {
        if (!A->has_comm) a.xr = xr_nocomm;
        else if (A->has_rem) a.xr = reinterpret_cast<const double *>(A->comm.d_xsend) - A->row_offset;   // own block, packed
        else a.xr = A->comm.d_xfull_r;
    }
    a.xl = xl;
    a.y = y;
    a.alpha = alpha;
    a.beta = beta;
    a.gamma = gamma;
    a.yin = A->ovr_yin;
    a.coef = A->ovr_coef;
    a.coef_mode = A->ovr_coef ? 1 : 0;
    a.partials = (red && !A->has_rem) ? A->d_partials : nullptr;
    a.swizzle = A->opts.xcd_swizzle;
    a.chunk_mult = A->chunk_mult;
    a.unroll = A->unroll;
    a.colmask = (A->debug & 1) ? 1023 : -1;
    if (A->debug & 1) {
        if (A->dbg.colmask) a.colmask = A->dbg.colmask;    // gather-window experiments (results wrong by design)
    }
    const bool prof = A->opts.profile != 0;
    if (async_gather && !A->has_rem) {          // nothing to overlap with: the single part needs the gathered x
        if (A->comm.allgather_wait(A->comm.ctx) != 0) {
            qbh::set_error("allgather_wait hook failed");
            return QBH_ECOMM;
        }
        if (packed && !realm) QBH_TRY(expand_packed());
        async_gather = false;
    }
    if (prof) {
        QBH_TRY(next_event_set(A));
        QBH_HIP(hipEventRecord(A->ev0, A->stream));
    }
    // complex128 values, complex vectors: the wave kernel; the real gather / all-real forms stay on the row kernel
    const bool wave = A->use_wave && a.xr == nullptr && a.y_re == nullptr;
    // The two passes of a Kronecker split take the dynamic ordered walk per XCD (DynWalk: C3 far pass 86 -> 64 GB, near pass
    // 92 -> 68 GB, 31.7 -> 31.0 ms); the unsplit wave kernel keeps the static chunked walk (xcd_swizzle 2), under which all XCDs
    // stream from ONE region -- ordered eighths cost it 34 -> 43 ms on C3.  xcd_swizzle 3 / QBH_WAVE_SWIZZLE choose by name.
    int wave_swz = A->opts.xcd_swizzle, wave_grid_used = A->wgrid;
    if (A->opts.wave_walk >= 0) wave_swz = A->opts.wave_walk;
    // qbh_opts.comm_reserve on a plain shard: the launch of the locally-owned columns runs while the gather is on the links; its
    // grid is persistent (wave and row kernels), so it leaves room for RCCL's kernel (280 registers per lane, see the split path):
    // at most c workgroups per CU, c - 1 on `reserve` of them, with c from the registers a wavefront of this launch can own at its
    // occupancy (512 / occ, an upper bound) so that 512 - (c - 1) * v >= 280
    auto comm_grid = [&](int g) -> int {
        if (!(A->has_comm && A->has_rem && async_gather && A->comm.nranks > 1 && A->opts.comm_reserve >= 0)) return g;
        if (A->ncu <= 0) {
            hipDeviceProp_t prop;
            A->ncu = (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        const int occ = g / A->ncu;
        if (occ < 1) return g;                   // does not fill the chip anyway
        const int v = std::max(8, (512 / occ) / 8 * 8), c = std::max(1, std::min(occ, 1 + 232 / v));
        const int reserve = A->opts.comm_reserve > 0 ? (A->opts.comm_reserve / 8) * 8 : 64;
        const int capped = std::min(g, c * A->ncu);
        return std::max(8, (std::max(capped / 2, capped - reserve) / 8) * 8);
    };
    if (wave && wave_swz == 3) {
        if (!A->d_wctr || A->opts.deterministic) wave_swz = 2;
        else QBH_HIP(hipMemsetAsync(A->d_wctr, 0, 3 * 128 * sizeof(unsigned long long), A->stream));
    }
    const bool kronc = A->kronc.active && realm && a.xr != nullptr && a.y_re != nullptr && !A->has_comm && !A->has_rem && !(A->debug & 1);
    int kronc_parts = 0;
    if (kronc) {
        // coded Kronecker split, all-real operation: tiled copy of the packed x, near launch (full epilogue), far launch
        // (tiled rows and columns, accumulates at orig(row), fused reductions of the finished y)
        qbh_csr::KronCoded &K = A->kronc;
        if (K.sl.active && K.table_route && A->dbg.mf_row != 0) {      // (mf_row=0: the debug switch that turns the row-staged kernel off)
            // T (x) 1 + 1 (x) T' + D recognised: the row-staged table kernel applies it from T, T' and one diagonal code per row --
            // no tiled copy, no far sums: x read (+ its neighbour rows through the caches), old y read, y written
            qbh::MfArgs m{};
            m.t = K.tables;
            m.row_begin = 0;
            m.nrows = A->nrows;
            m.xr = a.xr;
            m.y_re = a.y_re;
            m.alpha = a.alpha;
            m.beta = a.beta;
            m.gamma = a.gamma;
            m.partials = red ? A->d_partials : nullptr;
            m.dcode = K.sl.dcode;
            m.ddict = K.sl.d_dictr;
            K.xt_of = nullptr;
            QBH_TRY(qbh::launch_mf_hubbard(m, 0, A->stream, &kronc_parts));
        } else {
        if (K.xt_of != (const void *)a.xr) QBH_TRY(qbh::launch_kron_tile_re(a.xr, K.d_xt, A->nrows, K.t, A->stream));
        K.xt_of = nullptr;                           // an alias is good for one SpMV
        if (K.sl.active) {
            // sliced form: far pass (row sums in group order), near pass from the LDS-resident block of x with the whole epilogue
            QBH_TRY(qbh::launch_kronc(K.sl, A->d_dict, A->n_dict, K.d_xt, a.xr, a.y_re, a.alpha, a.beta, a.gamma, red ? A->d_partials : nullptr,
                                      reinterpret_cast<unsigned int *>(A->d_wctr), A->opts.deterministic != 0, &kronc_parts, A->stream));
        } else {
        qbh::SpmvArgs np = a;
        np.ia = K.near_p.d_ia;
        np.ja = K.near_p.d_ja;
        np.code = K.near_p.d_code;
        np.rb = K.near_p.d_rb;
        np.bp = K.near_p.d_bp;
        np.n_blocks = K.near_p.n_blocks;
        np.unroll = K.near_p.unroll;
        np.partials = nullptr;
        QBH_TRY(qbh::launch_spmv(np, A->kernel, K.near_p.npb, K.near_p.tpr, K.near_p.grid, A->stream));
        qbh::SpmvArgs fp = a;
        fp.ia = K.far_p.d_ia;
        fp.ja = K.far_p.d_ja;
        fp.code = K.far_p.d_code;
        fp.rb = K.far_p.d_rb;
        fp.bp = K.far_p.d_bp;
        fp.n_blocks = K.far_p.n_blocks;
        fp.unroll = K.far_p.unroll;
        fp.xr = K.d_xt;
        fp.beta = 1.0;
        fp.gamma = 0.0;
        fp.partials = red ? A->d_partials : nullptr;
        fp.swizzle = 1;                             // contiguous eighths of the bands per XCD
        fp.rowmap = 1;
        fp.kS = K.t.S;
        fp.kNU = K.t.NU;
        fp.kB = K.t.B;
        QBH_TRY(qbh::launch_spmv(fp, A->kernel, K.far_p.npb, K.far_p.tpr, K.far_p.grid, A->stream));
        kronc_parts = K.far_p.grid;
        }
        }
    } else if (wave) {
        a.wd = A->d_wd;
        a.n_wb = A->n_wb;
        a.swizzle = wave_swz;
        a.wctr = A->d_wctr;
        const int pipe = A->dbg.wave_pipelined;    // experiment: the pipelined kernel on an unsplit operator
        if (pipe && A->wtpr <= 8) {
            int ncu = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
            const int occ = std::max(1, qbh::wave2_kernel_occupancy(A->wtpr, 1));
            int64_t g = std::min<int64_t>((int64_t)occ * ncu, ((((A->n_wb + 3) >> 2) + 7) / 8) * 8);
            g = std::max<int64_t>(8, (g / 8) * 8);
            g = std::min<int64_t>(g, A->wgrid);            // the partial-sum buffer is sized for the wave kernel's grid
            wave_grid_used = (int)g;
            if (a.swizzle == 3) a.swizzle = 2;             // the experiment has no chunk-partial slots: static walk, per-workgroup partials
            QBH_TRY(qbh::launch_spmv_wave2(a, A->wtpr, 1, (int)g, A->stream));
        } else {
            QBH_TRY(qbh::launch_spmv_wave(a, A->wtpr, comm_grid(A->wgrid), A->stream));
        }
    } else {
        QBH_TRY(qbh::launch_spmv(a, A->kernel, A->npb, A->tpr, A->kernel == QBH_KERNEL_ROWS ? comm_grid(A->grid) : A->grid, A->stream));
    }
    if (prof) {
        QBH_HIP(hipEventRecord(A->ev1, A->stream));
        A->ev_pending = true;
    }
    int grid_last = kronc ? kronc_parts : wave ? wave_grid_used : A->grid;
    if (A->has_rem) {
        if (async_gather) {
            if (A->comm.allgather_wait(A->comm.ctx) != 0) {
                qbh::set_error("allgather_wait hook failed");
                return QBH_ECOMM;
            }
            if (packed && !realm) QBH_TRY(expand_packed());
        }
        const CsrPart &R = A->rem;
        a.ia = R.d_ia;
        a.ja = R.d_ja;
        a.val = R.d_val;
        a.code = R.d_code;
        a.rb = R.d_rb;
        a.bp = R.d_bp;
        a.n_blocks = R.n_blocks;
        a.xg = xg;
        if (realm) a.xr = A->has_comm ? A->comm.d_xfull_r : xr_nocomm;
        a.beta = 1.0;                       // accumulate onto the local part's result
        a.gamma = 0.0;
        a.yin = nullptr;
        a.coef_mode = A->ovr_coef ? 2 : 0;
        a.partials = red ? A->d_partials : nullptr;
        a.unroll = R.unroll;
        if (prof) QBH_HIP(hipEventRecord(A->ev2, A->stream));
        const bool wave_r = A->use_wave && a.xr == nullptr && a.y_re == nullptr;
        if (wave_r) {
            a.wd = R.d_wd;
            a.n_wb = R.n_wb;
            a.wctr = A->d_wctr + 128;          // the remote part's own counters (the local part may still be running)
            QBH_TRY(qbh::launch_spmv_wave(a, R.wtpr, R.wgrid, A->stream));
        } else {
            QBH_TRY(qbh::launch_spmv(a, A->kernel, R.npb, R.tpr, R.grid, A->stream));
        }
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev3, A->stream));
            A->ev_pending2 = true;
        }
        grid_last = wave_r ? R.wgrid : R.grid;
    }
    A->xr_of = nullptr;                      // the packed copy is consumed by exactly one SpMV
    A->stats.n_spmv++;
    if (realm) A->stats.n_spmv_real++;
    if (red && A->defer_red) {
        QBH_TRY(deferred_reduction(A, grid_last));
    } else if (red) {
        QBH_TRY(finish_reduction(A, grid_last, 3, red));
        if (prof) harvest_events(A);
    }
    return QBH_OK;
}

// Drivers call this at entry with the vectors of their recurrence: when the operator is real and all of them
// have exactly zero imaginary parts (on every rank), the x exchange carries only real parts for this solve.
int enable_real_wire(qbh_csr *A, std::initializer_list<const d2 *> vecs)
{
    A->real_wire = false;
    A->real_mode = false;
    A->xr_of = nullptr;
    if (A->has_comm && !A->comm.d_xfull_r) return QBH_OK;
    // split shards exchanging tiled blocks: qbh_opts.real_wire alone decides (their kernels have no real forms to select);
    // everything else: the real fast path and its forms
    const bool tiled = A->has_comm && A->kron.active && A->kron.comm_tiled;
    if (tiled) {
        if (!A->opts.real_wire) return QBH_OK;
    } else {
        if (!A->opts.real_fast_path) return QBH_OK;
        if (!(A->opts.real_forms & 1)) return QBH_OK;
    }
    double total = A->values_real ? 0.0 : 1.0;
    // every rank must take the same decision: sum the per-vector |Im|^2 (and the operator flag) over ranks
    for (const d2 *v : vecs) {
        double sq = 0.0;
        QBH_TRY(qbh::launch_imag_norm(v, A->nrows, A->d_partials, A->stream));
        QBH_TRY(finish_reduction(A, qbh::blas_grid(A->nrows), 1, &sq));
        total += sq;
    }
    double flag_sum = 0.0;
    {   // the operator flag also has to be agreed on
        const double mine = A->values_real ? 0.0 : 1.0;
        QBH_HIP(hipMemcpyAsync(A->d_partials, &mine, sizeof(double), hipMemcpyHostToDevice, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
        QBH_TRY(finish_reduction(A, 1, 1, &flag_sum));
    }
    if (total == 0.0 && flag_sum == 0.0) {
        QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream));
        A->real_wire = A->has_comm;
        // the row kernel can then gather 8-byte real parts (bit-identical result, half the x traffic)
        A->real_mode = A->kernel == QBH_KERNEL_ROWS && !tiled;
        if (!(A->opts.real_forms & 2)) A->real_mode = false;
        if (A->real_mode && !A->has_comm && !A->d_xr) QBH_HIP(qbh::dev_alloc(&A->d_xr, (size_t)A->ncols * sizeof(double)));
    }
    return QBH_OK;
}

// ... and this at exit: a non-zero imaginary part met while packing means results are wrong -> loud error.
int finish_real_wire(qbh_csr *A)
{
    A->xr_of = nullptr;
    if (!A->real_wire && !A->real_mode) return QBH_OK;
    A->real_wire = false;
    A->real_mode = false;
    int f = 0;
    QBH_HIP(hipMemcpyAsync(&f, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    const double mine = (double)f;
    double all = 0.0;
    QBH_HIP(hipMemcpyAsync(A->d_partials, &mine, sizeof(double), hipMemcpyHostToDevice, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    QBH_TRY(finish_reduction(A, 1, 1, &all));
    if (all != 0.0) {
        qbh::set_error("real wire format met a non-zero imaginary part (internal error)");
        return QBH_EHIP;
    }
    return QBH_OK;
}


int dotc_run(qbh_csr *A, const d2 *x, const d2 *y, double *res2)
{
    QBH_TRY(qbh::launch_dotc(x, y, A->nrows, A->d_partials, A->stream));
    return finish_reduction(A, qbh::blas_grid(A->nrows), 2, res2);
}


int axpy_norm_run(qbh_csr *A, d2 alpha, const d2 *x, d2 *y, double *nrm2sq)
{
    double *yr = packed_target(A);          // y is the next SpMV's x in every driver: emit its packed copy here
    if (d2 *yt = tiled_target(A)) {         // ... or, for a Kronecker split, its tiled copy
        QBH_TRY(qbh::launch_axpy_norm_tile(alpha, nullptr, x, y, yt, A->nrows, A->kron.t, A->d_partials, A->stream, nullptr, tiled_real(A), A->d_flag));
        A->kron.xt_of = y;
        A->xr_of = nullptr;
        return finish_reduction(A, qbh::blas_grid(A->nrows), 1, nrm2sq);
    }
    QBH_TRY(qbh::launch_axpy_norm(alpha, nullptr, x, y, A->nrows, A->d_partials, yr, A->d_flag, A->stream));
    A->xr_of = yr ? y : nullptr;
    A->kron.xt_of = nullptr;
    return finish_reduction(A, qbh::blas_grid(A->nrows), 1, nrm2sq);
}

// Single-GPU Lanczos step tail: y += (scale * d_scal[0]) x with d_scal[0] = <x, w> left on the device by a deferred
// spmv_run, then |y|^2; ONE copy + synchronisation returns both scalars (dot_out = d_scal[0], *nrm2sq).
int axpy_norm_deferred(qbh_csr *A, double scale, const d2 *x, d2 *y, double *dot_out, double *nrm2sq)
{
    double *yr = packed_target(A);
    if (d2 *yt = tiled_target(A)) {
        QBH_TRY(qbh::launch_axpy_norm_tile(d2{scale, 0.0}, A->d_scal, x, y, yt, A->nrows, A->kron.t, A->d_partials, A->stream, nullptr, tiled_real(A), A->d_flag));
        A->kron.xt_of = y;
        A->xr_of = nullptr;
    } else {
        QBH_TRY(qbh::launch_axpy_norm(d2{scale, 0.0}, A->d_scal, x, y, A->nrows, A->d_partials, yr, A->d_flag, A->stream));
        A->xr_of = yr ? y : nullptr;
        A->kron.xt_of = nullptr;
    }
    QBH_TRY(qbh::launch_reduce_partials(A->d_partials, qbh::blas_grid(A->nrows), 1, A->d_scal + 4, A->stream));
    QBH_HIP(hipMemcpyAsync(A->h_scal, A->d_scal, 5 * sizeof(double), hipMemcpyDeviceToHost, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    *dot_out = A->h_scal[0];
    *nrm2sq = A->h_scal[4];
    if (A->opts.profile) harvest_events(A);
    return QBH_OK;
}

int nrm2_run(qbh_csr *A, const d2 *x, double *nrm)
{
    double sq = 0.0;
    QBH_TRY(qbh::launch_nrm2sq(x, A->nrows, A->d_partials, A->stream));
    QBH_TRY(finish_reduction(A, qbh::blas_grid(A->nrows), 1, &sq));
    *nrm = std::sqrt(sq);
    return QBH_OK;
}

}  // namespace qbhapi

// ----------------------------------------------- device building blocks --------
extern "C" int qbh_spmv_dev(const qbh_csr *Ac, const qbh_z *d_x, qbh_z *d_y, double alpha, double beta,
                            double gamma, double *red)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x || !d_y) return QBH_EINVAL;
    Bind bind(A);
    return spmv_run(A, reinterpret_cast<const d2 *>(d_x), reinterpret_cast<d2 *>(d_y), alpha, beta, gamma, red);
}

extern "C" int qbh_dotc_dev(const qbh_csr *Ac, const qbh_z *d_x, const qbh_z *d_y, double *res)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x || !d_y || !res) return QBH_EINVAL;
    Bind bind(A);
    return dotc_run(A, reinterpret_cast<const d2 *>(d_x), reinterpret_cast<const d2 *>(d_y), res);
}

extern "C" int qbh_axpy_norm_dev(const qbh_csr *Ac, qbh_z alpha, const qbh_z *d_x, qbh_z *d_y, double *nrm2_sq)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x || !d_y || !nrm2_sq) return QBH_EINVAL;
    Bind bind(A);
    return axpy_norm_run(A, d2{alpha.re, alpha.im}, reinterpret_cast<const d2 *>(d_x),
                         reinterpret_cast<d2 *>(d_y), nrm2_sq);
}

extern "C" int qbh_scal_dev(const qbh_csr *Ac, double a, qbh_z *d_x)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x) return QBH_EINVAL;
    Bind bind(A);
    return qbh::launch_scal(a, reinterpret_cast<d2 *>(d_x), A->nrows, A->stream);
}

extern "C" int qbh_nrm2_dev(const qbh_csr *Ac, const qbh_z *d_x, double *nrm)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x || !nrm) return QBH_EINVAL;
    Bind bind(A);
    return nrm2_run(A, reinterpret_cast<const d2 *>(d_x), nrm);
}

// -------------------------------------------------- host-vector seam -----------
namespace qbhapi {
int multmv_host(qbh_csr *A, const qbh_z *x_host, qbh_z *y_host, double beta)
{
    if (!A || !x_host || !y_host) return QBH_EINVAL;
    if (A->has_comm || A->nrows != A->ncols) {
        qbh::set_error("qbh_multmv: host-vector seam needs an unsharded operator");
        return QBH_EUNSUPP;
    }
    Bind bind(A);
    const size_t bytes = (size_t)A->nrows * sizeof(d2);
    if (!A->d_stage_x) QBH_HIP(qbh::dev_alloc(&A->d_stage_x, bytes));
    if (!A->d_stage_y) QBH_HIP(qbh::dev_alloc(&A->d_stage_y, bytes));
    if (A->basis.kind != 0) {                     // the caller's order at the seam, the internal one in HBM
        QBH_TRY(vec_h2d(A, A->d_stage_x, x_host, A->nrows));
        if (beta != 0.0) QBH_TRY(vec_h2d(A, A->d_stage_y, y_host, A->nrows));
        QBH_TRY(spmv_run(A, A->d_stage_x, A->d_stage_y, 1.0, beta, 0.0, nullptr));
        return vec_d2h(A, y_host, A->d_stage_y, A->nrows);
    }
    QBH_HIP(hipMemcpyAsync(A->d_stage_x, x_host, bytes, hipMemcpyHostToDevice, A->stream));
    if (beta != 0.0) QBH_HIP(hipMemcpyAsync(A->d_stage_y, y_host, bytes, hipMemcpyHostToDevice, A->stream));
    QBH_TRY(spmv_run(A, A->d_stage_x, A->d_stage_y, 1.0, beta, 0.0, nullptr));
    QBH_HIP(hipMemcpyAsync(y_host, A->d_stage_y, bytes, hipMemcpyDeviceToHost, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}
}  // namespace qbhapi

extern "C" int qbh_multmv(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host)
{
    return multmv_host(const_cast<qbh_csr *>(A), x_host, y_host, 0.0);
}

extern "C" int qbh_multmv2(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host)
{
    return multmv_host(const_cast<qbh_csr *>(A), x_host, y_host, 1.0);
}
