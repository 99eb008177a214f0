// qbh_split.cpp -- the Kronecker split of a product-basis operator, built IN PLACE (complex128: kron_build, 2-byte columns,
// kron_restore) and its sibling for the library's default coded format (kronc_build).  Split out of qbh_api.cpp in round 5.
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

namespace qbhapi {
// everything of the split except the matrix arrays themselves (those are the handle's own d_ja / d_val, re-ordered in place)
void kron_free_aux(qbh_csr *A)
{
    qbh_csr::KronSplit &K = A->kron;
    for (void *q : {(void *)K.ia_n, (void *)K.ia_f, (void *)K.wd_n, (void *)K.wd_f, (void *)K.d_xt, (void *)K.d_far, (void *)K.ia_x, (void *)K.xrow,
                    (void *)K.wd_x, (void *)K.d_cls, (void *)K.c16_n, (void *)K.c16_f, (void *)K.d_chunk_red, (void *)K.d_need, (void *)K.d_send_list, (void *)K.d_vsend, (void *)K.d_vrecv})
        if (q) (void)hipFree(q);
    if (K.own_far) {
        if (K.ja_f) (void)hipFree(K.ja_f);
        if (K.val_f) (void)hipFree(K.val_f);
    }
    if (K.own_x) {
        if (K.ja_x) (void)hipFree(K.ja_x);
        if (K.val_x) (void)hipFree(K.val_x);
    }
    K = qbh_csr::KronSplit{};
}

qbh::KronParts kron_parts(const qbh_csr *A)
{
    const qbh_csr::KronSplit &K = A->kron;
    qbh::KronParts p{};
    p.ia = A->d_ia;
    p.ia_n = K.ia_n;
    p.fp = K.ia_f;
    p.ja_n = K.ja_n;
    p.ja_f = K.ja_f;
    p.c16_n = K.c16_n;
    p.c16_f = K.c16_f;
    p.val_n = K.val_n;
    p.val_f = K.val_f;
    p.ia_x = K.ia_x;
    p.xrow = K.xrow;
    p.n_xrows = K.n_xrows;
    p.ja_x = K.ja_x;
    p.val_x = K.val_x;
    p.map = K.map;
    return p;
}

qbh::KronCols kron_cols_one(int64_t S, int64_t NUg, int B)
{
    qbh::KronCols c{};
    c.S = S;
    c.B = B;
    c.nr = 1;
    c.cu[0] = 0;
    c.cu[1] = NUg;
    return c;
}

int wave_geometry_for(qbh_csr *A, const int64_t *ia, int64_t nr, int64_t nnz, double avg, bool slots, int ops, qbh::WaveDesc **wd_io, int64_t *nwb_o,
                      int *tpr_o, int *grid_o, int64_t shift)
{
    hipStream_t s = A->stream;
    QBH_TRY(qbh::launch_max_rowlen(ia, nr, (int64_t *)A->d_scal, s));
    int64_t maxlen = 0;
    QBH_HIP(hipMemcpyAsync(&maxlen, A->d_scal, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    const int64_t window = slots ? 512 : (maxlen <= 256) ? 505 - (maxlen > 0 ? maxlen - 1 : 0) : 249;
    const int64_t n_wb = std::max<int64_t>(1, (nnz + (slots ? shift : 0) + window - 1) / window);
    if (*wd_io) (void)hipFree(*wd_io);
    *wd_io = nullptr;
    QBH_HIP(hipMalloc(wd_io, (size_t)(n_wb + 2) * sizeof(qbh::WaveDesc)));
    if (slots) QBH_TRY(qbh::launch_build_slotdesc(ia, nr, nnz, *wd_io, n_wb, shift, s));
    else       QBH_TRY(qbh::launch_build_wavedesc(ia, nr, window, *wd_io, n_wb, s));
    const int tpr = avg <= 32 ? 2 : avg <= 64 ? 4 : 8;
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const int occ = std::max(1, ops >= 0 ? qbh::wave2_kernel_occupancy(tpr, ops) : qbh::wave_kernel_occupancy(tpr));
    int64_t g = std::min<int64_t>((int64_t)occ * ncu, ((((n_wb + 3) >> 2) + 7) / 8) * 8);
    g = std::max<int64_t>(8, (g / 8) * 8);
    *nwb_o = n_wb;
    *tpr_o = tpr;
    *grid_o = (int)g;
    return QBH_OK;
}

// wave-block geometry of the parts (pipelined kernel; the dense cross part of a several-class operator: the plain wave kernel)
int kron_geometry(qbh_csr *A)
{
    qbh_csr::KronSplit &K = A->kron;
    const int64_t n = A->nrows;
    QBH_TRY(wave_geometry_for(A, K.ia_n, n, K.nnz_n, (double)K.nnz_n / (double)n, false, K.map.nc > 1 ? 4 : 2, &K.wd_n, &K.nwb_n, &K.tpr_n, &K.grid_n));
    QBH_TRY(wave_geometry_for(A, K.ia_f, K.sliced ? K.n_groups : K.map.nfar_rows(), K.far_slots, (double)K.nnz_f / (double)n, K.sliced, K.sliced ? 3 : 0,
                              &K.wd_f, &K.nwb_f, &K.tpr_f, &K.grid_f));
    if (K.map.nc > 1) {
        std::vector<qbh::KronCls> hc((size_t)K.map.nc + 1);
        for (int c = 0; c <= K.map.nc; ++c)
            hc[(size_t)c] = qbh::KronCls{K.map.rbase[c], c < K.map.nc ? K.map.S[c] : 1, c < K.map.nc ? K.map.NU[c] : 0, K.map.fbase[c]};
        if (!K.d_cls) QBH_HIP(hipMalloc(&K.d_cls, hc.size() * sizeof(qbh::KronCls)));
        QBH_HIP(hipMemcpy(K.d_cls, hc.data(), hc.size() * sizeof(qbh::KronCls), hipMemcpyHostToDevice));
        QBH_TRY(qbh::launch_kron_desc_classes(K.wd_n, K.nwb_n, K.d_cls, K.map.nc, A->stream));
        if (K.nnz_x > 0)
            QBH_TRY(wave_geometry_for(A, K.ia_x, n, K.nnz_x, (double)K.nnz_x / (double)n, false, -1, &K.wd_x, &K.nwb_x, &K.tpr_x, &K.grid_x));
    }
    // slots of the near pass's chunk partials (zeroed once: every real chunk overwrites its slot in every launch)
    if (K.d_chunk_red) (void)hipFree(K.d_chunk_red);
    K.d_chunk_red = nullptr;
    K.n_chunk_slots = qbh::wave2_chunk_slots(K.nwb_n);
    QBH_HIP(hipMalloc(&K.d_chunk_red, (size_t)K.n_chunk_slots * 3 * sizeof(double)));
    QBH_HIP(hipMemsetAsync(K.d_chunk_red, 0, (size_t)K.n_chunk_slots * 3 * sizeof(double), A->stream));
    // 2-byte columns are relative to a base the block's descriptor names: fresh descriptors get it again
    if (K.c16_n) QBH_TRY(qbh::launch_kron_desc_c16(K.wd_n, K.nwb_n, K.t.S, false, false, A->stream));
    if (K.c16_f) QBH_TRY(qbh::launch_kron_desc_c16(K.wd_f, K.nwb_f, K.t.NU, true, false, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}

// 2-byte columns for the parts of a one-class split (qbh_opts.kron_cols16).  The two passes are bound by the rate of line
// requests, not by bytes (DESIGN-history 5.0b): the column stream is 16 of a block's ~150 lines as int32 and 8 as uint16.
//   near part: column - (first column of the shard + pad * S), pad = major index of the block's first row (a block of whole rows
//              with <= 512 entries reaches into the next major index at most: values < 2 S);
//   far part (sliced): target major index + (band - band of the block's first group) * NUg (NUg = major indices of the WHOLE
//              operator, also for a row shard); the element of the tiled x is band0 * 8 NUg + 8 * value + slot % 8.
// Each part is converted when every value fits 16 bits (checked on the device) and then lives in an allocation of its own; when
// both are and nothing else sits in the int32 array it is released (C3: 23.3 GB -> 11.6 GB of columns).  qbh_csr_download /
// kron_restore re-derive the int32 columns (k_kron_merge_rows): value mod S inside the row's block, value mod NU as the major index.
int kron_short_cols(qbh_csr *A)
{
    qbh_csr::KronSplit &K = A->kron;
    if (!A->opts.kron_cols16 || K.map.nc != 1 || !K.inplace || K.c16_n || K.c16_f) return QBH_OK;
    hipStream_t s = A->stream;
    const int64_t S = K.t.S, NU = K.t.NU;
    auto convert = [&](bool far, uint16_t **out) -> int {
        const int64_t cnt = far ? K.far_slots : K.nnz_n;
        uint16_t *c = nullptr;
        if (hipMalloc(&c, (size_t)(cnt + 64) * sizeof(uint16_t)) != hipSuccess) {
            (void)hipGetLastError();
            return QBH_OK;                               // no room: the part keeps its int32 columns
        }
        int bad = 0;
        int rc = qbh::launch_kron_desc_c16(far ? K.wd_f : K.wd_n, far ? K.nwb_f : K.nwb_n, far ? NU : S, far, false, s);
        hipError_t he = hipMemsetAsync(A->d_flag, 0, sizeof(int), s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemsetAsync(c + cnt, 0, 64 * sizeof(uint16_t), s);
        if (rc == QBH_OK && he == hipSuccess)
            rc = far ? qbh::launch_kron_c16_far(K.wd_f, K.ja_f, K.far_slots, K.NUg, c, A->d_flag, s)
                     : qbh::launch_kron_c16_near(K.wd_n, K.nwb_n, K.ja_n, S, A->row_offset, c, A->d_flag, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemcpyAsync(&bad, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipStreamSynchronize(s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemsetAsync(A->d_flag, 0, sizeof(int), s);
        if (rc != QBH_OK || he != hipSuccess || bad) {  // does not fit (or failed): back to the plain descriptors
            (void)hipGetLastError();
            (void)hipFree(c);
            const int rc2 = qbh::launch_kron_desc_c16(far ? K.wd_f : K.wd_n, far ? K.nwb_f : K.nwb_n, far ? NU : S, far, true, s);
            if (hipStreamSynchronize(s) != hipSuccess || rc2 != QBH_OK) return QBH_EHIP;
            return rc != QBH_OK ? rc : he != hipSuccess ? QBH_EHIP : QBH_OK;
        }
        *out = c;
        return QBH_OK;
    };
    if (K.nnz_n > 0 && 2 * S <= 65536) QBH_TRY(convert(false, &K.c16_n));
    // far: the sliced layout over the tiled order of the WHOLE vector -- also for a row shard: value = target major index (of the
    // whole operator) + (band - band of the block's first group) * NUg; under a communicator the gathered blocks are moved into that
    // order (k_kron_place), the columns never change
    if (K.sliced && !K.own_far && K.t.B == 8 && 2 * K.NUg <= 65536 && K.far_slots > 0)
        QBH_TRY(convert(true, &K.c16_f));
    if (K.c16_n && K.c16_f && A->own_arrays && (K.nnz_x == 0 || K.own_x)) {      // nothing is left in the int32 array
        (void)hipFree(A->d_ja);
        A->d_ja = nullptr;
    }
    if (K.c16_n) K.ja_n = nullptr;
    if (K.c16_f) K.ja_f = nullptr;
    return QBH_OK;
}

// H = H_near + H_far (+ H_cross) for an operator whose rows have a product structure (KronMap): index = major * S + minor with
// every entry changing either the minor index (near: inside the row's own block of S columns, an L2-sized window of x) or the
// major index alone (far: same minor index).  The far part is what makes a row-major sweep re-read x: every major index pulls in
// the x rows of all its neighbours (C3: 17 x 2.65 GB per SpMV).  Stored band-major over the minor index -- rows and columns in
// the tiled order of KronTile -- a band of 8 minor indices needs ONE 128-byte line per major index, and 8 consecutive far rows
// share every line they gather.  Per SpMV: x -> tiled copy (or written by the pass that produced x), far pass (row sums, tiled
// order), near pass (+ far result, fused epilogue).
// Round 4: the split REPLACES the CSR -- the handle's own d_ja / d_val are re-ordered in place into [near | far | cross] (same
// values, same columns -- 2 bytes each afterwards where they fit, kron_short_cols; peak during the conversion = the CSR + one copy of the far and cross parts), row
// shards made of whole major indices split the same way, and qbh_csr_download / kron_restore merge the parts back.  The choice
// is STRUCTURAL (verified on the device, never assumed, never timed): results do not depend on the box.  kron_split = 1 leaves
// operators below 1e8 nonzeros alone (three launches cost more than they save there); 2 splits whatever has the structure.
int kron_build(qbh_csr *A)
{
    if (A->kron.active) return QBH_OK;
    if (!A->use_wave || A->kind != 0 || A->has_rem || A->nnz <= 0 || !A->own_arrays || !A->d_val || A->kron_off) return QBH_OK;
    if (A->opts.kron_split == 0 || (A->debug & 1)) return QBH_OK;
    // the threshold is on the WHOLE operator (a shard's share scaled up): uneven shards must not decide differently
    if (A->opts.kron_split == 1 && (double)A->nnz * ((double)A->ncols / (double)A->nrows) < 1e8) return QBH_OK;
    if (A->opts.real_fast_path && A->values_real) return QBH_OK;      // the real-gather form of the row kernel needs the CSR
    const int64_t n = A->nrows;
    hipStream_t s = A->stream;
    qbh_csr::KronSplit &K = A->kron;
    const bool multi = A->basis.kind == QBH_BASIS_SPIN_SECTOR && A->basis.classes.nc > 1;
    if (multi) {
        if (A->nrows != A->ncols || A->row_offset != 0) return QBH_OK;
        K.map = A->basis.classes;
        K.map.sliced = 1;
        // qbh_opts.kron_cross_in_near = 0: the entries across the cut as a third pass of their own (k_spmv_wave, tiled columns);
        // default: they stay in the near part -- natural columns, gathers that miss -- which saves the third pass's reading of the vectors
        K.map.cross_near = A->opts.kron_cross_in_near ? 1 : 0;
        K.t = qbh::KronTile{K.map.S[0], K.map.NU[0], 8};
        K.U0 = 0;
        K.NUg = 0;
        K.cols = kron_cols_one(1, n, 8);
    } else {
        const int64_t S = A->opts.kron_minor;
        if (S <= 1 || S >= A->ncols || A->ncols % S != 0 || A->nrows % S != 0 || A->row_offset % S != 0) return QBH_OK;
        const int64_t NU = A->nrows / S, NUg = A->ncols / S, U0 = A->row_offset / S;
        // the structure is verified, never assumed: one entry that changes both indices and the operator stays unsplit
        QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        QBH_TRY(qbh::launch_kron_check2(A->d_ia, A->d_ja, n, S, U0, A->d_flag, s));
        int bad = 0;
        QBH_HIP(hipMemcpyAsync(&bad, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        QBH_HIP(hipStreamSynchronize(s));
        QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        if (bad) return QBH_OK;
        int B = 8;                                       // one 128-byte line of complex128 per (band, major index)
        while (B > 2 && (double)NUg * B * 16 > 2.5e6) B >>= 1;             // keep a band of x inside an XCD's L2
        {
            const int b = A->opts.kron_band;
            if (b == 2 || b == 4 || b == 8 || b == 16) B = b;
        }
        K.t = qbh::KronTile{S, NU, B};
        K.U0 = U0;
        K.NUg = NUg;
        K.cols = kron_cols_one(S, NUg, B);
        qbh::KronMap m{};
        m.nc = 1;
        m.B = B;
        m.U0 = U0;
        m.rbase[0] = 0;
        m.rbase[1] = n;
        m.S[0] = S;
        m.NU[0] = NU;
        m.fbase[0] = 0;
        m.fbase[1] = (S / B) * B * NU;
        m.cols = K.cols;
        const int want_sliced = A->opts.kron_sliced;     // 0 never, 1 when the padding is small, 2 whenever a group fits
        m.sliced = (want_sliced && B == 8 && S >= 8) ? 1 : 0;
        K.map = m;
    }
    int32_t *cn = nullptr, *cf = nullptr, *cx = nullptr, *tmp_c = nullptr, *tmpx_c = nullptr;
    d2 *tmp_v = nullptr, *tmpx_v = nullptr;
    void *chunk = nullptr;
    int32_t *d_rb = nullptr;
    int64_t *d_bp = nullptr;
    bool destructive = false;                        // the CSR is being re-ordered: a failure from here on is an error
    auto fail = [&](int code) {
        for (void *q : {(void *)cn, (void *)cf, (void *)cx, (void *)tmp_c, (void *)tmp_v, (void *)tmpx_c, (void *)tmpx_v, chunk, (void *)d_rb, (void *)d_bp})
            if (q) (void)hipFree(q);
        K.own_far = false;                           // tmp_c / tmp_v freed above
        K.ja_f = nullptr;
        K.val_f = nullptr;
        if (K.own_x && (tmpx_c == nullptr)) {        // already adopted: freed by kron_free_aux
        } else {
            K.own_x = false;
        }
        kron_free_aux(A);
        if (destructive && code == QBH_OK) code = QBH_EHIP;
        if (destructive) {
            qbh::set_error("Kronecker split: the in-place conversion failed half way; the operator is unusable");
            A->broken = true;
        }
        return code;
    };
#define KRON_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (void)hipGetLastError();                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? QBH_OK : QBH_EHIP); /* out of memory: stay unsplit */ \
        }                                                                                                \
    } while (0)
#define KRON_TRY(expr)                         \
    do {                                       \
        const int rc_ = (expr);                \
        if (rc_ != QBH_OK) return fail(rc_);   \
    } while (0)
    // ---- how many entries of every row go where ----
    int64_t nfr = K.map.nfar_rows();
    KRON_HIP(hipMalloc(&cn, (size_t)n * sizeof(int32_t)));
    KRON_HIP(hipMalloc(&cf, (size_t)std::max<int64_t>(nfr, n) * sizeof(int32_t)));
    KRON_HIP(hipMalloc(&cx, (size_t)n * sizeof(int32_t)));
    KRON_HIP(hipMemsetAsync(cf, 0, (size_t)std::max<int64_t>(nfr, n) * sizeof(int32_t), s));
    KRON_TRY(qbh::launch_kron_count3(A->d_ia, A->d_ja, n, K.map, cn, cf, cx, s));
    KRON_HIP(hipMalloc(&K.ia_n, (size_t)(n + 1) * sizeof(int64_t)));
    KRON_TRY(qbh::exclusive_scan(cn, n, K.ia_n, s));
    KRON_HIP(hipMemcpy(&K.nnz_n, K.ia_n + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    // far part: sliced (groups of 8 far rows, entries interleaved: every 8 consecutive stream elements are one 128-byte line of
    // the tiled x; groups padded to their longest row -- none for a product operator) while the padding stays under 1/8 of the
    // far entries and a group fits the wave tile; else plain rows in tiled order
    K.sliced = false;
    K.n_groups = nfr / 8;
    if (K.map.sliced) {
        int32_t *gw = nullptr;
        int64_t *gia = nullptr;
        KRON_HIP(hipMalloc(&gw, (size_t)std::max<int64_t>(K.n_groups, 1) * sizeof(int32_t)));
        int rc = qbh::launch_kron_group_width(cf, nfr, K.n_groups, gw, s);
        hipError_t he = rc == QBH_OK ? hipMalloc(&gia, (size_t)(K.n_groups + 1) * sizeof(int64_t)) : hipSuccess;
        if (rc == QBH_OK && he == hipSuccess) rc = qbh::exclusive_scan(gw, K.n_groups, gia, s);
        int64_t slots = 0, maxgw = 0, far_true = 0;
        int64_t *tmp_scan = nullptr;
        if (rc == QBH_OK && he == hipSuccess) he = hipMalloc(&tmp_scan, (size_t)(nfr + 1) * sizeof(int64_t));
        if (rc == QBH_OK && he == hipSuccess) rc = qbh::exclusive_scan(cf, nfr, tmp_scan, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemcpy(&far_true, tmp_scan + nfr, sizeof(int64_t), hipMemcpyDeviceToHost);
        if (tmp_scan) (void)hipFree(tmp_scan);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemcpy(&slots, gia + K.n_groups, sizeof(int64_t), hipMemcpyDeviceToHost);
        if (rc == QBH_OK && he == hipSuccess) rc = qbh::launch_max_rowlen(gia, K.n_groups, (int64_t *)A->d_scal, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemcpyAsync(&maxgw, A->d_scal, sizeof(int64_t), hipMemcpyDeviceToHost, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipStreamSynchronize(s);
        (void)hipFree(gw);
        if (rc != QBH_OK || he != hipSuccess) {
            if (gia) (void)hipFree(gia);
            (void)hipGetLastError();
            return fail(rc != QBH_OK ? rc : he == hipErrorOutOfMemory ? QBH_OK : QBH_EHIP);
        }
        const int want_sliced = A->opts.kron_sliced;
        K.nnz_f = far_true;
        if ((want_sliced == 2 || multi || slots - far_true <= far_true / 8) && maxgw <= 504 && slots < ((int64_t)1 << 40) && K.n_groups > 0) {
            K.ia_f = gia;
            K.sliced = true;
            K.far_slots = slots;
        } else {
            (void)hipFree(gia);
            if (multi) return fail(QBH_OK);           // several classes need the compact far rows of the sliced form
            K.map.sliced = 0;                         // plain rows in tiled order: every row has a far row id again
            K.map.fbase[1] = n;
            nfr = n;
            KRON_HIP(hipMemsetAsync(cf, 0, (size_t)n * sizeof(int32_t), s));
            KRON_TRY(qbh::launch_kron_count3(A->d_ia, A->d_ja, n, K.map, cn, cf, cx, s));
        }
    }
    if (!K.sliced) {
        KRON_HIP(hipMalloc(&K.ia_f, (size_t)(nfr + 1) * sizeof(int64_t)));
        KRON_TRY(qbh::exclusive_scan(cf, nfr, K.ia_f, s));
        KRON_HIP(hipMemcpy(&K.nnz_f, K.ia_f + nfr, sizeof(int64_t), hipMemcpyDeviceToHost));
        K.far_slots = K.nnz_f;
        K.n_groups = (nfr + 7) / 8;
    }
    // cross part: a compact list of the few rows that have one (one class), or row pointers over all rows (several classes)
    K.n_xrows = 0;
    if (multi) {
        KRON_HIP(hipMalloc(&K.ia_x, (size_t)(n + 1) * sizeof(int64_t)));
        KRON_TRY(qbh::exclusive_scan(cx, n, K.ia_x, s));
        KRON_HIP(hipMemcpy(&K.nnz_x, K.ia_x + n, sizeof(int64_t), hipMemcpyDeviceToHost));
        K.n_xrows = K.nnz_x > 0 ? n : 0;
    } else {
        int32_t *fl = nullptr, *cc = nullptr;
        int64_t *pos = nullptr;
        KRON_HIP(hipMalloc(&fl, (size_t)n * sizeof(int32_t)));
        hipError_t he = hipMalloc(&pos, (size_t)(n + 1) * sizeof(int64_t));
        int rc = he == hipSuccess ? qbh::launch_kron_flags(cx, n, fl, s) : QBH_OK;
        if (rc == QBH_OK && he == hipSuccess) rc = qbh::exclusive_scan(fl, n, pos, s);
        if (rc == QBH_OK && he == hipSuccess) he = hipMemcpy(&K.n_xrows, pos + n, sizeof(int64_t), hipMemcpyDeviceToHost);
        if (rc == QBH_OK && he == hipSuccess && K.n_xrows > 0) {
            he = hipMalloc(&K.xrow, (size_t)K.n_xrows * sizeof(int32_t));
            if (he == hipSuccess) he = hipMalloc(&cc, (size_t)K.n_xrows * sizeof(int32_t));
            if (he == hipSuccess) he = hipMalloc(&K.ia_x, (size_t)(K.n_xrows + 1) * sizeof(int64_t));
            if (he == hipSuccess) rc = qbh::launch_kron_xrows(cx, n, pos, K.xrow, cc, s);
            if (rc == QBH_OK && he == hipSuccess) rc = qbh::exclusive_scan(cc, K.n_xrows, K.ia_x, s);
            if (rc == QBH_OK && he == hipSuccess) he = hipMemcpy(&K.nnz_x, K.ia_x + K.n_xrows, sizeof(int64_t), hipMemcpyDeviceToHost);
        }
        (void)hipFree(fl);
        if (pos) (void)hipFree(pos);
        if (cc) (void)hipFree(cc);
        if (rc != QBH_OK || he != hipSuccess) {
            (void)hipGetLastError();
            return fail(rc != QBH_OK ? rc : he == hipErrorOutOfMemory ? QBH_OK : QBH_EHIP);
        }
    }
    for (int32_t **q : {&cn, &cf, &cx}) {
        (void)hipFree(*q);
        *q = nullptr;
    }
    if (K.nnz_f == 0 || K.nnz_n + K.nnz_f + K.nnz_x != A->nnz) return fail(QBH_OK);       // nothing far: the split buys nothing
    if (multi && (double)K.nnz_x > 0.4 * (double)A->nnz) return fail(QBH_OK);               // mostly unstructured: not worth three passes
    if (multi && K.map.cross_near && (double)K.nnz_f < 0.15 * (double)A->nnz) return fail(QBH_OK);     // ... nor two, when hardly anything is far
    // ---- everything the conversion needs is allocated BEFORE the CSR is touched ----
    KRON_TRY(qbh::launch_max_rowlen(A->d_ia, n, (int64_t *)A->d_scal, s));
    int64_t maxlen = 0;
    KRON_HIP(hipMemcpyAsync(&maxlen, A->d_scal, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    KRON_HIP(hipStreamSynchronize(s));
    const int64_t cw = std::max<int64_t>((int64_t)1 << 26, 4 * maxlen);             // nonzeros per compaction step
    const int64_t n_chunks = (A->nnz + cw - 1) / cw;
    KRON_HIP(hipMalloc(&tmp_v, (size_t)K.far_slots * sizeof(d2)));
    KRON_HIP(hipMalloc(&tmp_c, (size_t)K.far_slots * sizeof(int32_t)));
    if (K.nnz_x > 0) {
        KRON_HIP(hipMalloc(&tmpx_v, (size_t)K.nnz_x * sizeof(d2)));
        KRON_HIP(hipMalloc(&tmpx_c, (size_t)K.nnz_x * sizeof(int32_t)));
    }
    KRON_HIP(hipMalloc(&chunk, (size_t)(cw + maxlen) * sizeof(d2)));
    KRON_HIP(hipMalloc(&d_rb, (size_t)(n_chunks + 1) * sizeof(int32_t)));
    KRON_HIP(hipMalloc(&d_bp, (size_t)(n_chunks + 1) * sizeof(int64_t)));
    const int64_t far_len = multi ? K.map.nfar_rows() + 8 : n;          // one class: the slots of the narrow-band rows take the cross sums
    KRON_HIP(hipMalloc(&K.d_far, (size_t)far_len * sizeof(d2)));
    KRON_HIP(hipMemsetAsync(K.d_far, 0, (size_t)far_len * sizeof(d2), s));
    KRON_TRY(qbh::launch_build_rowblocks(A->d_ia, n, cw, d_rb, d_bp, n_chunks, s));
    std::vector<int32_t> rb((size_t)n_chunks + 1);
    KRON_HIP(hipMemcpyAsync(rb.data(), d_rb, rb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    KRON_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> nb((size_t)n_chunks + 1);            // near entries in front of each step's first row
    for (int64_t c = 0; c <= n_chunks; ++c) KRON_HIP(hipMemcpyAsync(&nb[(size_t)c], K.ia_n + rb[(size_t)c], sizeof(int64_t), hipMemcpyDeviceToHost, s));
    KRON_HIP(hipStreamSynchronize(s));
    // far and cross parts out of the intact CSR (values, then columns in the tiled order of the gathered x)
    KRON_TRY(qbh::launch_kron_far_fill(false, A->d_ia, A->d_ja, A->d_val, K.map, K.ia_f, K.n_groups, nullptr, tmp_v, s));
    KRON_TRY(qbh::launch_kron_far_fill(true, A->d_ia, A->d_ja, A->d_val, K.map, K.ia_f, K.n_groups, tmp_c, nullptr, s));
    if (K.nnz_x > 0) {
        const int64_t nx = multi ? n : K.n_xrows;
        KRON_TRY(qbh::launch_kron_part_gather_vals(2, A->d_ia, A->d_ja, A->d_val, 0, nx, K.map, K.ia_x, K.xrow, tmpx_v, s));
        KRON_TRY(qbh::launch_kron_part_gather_cols(2, A->d_ia, A->d_ja, 0, nx, K.map, K.ia_x, K.xrow, tmpx_c, s));
    }
    KRON_HIP(hipStreamSynchronize(s));
    // near part compacted towards the front of the arrays, step by step through the staging buffer (a step's destination
    // never reaches the source of a later step: near entries in front of a row <= all entries in front of it); the values
    // first -- their classification reads the columns
    destructive = true;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t r0 = rb[(size_t)c], r1 = rb[(size_t)c + 1], cnt = nb[(size_t)c + 1] - nb[(size_t)c];
        if (r1 <= r0 || cnt <= 0) continue;
        KRON_TRY(qbh::launch_kron_part_gather_vals(0, A->d_ia, A->d_ja, A->d_val, r0, r1, K.map, K.ia_n, nullptr, (d2 *)chunk, s));
        KRON_HIP(hipMemcpyAsync(A->d_val + nb[(size_t)c], chunk, (size_t)cnt * sizeof(d2), hipMemcpyDeviceToDevice, s));
    }
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t r0 = rb[(size_t)c], r1 = rb[(size_t)c + 1], cnt = nb[(size_t)c + 1] - nb[(size_t)c];
        if (r1 <= r0 || cnt <= 0) continue;
        KRON_TRY(qbh::launch_kron_part_gather_cols(0, A->d_ia, A->d_ja, r0, r1, K.map, K.ia_n, nullptr, (int32_t *)chunk, s));
        KRON_HIP(hipMemcpyAsync(A->d_ja + nb[(size_t)c], chunk, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    K.ja_n = A->d_ja;
    K.val_n = A->d_val;
    int64_t tail = K.nnz_n;
    // The blocks of the far stream are exact runs of 512 slots: they should start on 128-byte boundaries of BOTH arrays (8 lines
    // per 1 KB value load instead of 9, 2 per 256-byte column load instead of 3), i.e. the far part should begin a multiple of 32
    // entries behind the arrays' (aligned) base.  One class: the small cross part keeps the scratch arrays it was gathered into,
    // which leaves its entries' worth of slack behind the near part for that.
    const bool own_x = !multi && K.nnz_x >= 32 && !qbh::debug_sw().no_far_align;
    if (own_x && ((K.nnz_n + 31) / 32) * 32 + K.far_slots <= A->nnz) tail = ((K.nnz_n + 31) / 32) * 32;
    if (tail + K.far_slots + (own_x ? 0 : K.nnz_x) <= A->nnz) {    // no padding: the far part takes the space the far entries left
        KRON_HIP(hipMemcpyAsync(A->d_val + tail, tmp_v, (size_t)K.far_slots * sizeof(d2), hipMemcpyDeviceToDevice, s));
        KRON_HIP(hipMemcpyAsync(A->d_ja + tail, tmp_c, (size_t)K.far_slots * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        KRON_HIP(hipStreamSynchronize(s));
        (void)hipFree(tmp_v);
        (void)hipFree(tmp_c);
        K.ja_f = A->d_ja + tail;
        K.val_f = A->d_val + tail;
        K.own_far = false;
        tail += K.far_slots;
    } else {                                         // padded groups: the far part keeps its own (larger) arrays
        K.ja_f = tmp_c;
        K.val_f = tmp_v;
        K.own_far = true;
    }
    tmp_v = nullptr;
    tmp_c = nullptr;
    if (K.nnz_x > 0 && own_x) {                      // the cross part stays where it was gathered (a few MB)
        K.ja_x = tmpx_c;
        K.val_x = tmpx_v;
        K.own_x = true;
        tmpx_c = nullptr;
        tmpx_v = nullptr;
    } else if (K.nnz_x > 0) {                        // the cross part behind it (it always fits: its entries came out of these arrays)
        KRON_HIP(hipMemcpyAsync(A->d_val + tail, tmpx_v, (size_t)K.nnz_x * sizeof(d2), hipMemcpyDeviceToDevice, s));
        KRON_HIP(hipMemcpyAsync(A->d_ja + tail, tmpx_c, (size_t)K.nnz_x * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        KRON_HIP(hipStreamSynchronize(s));
        K.ja_x = A->d_ja + tail;
        K.val_x = A->d_val + tail;
        (void)hipFree(tmpx_v);
        (void)hipFree(tmpx_c);
        tmpx_v = nullptr;
        tmpx_c = nullptr;
    }
    KRON_HIP(hipStreamSynchronize(s));
    for (void **q : {&chunk, (void **)&d_rb, (void **)&d_bp}) {
        (void)hipFree(*q);
        *q = nullptr;
    }
    K.inplace = true;
    KRON_TRY(kron_geometry(A));
    KRON_TRY(kron_short_cols(A));
    if (qbh::debug_sw().print_ptrs)
        fprintf(stderr, "qbhip kron arrays: ia %p ja %p val %p | ia_n %p fp %p | ja_f %p val_f %p | wd_n %p wd_f %p | far %p | nnz_n %lld far_slots %lld\n", (void *)A->d_ia,
                (void *)A->d_ja, (void *)A->d_val, (void *)K.ia_n, (void *)K.ia_f, (void *)K.ja_f, (void *)K.val_f, (void *)K.wd_n, (void *)K.wd_f, (void *)K.d_far,
                (long long)K.nnz_n, (long long)K.far_slots);
#undef KRON_HIP
#undef KRON_TRY
    K.active = true;
    return QBH_OK;
}

// the CSR back out of the parts (new arrays, merged row by row: the original rows, bit for bit); the handle is unsplit
// afterwards and stays so.  Needs room for a second copy of the matrix while it runs.
int kron_restore(qbh_csr *A)
{
    if (!A->kron.active) return QBH_OK;
    hipStream_t s = A->stream;
    QBH_HIP(hipStreamSynchronize(s));
    int32_t *nja = nullptr;
    d2 *nval = nullptr;
    if (hipMalloc(&nja, (size_t)A->nnz * sizeof(int32_t)) != hipSuccess || hipMalloc(&nval, (size_t)A->nnz * sizeof(d2)) != hipSuccess) {
        (void)hipGetLastError();
        if (nja) (void)hipFree(nja);
        qbh::set_error("Kronecker split: no room to merge the parts back into a CSR (%.1f GB needed beside the operator)", A->nnz * 20e-9);
        return QBH_ENOMEM;
    }
    int rc = qbh::launch_kron_merge_rows(kron_parts(A), 0, A->nrows, nja, nval, 0, s);
    if (rc == QBH_OK && hipStreamSynchronize(s) != hipSuccess) rc = QBH_EHIP;
    if (rc != QBH_OK) {
        (void)hipFree(nja);
        (void)hipFree(nval);
        return rc;
    }
    if (A->d_ja) (void)hipFree(A->d_ja);
    (void)hipFree(A->d_val);
    A->d_ja = nja;
    A->d_val = nval;
    kron_free_aux(A);
    A->kron_off = true;
    return QBH_OK;
}

// ---- the split for the library's default form of a real operator: dictionary-coded values, packed-double vectors ----
void kronc_release(qbh_csr *A)
{
    qbh_csr::KronCoded &K = A->kronc;
    for (CsrPart *P : {&K.near_p, &K.far_p})
        for (void *q : {(void *)P->d_ia, (void *)P->d_ja, (void *)P->d_code, (void *)P->d_rb, (void *)P->d_bp})
            if (q) (void)hipFree(q);
    if (K.d_xt) (void)hipFree(K.d_xt);
    for (void *q : {(void *)K.tables.tgt_u, (void *)K.tables.val_u, (void *)K.tables.pk_d})
        if (q) (void)hipFree(q);
    for (void *q : {(void *)K.sl.gia_n, (void *)K.sl.gia_f, (void *)K.sl.ja_n, (void *)K.sl.ja_f, (void *)K.sl.code_n, (void *)K.sl.code_f, (void *)K.sl.d_far, (void *)K.sl.d_dictr, (void *)K.sl.tf_ptr, (void *)K.sl.dcode})
        if (q) (void)hipFree(q);
    K = qbh_csr::KronCoded{};
}

// The sliced form of the coded split (qbh_kronc.hip): both parts in groups of 16 rows, near columns relative to the major
// index's block (its x block lives in LDS during the near pass), far columns in the tiled order.  Needs 1-byte codes with a free
// code for the padding, the block of x (S doubles) inside one workgroup's LDS, and room for a second copy of the coded operator.
int kronc_build_sliced(qbh_csr *A, int64_t S, int64_t NU)
{
    const int64_t n = A->nrows;
    hipStream_t s = A->stream;
    if (A->dict_mode != 1 || A->code_w != 1 || A->n_dict > 255 || NU > 65535 || S > 20 * 1024 || qbh::kronc_near_lds_bytes(S) > (size_t)159 * 1024) return QBH_OK;
    qbh_csr::KronCoded &K = A->kronc;
    qbh::KroncSliced &L = K.sl;
    const int nb = (int)((S + 15) / 16);
    const int64_t G = (int64_t)nb * NU;
    int32_t *wn = nullptr, *wf = nullptr;
    auto fail = [&](int code) {
        if (wn) (void)hipFree(wn);
        if (wf) (void)hipFree(wf);
        kronc_release(A);
        return code;
    };
#define KS_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                  \
            return fail(e_ == hipErrorOutOfMemory ? QBH_OK : QBH_EHIP);               \
        }                                                                             \
    } while (0)
#define KS_TRY(expr)                           \
    do {                                       \
        const int rc_ = (expr);                \
        if (rc_ != QBH_OK) return fail(rc_);   \
    } while (0)
    // Is the far part T (x) 1 (the far entries of a row do not depend on its minor index: two-species models)?  Then it is kept as
    // T alone -- NU short rows, always in the L2 -- and the far pass has no stream.  QBH_KRONC_FAR_UNI=0: keep the general form.
    {
        int nonuni = 0;
        if (!(A->opts.kron_uniform & 1)) {
            nonuni = 1;
        } else {
            KS_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
            KS_TRY(qbh::launch_kronc_far_uniform(A->d_ia, A->d_ja, A->d_code, S, n, A->d_flag, s));
            KS_HIP(hipMemcpyAsync(&nonuni, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
            KS_HIP(hipStreamSynchronize(s));
            KS_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        }
        L.far_uni = nonuni == 0;
    }
    // ... and is the near part 1 (x) T' + D (off-diagonal near entries independent of the major index)?  Then T' is kept once -- nb
    // groups, in the L2 -- beside one diagonal code per row, and the near pass has no stream either.  QBH_KRONC_NEAR_UNI=0: general form.
    {
        int nonuni = 0;
        if (!(A->opts.kron_uniform & 2)) {
            nonuni = 1;
        } else {
            KS_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
            KS_TRY(qbh::launch_kronc_near_uniform(A->d_ia, A->d_ja, A->d_code, S, n, A->d_flag, s));
            KS_HIP(hipMemcpyAsync(&nonuni, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
            KS_HIP(hipStreamSynchronize(s));
            KS_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        }
        L.near_uni = nonuni == 0;
    }
    KS_HIP(hipMalloc(&wn, (size_t)G * sizeof(int32_t)));
    KS_HIP(hipMalloc(&wf, (size_t)G * sizeof(int32_t)));
    KS_TRY(qbh::launch_kronc_widths(A->d_ia, A->d_ja, S, NU, nb, wn, wf, s));
    const int64_t Gn = L.near_uni ? nb : G;                  // near groups stored
    if (L.near_uni) KS_TRY(qbh::launch_kronc_s_widths(A->d_ia, A->d_ja, S, nb, wn, s));
    KS_HIP(hipMalloc(&L.gia_n, (size_t)(Gn + 1) * sizeof(int64_t)));
    KS_TRY(qbh::exclusive_scan(wn, Gn, L.gia_n, s));
    if (L.far_uni) {
        KS_TRY(qbh::launch_kronc_t_widths(A->d_ia, A->d_ja, S, NU, wf, s));
        KS_HIP(hipMalloc(&L.tf_ptr, (size_t)(NU + 1) * sizeof(int64_t)));
        KS_TRY(qbh::exclusive_scan(wf, NU, L.tf_ptr, s));
        KS_HIP(hipMemcpy(&L.slots_f, L.tf_ptr + NU, sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {
        KS_HIP(hipMalloc(&L.gia_f, (size_t)(G + 1) * sizeof(int64_t)));
        KS_TRY(qbh::exclusive_scan(wf, G, L.gia_f, s));
        KS_HIP(hipMemcpy(&L.slots_f, L.gia_f + G, sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    (void)hipFree(wn);
    wn = nullptr;
    (void)hipFree(wf);
    wf = nullptr;
    KS_HIP(hipMemcpy(&L.slots_n, L.gia_n + Gn, sizeof(int64_t), hipMemcpyDeviceToHost));
    if (L.slots_f == 0 || L.slots_n + L.slots_f > 2 * A->nnz + 64 * G) return fail(QBH_OK);          // nothing far, or rows too ragged to pad
    {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)L.slots_n * 3 + (size_t)L.slots_f * 3 + (size_t)G * 16 * 8 + (size_t)n * 8 + ((size_t)1 << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need) return fail(QBH_OK);
    }
    constexpr size_t kPad = 1024;            // the passes read up to 8 x 64 slots past a group's end without clamping (qbh_kronc.hip)
    KS_HIP(hipMalloc(&L.ja_n, ((size_t)L.slots_n + kPad) * sizeof(uint16_t)));
    KS_HIP(hipMalloc(&L.code_n, (size_t)L.slots_n + kPad));
    KS_HIP(hipMalloc(&L.ja_f, ((size_t)L.slots_f + kPad) * sizeof(uint16_t)));
    KS_HIP(hipMalloc(&L.code_f, (size_t)L.slots_f + kPad));
    KS_HIP(hipMemsetAsync(L.ja_n + L.slots_n, 0, kPad * sizeof(uint16_t), s));
    KS_HIP(hipMemsetAsync(L.code_n + L.slots_n, 0, kPad, s));
    KS_HIP(hipMemsetAsync(L.ja_f + L.slots_f, 0, kPad * sizeof(uint16_t), s));
    KS_HIP(hipMemsetAsync(L.code_f + L.slots_f, 0, kPad, s));
    KS_HIP(hipMalloc(&L.d_far, (size_t)G * 16 * sizeof(double)));
    {
        std::vector<qbh::d2> hd((size_t)A->n_dict);
        std::vector<double> hr(256, 0.0);
        KS_HIP(hipMemcpy(hd.data(), A->d_dict, hd.size() * sizeof(qbh::d2), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hd.size(); ++i) hr[i] = hd[i].x;
        KS_HIP(hipMalloc(&L.d_dictr, 256 * sizeof(double)));
        KS_HIP(hipMemcpy(L.d_dictr, hr.data(), 256 * sizeof(double), hipMemcpyHostToDevice));
    }
    // 16 doubles of zeroed slack: the far pass gathers whole 16-wide lines even in the narrow last band (S % 16 != 0), whose
    // last line would otherwise end past the allocation
    KS_HIP(hipMalloc(&K.d_xt, (size_t)(n + 16) * sizeof(double)));
    KS_HIP(hipMemsetAsync(K.d_xt + n, 0, 16 * sizeof(double), s));
    if (!(L.near_uni && L.far_uni))
        KS_TRY(qbh::launch_kronc_fill(A->d_ia, A->d_ja, A->d_code, S, NU, nb, A->n_dict, L.near_uni ? nullptr : L.gia_n, L.ja_n, L.code_n, L.gia_f, L.ja_f,
                                      L.code_f, s));
    if (L.near_uni) {
        KS_TRY(qbh::launch_kronc_s_fill(A->d_ia, A->d_ja, A->d_code, S, nb, A->n_dict, L.gia_n, L.ja_n, L.code_n, s));
        KS_HIP(hipMalloc(&L.dcode, (size_t)n));
        KS_TRY(qbh::launch_kronc_dcode(A->d_ia, A->d_ja, A->d_code, n, A->n_dict, L.dcode, s));
    }
    if (L.far_uni) KS_TRY(qbh::launch_kronc_t_fill(A->d_ia, A->d_ja, A->d_code, S, NU, A->n_dict, L.tf_ptr, L.ja_f, L.code_f, s));
    if (!A->d_wctr) KS_HIP(qbh::dev_alloc(&A->d_wctr, qbh::kWctrRegions * 128 * sizeof(unsigned long long)));
    KS_HIP(hipStreamSynchronize(s));
#undef KS_HIP
#undef KS_TRY
    L.S = S;
    L.NU = NU;
    L.nb = nb;
    L.active = true;
    K.t = qbh::KronTile{S, NU, 16};
    K.active = true;
    if (L.near_uni && L.far_uni && (A->opts.kron_uniform & 4)) (void)kronc_table_route(A);      // best effort: the sliced passes stay otherwise
    return QBH_OK;
}

// An operator RECOGNISED as T (x) 1 + 1 (x) T' + D needs no stored matrix at all: T and T' are the two hop tables of the
// row-staged table kernel (k_mf_hubbard_row, DESIGN-history 4.6: one workgroup holds the row X[u][:] in LDS, T' gathers from LDS, T is
// ~17 coalesced row AXPYs), D one value code per row.  That kernel reads x once more than it must and writes y once -- three
// vector passes against the seven of the sliced split (tiled copy written + read, far sums written + read) -- and is the faster
// of the two wherever both apply (C3: 4.65 against 6.9 ms per SpMV, DESIGN-history 5.0d).  The tables are decoded on the host from the
// sliced structures (lane-major layouts of k_kronc_t_fill / k_kronc_s_fill) -- a few hundred KB -- and re-packed as ELL / packed
// words.  Not taken when the amplitudes do not fit the kernel's 15 codes, the rows its widths, or S its 24-bit targets.
int kronc_table_route(qbh_csr *A)
{
    qbh_csr::KronCoded &K = A->kronc;
    qbh::KroncSliced &L = K.sl;
    const int64_t S = L.S, NU = L.NU;
    if (!L.active || !L.near_uni || !L.far_uni || !L.dcode || S < 256 || S >= (1 << 24) || NU >= (1 << 24)) return QBH_OK;
    hipStream_t s = A->stream;
    std::vector<int64_t> tp((size_t)NU + 1), gs((size_t)L.nb + 1);
    std::vector<uint16_t> tcol((size_t)L.slots_f), scol((size_t)L.slots_n);
    std::vector<uint8_t> tcode((size_t)L.slots_f), scode((size_t)L.slots_n);
    std::vector<double> dictr(256);
    QBH_HIP(hipStreamSynchronize(s));
    QBH_HIP(hipMemcpy(tp.data(), L.tf_ptr, tp.size() * 8, hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(gs.data(), L.gia_n, gs.size() * 8, hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(tcol.data(), L.ja_f, tcol.size() * 2, hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(tcode.data(), L.code_f, tcode.size(), hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(scol.data(), L.ja_n, scol.size() * 2, hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(scode.data(), L.code_n, scode.size(), hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(dictr.data(), L.d_dictr, 256 * 8, hipMemcpyDeviceToHost));
    // amplitude codes of the table kernel: code 0 = 0.0, at most 15 others
    std::vector<double> amp(1, 0.0);
    auto amp_code = [&](double v) -> int {
        for (size_t c = 0; c < amp.size(); ++c)
            if (amp[c] == v) return (int)c;
        if (amp.size() == 16) return -1;
        amp.push_back(v);
        return (int)amp.size() - 1;
    };
    // T: entries of major index u at tp[u] + (k & 3) * nu + (k >> 2), nu = width / 4 (k_kronc_t_fill); zero-valued slots are padding
    std::vector<std::vector<std::pair<uint32_t, uint8_t>>> tu((size_t)NU), td((size_t)S);
    int wu = 0, wd = 0;
    for (int64_t u = 0; u < NU; ++u) {
        const int64_t base = tp[(size_t)u], w = tp[(size_t)u + 1] - base, nu = w >> 2;
        for (int64_t k = 0; k < w; ++k) {
            const int64_t at = base + (k & 3) * nu + (k >> 2);
            const double v = dictr[tcode[(size_t)at]];
            if (v == 0.0) continue;
            const int c = amp_code(v);
            if (c < 0) return QBH_OK;
            tu[(size_t)u].push_back({(uint32_t)tcol[(size_t)at], (uint8_t)c});
        }
        wu = std::max(wu, (int)tu[(size_t)u].size());
    }
    // T': group b holds the minor indices 16 b .. 16 b + 15; entry k of lane j at gs[b] + (16 (k & 3) + j) * (w / 4) + (k >> 2) (k_kronc_s_fill)
    for (int64_t b = 0; b < L.nb; ++b) {
        const int64_t base = gs[(size_t)b], w = (gs[(size_t)b + 1] - base) >> 4;
        for (int64_t j = 0; j < 16 && b * 16 + j < S; ++j)
            for (int64_t k = 0; k < w; ++k) {
                const int64_t at = base + (16 * (k & 3) + j) * (w >> 2) + (k >> 2);
                const double v = dictr[scode[(size_t)at]];
                if (v == 0.0) continue;
                const int c = amp_code(v);
                if (c < 0) return QBH_OK;
                td[(size_t)(b * 16 + j)].push_back({(uint32_t)scol[(size_t)at], (uint8_t)c});
            }
    }
    for (int64_t d = 0; d < S; ++d) wd = std::max(wd, (int)td[(size_t)d].size());
    wu = std::max(8, ((wu + 7) / 8) * 8);
    wd = std::max(8, ((wd + 7) / 8) * 8);
    if (wu > 64) return QBH_OK;                                   // kMfMaxUp of the row-staged kernel
    qbh::MfHubbard m;
    m.Nu = NU;
    m.Nd = S;
    m.wu = wu;
    m.wd = wd;
    m.U = 0.0;
    for (size_t c = 0; c < amp.size(); ++c) m.amp[c] = amp[c];
    std::vector<uint32_t> tgt((size_t)wu * NU), pk((size_t)wd * S);
    std::vector<uint8_t> val((size_t)wu * NU, 0);
    for (int k = 0; k < wu; ++k)
        for (int64_t u = 0; u < NU; ++u) tgt[(size_t)k * NU + u] = (uint32_t)u;                 // padding: (u itself, amplitude 0)
    for (int64_t u = 0; u < NU; ++u)
        for (size_t k = 0; k < tu[(size_t)u].size(); ++k) {
            tgt[k * (size_t)NU + (size_t)u] = tu[(size_t)u][k].first;
            val[k * (size_t)NU + (size_t)u] = tu[(size_t)u][k].second;
        }
    for (int64_t d = 0; d < S; ++d)
        for (int k = 0; k < wd; ++k) pk[((size_t)(k / 4) * S + d) * 4 + (k & 3)] = (uint32_t)d;   // padding: self, code 0
    for (int64_t d = 0; d < S; ++d)
        for (size_t k = 0; k < td[(size_t)d].size(); ++k)
            pk[((size_t)(k / 4) * S + d) * 4 + (k & 3)] = td[(size_t)d][k].first | ((uint32_t)td[(size_t)d][k].second << 24);
    auto drop = [&]() {
        for (void *q : {(void *)m.tgt_u, (void *)m.val_u, (void *)m.pk_d})
            if (q) (void)hipFree(q);
        (void)hipGetLastError();
        return QBH_OK;
    };
    if (hipMalloc(&m.tgt_u, tgt.size() * 4) != hipSuccess || hipMalloc(&m.val_u, val.size()) != hipSuccess || hipMalloc(&m.pk_d, pk.size() * 4) != hipSuccess)
        return drop();
    if (hipMemcpy(m.tgt_u, tgt.data(), tgt.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m.val_u, val.data(), val.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m.pk_d, pk.data(), pk.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return drop();
    K.tables = m;
    K.table_route = true;
    return QBH_OK;
}

// Same decomposition as kron_build, for the row kernel: the near part keeps rows and columns, the far part has rows AND
// columns in the tiled order of KronTile with B = 16 (one 128-byte line of doubles per major index and band); the far
// launch gathers from the tiled copy of the packed x and accumulates onto the near launch's result at orig(row).
// Measured slower than the unsplit operator (DESIGN-history 5.0b item 10); QBH_KRON_CODED=1 builds it for comparison.
int kronc_build(qbh_csr *A)
{
    kronc_release(A);
    // kron_split as for the complex128 form: 1 splits operators of 1e8 nonzeros and more, 2 whatever has the structure -- into the
    // sliced form (kronc_build_sliced) when its preconditions hold, else not at all.  QBH_KRON_CODED = 0 / 1 / 2 overrides
    // (1: the earlier form for the row kernel, measured slower than the unsplit operator; kept for comparison).
    int want = (A->opts.kron_split == 2 || (A->opts.kron_split == 1 && A->nnz >= 100000000)) ? 2 : 0;
    if (A->opts.kron_coded >= 0) want = A->opts.kron_coded;
    if (!want || A->opts.kron_split == 0 || !A->opts.real_fast_path) return QBH_OK;       // only the all-real operation runs it
    if (A->kernel != QBH_KERNEL_ROWS || A->d_code == nullptr || !A->values_real || A->kind != 0 || A->has_rem || A->nrows != A->ncols ||
        A->row_offset != 0 || A->nnz <= 0)
        return QBH_OK;
    const int64_t S = A->opts.kron_minor;
    if (S <= 1 || S >= A->nrows || A->nrows % S != 0) return QBH_OK;
    const int64_t NU = A->nrows / S, n = A->nrows;
    hipStream_t s = A->stream;
    {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)A->nnz * (4 + A->code_w) + (size_t)n * (8 + 16 + 16) + ((size_t)2 << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need) return QBH_OK;
    }
    QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
    QBH_TRY(qbh::launch_kron_check(A->d_ia, A->d_ja, n, S, A->d_flag, s));
    int bad = 0;
    QBH_HIP(hipMemcpyAsync(&bad, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
    if (bad) return QBH_OK;
    if (want == 2) return kronc_build_sliced(A, S, NU);
    qbh_csr::KronCoded &K = A->kronc;
    int B = 16;
    while (B > 2 && (double)NU * B * 8 > 2.5e6) B >>= 1;
    K.t = qbh::KronTile{S, NU, B};
    int32_t *cn = nullptr, *cf = nullptr;
    auto fail = [&](int code) {
        if (cn) (void)hipFree(cn);
        if (cf) (void)hipFree(cf);
        kronc_release(A);
        return code;
    };
#define KC_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                  \
            return fail(e_ == hipErrorOutOfMemory ? QBH_OK : QBH_EHIP);               \
        }                                                                             \
    } while (0)
#define KC_TRY(expr)                           \
    do {                                       \
        const int rc_ = (expr);                \
        if (rc_ != QBH_OK) return fail(rc_);   \
    } while (0)
    KC_HIP(hipMalloc(&cn, (size_t)n * sizeof(int32_t)));
    KC_HIP(hipMalloc(&cf, (size_t)n * sizeof(int32_t)));
    KC_TRY(qbh::launch_kron_count(A->d_ia, A->d_ja, n, K.t, cn, cf, s));
    KC_HIP(hipMalloc(&K.near_p.d_ia, (size_t)(n + 1) * sizeof(int64_t)));
    KC_HIP(hipMalloc(&K.far_p.d_ia, (size_t)(n + 1) * sizeof(int64_t)));
    KC_TRY(qbh::exclusive_scan(cn, n, K.near_p.d_ia, s));
    KC_TRY(qbh::exclusive_scan(cf, n, K.far_p.d_ia, s));
    (void)hipFree(cn);
    cn = nullptr;
    (void)hipFree(cf);
    cf = nullptr;
    KC_HIP(hipMemcpy(&K.near_p.nnz, K.near_p.d_ia + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    KC_HIP(hipMemcpy(&K.far_p.nnz, K.far_p.d_ia + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    if (K.far_p.nnz == 0 || K.near_p.nnz + K.far_p.nnz != A->nnz) return fail(QBH_OK);
    for (CsrPart *P : {&K.near_p, &K.far_p}) {
        KC_HIP(hipMalloc(&P->d_ja, std::max<size_t>((size_t)P->nnz, 1) * sizeof(int32_t)));
        KC_HIP(hipMalloc(&P->d_code, (size_t)P->nnz * A->code_w + 16));
        KC_HIP(hipMemsetAsync(P->d_code + (size_t)P->nnz * A->code_w, 0, 16, s));
    }
    KC_TRY(qbh::launch_kron_fill_codes(A->d_ia, A->d_ja, A->d_code, A->code_w, n, K.t, K.near_p.d_ia, K.near_p.d_ja, K.near_p.d_code,
                                       K.far_p.d_ia, K.far_p.d_ja, K.far_p.d_code, s));
    KC_HIP(hipMalloc(&K.d_xt, (size_t)n * sizeof(double)));
    for (CsrPart *P : {&K.near_p, &K.far_p})
        KC_TRY(setup_geometry(A, P->d_ia, P->nnz, A->dict_mode, &P->npb, &P->tpr, &P->unroll, &P->window, &P->n_blocks, &P->d_rb, &P->d_bp, &P->grid));
    KC_HIP(hipStreamSynchronize(s));
#undef KC_HIP
#undef KC_TRY
    K.active = true;
    return QBH_OK;
}

}  // namespace qbhapi
