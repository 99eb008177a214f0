// qbh_api.cpp -- the extern "C" surface of libqbhip.so: operator lifetime, the host-vector
// seam (MultMv / MultMv2), device building blocks, and the device-resident Lanczos and CG
// drivers.  Each entry point names the reference function it replaces in include/qbhip.h.
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
#include <vector>

#include "qbh_internal.hpp"

using qbh::d2;

// ------------------------------------------------------------------ errors -----
namespace qbh {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// QBH_DEBUG="key=value,key=value" (bare integer: flags); parsed again only when the variable's text has changed (a test
// harness may set it between two creations)
static void parse_debug(const char *e, DebugSw &d)
{
    d = DebugSw{};
    if (!e || !*e) return;
    struct Key { const char *name; int *i; long long *ll; };
    const Key keys[] = {{"flags", &d.flags, nullptr}, {"colmask", &d.colmask, nullptr}, {"tpr", &d.tpr, nullptr}, {"unroll", &d.unroll, nullptr},
                        {"grid", &d.grid, nullptr}, {"wave_tpr", &d.wave_tpr, nullptr}, {"chunk_mult", &d.chunk_mult, nullptr},
                        {"trace_create", &d.trace_create, nullptr}, {"trace_tune", &d.trace_tune, nullptr}, {"trace_dict", &d.trace_dict, nullptr},
                        {"print_ptrs", &d.print_ptrs, nullptr}, {"sec_walk", &d.sec_walk, nullptr}, {"sec_grid", &d.sec_grid, nullptr},
                        {"sec_unroll", &d.sec_unroll, nullptr}, {"wave_pipelined", &d.wave_pipelined, nullptr}, {"create_chunk", nullptr, &d.create_chunk},
                        {"force_ragged", &d.force_ragged, nullptr}, {"mf_row", &d.mf_row, nullptr}, {"mf_chunk", &d.mf_chunk, nullptr},
                        {"mf_window", &d.mf_window, nullptr}, {"kronc_abl", &d.kronc_abl, nullptr}, {"kronc_far_chunk", &d.kronc_far_chunk, nullptr},
                        {"kronc_far_ng", &d.kronc_far_ng, nullptr}, {"kronc_far_nt", &d.kronc_far_nt, nullptr}, {"no_far_align", &d.no_far_align, nullptr},
                        {"no_defer", &d.no_defer, nullptr}};
    const std::string all(e);
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string item = all.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) {
            if (item.find_first_not_of("0123456789") == std::string::npos) d.flags = atoi(item.c_str());
            else fprintf(stderr, "qbhip: QBH_DEBUG: '%s' is not key=value\n", item.c_str());
            continue;
        }
        const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
        bool known = false;
        for (const Key &key : keys)
            if (k == key.name) {
                if (key.i) *key.i = atoi(v.c_str());
                else *key.ll = atoll(v.c_str());
                known = true;
            }
        if (!known) fprintf(stderr, "qbhip: QBH_DEBUG: unknown key '%s'\n", k.c_str());
    }
}

const DebugSw &debug_sw()
{
    static std::mutex mu;
    static std::string seen = "\x01";           // never equal to a real value: the first call parses
    static DebugSw sw[2];                        // the previous one stays valid for a caller that still holds a reference to it
    static int cur = 0;
    const char *e = getenv("QBH_DEBUG");
    std::lock_guard<std::mutex> lock(mu);
    if (seen != (e ? e : "")) {
        seen = e ? e : "";
        cur ^= 1;
        parse_debug(e, sw[cur]);
    }
    return sw[cur];
}
}  // namespace qbh

extern "C" const char *qbh_last_error(void) { return qbh::g_err; }

extern "C" const char *qbh_strerror(int code)
{
    switch (code) {
    case QBH_OK:        return "success";
    case QBH_EINVAL:    return "invalid argument";
    case QBH_ENODEVICE: return "no HIP device available (libqbhip has no CPU fallback)";
    case QBH_EHIP:      return "HIP runtime call failed";
    case QBH_ENOMEM:    return "out of memory";
    case QBH_ENOTHERM:  return "matrix is not Hermitian";
    case QBH_ECOMM:     return "communicator hook failed";
    case QBH_ENOTNORM:  return "Lanczos start vector is not normalised";
    case QBH_ENOCONV:   return "tridiagonal eigen-solver did not converge";
    case QBH_EUNSUPP:   return "not supported";
    default:            return "unknown error";
    }
}

extern "C" int qbh_version(void) { return QBH_VERSION; }

extern "C" int qbh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

namespace {
std::mutex g_defaults_mu;
bool g_have_defaults = false;
qbh_opts g_defaults;
}  // namespace

// process-wide defaults: what qbh_opts_default returns and what a NULL `opts` argument means from now on (NULL: back to the
// built-in ones).  For a host whose constructor call cannot carry options (the reference's csr_mat(lil_mat&))
extern "C" void qbh_opts_set_default(const qbh_opts *o)
{
    std::lock_guard<std::mutex> lock(g_defaults_mu);
    g_have_defaults = o != nullptr;
    if (o) {
        g_defaults = *o;
        g_defaults.device = -1;          // a default names no device and no stream: those belong to one call
        g_defaults.stream = nullptr;
    }
}

extern "C" void qbh_opts_default(qbh_opts *o)
{
    if (!o) return;
    {
        std::lock_guard<std::mutex> lock(g_defaults_mu);
        if (g_have_defaults) {
            *o = g_defaults;
            return;
        }
    }
    qbh::opts_builtin(o);
}

// the built-in defaults, whatever qbh_opts_set_default says: what a NULL `opts` means for operators the library generates or
// adopts itself (a host's hint about the basis of ITS arrays must not reach them)
void qbh::opts_builtin(qbh_opts *o)
{
    o->device = -1;
    o->stream = nullptr;
    o->spmv_kernel = QBH_KERNEL_AUTO;
    o->nnz_per_block = 0;
    o->xcd_swizzle = 2;
    o->value_dict = 1;      // lossless; falls back to plain storage by itself
    o->profile = 0;
    o->check_hermitian = 1;
    o->real_fast_path = 1;
    o->kron_split = 1;
    o->kron_cols16 = 1;
    o->kron_sliced = 1;
    o->kron_band = 0;
    o->kron_cross_in_near = 1;
    o->kron_coded = -1;
    o->kron_uniform = 3;
    o->gather_parts = 0;
    o->wave_walk = -1;
    o->tile_fold = 1;
    o->autotune = 1;
    o->shard_split = 1;
    o->real_forms = 7;
    o->basis_detect = 1;
    o->kron_minor = 0;
    o->deterministic = 0;
    o->basis_kind = QBH_BASIS_NONE;
    o->n_sites = o->n_up = o->n_dn = 0;
}

// ------------------------------------------------------------ operator ---------
namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int require_device(const qbh_opts *opts, int *dev_out)
{
    int n = qbh_device_count();
    if (n <= 0) {
        qbh::set_error("no HIP device visible; libqbhip has no CPU fallback");
        return QBH_ENODEVICE;
    }
    int dev = -1;
    if (opts && opts->device >= 0) {
        if (opts->device >= n) {
            qbh::set_error("device %d requested but only %d visible", opts->device, n);
            return QBH_EINVAL;
        }
        dev = opts->device;
        QBH_HIP(hipSetDevice(dev));
    } else {
        QBH_HIP(hipGetDevice(&dev));
    }
    *dev_out = dev;
    return QBH_OK;
}

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

int try_value_dict(qbh_csr *A);

// row-block geometry of one part (ia of length nrows+1, nnz entries)
int setup_geometry(qbh_csr *A, const int64_t *d_ia, int64_t nnz, int dict_mode, int *npb_o, int *tpr_o, int *unroll_o,
                   int64_t *window_o, int64_t *n_blocks_o, int32_t **d_rb_o, int64_t **d_bp_o, int *grid_o)
{
    hipStream_t s = A->stream;
    const qbh_opts &o = A->opts;
    // longest row -> nnz window of a row block
    QBH_TRY(qbh::launch_max_rowlen(d_ia, A->nrows, (int64_t *)A->d_scal, s));
    int64_t maxlen = 0;
    QBH_HIP(hipMemcpyAsync(&maxlen, A->d_scal, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    const double avg = A->nrows > 0 ? (double)nnz / (double)A->nrows : 0.0;
    int npb, tpr, unroll = 4;
    const bool coded = dict_mode != 0;
    if (A->kernel == QBH_KERNEL_ROWS) {
        // measured on C3: coded 8192/P=1/8 gathers in flight is HBM-bound on its real
        // traffic; the plain kernel stages 16-byte values and is limited to 2048 by LDS occupancy.
        // two-byte codes: 4096 keeps three workgroups per CU next to the 16 KB LDS dictionary
        npb = dict_mode == 1 ? 8192 : dict_mode >= 2 ? 4096 : 2048;
        const double cap_rows = 0.75 * qbh::kRowCap * (avg > 1.0 ? avg : 1.0);   // keep rows/block under kRowCap
        while (npb > 1024 && (double)npb > cap_rows) npb >>= 1;
        if (o.nnz_per_block > 0) npb = o.nnz_per_block;
        tpr = avg <= 64 ? 1 : avg <= 128 ? 2 : avg <= 256 ? 4 : 8;
        if (!coded && avg > 12) tpr = avg <= 96 ? 4 : 8;
        unroll = tpr == 1 ? 8 : 4;
    } else {
        npb = o.nnz_per_block > 0 ? o.nnz_per_block : 2048;
        if (A->kernel == QBH_KERNEL_VECTOR) tpr = avg <= 24 ? 2 : avg <= 96 ? 4 : avg <= 192 ? 8 : avg <= 512 ? 16 : avg <= 2048 ? 32 : 64;
        else                                tpr = avg <= 3 ? 1 : avg <= 8 ? 2 : avg <= 48 ? 4 : avg <= 128 ? 8 : 16;
    }
    if (npb != 1024 && npb != 2048 && npb != 4096 && !(npb == 8192 && A->kernel == QBH_KERNEL_ROWS && dict_mode == 1)) {
        qbh::set_error("nnz_per_block must be 1024, 2048 or 4096 (8192: row kernel with value dictionary only)");
        return QBH_EINVAL;
    }
    if (qbh::debug_sw().tpr) tpr = qbh::debug_sw().tpr;              // tuning experiments
    if (qbh::debug_sw().unroll) unroll = qbh::debug_sw().unroll;
    // a block holds the rows that START inside its window, so it can exceed the window by
    // one row; keep window + maxlen - 1 <= npb when rows are short, otherwise let the
    // oversized-block path take the few long rows.
    int64_t window = (maxlen <= npb / 2) ? npb - (maxlen > 0 ? maxlen - 1 : 0) : npb / 2;
    int64_t n_blocks = std::max<int64_t>(1, (nnz + window - 1) / window);
    QBH_HIP(qbh::dev_alloc(d_rb_o, (size_t)(n_blocks + 1) * sizeof(int32_t)));
    QBH_HIP(qbh::dev_alloc(d_bp_o, (size_t)(n_blocks + 1) * sizeof(int64_t)));
    QBH_TRY(qbh::launch_build_rowblocks(d_ia, A->nrows, window, *d_rb_o, *d_bp_o, n_blocks, s));
    *npb_o = npb;
    *tpr_o = tpr;
    *unroll_o = unroll;
    *window_o = window;
    *n_blocks_o = n_blocks;
    *grid_o = qbh::spmv_grid(A->kernel, n_blocks, A->nrows, tpr);
    if (A->kernel == QBH_KERNEL_ROWS) {
        // persistent launch: exactly the workgroups that are resident at once (measured on C3: 768 = 3 per
        // CU runs 8 % faster than an oversubscribed 4096 because the chunked XCD walk then keeps every
        // XCD on ONE contiguous chunk of row blocks)
        const int occ = qbh::rows_kernel_occupancy(npb, tpr, unroll, dict_mode);
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        if (occ > 0) {
            int64_t g = (int64_t)occ * ncu;
            g = std::min<int64_t>(g, ((n_blocks + 7) / 8) * 8);
            *grid_o = (int)std::max<int64_t>(8, (g / 8) * 8);
        }
    }
    if (A->kernel == QBH_KERNEL_VECTOR) {
        // persistent launch as well: resident workgroups only, so that the chunked XCD walk applies
        const int occ = qbh::vector_kernel_occupancy(tpr, unroll, coded);
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        if (occ > 0) {
            const int rpb = qbh::kBlock / tpr;
            const int64_t units = (A->nrows + rpb - 1) / rpb;
            int64_t g = std::min<int64_t>((int64_t)occ * ncu, ((units + 7) / 8) * 8);
            *grid_o = (int)std::max<int64_t>(8, (g / 8) * 8);
        }
    }
    if (qbh::debug_sw().grid >= 8) *grid_o = (qbh::debug_sw().grid / 8) * 8;      // tuning experiments
    return QBH_OK;
}

// A genuine row shard (nrows < ncols) is split into locally-owned columns and remote columns so the
// local part can run while the all-gather of x is in flight.
int split_shard(qbh_csr *A)
{
    if (A->nrows == A->ncols || A->nnz == 0) return QBH_OK;
    if (!A->opts.shard_split) return QBH_OK;
    hipStream_t s = A->stream;
    {   // the split holds a second copy of the shard until the original is released: skip it (one launch per SpMV, no
        // overlap with the gather) rather than fail when HBM cannot hold both.  The decision is per rank (ranks whose
        // shard does not fit keep the single launch); both forms take part in the same collectives, so ranks may differ
        size_t free_b = 0, total_b = 0;
        const size_t per_nnz = 4 + (A->d_code ? (size_t)A->code_w : sizeof(d2));
        const size_t need = (size_t)A->nnz * per_nnz + (size_t)A->nrows * 20 + ((size_t)1 << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need) return QBH_OK;
    }
    const int32_t lo = (int32_t)A->row_offset, hi = (int32_t)(A->row_offset + A->nrows);
    int32_t *cnt = nullptr;
    int64_t *ia0 = nullptr, *ia1 = nullptr;
    int32_t *ja0 = nullptr, *ja1 = nullptr;
    d2 *v0 = nullptr, *v1 = nullptr;
    uint8_t *c0 = nullptr, *c1 = nullptr;
    auto drop = [&](int code) {                 // nothing of a failed split survives (the operator keeps its one part)
        for (void *q : {(void *)cnt, (void *)ia0, (void *)ia1, (void *)ja0, (void *)ja1, (void *)v0, (void *)v1, (void *)c0, (void *)c1})
            if (q) (void)hipFree(q);
        return code;
    };
#define SPLIT_HIP(call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            qbh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);       \
            (void)hipGetLastError();                                                                         \
            return drop(e_ == hipErrorOutOfMemory ? QBH_ENOMEM : QBH_EHIP);                                  \
        }                                                                                                    \
    } while (0)
#define SPLIT_TRY(expr)                        \
    do {                                       \
        const int rc_ = (expr);                \
        if (rc_ != QBH_OK) return drop(rc_);   \
    } while (0)
    SPLIT_HIP(qbh::dev_alloc(&cnt, (size_t)A->nrows * sizeof(int32_t)));
    SPLIT_HIP(qbh::dev_alloc(&ia0, (size_t)(A->nrows + 1) * sizeof(int64_t)));
    SPLIT_HIP(qbh::dev_alloc(&ia1, (size_t)(A->nrows + 1) * sizeof(int64_t)));
    SPLIT_TRY(qbh::launch_split_count(A->d_ia, A->d_ja, A->nrows, lo, hi, cnt, s));
    SPLIT_TRY(qbh::exclusive_scan(cnt, A->nrows, ia0, s));
    (void)hipFree(cnt);
    cnt = nullptr;
    int64_t nnz0 = 0;
    SPLIT_HIP(hipMemcpy(&nnz0, ia0 + A->nrows, sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t nnz1 = A->nnz - nnz0;
    if (nnz1 == 0) return drop(QBH_OK);        // nothing remote (block-diagonal shard): keep one part
    const bool coded = A->d_code != nullptr;
    SPLIT_HIP(qbh::dev_alloc(&ja0, std::max<size_t>((size_t)nnz0, 1) * sizeof(int32_t)));
    SPLIT_HIP(qbh::dev_alloc(&ja1, (size_t)nnz1 * sizeof(int32_t)));
    if (coded) {
        SPLIT_HIP(qbh::dev_alloc(&c0, (size_t)nnz0 * A->code_w + 16));
        SPLIT_HIP(qbh::dev_alloc(&c1, (size_t)nnz1 * A->code_w + 16));
        SPLIT_HIP(hipMemsetAsync(c0 + (size_t)nnz0 * A->code_w, 0, 16, s));
        SPLIT_HIP(hipMemsetAsync(c1 + (size_t)nnz1 * A->code_w, 0, 16, s));
    } else {
        SPLIT_HIP(qbh::dev_alloc(&v0, std::max<size_t>((size_t)nnz0, 1) * sizeof(d2)));
        SPLIT_HIP(qbh::dev_alloc(&v1, (size_t)nnz1 * sizeof(d2)));
    }
    SPLIT_TRY(qbh::launch_split_fill(A->d_ia, A->d_ja, A->d_val, A->d_code, A->nrows, lo, hi, ia0, ja0, v0, c0, ia1, ja1, v1, c1,
                                      A->code_w, s));
    SPLIT_HIP(hipStreamSynchronize(s));
#undef SPLIT_HIP
#undef SPLIT_TRY
    if (A->own_arrays) {
        (void)hipFree(A->d_ia);
        (void)hipFree(A->d_ja);
        if (A->d_val) (void)hipFree(A->d_val);
    }
    if (A->d_code) (void)hipFree(A->d_code);         // the code array is always library-owned
    A->own_arrays = true;
    A->d_ia = ia0;
    A->d_ja = ja0;
    A->d_val = v0;
    A->d_code = c0;
    A->nnz = nnz0;
    A->rem.d_ia = ia1;
    A->rem.d_ja = ja1;
    A->rem.d_val = v1;
    A->rem.d_code = c1;
    A->rem.nnz = nnz1;
    A->has_rem = true;
    return QBH_OK;
}


// ---------------------------------------------------------------------------------- Kronecker split ----
}  // namespace
namespace qbh {
hipError_t device_alloc(void **p, size_t bytes) { return hipMalloc(p, bytes); }
}  // namespace qbh
namespace {
// everything of the split except the matrix arrays themselves (those are the handle's own d_ja / d_val, re-ordered in place)
void kron_free_aux(qbh_csr *A)
{
    qbh_csr::KronSplit &K = A->kron;
    for (void *q : {(void *)K.ia_n, (void *)K.ia_f, (void *)K.wd_n, (void *)K.wd_f, (void *)K.d_xt, (void *)K.d_far, (void *)K.ia_x, (void *)K.xrow,
                    (void *)K.wd_x, (void *)K.d_cls, (void *)K.c16_n, (void *)K.c16_f})
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
                      int *tpr_o, int *grid_o, int64_t shift = 0)
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
    // 2-byte columns are relative to a base the block's descriptor names: fresh descriptors get it again
    if (K.c16_n) QBH_TRY(qbh::launch_kron_desc_c16(K.wd_n, K.nwb_n, K.t.S, false, false, A->stream));
    if (K.c16_f) QBH_TRY(qbh::launch_kron_desc_c16(K.wd_f, K.nwb_f, K.t.NU, true, false, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}

// 2-byte columns for the parts of a one-class split (qbh_opts.kron_cols16).  The two passes are bound by the rate of line
// requests, not by bytes: the column stream is 16 of a block's ~150 lines as int32 and 8 as uint16.
//   near part: column - (first column of the shard + pad * S), pad = major index of the block's first row (a block of whole rows
//              with <= 512 entries reaches into the next major index at most: values < 2 S);
//   far part (sliced, whole operator): target major index + (band - band of the block's first group) * NU; the element of the
//              tiled x is band0 * 8 NU + 8 * value + slot % 8.
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
            rc = far ? qbh::launch_kron_c16_far(K.wd_f, K.ja_f, K.far_slots, NU, c, A->d_flag, s)
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
    // far: the sliced layout over the tiled order of the WHOLE vector (a shard under a communicator gathers rank-major blocks)
    if (K.sliced && !K.own_far && K.t.B == 8 && A->nrows == A->ncols && A->row_offset == 0 && 2 * NU <= 65536 && K.far_slots > 0)
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
// values, same columns, same 20 B per nonzero; peak during the conversion = the CSR + one copy of the far and cross parts), row
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
    return QBH_OK;
}

// Same decomposition as kron_build, for the row kernel: the near part keeps rows and columns, the far part has rows AND
// columns in the tiled order of KronTile with B = 16 (one 128-byte line of doubles per major index and band); the far
// launch gathers from the tiled copy of the packed x and accumulates onto the near launch's result at orig(row).
// Measured slower than the unsplit operator; QBH_KRON_CODED=1 builds it for comparison.
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

// wave-block geometry of one part for k_spmv_wave (uncoded complex128 values)
int setup_wave_geometry(qbh_csr *A, const int64_t *d_ia, int64_t nnz, qbh::WaveDesc **d_wd_o, int64_t *n_wb_o, int *tpr_o, int *grid_o)
{
    hipStream_t s = A->stream;
    QBH_TRY(qbh::launch_max_rowlen(d_ia, A->nrows, (int64_t *)A->d_scal, s));
    int64_t maxlen = 0;
    QBH_HIP(hipMemcpyAsync(&maxlen, A->d_scal, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    // a block holds the rows that START inside its window: window + maxlen - 1 <= 512 when rows are short; rows longer
    // than half a tile make some blocks exceed it and those take the row-at-a-time path of the kernel
    // (7 slots of the tile are kept free: the kernel reads the stream from the 128-byte boundary below the block)
    const int64_t window = (maxlen <= 256) ? 505 - (maxlen > 0 ? maxlen - 1 : 0) : 249;
    const int64_t n_wb = std::max<int64_t>(1, (nnz + window - 1) / window);
    QBH_HIP(qbh::dev_alloc(d_wd_o, (size_t)(n_wb + 2) * sizeof(qbh::WaveDesc)));
    QBH_TRY(qbh::launch_build_wavedesc(d_ia, A->nrows, window, *d_wd_o, n_wb, s));
    const double avg = A->nrows > 0 ? (double)nnz / (double)A->nrows : 0.0;
    int tpr = avg <= 32 ? 2 : avg <= 64 ? 4 : avg <= 128 ? 8 : 16;      // rows per pass = 64 / tpr >= rows per block
    {
        const int t = qbh::debug_sw().wave_tpr;
        if (t == 2 || t == 4 || t == 8 || t == 16) tpr = t;
    }
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const int occ = std::max(1, qbh::wave_kernel_occupancy(tpr));
    int64_t g = std::min<int64_t>((int64_t)occ * ncu, ((((n_wb + 3) >> 2) + 7) / 8) * 8);
    g = std::max<int64_t>(8, (g / 8) * 8);
    if (qbh::debug_sw().grid >= 8) g = (qbh::debug_sw().grid / 8) * 8;
    *n_wb_o = n_wb;
    *tpr_o = tpr;
    *grid_o = (int)g;
    return QBH_OK;
}

// row-block geometry of the part(s) and the partial-sum workspace; callable again after the shard has been split
int build_geometry(qbh_csr *A)
{
    for (void *q : {(void *)A->d_rb, (void *)A->d_bp, (void *)A->rem.d_rb, (void *)A->rem.d_bp, (void *)A->d_wd, (void *)A->rem.d_wd})
        if (q) (void)hipFree(q);
    A->d_rb = nullptr;
    A->d_bp = nullptr;
    A->rem.d_rb = nullptr;
    A->rem.d_bp = nullptr;
    A->d_wd = nullptr;
    A->rem.d_wd = nullptr;
    // complex128 values (no dictionary): the wave kernel, unless the row kernel was asked for by name
    A->use_wave = A->kernel == QBH_KERNEL_ROWS && A->dict_mode == 0 && A->d_val != nullptr && A->opts.spmv_kernel != QBH_KERNEL_ROWS;
    if (A->tuned == 0) A->use_wave = false;            // timed at creation (autotune_kernel): the row kernel won
    // an operator on a product basis is re-ordered in place into its two parts: it has no CSR geometry afterwards
    QBH_TRY(kron_build(A));
    int grid_max = 0;
    if (A->kron.active) {
        grid_max = std::max(A->kron.grid_n, A->kron.grid_f);
        A->grid = A->wgrid = grid_max;
        A->n_blocks = A->n_wb = 0;
    } else {
        QBH_TRY(setup_geometry(A, A->d_ia, A->nnz, A->dict_mode, &A->npb, &A->tpr, &A->unroll, &A->window, &A->n_blocks, &A->d_rb,
                               &A->d_bp, &A->grid));
        grid_max = A->grid;
        if (A->has_rem) {
            CsrPart &R = A->rem;
            QBH_TRY(setup_geometry(A, R.d_ia, R.nnz, A->dict_mode, &R.npb, &R.tpr, &R.unroll, &R.window, &R.n_blocks, &R.d_rb, &R.d_bp,
                                   &R.grid));
            grid_max = std::max(grid_max, R.grid);
        }
        if (A->use_wave) {
            QBH_TRY(setup_wave_geometry(A, A->d_ia, A->nnz, &A->d_wd, &A->n_wb, &A->wtpr, &A->wgrid));
            grid_max = std::max(grid_max, A->wgrid);
            if (A->has_rem) {
                CsrPart &R = A->rem;
                QBH_TRY(setup_wave_geometry(A, R.d_ia, R.nnz, &R.d_wd, &R.n_wb, &R.wtpr, &R.wgrid));
                grid_max = std::max(grid_max, R.wgrid);
            }
        }
        QBH_TRY(kronc_build(A));
        if (A->kronc.active) grid_max = std::max(grid_max, std::max(A->kronc.near_p.grid, A->kronc.far_p.grid));
    }
    const size_t nparts = (size_t)std::max(grid_max, qbh::kMaxRedBlocks);
    if (A->d_partials) (void)hipFree(A->d_partials);
    A->d_partials = nullptr;
    QBH_HIP(qbh::dev_alloc(&A->d_partials, nparts * 16 * sizeof(double)));
    return QBH_OK;
}

int spmv_run(qbh_csr *A, const d2 *x, d2 *y, double alpha, double beta, double gamma, double *red);

// QBH_KERNEL_AUTO on complex128 values of an operator WITHOUT a product structure: the wave kernel wins where the gathers hit
// the caches (1-D operators, kagome), the row kernel where they miss (momentum sectors: its lanes-to-rows gathers and larger
// blocks cost fewer line fetches) -- so the two are TIMED on the operator itself and the faster one is kept.  Not timed (the
// choice is then the same on every box and results are bit-reproducible from run to run): operators below 1e7 nonzeros, row
// shards, a Kronecker split (structural), qbh_opts.deterministic, QBH_NO_AUTOTUNE=1.  qbh_csr_info.tuned / tune_ms_* report
// what happened.
int autotune_kernel(qbh_csr *A)
{
    if (A->opts.spmv_kernel != QBH_KERNEL_AUTO || !A->use_wave || A->kind != 0 || A->has_rem || A->has_comm || A->nnz < 10000000) return QBH_OK;
    if (A->kron.active || A->opts.deterministic || A->nrows != A->ncols) return QBH_OK;
    if (!A->opts.autotune) return QBH_OK;
    hipStream_t s = A->stream;
    d2 *x = nullptr, *y = nullptr;
    if (qbh::dev_alloc(&x, (size_t)A->ncols * sizeof(d2)) != hipSuccess || qbh::dev_alloc(&y, (size_t)A->nrows * sizeof(d2)) != hipSuccess) {
        (void)hipGetLastError();
        if (x) (void)hipFree(x);
        return QBH_OK;                     // no room for the trial vectors: keep the default choice
    }
    auto done = [&](int code) {
        (void)hipFree(x);
        (void)hipFree(y);
        return code;
    };
    // the Lanczos form of the call (old y read, reductions fused) on random vectors: what the solvers issue
    if (qbh::launch_randomize(x, nullptr, A->ncols, 0, 12345u, A->d_partials, s) != QBH_OK) return done(QBH_OK);
    if (qbh::launch_fill_const(y, A->nrows, 0.25, s) != QBH_OK) return done(QBH_OK);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(QBH_OK);
    const int saved_profile = A->opts.profile;
    A->opts.profile = 0;
    // one warm launch, then the FASTEST of three timed launches per form (clock ramps and page faults only ever add time);
    // the wave kernel is the default and is given up only for a form that is at least 3 % faster
    double t_mode[2] = {1e300, 1e300};
    for (int mode = 0; mode < 2; ++mode) {                              // 0 row kernel | 1 wave kernel
        A->use_wave = mode != 0;
        double red[3];
        bool ok = spmv_run(A, x, y, 0.5, -0.25, 0.0, red) == QBH_OK;                                   // warm
        for (int r = 0; r < 3 && ok; ++r) {
            float ms = 0.f;
            ok = hipEventRecord(e0, s) == hipSuccess && spmv_run(A, x, y, 0.5, -0.25, 0.0, red) == QBH_OK &&
                 hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (ok && ms < t_mode[mode]) t_mode[mode] = ms;
        }
    }
    int best_mode = 1;
    if (qbh::debug_sw().trace_tune)
        fprintf(stderr, "qbhip autotune: dim %lld nnz %lld: row kernel %.3f ms, wave kernel %.3f ms\n", (long long)A->nrows,
                (long long)A->nnz, t_mode[0], t_mode[1]);
    if (t_mode[0] < 0.97 * t_mode[best_mode]) best_mode = 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    A->opts.profile = saved_profile;
    A->use_wave = best_mode != 0;
    A->tuned = best_mode != 0 ? 1 : 0;
    A->tune_ms[0] = t_mode[0];
    A->tune_ms[1] = t_mode[1];
    A->stats = qbh_stats{};
    A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    A->xr_of = nullptr;
    return done(QBH_OK);
}

// geometry + workspace once the CSR arrays are in HBM
int finalize(qbh_csr *A)
{
    hipStream_t s = A->stream;
    const qbh_opts &o = A->opts;
    QBH_HIP(qbh::dev_alloc(&A->d_scal, 16 * sizeof(double)));
    QBH_HIP(hipHostMalloc(&A->h_scal, 16 * sizeof(double)));
    QBH_HIP(qbh::dev_alloc(&A->d_wctr, qbh::kWctrRegions * 128 * sizeof(unsigned long long)));
    QBH_HIP(hipEventCreate(&A->ev0));
    QBH_HIP(hipEventCreate(&A->ev1));
    QBH_HIP(hipEventCreate(&A->ev2));
    QBH_HIP(hipEventCreate(&A->ev3));
    A->nnz_total = A->nnz;
    // QBH_KERNEL_WAVE is the uncoded member of the row-kernel family (coded values and the real gather stay on k_spmv_rows)
    A->kernel = (o.spmv_kernel == QBH_KERNEL_VECTOR) ? QBH_KERNEL_VECTOR
              : (o.spmv_kernel == QBH_KERNEL_STREAM) ? QBH_KERNEL_STREAM : QBH_KERNEL_ROWS;

    QBH_HIP(qbh::dev_alloc(&A->d_flag, sizeof(int)));
    QBH_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
    // the caller told us what its index means (qbh_opts.basis_kind): hold the operator in the order the kernels like, keep
    // the map for the vector seams; a hint that does not describe the matrix changes nothing
    if (o.basis_kind != QBH_BASIS_NONE && A->basis.kind == 0) {
        bool applied = false;
        QBH_TRY(qbh::basis_to_internal(A, o.basis_kind, o.n_sites, o.n_up, o.n_dn, &applied));
    } else if (o.basis_kind == QBH_BASIS_NONE && o.basis_detect && o.kron_minor == 0 && o.kron_split != 0 && A->basis.kind == 0 &&
               (o.kron_split == 2 || A->nnz >= 100000000)) {
        // nobody said what the index means (the reference's constructor cannot): look for a two-species basis of this dimension
        bool applied = false;
        const double t_d = now_ms();
        QBH_TRY(qbh::basis_detect(A, &applied));
        A->detect_ms = now_ms() - t_d;
    }
    // value dictionary first: it decides how much LDS a row block needs
    QBH_TRY(try_value_dict(A));
    const bool coded = A->d_code != nullptr;
    // is every stored value real?  (enables the 8-byte wire format of the x exchange)
    if (coded) {
        A->code_w = A->n_dict <= 256 ? 1 : 2;
        A->dict_mode = A->n_dict <= 256 ? 1 : A->n_dict <= qbh::kDictLds ? 2 : 3;
        std::vector<d2> dict((size_t)A->n_dict);
        QBH_HIP(hipMemcpy(dict.data(), A->d_dict, dict.size() * sizeof(d2), hipMemcpyDeviceToHost));
        A->values_real = true;
        for (int i = 0; i < A->n_dict; ++i) A->values_real = A->values_real && dict[(size_t)i].y == 0.0;
    } else if (A->d_val && A->nnz > 0) {
        double *tmp = nullptr;
        QBH_HIP(qbh::dev_alloc(&tmp, (size_t)qbh::kMaxRedBlocks * sizeof(double)));
        QBH_TRY(qbh::launch_imag_norm(A->d_val, A->nnz, tmp, s));
        std::vector<double> hp((size_t)qbh::blas_grid(A->nnz));
        QBH_HIP(hipMemcpyAsync(hp.data(), tmp, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        QBH_HIP(hipStreamSynchronize(s));
        (void)hipFree(tmp);
        double sum = 0.0;
        for (double v : hp) sum += v;
        A->values_real = (sum == 0.0);
    }
    // The local / remote column split of a row shard only pays under a communicator (the local part runs while the
    // all-gather is in flight): it is made when one is attached (qbh_csr_set_comm), not here -- a shard that is driven
    // with a full-length x and no communicator keeps ONE part and one launch per SpMV.
    QBH_TRY(build_geometry(A));
    QBH_HIP(hipStreamSynchronize(s));
    A->stats = qbh_stats{};
    A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    QBH_TRY(autotune_kernel(A));
    return QBH_OK;
}

int new_handle(qbh_csr **out, const qbh_opts *opts, bool host_arrays = false)
{
    int dev = 0;
    QBH_TRY(require_device(opts, &dev));
    qbh_csr *A = new (std::nothrow) qbh_csr();
    if (!A) return QBH_ENOMEM;
    if (opts) A->opts = *opts;
    else if (host_arrays) qbh_opts_default(&A->opts);      // process-wide defaults: the host-array entry points they are documented for
    else qbh::opts_builtin(&A->opts);
    A->device = dev;
    A->debug = qbh::debug_sw().flags;                                   // timing experiments only
    if (qbh::debug_sw().chunk_mult > 0) A->chunk_mult = qbh::debug_sw().chunk_mult;
    if (A->opts.stream) {
        A->stream = (hipStream_t)A->opts.stream;
        A->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&A->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            qbh::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete A;
            return QBH_EHIP;
        }
        A->own_stream = true;
    }
    *out = A;
    return QBH_OK;
}

// dictionary-code the value stream on the device when there are at most 256 (1-byte codes) or
// 65536 (2-byte codes, row kernel) distinct values (exact, lossless); otherwise the operator silently stays uncoded.
int try_value_dict(qbh_csr *A)
{
    if (!A->opts.value_dict || !A->d_val || A->d_code || A->nnz <= 0) return QBH_OK;
    // two-byte codes (<= 65536 distinct values) are a feature of the row kernel
    const int cap = (A->kernel == QBH_KERNEL_ROWS && A->opts.value_dict != 2) ? 65536 : 256;
    uint8_t *code = nullptr;
    d2 *dict = nullptr;
    int n = 0;
    int rc = qbh::build_value_dict(A->d_val, A->nnz, cap, &code, &dict, &n, A->stream);
    if (rc != QBH_OK || n == 0) return rc;
    A->d_code = code;
    A->d_dict = dict;
    A->n_dict = n;
    if (A->own_arrays) (void)hipFree(A->d_val);     // the 16 B/nnz array is no longer needed
    A->d_val = nullptr;
    return QBH_OK;
}

}  // namespace

extern "C" void qbh_csr_destroy(qbh_csr *A)
{
    if (!A) return;
    Bind bind(A);
    if (A->stream) (void)hipStreamSynchronize(A->stream);
    if (A->own_arrays) {
        if (A->d_ia) (void)hipFree(A->d_ia);
        if (A->d_ja) (void)hipFree(A->d_ja);
        if (A->d_val) (void)hipFree(A->d_val);
    }
    if (A->d_code) (void)hipFree(A->d_code);
    if (A->d_dict) (void)hipFree(A->d_dict);
    if (A->d_rb) (void)hipFree(A->d_rb);
    if (A->d_bp) (void)hipFree(A->d_bp);
    if (A->d_wd) (void)hipFree(A->d_wd);
    kronc_release(A);
    if (A->rem.d_wd) (void)hipFree(A->rem.d_wd);
    kron_free_aux(A);
    if (A->rem.d_ia) (void)hipFree(A->rem.d_ia);
    if (A->rem.d_ja) (void)hipFree(A->rem.d_ja);
    if (A->rem.d_val) (void)hipFree(A->rem.d_val);
    if (A->rem.d_code) (void)hipFree(A->rem.d_code);
    if (A->rem.d_rb) (void)hipFree(A->rem.d_rb);
    if (A->rem.d_bp) (void)hipFree(A->rem.d_bp);
    qbh::release_native_comm(A);
    if (A->d_flag) (void)hipFree(A->d_flag);
    if (A->basis.d_map) (void)hipFree(A->basis.d_map);
    if (A->basis.d_stage) (void)hipFree(A->basis.d_stage);
    if (A->d_xr) (void)hipFree(A->d_xr);
    if (A->kind == 1) {
        (void)hipFree(A->mf.cfg_u);
        (void)hipFree(A->mf.cfg_d);
        (void)hipFree(A->mf.tgt_u);
        (void)hipFree(A->mf.tgt_d);
        (void)hipFree(A->mf.val_u);
        (void)hipFree(A->mf.val_d);
        if (A->mf.pk_d) (void)hipFree(A->mf.pk_d);
    }
    if (A->mfsec) {
        for (void *q : {(void *)A->mfsec->blk, (void *)A->mfsec->hop, (void *)A->mfsec->item, (void *)A->mfsec->ucfg,
                        (void *)A->mfsec->upell, (void *)A->mfsec->prank, (void *)A->mfsec->rrow, (void *)A->mfsec->ria,
                        (void *)A->mfsec->rja, (void *)A->mfsec->rval, (void *)A->d_mfsec})
            if (q) (void)hipFree(q);
        delete A->mfsec;
        A->mfsec = nullptr;
    }
    if (A->kind == 2) {
        (void)hipFree(A->mfh.binom);
        (void)hipFree(A->mfh.chunk);
        (void)hipFree(A->mfh.mask);
        (void)hipFree(A->mfh.offd);
        (void)hipFree(A->mfh.diag);
    }
    if (A->ev2) (void)hipEventDestroy(A->ev2);
    if (A->ev3) (void)hipEventDestroy(A->ev3);
    if (A->d_partials) (void)hipFree(A->d_partials);
    if (A->d_scal) (void)hipFree(A->d_scal);
    if (A->d_wctr) (void)hipFree(A->d_wctr);
    if (A->h_scal) (void)hipHostFree(A->h_scal);
    if (A->d_stage_x) (void)hipFree(A->d_stage_x);
    if (A->d_stage_y) (void)hipFree(A->d_stage_y);
    if (A->ev0) (void)hipEventDestroy(A->ev0);
    if (A->ev1) (void)hipEventDestroy(A->ev1);
    if (A->own_stream && A->stream) (void)hipStreamDestroy(A->stream);
    delete A;
}

// Shared body of qbh_csr_create / qbh_csr_create_rows: validation on the host (threads), then the chunked upload and the
// upper -> full expansion on the device (qbh_build.hip); no second host copy of the matrix is made.
static int create_from_host(qbh_csr **out, int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia, const int64_t *ja,
                            const qbh_z *val, int64_t r0, int64_t r1, const qbh_opts *opts, const char *who)
{
    if (!out || !ia || !ja || !val || dim <= 0 || nnz <= 0) {
        qbh::set_error("%s: null pointer or non-positive size", who);
        return QBH_EINVAL;
    }
    if (dim >= (int64_t)std::numeric_limits<int32_t>::max()) {
        qbh::set_error("%s: dim %lld does not fit the int32 column index of one GPU shard", who, (long long)dim);
        return QBH_EUNSUPP;
    }
    if (r0 < 0 || r1 > dim || r0 >= r1) {
        qbh::set_error("%s: row range [%lld, %lld) outside [0, %lld)", who, (long long)r0, (long long)r1, (long long)dim);
        return QBH_EINVAL;
    }
    const double t0 = now_ms();
    const bool trace = qbh::debug_sw().trace_create != 0;
    QBH_TRY(qbh::validate_host_csr(dim, nnz, sym_upper, ia, ja));
    if (trace) fprintf(stderr, "[qbh_csr_create] %-22s %8.2f ms\n", "host validation", now_ms() - t0);
    const d2 *hv = reinterpret_cast<const d2 *>(val);
    qbh_opts eff;                             // NULL means the process-wide defaults here
    if (opts) eff = *opts;
    else qbh_opts_default(&eff);
    if (!sym_upper && eff.check_hermitian) QBH_TRY(qbh::check_hermitian_host(dim, ia, ja, hv));

    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts, true));
    A->nrows = r1 - r0;
    A->ncols = dim;
    A->row_offset = r0;
    A->own_arrays = true;
    int rc = qbh::build_shard_from_host(dim, nnz, sym_upper, ia, ja, hv, r0, r1, A->stream, &A->d_ia, &A->d_ja, &A->d_val, &A->nnz,
                                        &A->create_ms_upload);
    const double t1 = now_ms();
    if (rc == QBH_OK) rc = finalize(A);
    if (rc != QBH_OK) {
        qbh_csr_destroy(A);
        return rc;
    }
    if (trace) fprintf(stderr, "[qbh_csr_create] %-22s %8.2f ms\n", "finalize (geometry)", now_ms() - t1);
    A->create_ms = now_ms() - t0;
    A->create_bytes_in = nnz * 24 + (dim + 1) * 8;
    *out = A;
    return QBH_OK;
}

extern "C" int qbh_csr_create(qbh_csr **out, int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia,
                              const int64_t *ja, const qbh_z *val, const qbh_opts *opts)
{
    return create_from_host(out, dim, nnz, sym_upper, ia, ja, val, 0, dim, opts, "qbh_csr_create");
}

extern "C" int qbh_csr_create_rows(qbh_csr **out, int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia,
                                   const int64_t *ja, const qbh_z *val, int64_t row_begin, int64_t row_end,
                                   const qbh_opts *opts)
{
    return create_from_host(out, dim, nnz, sym_upper, ia, ja, val, row_begin, row_end, opts, "qbh_csr_create_rows");
}

extern "C" int qbh_balanced_row_cuts(int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia, const int64_t *ja,
                                     int nranks, int64_t *cuts)
{
    if (!ia || !ja || !cuts || dim <= 0 || nnz <= 0 || nranks < 1) {
        qbh::set_error("qbh_balanced_row_cuts: invalid argument");
        return QBH_EINVAL;
    }
    QBH_TRY(qbh::validate_host_csr(dim, nnz, sym_upper, ia, ja));
    return qbh::balanced_row_cuts(dim, nnz, sym_upper, ia, ja, nranks, cuts);
}

extern "C" int qbh_csr_create_device(qbh_csr **out, int64_t nrows, int64_t ncols, int64_t row_offset,
                                     int64_t nnz, int64_t *d_ia, int32_t *d_ja, qbh_z *d_val,
                                     int take_ownership, const qbh_opts *opts)
{
    if (!out || !d_ia || !d_ja || !d_val || nrows <= 0 || ncols <= 0 || nnz < 0 || row_offset < 0 ||
        row_offset + nrows > ncols) {
        qbh::set_error("qbh_csr_create_device: invalid argument");
        return QBH_EINVAL;
    }
    if (ncols >= (int64_t)std::numeric_limits<int32_t>::max()) {
        qbh::set_error("ncols %lld does not fit int32 columns", (long long)ncols);
        return QBH_EUNSUPP;
    }
    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts));
    A->nrows = nrows;
    A->ncols = ncols;
    A->row_offset = row_offset;
    A->nnz = nnz;
    A->d_ia = d_ia;
    A->d_ja = d_ja;
    A->d_val = reinterpret_cast<d2 *>(d_val);
    A->own_arrays = take_ownership != 0;
    int rc = finalize(A);
    if (rc != QBH_OK) {
        // take_ownership: the arrays belong to the library from the moment of the call (finalize may already have replaced
        // them by the split shard), so they are released here and the caller must not free them.  Borrowed arrays that
        // finalize left untouched stay with the caller.
        if (!take_ownership && A->d_ia == d_ia) A->own_arrays = false;
        if (!take_ownership && A->d_ia == d_ia) A->d_ia = nullptr, A->d_ja = nullptr, A->d_val = nullptr;
        qbh_csr_destroy(A);
        return rc;
    }
    *out = A;
    return QBH_OK;
}

int qbh::adopt_coded_csr(qbh_csr **out, int64_t nrows, int64_t ncols, int64_t row_offset, int64_t nnz, int64_t *d_ia,
                         int32_t *d_ja, uint8_t *d_code, qbh::d2 *d_dict, int n_dict, const qbh_opts *opts)
{
    if (!out || !d_ia || !d_ja || !d_code || !d_dict || n_dict <= 0 || n_dict > 65536 || nrows <= 0 || nnz < 0 ||
        row_offset < 0 || row_offset + nrows > ncols || ncols >= (int64_t)std::numeric_limits<int32_t>::max()) {
        qbh::set_error("adopt_coded_csr: invalid argument");
        return QBH_EINVAL;
    }
    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts));
    A->nrows = nrows;
    A->ncols = ncols;
    A->row_offset = row_offset;
    A->nnz = nnz;
    A->d_ia = d_ia;
    A->d_ja = d_ja;
    A->d_code = d_code;
    A->d_dict = d_dict;
    A->n_dict = n_dict;
    A->own_arrays = true;
    int rc = finalize(A);
    if (rc != QBH_OK) {
        qbh_csr_destroy(A);      // the handle owned every array from the moment of the call: all released here
        return rc;
    }
    *out = A;
    return QBH_OK;
}

int qbh::adopt_mf_sector(qbh_csr **out, qbh::MfSec *host_tables, qbh::MfSec *dev_tables, int64_t dim, int64_t nnz_equiv,
                         const qbh_opts *opts)
{
    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts));
    A->nrows = A->ncols = dim;
    A->row_offset = 0;
    A->nnz = A->nnz_total = nnz_equiv;
    A->kernel = QBH_KERNEL_ROWS;
    A->values_real = host_tables->all_real;
    A->n_blocks = (dim + qbh::kBlock - 1) / qbh::kBlock;
    A->grid = (int)std::min<int64_t>(A->n_blocks, 256 * 8);
    auto fail = [&](int code) {
        qbh_csr_destroy(A);
        return code;
    };
    if (qbh::dev_alloc(&A->d_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipHostMalloc(&A->h_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (qbh::dev_alloc(&A->d_flag, sizeof(int)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipMemset(A->d_flag, 0, sizeof(int)) != hipSuccess) return fail(QBH_EHIP);
    if (hipEventCreate(&A->ev0) != hipSuccess || hipEventCreate(&A->ev1) != hipSuccess ||
        hipEventCreate(&A->ev2) != hipSuccess || hipEventCreate(&A->ev3) != hipSuccess)
        return fail(QBH_EHIP);
    if (qbh::dev_alloc(&A->d_partials, (size_t)qbh::kMaxRedBlocks * 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    A->stats = qbh_stats{};
    A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    A->kind = 3;                // from here on the handle owns the tables (on any failure above the caller still does)
    A->mfsec = host_tables;
    A->d_mfsec = dev_tables;
    *out = A;
    return QBH_OK;
}

int qbh::adopt_mf_hubbard(qbh_csr **out, const qbh::MfHubbard &t, int64_t nrows, int64_t ncols, int64_t row_offset,
                          int64_t nnz_equiv, const qbh_opts *opts)
{
    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts));
    A->nrows = nrows;
    A->ncols = ncols;
    A->row_offset = row_offset;
    A->nnz = A->nnz_total = nnz_equiv;         // what the CSR of the same operator would hold (for the byte accounting)
    A->kernel = QBH_KERNEL_ROWS;
    A->values_real = true;                     // t and U are real
    A->n_blocks = (nrows + qbh::kBlock - 1) / qbh::kBlock;
    A->grid = (int)std::min<int64_t>(A->n_blocks, 256 * 8);
    auto fail = [&](int code) {
        qbh_csr_destroy(A);
        return code;
    };
    if (qbh::dev_alloc(&A->d_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipHostMalloc(&A->h_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (qbh::dev_alloc(&A->d_flag, sizeof(int)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipMemset(A->d_flag, 0, sizeof(int)) != hipSuccess) return fail(QBH_EHIP);
    if (hipEventCreate(&A->ev0) != hipSuccess || hipEventCreate(&A->ev1) != hipSuccess ||
        hipEventCreate(&A->ev2) != hipSuccess || hipEventCreate(&A->ev3) != hipSuccess)
        return fail(QBH_EHIP);
    const size_t nparts = (size_t)std::max(A->grid, qbh::kMaxRedBlocks);
    if (qbh::dev_alloc(&A->d_partials, nparts * 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    A->stats = qbh_stats{};
    A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    A->kind = 1;                // from here on the handle owns the tables (on any failure above the caller still does)
    A->mf = t;
    *out = A;
    return QBH_OK;
}

int qbh::adopt_mf_heis(qbh_csr **out, const qbh::MfHeis &t, int64_t nrows, int64_t ncols, int64_t row_offset,
                          int64_t nnz_equiv, const qbh_opts *opts)
{
    qbh_csr *A = nullptr;
    QBH_TRY(new_handle(&A, opts));
    A->nrows = nrows;
    A->ncols = ncols;
    A->row_offset = row_offset;
    A->nnz = A->nnz_total = nnz_equiv;         // what the CSR of the same operator would hold (for the byte accounting)
    A->kernel = QBH_KERNEL_ROWS;
    A->values_real = true;                     // J is real
    A->n_blocks = (nrows + qbh::kBlock - 1) / qbh::kBlock;
    A->grid = (int)std::min<int64_t>(A->n_blocks, 256 * 8);
    auto fail = [&](int code) {
        qbh_csr_destroy(A);
        return code;
    };
    if (qbh::dev_alloc(&A->d_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipHostMalloc(&A->h_scal, 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    if (qbh::dev_alloc(&A->d_flag, sizeof(int)) != hipSuccess) return fail(QBH_ENOMEM);
    if (hipMemset(A->d_flag, 0, sizeof(int)) != hipSuccess) return fail(QBH_EHIP);
    if (hipEventCreate(&A->ev0) != hipSuccess || hipEventCreate(&A->ev1) != hipSuccess ||
        hipEventCreate(&A->ev2) != hipSuccess || hipEventCreate(&A->ev3) != hipSuccess)
        return fail(QBH_EHIP);
    const size_t nparts = (size_t)std::max(A->grid, qbh::kMaxRedBlocks);
    if (qbh::dev_alloc(&A->d_partials, nparts * 16 * sizeof(double)) != hipSuccess) return fail(QBH_ENOMEM);
    A->stats = qbh_stats{};
    A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    A->kind = 2;                // from here on the handle owns the tables (on any failure above the caller still does)
    A->mfh = t;
    *out = A;
    return QBH_OK;
}

extern "C" int qbh_csr_get_info(const qbh_csr *A, qbh_csr_info *info)
{
    if (!A || !info) return QBH_EINVAL;
    info->nrows = A->nrows;
    info->ncols = A->ncols;
    info->row_offset = A->row_offset;
    const int64_t nnz = A->nnz_total, nb = A->n_blocks + (A->has_rem ? A->rem.n_blocks : 0);
    info->nnz = nnz;
    info->n_blocks = nb;
    info->bytes_matrix = (A->nrows + 1) * 8 * (A->has_rem ? 2 : 1) + nnz * 4 + (A->d_code ? nnz + 256 * 16 : nnz * 16) +
                         (nb + 2) * 12;
    info->bytes_algorithmic = nnz * 20 + (A->nrows + 1) * 8 + A->nrows * 32;
    if (A->kind == 1)
        info->bytes_matrix = (A->mf.Nu * A->mf.wu + A->mf.Nd * A->mf.wd) * 5 + A->mf.Nd * A->mf.wd * 4 + (A->mf.Nu + A->mf.Nd) * 4;
    if (A->kind == 2)
        info->bytes_matrix = ((int64_t)(A->mfh.n_sites + 1) * (A->mfh.n_dn + 1) + (int64_t)A->mfh.n_chunks * (A->mfh.n_dn + 1) * 64 +
                              3 * (int64_t)A->mfh.n_bonds) * 8;
    if (A->kind == 3 && A->mfsec) {
        const qbh::MfSec &m = *A->mfsec;
        info->bytes_matrix = m.n_blocks * (int64_t)sizeof(qbh::MfSecBlock) + m.n_items * 8 + m.cu * 4 * (1 + m.w_up + m.n_trans) +
                             m.n_rrows * 12 + m.rnnz * 20;
    }
    info->kernel = A->kind != 0 ? QBH_KERNEL_MATRIX_FREE : A->use_wave ? QBH_KERNEL_WAVE : A->kernel;
    info->value_dict = A->d_code ? A->n_dict : 0;
    info->device = A->device;
    info->stream = (void *)A->stream;
    info->create_ms = A->create_ms;
    info->create_bytes_in = A->create_bytes_in;
    info->kron_minor = A->kron.active ? A->kron.t.S : 0;
    info->kron_far_nnz = A->kron.active ? A->kron.nnz_f : 0;
    info->kron_band = A->kron.active ? A->kron.t.B : 0;
    info->kron_sliced = A->kron.active && A->kron.sliced ? 1 : 0;
    info->kron_inplace = A->kron.active && A->kron.inplace ? 1 : 0;
    info->kron_classes = 0;
    info->kron_cols16 = 0;
    info->kron_cross_nnz = 0;
    info->gather_parts = A->has_comm ? ((A->kron.active && A->kron.comm_tiled) ? A->kron.n_parts : 1) : 0;
    if (A->kron.active) {
        const qbh_csr::KronSplit &K = A->kron;
        info->n_blocks = K.nwb_n + K.nwb_f + K.nwb_x;
        info->bytes_matrix = (A->nrows + 1) * 16 + ((K.sliced ? K.n_groups : A->nrows) + 1) * 8 + (K.n_xrows + 1) * (K.xrow ? 12 : 8) +
                             (K.own_far ? nnz + K.far_slots : nnz) * 20 + (K.nwb_n + K.nwb_f + K.nwb_x + 6) * 16;
        // columns as they are held: the int32 array while anything lives in it, 2 bytes per entry of a converted part
        info->bytes_matrix += (A->d_ja ? 0 : -4 * nnz) + 2 * (K.c16_n ? K.nnz_n : 0) + 2 * (K.c16_f ? K.far_slots : 0) + ((!A->d_ja && K.own_x) ? 4 * K.nnz_x : 0);
        info->kron_cols16 = (K.c16_n ? 1 : 0) | (K.c16_f ? 2 : 0);
        info->kron_classes = K.map.nc;
        info->kron_cross_nnz = K.nnz_x;
        for (int c = 0; c < K.map.nc; ++c) info->kron_minor = std::max<int64_t>(info->kron_minor, K.map.S[c]);      // several classes: the largest block
    }
    info->tuned = A->tuned;
    info->tune_ms_rows = A->tune_ms[0];
    info->tune_ms_wave = A->tune_ms[1];
    info->basis_internal = A->basis.kind;
    info->basis_detected = A->basis.detected ? 1 : 0;
    info->basis_n_sites = A->basis.kind ? A->basis.n_sites : 0;
    info->basis_n_up = A->basis.kind ? A->basis.n_up : 0;
    info->basis_n_dn = A->basis.kind ? A->basis.n_dn : 0;
    info->basis_detect_ms = A->detect_ms;
    if (A->kronc.active) {                       // the coded form of the split (row kernel, packed-double vectors)
        info->kron_minor = A->kronc.t.S;
        info->kron_far_nnz = A->kronc.sl.active ? A->kronc.sl.slots_f : A->kronc.far_p.nnz;      // sliced: stored far entries (padding included)
        info->kron_sliced = A->kronc.sl.active ? 1 : 0;
        info->kron_band = A->kronc.t.B;
    }
    return QBH_OK;
}

// Declare what the index of an EXISTING operator means (the same thing qbh_opts.basis_kind says at creation): the operator is
// re-ordered internally, its geometry rebuilt; device vectors made before the call are in the old order.
extern "C" int qbh_csr_set_basis(qbh_csr *A, int basis_kind, int n_sites, int n_up, int n_dn)
{
    if (!A) return QBH_EINVAL;
    if (A->has_comm || A->has_rem || A->kron.active || A->kronc.active || A->d_code || A->basis.kind != 0 || A->kind != 0) {
        qbh::set_error("qbh_csr_set_basis: needs a plain, unsharded complex128 CSR (value_dict = 0, kron_split = 0 at creation) without a communicator");
        return QBH_EUNSUPP;
    }
    Bind bind(A);
    QBH_HIP(hipStreamSynchronize(A->stream));
    bool applied = false;
    if (basis_kind == QBH_BASIS_DETECT) {
        const double t_d = now_ms();
        QBH_TRY(qbh::basis_detect(A, &applied));
        A->detect_ms = now_ms() - t_d;
    } else {
        QBH_TRY(qbh::basis_to_internal(A, basis_kind, n_sites, n_up, n_dn, &applied));
    }
    if (!applied) return QBH_OK;
    A->opts.basis_kind = A->basis.kind;
    A->opts.n_sites = A->basis.n_sites;
    A->opts.n_up = A->basis.n_up;
    A->opts.n_dn = A->basis.n_dn;
    if (A->opts.kron_split == 0) A->opts.kron_split = 1;
    A->tuned = -1;
    QBH_TRY(build_geometry(A));
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}

// The gather in parts (qbh_comm::allgather_part_begin): the far pass sweeps the gathered x band range by band range, a band
// range is one contiguous piece of every rank's tiled block, so the far pass of the first range can run while the later ranges
// are still on the links -- the step then costs max(wire, near + far) instead of max(wire, near) + far.  Default: 4 parts when
// there are ranks to receive from (QBH_GATHER_PARTS overrides; 1 = the single gather), none when the communicator has no
// part hooks (the Python ShardComm) or the far part is not sliced.
// what THIS rank could do (1 = the single gather); the ranks then take the smallest proposal (qbh_csr_set_comm): a rank that
// issued one whole-block group while its peers issue four part groups would hang the exchange
int kron_parts_wanted(const qbh_csr *A, const qbh_comm *comm)
{
    const qbh_csr::KronSplit &K = A->kron;
    if (!K.active || !K.sliced || !comm->allgather_part_begin || !comm->allgather_part_wait || !A->d_wctr || K.nwb_f <= 0) return 1;
    int64_t want = comm->nranks > 1 ? 4 : 1;
    if (A->opts.gather_parts > 0) want = A->opts.gather_parts;
    const int64_t nfb = K.t.S / K.t.B;                       // full bands (the far pass covers exactly these)
    return (int)std::max<int64_t>(1, std::min<int64_t>({want, 8, nfb}));
}

int kron_gather_parts(qbh_csr *A, const qbh_comm *comm, int64_t want)
{
    qbh_csr::KronSplit &K = A->kron;
    K.n_parts = 1;
    K.part_off_len.clear();
    const int64_t nfb = K.t.S / K.t.B;
    if (want <= 1 || kron_parts_wanted(A, comm) < want) return QBH_OK;
    int64_t band[9];
    for (int64_t k = 0; k <= want; ++k) band[k] = k * nfb / want;
    K.part_blk[0] = 0;
    for (int64_t k = 1; k < want; ++k) {                     // first slot of the range's first group -> the block that holds it
        int64_t slot = 0;
        QBH_HIP(hipMemcpy(&slot, K.ia_f + band[k] * K.t.NU, sizeof(int64_t), hipMemcpyDeviceToHost));
        K.part_blk[k] = std::min<int64_t>(slot / 512, K.nwb_f);
    }
    K.part_blk[want] = K.nwb_f;
    K.part_off_len.assign((size_t)want * 2 * (size_t)comm->nranks, 0);
    for (int64_t k = 0; k < want; ++k)
        for (int q = 0; q < comm->nranks; ++q) {
            const int64_t nu = K.cols.cu[q + 1] - K.cols.cu[q];
            const int64_t off = band[k] * K.t.B * nu;
            const int64_t end = k == want - 1 ? nu * K.t.S : band[k + 1] * K.t.B * nu;
            K.part_off_len[((size_t)k * (size_t)comm->nranks + (size_t)q) * 2] = off;
            K.part_off_len[((size_t)k * (size_t)comm->nranks + (size_t)q) * 2 + 1] = end - off;
        }
    K.n_parts = (int)want;
    return QBH_OK;
}

extern "C" int qbh_csr_set_comm(qbh_csr *A, const qbh_comm *comm)
{
    if (!A) return QBH_EINVAL;
    if (!comm || comm->nranks < 1) {          // NULL detaches; a 1-rank communicator is valid (hooks still run)
        A->has_comm = false;
        if (A->kron.active && A->kron.comm_tiled) {       // far columns back to the tiled order of the whole vector
            Bind bind(A);
            const qbh::KronCols one = kron_cols_one(A->kron.t.S, A->kron.NUg, A->kron.t.B);
            if (A->kron.ja_f) QBH_TRY(qbh::launch_kron_remap_cols(A->kron.ja_f, A->kron.far_slots, A->kron.cols, one, A->stream));
            QBH_TRY(qbh::launch_kron_remap_cols(A->kron.ja_x, A->kron.nnz_x, A->kron.cols, one, A->stream));
            QBH_HIP(hipStreamSynchronize(A->stream));
            A->kron.cols = one;
            A->kron.map.cols = one;
            A->kron.comm_tiled = false;
            A->kron.n_parts = 1;
            A->kron.xt_of = nullptr;
        }
        return QBH_OK;
    }
    if (A->kind == 3) {
        qbh::set_error("qbh_csr_set_comm: the matrix-free sector operator is a single-GPU form");
        return QBH_EUNSUPP;
    }
    if (!comm->d_xsend || !comm->d_xfull || !comm->d_scal || !comm->allgather_x || !comm->allreduce_sum ||
        comm->rank < 0 || comm->rank >= comm->nranks || comm->nblk < A->nrows) {
        // without buffers and hooks there is nothing to tell the peers with: the one failure that stays local
        qbh::set_error("qbh_csr_set_comm: incomplete communicator (rank %d/%d nblk %lld nrows %lld)", comm->rank, comm->nranks,
                       (long long)comm->nblk, (long long)A->nrows);
        return QBH_EINVAL;
    }
    // This call is COLLECTIVE for nranks > 1: every rank's local verdict travels through the communicator's own all-reduce
    // before anything is decided, so that no rank returns early while its peers wait in a collective, and the form of the
    // exchange (tiled blocks or plain, how many parts) is the same everywhere by construction.
    int local_err = QBH_OK;
    std::vector<int64_t> cuts_new;
    int64_t full_new = 0;
    if (comm->row_cuts) {
        const int64_t *c = comm->row_cuts;
        bool ok = c[0] == 0 && c[comm->nranks] == A->ncols && c[comm->rank] == A->row_offset &&
                  c[comm->rank + 1] - c[comm->rank] == A->nrows;
        for (int q = 0; q < comm->nranks && ok; ++q) ok = c[q + 1] >= c[q] && c[q + 1] - c[q] <= comm->nblk;
        if (!ok) {
            qbh::set_error("qbh_csr_set_comm: row_cuts do not describe this shard (rank %d/%d rows [%lld, %lld))", comm->rank,
                           comm->nranks, (long long)A->row_offset, (long long)(A->row_offset + A->nrows));
            local_err = QBH_EINVAL;
        } else {
            cuts_new.assign(c, c + comm->nranks + 1);
            full_new = A->ncols;
        }
    } else {
        if (comm->nblk * comm->rank != A->row_offset || comm->nblk * comm->nranks < A->ncols) {
            qbh::set_error("qbh_csr_set_comm: inconsistent communicator (rank %d/%d nblk %lld row_offset %lld)",
                           comm->rank, comm->nranks, (long long)comm->nblk, (long long)A->row_offset);
            local_err = QBH_EINVAL;
        }
        full_new = comm->nblk * (int64_t)comm->nranks;
    }
    Bind bind(A);
    // sums of indicators over the ranks: [0] failures, [1] ranks that can exchange tiled blocks, [2 + k] ranks proposing k + 1 parts
    auto agree = [&](double (&v)[12]) -> int {
        if (comm->nranks == 1) return QBH_OK;
        QBH_HIP(hipMemcpyAsync(comm->d_scal, v, sizeof(v), hipMemcpyHostToDevice, A->stream));
        if (comm->allreduce_sum(comm->ctx, 0, 12) != 0) {
            qbh::set_error("qbh_csr_set_comm: allreduce_sum hook failed");
            return QBH_ECOMM;
        }
        QBH_HIP(hipMemcpyAsync(v, comm->d_scal, sizeof(v), hipMemcpyDeviceToHost, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
        return QBH_OK;
    };
    qbh_csr::KronSplit &K = A->kron;
    const int64_t S = K.active ? K.t.S : 1;
    bool mine = local_err == QBH_OK && A->kind == 0 && K.active && K.map.nc == 1 && comm->nranks <= qbh::kKronMaxRanks && !(K.c16_f && comm->nranks > 1);
    if (mine) {
        if (comm->row_cuts) {
            for (int q = 0; q <= comm->nranks; ++q) mine = mine && comm->row_cuts[q] % S == 0;
        } else {
            mine = comm->nblk % S == 0;
        }
    }
    const int my_parts = mine ? kron_parts_wanted(A, comm) : 1;
    double v[12] = {0};
    v[0] = local_err != QBH_OK ? 1.0 : 0.0;
    v[1] = mine ? 1.0 : 0.0;
    v[2 + (my_parts - 1)] = 1.0;
    QBH_TRY(agree(v));
    if (v[0] > 0.0) {
        if (local_err == QBH_OK) qbh::set_error("qbh_csr_set_comm: a peer rank rejected the communicator (its qbh_last_error says why)");
        return local_err != QBH_OK ? local_err : QBH_ECOMM;
    }
    A->comm_cuts = cuts_new;
    A->comm_full = full_new;
    const bool all_tiled = v[1] == (double)comm->nranks;
    int parts = 1;
    for (int k = 0; k < 8; ++k)
        if (v[2 + k] > 0.0) {
            parts = k + 1;                       // the smallest proposal
            break;
        }
    if (A->kind == 0) {
        // A shard split in place (kron_build) exchanges the TILED copy of its block -- which only works when every rank does:
        // cuts at whole major indices and every operator split.  Without agreement a split shard is merged back into its CSR
        // and takes the generic path below; whether that worked is agreed on once more (a rank out of memory there must not
        // leave its peers attached and waiting in their first gather).
        if (K.active && all_tiled) {
            qbh::KronCols to{};
            to.S = S;
            to.B = K.t.B;
            to.nr = comm->nranks;
            for (int q = 0; q <= comm->nranks; ++q) {
                const int64_t cut = comm->row_cuts ? comm->row_cuts[q] : std::min<int64_t>((int64_t)q * comm->nblk, A->ncols);
                to.cu[q] = cut / S;
            }
            if (K.ja_f) QBH_TRY(qbh::launch_kron_remap_cols(K.ja_f, K.far_slots, K.cols, to, A->stream));      // 2-byte far columns: one rank, nothing moves
            QBH_TRY(qbh::launch_kron_remap_cols(K.ja_x, K.nnz_x, K.cols, to, A->stream));
            QBH_HIP(hipStreamSynchronize(A->stream));
            K.cols = to;
            K.map.cols = to;
            K.comm_tiled = true;
            K.xt_of = nullptr;
            QBH_TRY(kron_gather_parts(A, comm, parts));
        } else if (!all_tiled) {
            int rrc = QBH_OK;
            if (K.active) {
                rrc = kron_restore(A);
                if (rrc == QBH_OK) rrc = build_geometry(A);
            }
            double w[12] = {0};
            w[0] = rrc != QBH_OK ? 1.0 : 0.0;
            const int arc = agree(w);
            if (rrc != QBH_OK || arc != QBH_OK || w[0] > 0.0) {
                if (rrc == QBH_OK && arc == QBH_OK) qbh::set_error("qbh_csr_set_comm: a peer rank could not merge its split operator back into a CSR");
                return rrc != QBH_OK ? rrc : arc != QBH_OK ? arc : QBH_ECOMM;
            }
        }
    }
    if (A->kind == 0 && !A->kron.active && !A->has_rem && A->nrows < A->ncols) {      // first communicator on a stored row shard: split it now
        // the split comes FIRST and the communicator is committed only when it succeeded: a failure of the split itself (out
        // of memory) leaves the operator exactly as it was, unattached, with its single-part geometry; a failure AFTER it
        // (geometry of the two parts) leaves a handle that refuses every further SpMV
        Bind bind(A);
        QBH_HIP(hipStreamSynchronize(A->stream));
        QBH_TRY(split_shard(A));
        if (A->has_rem) {
            const int rc = build_geometry(A);
            if (rc != QBH_OK) {                     // the shard IS split but has no geometry: nothing can run on it any more
                A->has_comm = false;
                A->broken = true;
                return rc;
            }
        }
        QBH_HIP(hipStreamSynchronize(A->stream));
    }
    A->comm = *comm;
    A->comm.row_cuts = A->comm_cuts.empty() ? nullptr : A->comm_cuts.data();
    A->has_comm = true;
    return QBH_OK;
}

extern "C" int qbh_sync(const qbh_csr *A)
{
    if (!A) return QBH_EINVAL;
    Bind bind(A);
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}

extern "C" int qbh_get_stats(const qbh_csr *Ac, qbh_stats *s, int reset)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A) return QBH_EINVAL;
    qbh::harvest_native_comm(A);
    if (s) {
        *s = A->stats;
        if (A->stats.n_spmv == 0) s->ms_spmv_min = 0.0;
    }
    if (reset) {
        A->stats = qbh_stats{};
        A->stats.ms_spmv_min = std::numeric_limits<double>::infinity();
    }
    return QBH_OK;
}

// ------------------------------------------------- reductions / scalars --------
namespace {

inline double *scal_buf(qbh_csr *A) { return A->has_comm ? A->comm.d_scal : A->d_scal; }

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

void harvest_events(qbh_csr *A)
{
    float ms = 0.f, total = 0.f;
    bool any = false;
    if (A->ev_pending) {
        if (hipEventSynchronize(A->ev1) == hipSuccess && hipEventElapsedTime(&ms, A->ev0, A->ev1) == hipSuccess) {
            total += ms;
            any = true;
        }
        A->ev_pending = false;
    }
    if (A->ev_pending2) {
        if (hipEventSynchronize(A->ev3) == hipSuccess && hipEventElapsedTime(&ms, A->ev2, A->ev3) == hipSuccess) {
            total += ms;
            any = true;
        }
        A->ev_pending2 = false;
    }
    if (any) {
        A->stats.ms_spmv += total;
        if (total < A->stats.ms_spmv_min) A->stats.ms_spmv_min = total;
    }
}

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
    if (A->has_comm) return A->real_wire ? nullptr : reinterpret_cast<d2 *>(A->comm.d_xsend);
    return (A->nrows == A->ncols && A->kron.xt_cap >= A->nrows) ? A->kron.d_xt : nullptr;
}
// Only a driver knows that nothing else writes its vectors between the pass that produces x and the SpMV that reads it
// (a caller of the building-block entry points may scale or overwrite a vector in between): the drivers hold this guard.
// the coded split's tiled x (packed doubles): written by the all-real Lanczos step's axpy when the operator runs that form
inline double *kronc_tiled_target(const qbh_csr *A)
{
    return (A->kronc.active && A->kronc.sl.active && !A->has_comm && !A->has_rem && A->opts.tile_fold) ? A->kronc.d_xt : nullptr;
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
    int kron_swz = A->opts.deterministic ? 2 : 3;           // dynamic ordered walk per XCD unless the caller wants static walks
    if (A->opts.wave_walk >= 0) kron_swz = A->opts.wave_walk;
    if (kron_swz == 3) {
        if (!A->d_wctr) kron_swz = 2;
        else QBH_HIP(hipMemsetAsync(A->d_wctr, 0, (size_t)(comm && K.n_parts > 1 ? qbh::kWctrRegions : 3) * 128 * sizeof(unsigned long long), s));
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
    const d2 *xt = nullptr;                                  // what the far pass gathers from
    bool async_gather = false;
    if (comm) {
        d2 *send = reinterpret_cast<d2 *>(A->comm.d_xsend);
        if (K.xt_of != (const void *)x) QBH_TRY(qbh::launch_kron_tile(x, send, A->nrows, K.t, s));
        K.xt_of = nullptr;
        async_gather = A->comm.allgather_begin && A->comm.allgather_wait;
        int hrc = 0;
        if (K.n_parts > 1) {                                 // band ranges one after another: the far pass follows them (below)
            for (int k = 0; k < K.n_parts && hrc == 0; ++k)
                hrc = A->comm.allgather_part_begin(A->comm.ctx, k, K.n_parts, K.part_off_len.data() + (size_t)k * 2 * (size_t)A->comm.nranks);
        } else {
            hrc = async_gather ? A->comm.allgather_begin(A->comm.ctx, 0) : A->comm.allgather_x(A->comm.ctx, 0);
        }
        if (hrc != 0) {
            qbh::set_error("allgather hook failed");
            return QBH_ECOMM;
        }
        A->stats.n_gather++;
        xt = reinterpret_cast<const d2 *>(A->comm.d_xfull);
    } else {
        if (K.xt_cap < A->ncols) {                           // first use: the tiled copy of the full-length x
            if (K.d_xt) (void)hipFree(K.d_xt);
            K.d_xt = nullptr;
            K.xt_cap = 0;
            QBH_HIP(qbh::dev_alloc(&K.d_xt, (size_t)A->ncols * sizeof(d2)));
            K.xt_cap = A->ncols;
            if (qbh::debug_sw().print_ptrs) fprintf(stderr, "qbhip kron xt %p x %p y %p\n", (void *)K.d_xt, (const void *)x, (void *)y);
            K.xt_of = nullptr;
        }
        if (prof) {
            harvest_events(A);
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
        } else {
            nr.kcls = K.d_cls;
            nr.partials = (red && K.nnz_x == 0) ? A->d_partials : nullptr;
            QBH_TRY(qbh::launch_spmv_wave2(nr, K.tpr_n, 4, K.grid_n, s));
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
            harvest_events(A);
            QBH_HIP(hipEventRecord(A->ev0, s));
        }
        nr.far = nullptr;
        nr.partials = nullptr;
        QBH_TRY(qbh::launch_spmv_wave2(nr, K.tpr_n, 1, K.grid_n, s));        // y = alpha H_near x + beta y + gamma x
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev1, s));
            A->ev_pending = true;
        }
        if (K.n_parts > 1) {
            // every band range of the far part as soon as its piece of the gathered x is there; a block that straddles a range
            // boundary belongs to the later range (the pieces complete in order), its cut groups add up through the atomics
            QBH_TRY(qbh::launch_zero_cut_groups(K.wd_f, K.nwb_f, f.nrows, K.d_far, s));
            for (int k = 0; k < K.n_parts; ++k) {
                if (A->comm.allgather_part_wait(A->comm.ctx, k) != 0) {
                    qbh::set_error("allgather_part_wait hook failed");
                    return QBH_ECOMM;
                }
                if (prof && k == 0) QBH_HIP(hipEventRecord(A->ev2, s));
                qbh::SpmvArgs fk = f;
                fk.wd = K.wd_f + K.part_blk[k];
                fk.n_wb = K.part_blk[k + 1] - K.part_blk[k];
                fk.wctr = A->d_wctr ? A->d_wctr + (3 + k) * 128 : nullptr;
                if (fk.n_wb > 0) QBH_TRY(qbh::launch_spmv_wave2(fk, K.tpr_f, 3, (int)std::min<int64_t>(K.grid_f, std::max<int64_t>(fk.n_wb, 8)), s));
            }
        } else {
            if (async_gather && A->comm.allgather_wait(A->comm.ctx) != 0) {
                qbh::set_error("allgather_wait hook failed");
                return QBH_ECOMM;
            }
            if (prof) QBH_HIP(hipEventRecord(A->ev2, s));
            if (K.sliced) QBH_TRY(qbh::launch_zero_cut_groups(K.wd_f, K.nwb_f, f.nrows, K.d_far, s));
            QBH_TRY(qbh::launch_spmv_wave2(f, K.tpr_f, K.sliced ? 3 : 0, K.grid_f, s));
        }
        QBH_TRY(qbh::launch_kron_cross_rows(K.ia_x, K.xrow, K.n_xrows, K.ja_x, K.val_x, xt, K.t, K.d_far, s));
        QBH_TRY(qbh::launch_kron_combine(K.d_far, K.t, xl, y, A->nrows, alpha, red ? A->d_partials : nullptr, &nparts, s));
        if (prof) {
            QBH_HIP(hipEventRecord(A->ev3, s));
            A->ev_pending2 = true;
        }
    }
    A->xr_of = nullptr;
    A->stats.n_spmv++;
    if (red && A->defer_red) {
        QBH_TRY(qbh::launch_reduce_partials(A->d_partials, nparts, 3, A->d_scal, s));
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
            harvest_events(A);
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
            const int sec_walk = qbh::debug_sw().sec_walk;
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
            QBH_TRY(qbh::launch_reduce_partials(A->d_partials, mf_parts, 3, A->d_scal, A->stream));
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
    a.partials = (red && !A->has_rem) ? A->d_partials : nullptr;
    a.swizzle = A->opts.xcd_swizzle;
    a.chunk_mult = A->chunk_mult;
    a.unroll = A->unroll;
    a.colmask = (A->debug & 1) ? 1023 : -1;
    if (A->debug & 1) {
        if (qbh::debug_sw().colmask) a.colmask = qbh::debug_sw().colmask;    // gather-window experiments (results wrong by design)
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
        harvest_events(A);
        QBH_HIP(hipEventRecord(A->ev0, A->stream));
    }
    // complex128 values, complex vectors: the wave kernel; the real gather / all-real forms stay on the row kernel
    const bool wave = A->use_wave && a.xr == nullptr && a.y_re == nullptr;
    // The two passes of a Kronecker split take the dynamic ordered walk per XCD (DynWalk: C3 far pass 86 -> 64 GB, near pass
    // 92 -> 68 GB, 31.7 -> 31.0 ms); the unsplit wave kernel keeps the static chunked walk (xcd_swizzle 2), under which all XCDs
    // stream from ONE region -- ordered eighths cost it 34 -> 43 ms on C3.  xcd_swizzle 3 / QBH_WAVE_SWIZZLE choose by name.
    int wave_swz = A->opts.xcd_swizzle, wave_grid_used = A->wgrid;
    if (A->opts.wave_walk >= 0) wave_swz = A->opts.wave_walk;
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
    } else if (wave) {
        a.wd = A->d_wd;
        a.n_wb = A->n_wb;
        a.swizzle = wave_swz;
        a.wctr = A->d_wctr;
        const int pipe = qbh::debug_sw().wave_pipelined;    // experiment: the pipelined kernel on an unsplit operator
        if (pipe && A->wtpr <= 8) {
            int ncu = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, A->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
            const int occ = std::max(1, qbh::wave2_kernel_occupancy(A->wtpr, 1));
            int64_t g = std::min<int64_t>((int64_t)occ * ncu, ((((A->n_wb + 3) >> 2) + 7) / 8) * 8);
            g = std::max<int64_t>(8, (g / 8) * 8);
            g = std::min<int64_t>(g, A->wgrid);            // the partial-sum buffer is sized for the wave kernel's grid
            wave_grid_used = (int)g;
            QBH_TRY(qbh::launch_spmv_wave2(a, A->wtpr, 1, (int)g, A->stream));
        } else {
            QBH_TRY(qbh::launch_spmv_wave(a, A->wtpr, A->wgrid, A->stream));
        }
    } else {
        QBH_TRY(qbh::launch_spmv(a, A->kernel, A->npb, A->tpr, A->grid, A->stream));
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
        QBH_TRY(qbh::launch_reduce_partials(A->d_partials, grid_last, 3, A->d_scal, A->stream));
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
    if (!A->opts.real_fast_path) return QBH_OK;
    if (!(A->opts.real_forms & 1)) return QBH_OK;
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
        A->real_mode = A->kernel == QBH_KERNEL_ROWS;
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

struct WireGuard {            // whatever path a driver leaves by, the next call starts with the complex wire
    qbh_csr *A;
    ~WireGuard() { A->real_wire = false; A->real_mode = false; A->xr_of = nullptr; }
};

int dotc_run(qbh_csr *A, const d2 *x, const d2 *y, double *res2)
{
    QBH_TRY(qbh::launch_dotc(x, y, A->nrows, A->d_partials, A->stream));
    return finish_reduction(A, qbh::blas_grid(A->nrows), 2, res2);
}

// where the packed real parts of the next gather source go in the real fast path (nullptr: not active)
inline double *packed_target(qbh_csr *A)
{
    if (!A->real_mode || A->kernel != QBH_KERNEL_ROWS) return nullptr;
    if (A->has_comm) return A->real_wire ? reinterpret_cast<double *>(A->comm.d_xsend) : nullptr;
    return (A->nrows == A->ncols) ? A->d_xr : nullptr;
}

int axpy_norm_run(qbh_csr *A, d2 alpha, const d2 *x, d2 *y, double *nrm2sq)
{
    double *yr = packed_target(A);          // y is the next SpMV's x in every driver: emit its packed copy here
    if (d2 *yt = tiled_target(A)) {         // ... or, for a Kronecker split, its tiled copy
        QBH_TRY(qbh::launch_axpy_norm_tile(alpha, nullptr, x, y, yt, A->nrows, A->kron.t, A->d_partials, A->stream));
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
        QBH_TRY(qbh::launch_axpy_norm_tile(d2{scale, 0.0}, A->d_scal, x, y, yt, A->nrows, A->kron.t, A->d_partials, A->stream));
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

}  // namespace

// -------------------------------------------------------- device vectors -------
extern "C" int qbh_vec_alloc(qbh_z **d_out, int64_t n)
{
    if (!d_out || n <= 0) return QBH_EINVAL;
    if (qbh_device_count() <= 0) {
        qbh::set_error("no HIP device visible");
        return QBH_ENODEVICE;
    }
    QBH_HIP(qbh::dev_alloc((void **)d_out, (size_t)n * sizeof(qbh_z)));
    return QBH_OK;
}

extern "C" int qbh_vec_free(qbh_z *d)
{
    if (d) QBH_HIP(hipFree(d));
    return QBH_OK;
}

namespace {
// Host <-> device copies of vectors.  An operator held in another order than the caller's (qbh_opts.basis_kind) keeps its
// device vectors in the INTERNAL order: these two seams -- every host vector passes one of them -- translate, so callers see
// their own order throughout.  n must then be a whole number of vectors.
int vec_h2d(qbh_csr *A, d2 *d_dst, const void *h_src, int64_t n)
{
    if (A->basis.kind == 0) {
        QBH_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)n * sizeof(d2), hipMemcpyHostToDevice, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
        return QBH_OK;
    }
    const int64_t dim = A->nrows;
    if (n % dim != 0) {
        qbh::set_error("vector copy of %lld elements: an operator with a basis map moves whole vectors (%lld elements)", (long long)n, (long long)dim);
        return QBH_EINVAL;
    }
    if (!A->basis.d_stage) QBH_HIP(qbh::dev_alloc(&A->basis.d_stage, (size_t)dim * sizeof(d2)));
    for (int64_t j = 0; j < n / dim; ++j) {
        QBH_HIP(hipMemcpyAsync(A->basis.d_stage, (const char *)h_src + (size_t)j * (size_t)dim * sizeof(d2), (size_t)dim * sizeof(d2),
                               hipMemcpyHostToDevice, A->stream));
        QBH_TRY(qbh::launch_basis_scatter(A->basis.d_map, A->basis.d_stage, d_dst + (size_t)j * (size_t)dim, dim, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
    }
    return QBH_OK;
}
int vec_d2h(qbh_csr *A, void *h_dst, const d2 *d_src, int64_t n)
{
    if (A->basis.kind == 0) {
        QBH_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)n * sizeof(d2), hipMemcpyDeviceToHost, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
        return QBH_OK;
    }
    const int64_t dim = A->nrows;
    if (n % dim != 0) {
        qbh::set_error("vector copy of %lld elements: an operator with a basis map moves whole vectors (%lld elements)", (long long)n, (long long)dim);
        return QBH_EINVAL;
    }
    if (!A->basis.d_stage) QBH_HIP(qbh::dev_alloc(&A->basis.d_stage, (size_t)dim * sizeof(d2)));
    for (int64_t j = 0; j < n / dim; ++j) {
        QBH_TRY(qbh::launch_basis_gather(A->basis.d_map, d_src + (size_t)j * (size_t)dim, A->basis.d_stage, dim, A->stream));
        QBH_HIP(hipMemcpyAsync((char *)h_dst + (size_t)j * (size_t)dim * sizeof(d2), A->basis.d_stage, (size_t)dim * sizeof(d2),
                               hipMemcpyDeviceToHost, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
    }
    return QBH_OK;
}
}  // namespace

extern "C" int qbh_vec_upload(const qbh_csr *A, qbh_z *d_dst, const qbh_z *h_src, int64_t n)
{
    if (!A || !d_dst || !h_src || n < 0) return QBH_EINVAL;
    Bind bind(A);
    return vec_h2d(const_cast<qbh_csr *>(A), reinterpret_cast<d2 *>(d_dst), h_src, n);
}

extern "C" int qbh_vec_download(const qbh_csr *A, qbh_z *h_dst, const qbh_z *d_src, int64_t n)
{
    if (!A || !h_dst || !d_src || n < 0) return QBH_EINVAL;
    Bind bind(A);
    return vec_d2h(const_cast<qbh_csr *>(A), h_dst, reinterpret_cast<const d2 *>(d_src), n);
}

extern "C" int qbh_vec_zero(const qbh_csr *A, qbh_z *d, int64_t n)
{
    if (!A || !d || n < 0) return QBH_EINVAL;
    Bind bind(A);
    QBH_HIP(hipMemsetAsync(d, 0, (size_t)n * sizeof(qbh_z), A->stream));
    return QBH_OK;
}

extern "C" int qbh_vec_randomize(const qbh_csr *Ac, qbh_z *d_x, uint32_t seed)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !d_x) return QBH_EINVAL;
    Bind bind(A);
    d2 *x = reinterpret_cast<d2 *>(d_x);
    if (seed == 0) {   // src/miscellaneous.cc:374-376
        return qbh::launch_fill_const(x, A->nrows, std::sqrt(1.0 / (double)A->ncols), A->stream);
    }
    const int64_t nruns = (A->nrows + 15) / 16;
    d2 *xr = x;                                  // the Lehmer stream is indexed by the CALLER's element number
    if (A->basis.kind != 0) {
        if (!A->basis.d_stage) QBH_HIP(qbh::dev_alloc(&A->basis.d_stage, (size_t)A->nrows * sizeof(d2)));
        xr = A->basis.d_stage;
    }
    QBH_TRY(qbh::launch_randomize(xr, nullptr, A->nrows, A->has_comm ? A->row_offset : 0, seed, A->d_partials, A->stream));
    double sq = 0.0;
    QBH_TRY(finish_reduction(A, qbh::blas_grid(nruns), 1, &sq));
    if (xr != x) QBH_TRY(qbh::launch_basis_scatter(A->basis.d_map, xr, x, A->nrows, A->stream));
    return qbh::launch_scal(1.0 / std::sqrt(sq), x, A->nrows, A->stream);
}

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
namespace {
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
}  // namespace

extern "C" int qbh_multmv(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host)
{
    return multmv_host(const_cast<qbh_csr *>(A), x_host, y_host, 0.0);
}

extern "C" int qbh_multmv2(const qbh_csr *A, const qbh_z *x_host, qbh_z *y_host)
{
    return multmv_host(const_cast<qbh_csr *>(A), x_host, y_host, 1.0);
}

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
        const bool no_defer = qbh::debug_sw().no_defer != 0;                   // A/B switch
        if (!A->has_comm && !no_defer) {
            // one GPU: <u, w> stays on the device and feeds the axpy directly; one host synchronisation per step
            A->defer_red = true;
            const int rc1 = spmv_run(A, vpt(mcur - 1), vpt(mcur), sc[sx], -bprev * sc[sy], 0.0, red);
            A->defer_red = false;
            QBH_TRY(rc1);
            double dot = 0.0;
            QBH_TRY(axpy_norm_deferred(A, -sc[sx] * sc[sx], vpt(mcur - 1), vpt(mcur), &dot, &sq));
            a[mcur - 1] = sc[sx] * dot;
            b[mcur] = std::sqrt(sq);
            sc[sy] = 1.0 / b[mcur];
            return QBH_OK;
        }
        QBH_TRY(spmv_run(A, vpt(mcur - 1), vpt(mcur), sc[sx], -bprev * sc[sy], 0.0, red));
        a[mcur - 1] = sc[sx] * red[0];
        // w -= a v_{m-1} ; b = |w|                                                         K5+K6
        QBH_TRY(axpy_norm_run(A, d2{-a[mcur - 1] * sc[sx], 0.0}, vpt(mcur - 1), vpt(mcur), &sq));
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
    if (k == 0) {                                          // :167-191
        b[0] = 0.0;
        QBH_TRY(step(1, 0.0));
        m = ++k;
        --np;
    }

    std::vector<double> w((size_t)mm + 8), zl((size_t)mm + 8), ws((size_t)mm + 8);     // ws always holds four Ritz values
    int rc = QBH_OK;
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
            // the four lowest Ritz values and the last component of the lowest Ritz vector: all the test below
            // uses of hess_eigen's full decomposition (src/lanczos.cc:229-231), in O(m) instead of O(m^2..m^3)
            double zl0 = 0.0;
            const int nsm = (int)std::min<int64_t>(4, m);
            for (int q = 0; q < 4; ++q) ws[(size_t)q] = 0.0;
            rc = qbh::tridiag_lowest(m, a, b + 1, nsm, ws.data(), &zl0);
            if (rc == QBH_ENOCONV) {           // overflow guard of the twisted factorisation: fall back to QL
                rc = qbh::tridiag_eigen_lastrow(m, a, b + 1, w.data(), zl.data());
                if (rc != QBH_OK) break;
                int64_t imin = 0;
                for (int64_t j = 1; j < m; ++j)
                    if (w[j] < w[imin]) imin = j;
                std::copy(w.begin(), w.begin() + m, ws.begin());
                std::partial_sort(ws.begin(), ws.begin() + nsm, ws.begin() + m);
                zl0 = zl[(size_t)imin];
            }
            if (rc != QBH_OK) break;
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
                if (cnt_accuE0 > 15 && accuracy < prec) break;   // :240
            }
            theta0_prev = ritz0;
            theta1_prev = ritz1;
        }
    } while (m < mm);
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
    QBH_TRY(qbh::launch_randomize(nullptr, d_x, A->nrows, A->has_comm ? A->row_offset : 0, seed, A->d_partials, A->stream));
    double sq = 0.0;
    QBH_TRY(finish_reduction(A, qbh::blas_grid(nruns), 1, &sq));
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
            QBH_TRY(spmv_run(A, p, pp, 1.0, 0.0, machine_prec - E0, red));
            const double den = red[0] * red[0] + red[1] * red[1];
            const d2 alpha = {accu * accu * red[0] / den, -accu * accu * red[1] / den};
            QBH_TRY(qbh::launch_cg_update(alpha, p, pp, v, r, n, A->d_partials, A->stream));   // :324-325
            QBH_TRY(finish_reduction(A, qbh::blas_grid(n), 1, &sq));
            const double beta = std::sqrt(sq) / accu;                                        // :326
            {
                double *pr = packed_target(A);      // p is the next SpMV's x: emit its packed real copy in the same pass
                if (d2 *pt = tiled_target(A)) {     // ... or its tiled copy (Kronecker split)
                    QBH_TRY(qbh::launch_xpby_tile(r, beta * beta, p, pt, n, A->kron.t, A->stream));
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
            // DGKS-style correction, only when the removed components outweigh the remainder by more than
            // 7x (error amplification |w|/|w'|); a Hamiltonian with a large diagonal would otherwise trigger
            // it on every step because alpha^2 dominates |w|^2
            if (b2 < 0.02 * wnorm2) {
                double corr = 0.0;
                rc = cgs_pass(w, j + 1, &corr, &b2);
                if (rc != QBH_OK) break;
                alpha += corr;
            }
            const double beta = std::sqrt(b2);
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
    if (info) {
        info->n_matvec = A->stats.n_spmv - spmv0;
        info->ms_spmv = A->stats.ms_spmv - ms_spmv0;
        info->ms_total = now_ms() - t_start;
        info->n_reorth = restarts;                           // number of restarts (ARPACK's niter)
    }
    return rc;
}

// ------------------------------------------------------------- download ---------
namespace {
// rows [r0, r1) of one part -> host (ia rebased to 0)
int download_part(const qbh_csr *A, const int64_t *d_ia, const int32_t *d_ja, const d2 *d_val, const uint8_t *d_code, int64_t r0,
                  int64_t r1, std::vector<int64_t> &ia, std::vector<int32_t> &ja, std::vector<d2> &val, bool want_val)
{
    ia.resize((size_t)(r1 - r0 + 1));
    QBH_HIP(hipMemcpy(ia.data(), d_ia + r0, ia.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t p0 = ia.front(), p1 = ia.back();
    for (auto &v : ia) v -= p0;
    ja.resize((size_t)(p1 - p0));
    if (p1 > p0) QBH_HIP(hipMemcpy(ja.data(), d_ja + p0, ja.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (!want_val) return QBH_OK;
    val.resize((size_t)(p1 - p0));
    if (p1 == p0) return QBH_OK;
    if (d_code) {                              // decode the dictionary-coded stream
        std::vector<d2> dict((size_t)A->n_dict);
        QBH_HIP(hipMemcpy(dict.data(), A->d_dict, dict.size() * sizeof(d2), hipMemcpyDeviceToHost));
        if (A->code_w == 2) {
            std::vector<uint16_t> code(val.size());
            QBH_HIP(hipMemcpy(code.data(), d_code + 2 * p0, code.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < code.size(); ++i) val[i] = dict[code[i]];
        } else {
            std::vector<uint8_t> code(val.size());
            QBH_HIP(hipMemcpy(code.data(), d_code + p0, code.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < code.size(); ++i) val[i] = dict[code[i]];
        }
    } else {
        QBH_HIP(hipMemcpy(val.data(), d_val + p0, val.size() * sizeof(d2), hipMemcpyDeviceToHost));
    }
    return QBH_OK;
}
}  // namespace

extern "C" int qbh_csr_download(const qbh_csr *A, int64_t r0, int64_t r1, int64_t *ia, int32_t *ja, qbh_z *val)
{
    if (!A || r0 < 0 || r1 < r0 || r1 > A->nrows) return QBH_EINVAL;
    if (A->kind != 0) {
        qbh::set_error("qbh_csr_download: the operator is matrix-free (no stored CSR)");
        return QBH_EUNSUPP;
    }
    if (A->basis.kind != 0) {
        qbh::set_error("qbh_csr_download: the operator is held in another basis order than the caller's (qbh_opts.basis_kind); "
                       "its rows are not the caller's rows");
        return QBH_EUNSUPP;
    }
    Bind bind(A);
    QBH_HIP(hipStreamSynchronize(A->stream));
    if (A->kron.active) {                    // split in place: the rows are merged back on the device (columns ascending)
        std::vector<int64_t> hia((size_t)(r1 - r0 + 1));
        QBH_HIP(hipMemcpy(hia.data(), A->d_ia + r0, hia.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
        const int64_t p0 = hia.front(), cnt = hia.back() - hia.front();
        if (ia)
            for (size_t i = 0; i < hia.size(); ++i) ia[i] = hia[i] - p0;
        if (cnt == 0 || (!ja && !val)) return QBH_OK;
        int32_t *tj = nullptr;
        d2 *tv = nullptr;
        QBH_HIP(qbh::dev_alloc(&tj, (size_t)cnt * sizeof(int32_t)));
        hipError_t he = qbh::dev_alloc(&tv, (size_t)cnt * sizeof(d2));
        int rc = he == hipSuccess ? qbh::launch_kron_merge_rows(kron_parts(A), r0, r1, tj, tv, p0, A->stream) : QBH_ENOMEM;
        if (rc == QBH_OK && ja) he = hipMemcpyAsync(ja, tj, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, A->stream);
        if (rc == QBH_OK && he == hipSuccess && val) he = hipMemcpyAsync(val, tv, (size_t)cnt * sizeof(d2), hipMemcpyDeviceToHost, A->stream);
        if (rc == QBH_OK && he == hipSuccess) he = hipStreamSynchronize(A->stream);
        (void)hipFree(tj);
        if (tv) (void)hipFree(tv);
        if (rc == QBH_OK && he != hipSuccess) {
            qbh::set_error("qbh_csr_download: %s", hipGetErrorString(he));
            (void)hipGetLastError();
            rc = QBH_EHIP;
        }
        return rc;
    }
    std::vector<int64_t> ia0, ia1;
    std::vector<int32_t> ja0, ja1;
    std::vector<d2> v0, v1;
    const bool want_val = val != nullptr;
    QBH_TRY(download_part(A, A->d_ia, A->d_ja, A->d_val, A->d_code, r0, r1, ia0, ja0, v0, want_val));
    if (A->has_rem)
        QBH_TRY(download_part(A, A->rem.d_ia, A->rem.d_ja, A->rem.d_val, A->rem.d_code, r0, r1, ia1, ja1, v1, want_val));
    // merge the two column-sorted parts of every row back into one ascending row
    int64_t q = 0;
    for (int64_t r = 0; r < r1 - r0; ++r) {
        if (ia) ia[r] = q;
        int64_t a = ia0[r], ae = ia0[r + 1], b = A->has_rem ? ia1[r] : 0, be = A->has_rem ? ia1[r + 1] : 0;
        while (a < ae || b < be) {
            const bool take0 = b >= be || (a < ae && ja0[a] <= ja1[b]);
            if (ja) ja[q] = take0 ? ja0[a] : ja1[b];
            if (val) {
                const d2 v = take0 ? v0[a] : v1[b];
                val[q].re = v.x;
                val[q].im = v.y;
            }
            if (take0) ++a;
            else ++b;
            ++q;
        }
    }
    if (ia) ia[r1 - r0] = q;
    return QBH_OK;
}
