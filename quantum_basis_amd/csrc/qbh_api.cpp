// qbh_api.cpp -- the extern "C" surface of libqbhip.so, part 1: errors, options, operator lifetime (create / adopt / destroy /
// info / set_basis), geometry of the unsplit forms, device vectors, download.  The other parts: qbh_split.cpp, qbh_commattach.cpp,
// qbh_spmv.cpp, qbh_solvers.cpp (qbh_api_priv.hpp says what each holds).  Each entry point names the reference function it
// replaces in include/qbhip.h.
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

#include "qbh_api_priv.hpp"

using qbh::d2;
using namespace qbhapi;

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
                        {"print_ptrs", &d.print_ptrs, nullptr}, {"sec_walk", &d.sec_walk, nullptr}, {"sec_grid", &d.sec_grid, nullptr}, {"sec_tile", &d.sec_tile, nullptr},
                        {"sec_unroll", &d.sec_unroll, nullptr}, {"wave_pipelined", &d.wave_pipelined, nullptr}, {"create_chunk", nullptr, &d.create_chunk},
                        {"force_ragged", &d.force_ragged, nullptr}, {"mf_row", &d.mf_row, nullptr}, {"mf_chunk", &d.mf_chunk, nullptr},
                        {"mf_window", &d.mf_window, nullptr}, {"kronc_abl", &d.kronc_abl, nullptr}, {"kronc_far_chunk", &d.kronc_far_chunk, nullptr},
                        {"kronc_far_ng", &d.kronc_far_ng, nullptr}, {"kronc_far_nt", &d.kronc_far_nt, nullptr}, {"no_far_align", &d.no_far_align, nullptr},
                        {"no_defer", &d.no_defer, nullptr}, {"pipe_nospec", &d.pipe_nospec, nullptr}, {"host_delay_us", &d.host_delay_us, nullptr},
                        {"side_noprio", &d.side_noprio, nullptr}, {"comm_reserve", &d.comm_reserve, nullptr}, {"comm_far_cap", &d.comm_far_cap, nullptr}};
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

namespace qbhapi {
std::mutex g_defaults_mu;
bool g_have_defaults = false;
qbh_opts g_defaults;
}  // namespace qbhapi

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
    o->kron_uniform = 7;
    o->gather_parts = 0;
    o->wave_walk = -1;
    o->tile_fold = 1;
    o->autotune = 1;
    o->shard_split = 1;
    o->real_forms = 7;
    o->basis_detect = 1;
    o->sector_orbit = 1;
    o->lanczos_pipeline = 1;
    o->real_wire = 1;
    o->sector_cut = 0;
    o->comm_reserve = 0;
    o->sparse_gather = 1;
    o->major_partition = 0;
    o->kron_minor = 0;
    o->deterministic = 0;
    o->basis_kind = QBH_BASIS_NONE;
    o->n_sites = o->n_up = o->n_dn = 0;
}

// ------------------------------------------------------------ operator ---------
namespace qbhapi {

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
}  // namespace qbhapi
namespace qbh {
hipError_t device_alloc(void **p, size_t bytes) { return hipMalloc(p, bytes); }
}  // namespace qbh
namespace qbhapi {
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

int new_handle(qbh_csr **out, const qbh_opts *opts, bool host_arrays)
{
    int dev = 0;
    QBH_TRY(require_device(opts, &dev));
    qbh_csr *A = new (std::nothrow) qbh_csr();
    if (!A) return QBH_ENOMEM;
    if (opts) A->opts = *opts;
    else if (host_arrays) qbh_opts_default(&A->opts);      // process-wide defaults: the host-array entry points they are documented for
    else qbh::opts_builtin(&A->opts);
    A->device = dev;
    A->dbg = qbh::debug_sw();                 // snapshot: the hot paths (every SpMV, every Lanczos step) read the handle's copy, never the environment
    A->debug = A->dbg.flags;                                            // timing experiments only
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

}  // namespace qbhapi

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
                        (void *)A->mfsec->upell, (void *)A->mfsec->prank, (void *)A->mfsec->oid, (void *)A->mfsec->oek, (void *)A->mfsec->tpar,
                        (void *)A->mfsec->usgn, (void *)A->mfsec->utab, (void *)A->mfsec->uext, (void *)A->mfsec->rrow, (void *)A->mfsec->ria,
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
    for (auto &o : A->ev_old)
        for (hipEvent_t e : o.e)
            if (e) (void)hipEventDestroy(e);
    if (A->d_major_inv) (void)hipFree(A->d_major_inv);
    if (A->lz.d_buf) (void)hipFree(A->lz.d_buf);
    if (A->lz.d_state) (void)hipFree(A->lz.d_state);
    if (A->lz.h_log) (void)hipHostFree(A->lz.h_log);
    for (hipEvent_t e : A->lz.ev)
        if (e) (void)hipEventDestroy(e);
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
    if (hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream) != hipSuccess) return fail(QBH_EHIP);      // on the handle's stream: the null stream is not ordered with it
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
    if (hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream) != hipSuccess) return fail(QBH_EHIP);      // on the handle's stream: the null stream is not ordered with it
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
    if (hipMemsetAsync(A->d_flag, 0, sizeof(int), A->stream) != hipSuccess) return fail(QBH_EHIP);      // on the handle's stream: the null stream is not ordered with it
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
        info->bytes_matrix = m.n_blocks * (int64_t)sizeof(qbh::MfSecBlock) + m.n_items * 8 + m.n_rrows * 12 + m.rnnz * 20 +
                             (m.orbit ? m.cu * 26 + m.n_orb * m.w_orb * 6 : m.cu * 4 * (1 + m.w_up + m.n_trans));
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
    info->wire_element_bytes = A->has_comm ? A->wire_bytes_last : 0;
    info->major_partition = A->d_major_inv ? A->major_parts : 0;
    info->gather_sparse = (A->has_comm && A->kron.active && A->kron.comm_tiled && A->kron.sparse) ? 1 : 0;
    info->gather_needed_frac = (A->has_comm && A->kron.active && A->kron.comm_tiled) ? A->kron.need_frac : 1.0;
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
    info->kron_table_kernel = (A->kronc.active && A->kronc.table_route) ? 1 : 0;
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

extern "C" int qbh_sync(const qbh_csr *A)
{
    if (!A) return QBH_EINVAL;
    Bind bind(A);
    QBH_HIP(hipStreamSynchronize(A->stream));
    return QBH_OK;
}

extern "C" int qbh_csr_set_option(qbh_csr *A, const char *name, int value)
{
    if (!A || !name) return QBH_EINVAL;
    const std::string nm(name);
    if (nm == "lanczos_pipeline") A->opts.lanczos_pipeline = value;
    else if (nm == "profile") {
        Bind bind(A);
        harvest_events(A);                  // nothing stays pending across the change
        A->opts.profile = value;
    }
    else if (nm == "tile_fold") A->opts.tile_fold = value;
    else if (nm == "comm_reserve") A->opts.comm_reserve = value;          // a launch-geometry choice: read at every SpMV (every rank must set the same)
    else {
        qbh::set_error("qbh_csr_set_option: '%s' is not an option that can change after creation", name);
        return QBH_EINVAL;
    }
    return QBH_OK;
}

extern "C" int qbh_csr_major_order(const qbh_csr *A, int32_t *generator_major, int64_t n_major)
{
    if (!A || !generator_major) return QBH_EINVAL;
    if (!A->d_major_inv) {
        qbh::set_error("qbh_csr_major_order: the operator is in the generator's own order");
        return QBH_EUNSUPP;
    }
    if (n_major != A->major_n) return QBH_EINVAL;
    Bind bind(A);
    QBH_HIP(hipMemcpy(generator_major, A->d_major_inv, (size_t)n_major * sizeof(int32_t), hipMemcpyDeviceToHost));
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

namespace qbhapi {
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
}  // namespace qbhapi

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

extern "C" int qbh_vec_to_internal(const qbh_csr *A, qbh_z *d_dst, const qbh_z *d_src)
{
    if (!A || !d_dst || !d_src || d_dst == d_src) return QBH_EINVAL;
    Bind bind(A);
    if (A->basis.kind == 0) {
        QBH_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)A->nrows * sizeof(qbh_z), hipMemcpyDeviceToDevice, A->stream));
        return QBH_OK;
    }
    return qbh::launch_basis_scatter(A->basis.d_map, reinterpret_cast<const d2 *>(d_src), reinterpret_cast<d2 *>(d_dst), A->nrows, A->stream);
}

extern "C" int qbh_vec_from_internal(const qbh_csr *A, qbh_z *d_dst, const qbh_z *d_src)
{
    if (!A || !d_dst || !d_src || d_dst == d_src) return QBH_EINVAL;
    Bind bind(A);
    if (A->basis.kind == 0) {
        QBH_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)A->nrows * sizeof(qbh_z), hipMemcpyDeviceToDevice, A->stream));
        return QBH_OK;
    }
    return qbh::launch_basis_gather(A->basis.d_map, reinterpret_cast<const d2 *>(d_src), reinterpret_cast<d2 *>(d_dst), A->nrows, A->stream);
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
    if (A->d_major_inv)      // partition order of the major indices: element (u, d) is drawn at position (generator's u) * S + d of the stream
        QBH_TRY(qbh::launch_randomize(xr, nullptr, A->nrows, 0, seed, A->d_partials, A->stream, A->d_major_inv + A->row_offset / A->major_S, A->major_S));
    else
    QBH_TRY(qbh::launch_randomize(xr, nullptr, A->nrows, A->has_comm ? A->row_offset : 0, seed, A->d_partials, A->stream));
    double sq = 0.0;
    QBH_TRY(finish_reduction(A, qbh::blas_grid(nruns), 1, &sq));
    if (xr != x) QBH_TRY(qbh::launch_basis_scatter(A->basis.d_map, xr, x, A->nrows, A->stream));
    return qbh::launch_scal(1.0 / std::sqrt(sq), x, A->nrows, A->stream);
}


// ------------------------------------------------------------- download ---------
namespace qbhapi {
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
}  // namespace qbhapi

static int download_rows_internal(const qbh_csr *A, int64_t r0, int64_t r1, int64_t *ia, int32_t *ja, qbh_z *val);

extern "C" int qbh_csr_download(const qbh_csr *A, int64_t r0, int64_t r1, int64_t *ia, int32_t *ja, qbh_z *val)
{
    if (!A || r0 < 0 || r1 < r0 || r1 > A->nrows) return QBH_EINVAL;
    if (A->kind != 0) {
        qbh::set_error("qbh_csr_download: the operator is matrix-free (no stored CSR)");
        return QBH_EUNSUPP;
    }
    if (A->basis.kind == 0) return download_rows_internal(A, r0, r1, ia, ja, val);
    // The operator is held in another order than the caller's (a named / detected basis, a cut sector): the CALLER's rows come
    // back -- row r = internal row g(r), column c -> the caller index of that internal column, value times the two signs, columns
    // ascending again (H_caller[r, c] = s_r s_c H_internal[g(r), g(c)]).  Done on the host from the whole internal operator: a
    // seam for tests and tools, 20 B per nonzero of host memory (QBH_ENOMEM when that does not fit).
    if (A->nrows != A->ncols) {
        qbh::set_error("qbh_csr_download: a basis map on a row shard");
        return QBH_EUNSUPP;
    }
    try {
        const int64_t n = A->nrows;
        std::vector<int64_t> iai((size_t)n + 1);
        QBH_TRY(download_rows_internal(A, 0, n, iai.data(), nullptr, nullptr));
        const int64_t nnz_i = iai[(size_t)n];
        std::vector<int32_t> jai((size_t)std::max<int64_t>(nnz_i, 1));
        std::vector<qbh_z> vai(val ? (size_t)std::max<int64_t>(nnz_i, 1) : 0);
        QBH_TRY(download_rows_internal(A, 0, n, iai.data(), jai.data(), val ? vai.data() : nullptr));
        std::vector<uint32_t> hm((size_t)n);
        {
            Bind bind(A);
            QBH_HIP(hipMemcpy(hm.data(), A->basis.d_map, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        std::vector<int32_t> inv((size_t)n);
        for (int64_t r = 0; r < n; ++r) inv[(size_t)(hm[(size_t)r] & 0x7FFFFFFFu)] = (int32_t)r;
        std::vector<std::pair<int32_t, qbh_z>> row;
        int64_t q = 0;
        for (int64_t r = r0; r < r1; ++r) {
            const uint32_t m = hm[(size_t)r];
            const int64_t g = m & 0x7FFFFFFFu;
            const bool sr = (m >> 31) != 0;
            row.clear();
            for (int64_t k = iai[(size_t)g]; k < iai[(size_t)g + 1]; ++k) {
                const int32_t c = inv[(size_t)jai[(size_t)k]];
                qbh_z v{0.0, 0.0};
                if (val) {
                    v = vai[(size_t)k];
                    if (sr != ((hm[(size_t)c] >> 31) != 0)) {
                        v.re = -v.re;
                        v.im = -v.im;
                    }
                }
                row.emplace_back(c, v);
            }
            std::sort(row.begin(), row.end(), [](const std::pair<int32_t, qbh_z> &a, const std::pair<int32_t, qbh_z> &b) { return a.first < b.first; });
            if (ia) ia[r - r0] = q;
            for (const auto &e : row) {
                if (ja) ja[q] = e.first;
                if (val) val[q] = e.second;
                ++q;
            }
        }
        if (ia) ia[r1 - r0] = q;
    } catch (const std::bad_alloc &) {
        qbh::set_error("qbh_csr_download: no host memory for the whole internal operator (basis-mapped download)");
        return QBH_ENOMEM;
    }
    return QBH_OK;
}

static int download_rows_internal(const qbh_csr *A, int64_t r0, int64_t r1, int64_t *ia, int32_t *ja, qbh_z *val)
{
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
