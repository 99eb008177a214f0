// qbh_build.hip -- the host-assembled CSR of the reference (csr_mat<T>: int64 ia/ja, complex128 val, Hermitian-upper by
// default; src/qbasis.h:979-985, src/sparse.cc:202-260) -> one row shard of the FULL operator in HBM.
//
// The host arrays are never copied on the host: they are streamed once (twice for Hermitian-upper input: a column-only
// counting pass, then the fill pass) through two pinned staging buffers of a few MB -- int64 columns narrowed to int32
// by a pool of host threads while the previous chunk is in flight -- and the upper -> full expansion, which is a
// transpose of the strictly upper part, happens on the device:
//   pass 1  k_mirror_count   every entry (r, c), r < c, adds one to the length of row c   (atomic int32 counters)
//           row lengths -> exclusive scan -> row pointers of the shard
//   pass 2  k_fill           entry (r, c, v): the row's own entries keep their order behind the mirrored ones;
//                            the mirrored copy (c, r, conj v) takes the next free slot of row c (atomic cursor)
//           k_sort_lower     the mirrored entries of each row are put in ascending column order (they arrive in
//                            atomic order; columns within a row are unique, so the result is deterministic)
// A row shard [r0, r1) only keeps what lands in its rows, so the same code serves qbh_csr_create (the whole operator)
// and qbh_csr_create_rows (one rank's block of the unchanged host CSR).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "qbh_internal.hpp"

namespace qbh {

// ------------------------------------------------------------ host thread pool ----
int host_threads()
{
    static int n = [] {
        int t = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("QBH_HOST_THREADS")) t = atoi(e);
        return std::max(1, std::min(t, 32));
    }();
    return n;
}

// fn(begin, end) over [0, n) in contiguous slices, one per thread
template <typename F>
static void parallel_slices(int64_t n, int64_t min_per_thread, F fn)
{
    int t = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (t <= 1) {
        fn((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)t);
    const int64_t per = (n + t - 1) / t;
    for (int i = 0; i < t; ++i) {
        const int64_t b = std::min<int64_t>(n, (int64_t)i * per), e = std::min<int64_t>(n, b + per);
        if (b < e) th.emplace_back([=] { fn(b, e); });
    }
    for (auto &x : th) x.join();
}

// Argument validation of qbh_csr_create, before anything touches the device (the reference asserts / exits at
// construction, src/sparse.cc:202-256): ia monotone, columns in range, col >= row for Hermitian-upper storage.
int validate_host_csr(int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia, const int64_t *ja)
{
    if (ia[0] != 0 || ia[dim] != nnz) {
        set_error("qbh_csr_create: ia[0] must be 0 and ia[dim] must equal nnz (zero-based CSR)");
        return QBH_EINVAL;
    }
    std::atomic<int64_t> bad_row{-1}, bad_kind{0}, bad_col{0};
    parallel_slices(dim, 4096, [&](int64_t b, int64_t e) {
        for (int64_t r = b; r < e && bad_row.load(std::memory_order_relaxed) < 0; ++r) {
            if (ia[r + 1] < ia[r] || ia[r] < 0 || ia[r + 1] > nnz) {
                bad_kind = 1;
                bad_row = r;
                return;
            }
            for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
                const int64_t c = ja[p];
                if (c < 0 || c >= dim || (sym_upper && c < r)) {
                    bad_kind = 2;
                    bad_col = c;
                    bad_row = r;
                    return;
                }
            }
        }
    });
    if (bad_row.load() >= 0) {
        if (bad_kind.load() == 1) set_error("qbh_csr_create: ia not monotone at row %lld", (long long)bad_row.load());
        else set_error("qbh_csr_create: bad column %lld in row %lld", (long long)bad_col.load(), (long long)bad_row.load());
        return QBH_EINVAL;
    }
    return QBH_OK;
}

// src/sparse.cc:235-256: every (r,c) of a full-storage matrix needs (c,r) == conj within sparse_precision
int check_hermitian_host(int64_t dim, const int64_t *ia, const int64_t *ja, const d2 *hv)
{
    std::atomic<int64_t> bad_r{-1}, bad_c{-1};
    parallel_slices(dim, 2048, [&](int64_t b, int64_t e) {
        for (int64_t r = b; r < e && bad_r.load(std::memory_order_relaxed) < 0; ++r)
            for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
                const int64_t c = ja[p];
                if (c == r) continue;
                const int64_t *lo = std::lower_bound(ja + ia[c], ja + ia[c + 1], r);
                int64_t q = lo - ja;
                if (q == ia[c + 1] || ja[q] != r) {           // unsorted row: linear search
                    for (q = ia[c]; q < ia[c + 1] && ja[q] != r; ++q) {}
                }
                if (q == ia[c + 1] || std::hypot(hv[p].x - hv[q].x, hv[p].y + hv[q].y) > QBH_SPARSE_PRECISION) {
                    bad_c = c;
                    bad_r = r;
                    return;
                }
            }
    });
    if (bad_r.load() >= 0) {
        set_error("Hermitian check failed at (row, col) = (%lld, %lld)", (long long)bad_r.load(), (long long)bad_c.load());
        return QBH_ENOTHERM;
    }
    return QBH_OK;
}

// ------------------------------------------------------------------ kernels -----
namespace {

// row of global nonzero index g: the last row whose pointer is <= g, searched in ia[lo .. hi] (ia is rebased: ia[i]
// belongs to host row row0 + i)
__device__ __forceinline__ int64_t row_of(const int64_t *ia, int64_t lo, int64_t hi, int64_t g)
{
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ia[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// chunk = nonzeros [g0, g0 + n) of the host matrix, columns already narrowed to int32.  rlo / rhi bracket the rows the
// chunk touches (indices into the rebased ia).
__global__ __launch_bounds__(kBlock) void k_mirror_count(const int32_t *ja, int64_t g0, int n, const int64_t *ia, int64_t rlo,
                                                         int64_t rhi, int64_t row0, int64_t r0, int64_t r1, int32_t *lowcnt)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t r = row0 + row_of(ia, rlo, rhi, g0 + i);
    const int64_t c = ja[i];
    if (c != r && c >= r0 && c < r1) atomicAdd(&lowcnt[c - r0], 1);
}

__global__ __launch_bounds__(kBlock) void k_row_total(const int64_t *ia_own, const int32_t *lowcnt, int64_t nloc, int32_t *cnt)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < nloc; r += stride)
        cnt[r] = (lowcnt ? lowcnt[r] : 0) + (int32_t)(ia_own[r + 1] - ia_own[r]);
}

__global__ __launch_bounds__(kBlock) void k_fill(const int32_t *ja, const d2 *val, int64_t g0, int n, const int64_t *ia, int64_t rlo,
                                                 int64_t rhi, int64_t row0, int64_t r0, int64_t r1, int sym, const int64_t *ia_f,
                                                 const int32_t *lowcnt, int32_t *cursor, int32_t *ja_f, d2 *val_f)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t g = g0 + i;
    const int64_t ri = row_of(ia, rlo, rhi, g);
    const int64_t r = row0 + ri;
    const int32_t c = ja[i];
    const d2 v = val[i];
    if (r >= r0 && r < r1) {
        const int64_t dst = ia_f[r - r0] + (lowcnt ? lowcnt[r - r0] : 0) + (g - ia[ri]);
        ja_f[dst] = c;
        val_f[dst] = v;
    }
    if (sym && c != r && c >= r0 && c < r1) {
        const int64_t dst = ia_f[c - r0] + atomicAdd(&cursor[c - r0], 1);
        ja_f[dst] = (int32_t)r;
        val_f[dst] = d2{v.x, -v.y};
    }
}

// ascending columns inside the mirrored (lower) part of every row.  Short segments: one lane, insertion sort in place.
__global__ __launch_bounds__(kBlock) void k_sort_lower_short(const int64_t *ia_f, const int32_t *lowcnt, int64_t nloc, int32_t *ja_f,
                                                             d2 *val_f, int cap)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < nloc; r += stride) {
        const int L = lowcnt[r];
        if (L < 2 || L > cap) continue;
        int32_t *c = ja_f + ia_f[r];
        d2 *v = val_f + ia_f[r];
        for (int i = 1; i < L; ++i) {
            const int32_t ck = c[i];
            const d2 vk = v[i];
            int j = i - 1;
            while (j >= 0 && c[j] > ck) {
                c[j + 1] = c[j];
                v[j + 1] = v[j];
                --j;
            }
            c[j + 1] = ck;
            v[j + 1] = vk;
        }
    }
}

// long segments (> cap): one workgroup per row, rank sort through a scratch copy (rank = number of smaller columns)
__global__ __launch_bounds__(kBlock) void k_sort_lower_long(const int64_t *ia_f, const int32_t *lowcnt, const int64_t *rows, int64_t nrows_long,
                                                            int32_t *ja_f, d2 *val_f, int32_t *tmp_c, d2 *tmp_v, const int64_t *tmp_off)
{
    for (int64_t k = blockIdx.x; k < nrows_long; k += gridDim.x) {
        const int64_t r = rows[k];
        const int L = lowcnt[r];
        int32_t *c = ja_f + ia_f[r];
        d2 *v = val_f + ia_f[r];
        int32_t *tc = tmp_c + tmp_off[k];
        d2 *tv = tmp_v + tmp_off[k];
        for (int i = threadIdx.x; i < L; i += kBlock) {
            tc[i] = c[i];
            tv[i] = v[i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L; i += kBlock) {
            const int32_t ci = tc[i];
            int rank = 0;
            for (int j = 0; j < L; ++j) rank += tc[j] < ci ? 1 : 0;
            c[rank] = ci;
            v[rank] = tv[i];
        }
        __syncthreads();
    }
}

double wall_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// two pinned staging buffers + two device chunk buffers; the host threads fill buffer b while the copy / kernel of
// buffer b^1 run
struct Stager {
    int64_t   cap = 0;
    int32_t  *h_ja[2] = {nullptr, nullptr}, *d_ja[2] = {nullptr, nullptr};
    d2       *h_val[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool      pending[2] = {false, false};
    ~Stager()
    {
        for (int b = 0; b < 2; ++b) {
            if (h_ja[b]) (void)hipHostFree(h_ja[b]);
            if (h_val[b]) (void)hipHostFree(h_val[b]);
            if (d_ja[b]) (void)hipFree(d_ja[b]);
            if (d_val[b]) (void)hipFree(d_val[b]);
            if (done[b]) (void)hipEventDestroy(done[b]);
        }
    }
    int init(int64_t chunk, bool with_val)
    {
        cap = chunk;
        for (int b = 0; b < 2; ++b) {
            QBH_HIP(hipHostMalloc(&h_ja[b], (size_t)chunk * sizeof(int32_t)));
            QBH_HIP(qbh::dev_alloc(&d_ja[b], (size_t)chunk * sizeof(int32_t)));
            if (with_val) {
                QBH_HIP(hipHostMalloc(&h_val[b], (size_t)chunk * sizeof(d2)));
                QBH_HIP(qbh::dev_alloc(&d_val[b], (size_t)chunk * sizeof(d2)));
            }
            QBH_HIP(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        }
        return QBH_OK;
    }
};

}  // namespace

// Rows [r0, r1) of the full operator from the host CSR (Hermitian-upper when sym != 0).  Outputs are hipMalloc'ed.
int build_shard_from_host(int64_t dim, int64_t nnz, int sym, const int64_t *ia, const int64_t *ja, const d2 *val, int64_t r0,
                          int64_t r1, hipStream_t s, int64_t **d_ia_out, int32_t **d_ja_out, d2 **d_val_out, int64_t *nnz_out,
                          double *ms_out)
{
    const double t_begin = wall_ms();
    const int64_t nloc = r1 - r0;
    // host rows whose entries can land in the shard: with Hermitian-upper storage (col >= row) rows [0, r1)
    const int64_t row0 = sym ? 0 : r0;
    const int64_t g_begin = ia[row0], g_end = ia[r1];
    const int64_t n_ia = r1 - row0 + 1;

    int64_t *d_ia = nullptr, *d_ia_f = nullptr;
    int32_t *d_low = nullptr, *d_cnt = nullptr, *d_cur = nullptr, *d_ja_f = nullptr;
    d2 *d_val_f = nullptr;
    auto cleanup = [&](bool all) {
        if (d_ia) (void)hipFree(d_ia);
        if (d_low) (void)hipFree(d_low);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_cur) (void)hipFree(d_cur);
        if (all) {
            if (d_ia_f) (void)hipFree(d_ia_f);
            if (d_ja_f) (void)hipFree(d_ja_f);
            if (d_val_f) (void)hipFree(d_val_f);
        }
    };
#define QBH_B(call)                                                                           \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            cleanup(true);                                                                    \
            return (_e == hipErrorOutOfMemory) ? QBH_ENOMEM : QBH_EHIP;                       \
        }                                                                                     \
    } while (0)
#define QBH_BT(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != QBH_OK) {          \
            cleanup(true);            \
            return _rc;               \
        }                             \
    } while (0)

    // the row pointers of the host rows involved (pageable -> device; 8 B per row, small next to the matrix)
    QBH_B(qbh::dev_alloc(&d_ia, (size_t)n_ia * sizeof(int64_t)));
    QBH_B(hipMemcpyAsync(d_ia, ia + row0, (size_t)n_ia * sizeof(int64_t), hipMemcpyHostToDevice, s));
    const int64_t *d_ia_own = d_ia + (r0 - row0);        // pointers of the shard's own rows

    // staging chunk: 4 M nonzeros for large matrices; small ones use ~1/8 of their range so that pinning the staging
    // buffers (the fixed cost of a small create) stays cheap and the upload still pipelines
    int64_t chunk = std::min<int64_t>(4 << 20, std::max<int64_t>(256 << 10, (g_end - g_begin + 7) / 8));
    if (debug_sw().create_chunk > 0) chunk = std::max<int64_t>(1024, debug_sw().create_chunk);
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(g_end - g_begin, 1));
    const bool trace = debug_sw().trace_create != 0;
    double t_mark = wall_ms();
    auto mark = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        const double t = wall_ms();
        fprintf(stderr, "[qbh_csr_create] %-22s %8.2f ms\n", what, t - t_mark);
        t_mark = t;
    };
    Stager st;
    QBH_BT(st.init(chunk, true));
    mark("staging buffers");

    // one streaming pass over the host nonzeros [g_begin, g_end): stage(b, g, n) fills the pinned buffers, launch(b, g, n)
    // enqueues copy + kernel
    auto stream_pass = [&](bool with_val, auto launch) -> int {
        int64_t k = 0;
        for (int64_t g = g_begin; g < g_end; g += chunk, ++k) {
            const int b = (int)(k & 1);
            const int n = (int)std::min<int64_t>(chunk, g_end - g);
            if (st.pending[b]) {
                hipError_t e = hipEventSynchronize(st.done[b]);
                if (e != hipSuccess) {
                    set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
                    return QBH_EHIP;
                }
                st.pending[b] = false;
            }
            int32_t *hj = st.h_ja[b];
            d2 *hv = st.h_val[b];
            parallel_slices(n, 1 << 16, [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) hj[i] = (int32_t)ja[g + i];
                if (with_val) std::memcpy(hv + lo, val + g + lo, (size_t)(hi - lo) * sizeof(d2));
            });
            QBH_HIP(hipMemcpyAsync(st.d_ja[b], hj, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
            if (with_val) QBH_HIP(hipMemcpyAsync(st.d_val[b], hv, (size_t)n * sizeof(d2), hipMemcpyHostToDevice, s));
            // rows touched by this chunk (host binary search on ia; brackets the device search)
            const int64_t rlo = (std::upper_bound(ia + row0, ia + r1 + 1, g) - (ia + row0)) - 1;
            const int64_t rhi = (std::upper_bound(ia + row0, ia + r1 + 1, g + n - 1) - (ia + row0));
            launch(b, g, n, std::max<int64_t>(rlo, 0), std::min<int64_t>(rhi, n_ia - 1));
            QBH_HIP(hipGetLastError());
            QBH_HIP(hipEventRecord(st.done[b], s));
            st.pending[b] = true;
        }
        QBH_HIP(hipStreamSynchronize(s));
        st.pending[0] = st.pending[1] = false;
        return QBH_OK;
    };

    if (sym) {
        QBH_B(qbh::dev_alloc(&d_low, (size_t)nloc * sizeof(int32_t)));
        QBH_B(hipMemsetAsync(d_low, 0, (size_t)nloc * sizeof(int32_t), s));
        QBH_BT(stream_pass(false, [&](int b, int64_t g, int n, int64_t rlo, int64_t rhi) {
            hipLaunchKernelGGL(k_mirror_count, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, st.d_ja[b], g, n, d_ia, rlo, rhi,
                               row0, r0, r1, d_low);
        }));
    }
    mark("pass 1 (mirror count)");
    QBH_B(qbh::dev_alloc(&d_cnt, (size_t)nloc * sizeof(int32_t)));
    hipLaunchKernelGGL(k_row_total, dim3(blas_grid(nloc)), dim3(kBlock), 0, s, d_ia_own, d_low, nloc, d_cnt);
    QBH_B(hipGetLastError());
    QBH_B(qbh::dev_alloc(&d_ia_f, (size_t)(nloc + 1) * sizeof(int64_t)));
    QBH_BT(exclusive_scan(d_cnt, nloc, d_ia_f, s));
    int64_t nnz_f = 0;
    QBH_B(hipMemcpy(&nnz_f, d_ia_f + nloc, sizeof(int64_t), hipMemcpyDeviceToHost));
    (void)hipFree(d_cnt);
    d_cnt = nullptr;
    QBH_B(qbh::dev_alloc(&d_ja_f, std::max<size_t>((size_t)nnz_f, 1) * sizeof(int32_t)));
    QBH_B(qbh::dev_alloc(&d_val_f, std::max<size_t>((size_t)nnz_f, 1) * sizeof(d2)));
    if (sym) {
        QBH_B(qbh::dev_alloc(&d_cur, (size_t)nloc * sizeof(int32_t)));
        QBH_B(hipMemsetAsync(d_cur, 0, (size_t)nloc * sizeof(int32_t), s));
    }
    QBH_BT(stream_pass(true, [&](int b, int64_t g, int n, int64_t rlo, int64_t rhi) {
        hipLaunchKernelGGL(k_fill, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, st.d_ja[b], st.d_val[b], g, n, d_ia, rlo, rhi,
                           row0, r0, r1, sym, d_ia_f, d_low, d_cur, d_ja_f, d_val_f);
    }));
    mark("scan + pass 2 (fill)");
    if (sym) {
        constexpr int kShortCap = 96;
        hipLaunchKernelGGL(k_sort_lower_short, dim3(blas_grid(nloc)), dim3(kBlock), 0, s, d_ia_f, d_low, nloc, d_ja_f, d_val_f, kShortCap);
        QBH_B(hipGetLastError());
        // rows with a long mirrored part are rare in this domain (a Hamiltonian row holds tens of entries): list them on
        // the host and rank-sort them one workgroup per row
        std::vector<int32_t> low((size_t)nloc);
        QBH_B(hipMemcpyAsync(low.data(), d_low, (size_t)nloc * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        QBH_B(hipStreamSynchronize(s));
        std::vector<int64_t> rows, off;
        int64_t tot = 0;
        for (int64_t r = 0; r < nloc; ++r)
            if (low[(size_t)r] > kShortCap) {
                rows.push_back(r);
                off.push_back(tot);
                tot += low[(size_t)r];
            }
        if (!rows.empty()) {
            int64_t *d_rows = nullptr, *d_off = nullptr;
            int32_t *tc = nullptr;
            d2 *tv = nullptr;
            QBH_B(qbh::dev_alloc(&d_rows, rows.size() * sizeof(int64_t)));
            QBH_B(qbh::dev_alloc(&d_off, rows.size() * sizeof(int64_t)));
            QBH_B(qbh::dev_alloc(&tc, (size_t)tot * sizeof(int32_t)));
            QBH_B(qbh::dev_alloc(&tv, (size_t)tot * sizeof(d2)));
            QBH_B(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            QBH_B(hipMemcpy(d_off, off.data(), rows.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_sort_lower_long, dim3((unsigned)std::min<size_t>(rows.size(), 4096)), dim3(kBlock), 0, s, d_ia_f, d_low,
                               d_rows, (int64_t)rows.size(), d_ja_f, d_val_f, tc, tv, d_off);
            hipError_t e = hipStreamSynchronize(s);
            (void)hipFree(d_rows);
            (void)hipFree(d_off);
            (void)hipFree(tc);
            (void)hipFree(tv);
            QBH_B(e);
        }
    }
    QBH_B(hipStreamSynchronize(s));
    mark("sort mirrored parts");
    cleanup(false);
#undef QBH_B
#undef QBH_BT
    *d_ia_out = d_ia_f;
    *d_ja_out = d_ja_f;
    *d_val_out = d_val_f;
    *nnz_out = nnz_f;
    if (ms_out) *ms_out = wall_ms() - t_begin;
    return QBH_OK;
}

// Row cuts balanced by the nonzeros of the FULL operator (momentum sectors hold decoupled one-entry rows, so
// uniform row blocks are not nnz-balanced): full row lengths on the host (own entries + mirrored ones), prefix sums,
// cut p at the first row where the running count reaches p/nranks of the total.
int balanced_row_cuts(int64_t dim, int64_t nnz, int sym, const int64_t *ia, const int64_t *ja, int nranks, int64_t *cuts)
{
    (void)nnz;
    std::vector<std::atomic<int32_t>> extra(sym ? (size_t)dim : 0);
    if (sym) {
        for (auto &x : extra) x.store(0, std::memory_order_relaxed);
        parallel_slices(dim, 4096, [&](int64_t b, int64_t e) {
            for (int64_t r = b; r < e; ++r)
                for (int64_t p = ia[r]; p < ia[r + 1]; ++p)
                    if (ja[p] != r) extra[(size_t)ja[p]].fetch_add(1, std::memory_order_relaxed);
        });
    }
    int64_t total = 0;
    std::vector<int64_t> pre((size_t)dim + 1);
    pre[0] = 0;
    for (int64_t r = 0; r < dim; ++r) {
        total += (ia[r + 1] - ia[r]) + (sym ? extra[(size_t)r].load(std::memory_order_relaxed) : 0);
        pre[(size_t)r + 1] = total;
    }
    cuts[0] = 0;
    for (int p = 1; p < nranks; ++p) {
        const int64_t want = (int64_t)((double)total * p / nranks);
        int64_t r = std::lower_bound(pre.begin(), pre.end(), want) - pre.begin();
        r = std::max<int64_t>(r, cuts[p - 1]);
        cuts[p] = std::min<int64_t>(r, dim);
    }
    cuts[nranks] = dim;
    return QBH_OK;
}

}  // namespace qbh
