// qbh_reorder.hip -- a device-generated benchmark operator re-expressed in the REFERENCE's basis order and fermion
// convention, on the device (measurement harness, SURVEY 7 hard-part 1: the CSR kernel must be measured on the order the
// unchanged reference host code would hand over, at the sizes that code cannot assemble).
//
// The reference sorts its basis by (sub_b, sub_a): the odd sites compacted into one integer, then the even sites
// (sort_basis_Lin_order, src/basis.cc:1144-1190; split unzipper_basis, :971-996); a state's row index is its position in
// that order (j = Lin_Ja[i_a] + Lin_Jb[i_b], src/model.cc:665-670).  Local states (src/basis.cc:52-83): spin-1/2 one bit
// per site (1 = down); electron two bits per site (bit 0 up, bit 1 down), fermion operators ordered by site
// (src/basis.cc:2650-2664).  The generators (qbh_gen.hip) index by colexicographic rank and put all up operators before
// all down operators, so H_ref = P D H_gen D P^T with a permutation P and a diagonal of signs D.
//
//   keys     one 64-bit key (sub_b, sub_a) per generator index          k_ref_keys
//   sort     radix sort of (key, generator index) pairs                  hipcub::DeviceRadixSort
//   pos      position of every generator index in the sorted order, sign in bit 31
//   fill     one thread per reference row: columns mapped through pos, ranked inside the row through a transposed LDS tile
//            (rows are <= 64 entries for every benchmark family; longer rows take an insertion sort in place)
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "qbh_internal.hpp"

namespace qbh {

namespace {

struct RefOrderArgs {
    int kind;          // 0 spin-1/2 (one pattern of n_dn bits), 1 electron (n_up, n_dn patterns)
    int n_sites, n_up, n_dn;
    int64_t dim, n_minor;              // n_minor = C(n_sites, n_dn): generator index = rank_up * n_minor + rank_dn
    const uint64_t *binom;             // [33 * 33] C(p, k) at p * 33 + k (device)
};

__device__ __forceinline__ uint32_t colex_unrank(const RefOrderArgs &a, int64_t r, int k)
{
    uint32_t bits = 0;
    for (int p = a.n_sites - 1; p >= 0 && k > 0; --p) {
        const int64_t c = (int64_t)a.binom[p * 33 + k];
        if (c <= r) {
            bits |= 1u << p;
            r -= c;
            --k;
        }
    }
    return bits;
}

// sites of one parity compacted (bps bits per site)
__device__ __forceinline__ uint64_t compact_sites(uint64_t word, int n_sites, int bps, int parity)
{
    uint64_t out = 0;
    const uint64_t mask = (1ull << bps) - 1;
    int k = 0;
    for (int s = parity; s < n_sites; s += 2, ++k) out |= ((word >> (s * bps)) & mask) << (k * bps);
    return out;
}

__global__ __launch_bounds__(256) void k_ref_keys(RefOrderArgs a, uint64_t *keys, int32_t *vals, uint8_t *sign)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.dim; g += (int64_t)gridDim.x * blockDim.x) {
        uint64_t word;
        int bps;
        uint8_t sg = 0;
        if (a.kind == 0) {
            word = colex_unrank(a, g, a.n_dn);
            bps = 1;
        } else {
            const uint32_t up = colex_unrank(a, g / a.n_minor, a.n_up), dn = colex_unrank(a, g % a.n_minor, a.n_dn);
            word = 0;
            for (int s = 0; s < a.n_sites; ++s) word |= ((uint64_t)((up >> s) & 1) << (2 * s)) | ((uint64_t)((dn >> s) & 1) << (2 * s + 1));
            bps = 2;
            // all-up-then-all-down ordering -> site ordering: every down operator at site i passes the up operators at sites j > i
            int swaps = 0;
            for (int i = 0; i < a.n_sites; ++i)
                if ((dn >> i) & 1) swaps += __popc(up >> (i + 1));
            sg = (uint8_t)(swaps & 1);
        }
        const int na = (a.n_sites + 1) / 2;                          // even sites
        const uint64_t sub_a = compact_sites(word, a.n_sites, bps, 0), sub_b = compact_sites(word, a.n_sites, bps, 1);
        keys[g] = (sub_b << (na * bps)) | sub_a;
        vals[g] = (int32_t)g;
        sign[g] = sg;
    }
}

__global__ __launch_bounds__(256) void k_ref_pos(const int32_t *order, const uint8_t *sign, int64_t dim, uint32_t *pos, const int64_t *ia,
                                                 int32_t *cnt)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < dim; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = order[r];
        pos[g] = (uint32_t)r | ((uint32_t)sign[g] << 31);
        cnt[r] = (int32_t)(ia[g + 1] - ia[g]);
    }
}

constexpr int kRefMaxRow = 64;

__global__ __launch_bounds__(256) void k_ref_fill(const int32_t *order, const uint32_t *pos, int64_t dim, const int64_t *ia, const int32_t *ja,
                                                  const d2 *val, const int64_t *ia_r, int32_t *ja_r, d2 *val_r)
{
    __shared__ uint32_t tile[kRefMaxRow * 256];                      // element k of thread t at [k * 256 + t]: conflict-free
    const int t = threadIdx.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + t; r < dim; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = order[r];
        const uint32_t pg = pos[g];
        const int64_t src = ia[g], dst = ia_r[r];
        const int len = (int)(ia[g + 1] - src);
        if (len <= kRefMaxRow) {
            for (int k = 0; k < len; ++k) tile[k * 256 + t] = pos[ja[src + k]];
            for (int k = 0; k < len; ++k) {
                const uint32_t mine = tile[k * 256 + t], mc = mine & 0x7FFFFFFFu;
                int rank = 0;
                for (int j = 0; j < len; ++j) rank += (tile[j * 256 + t] & 0x7FFFFFFFu) < mc;
                const d2 v = val[src + k];
                ja_r[dst + rank] = (int32_t)mc;
                val_r[dst + rank] = ((mine ^ pg) >> 31) ? -v : v;
            }
        } else {
            for (int k = 0; k < len; ++k) {                          // insertion sort in place (long rows: not a benchmark case)
                const uint32_t p = pos[ja[src + k]];
                const int32_t c = (int32_t)(p & 0x7FFFFFFFu);
                d2 v = val[src + k];
                if ((p ^ pg) >> 31) v = -v;
                int64_t q = dst + k;
                while (q > dst && ja_r[q - 1] > c) {
                    ja_r[q] = ja_r[q - 1];
                    val_r[q] = val_r[q - 1];
                    --q;
                }
                ja_r[q] = c;
                val_r[q] = v;
            }
        }
    }
}

// inverse direction (qbh_opts.basis_kind): reference row r holds generator index g = order[r]
__global__ __launch_bounds__(256) void k_inv_maps(const int32_t *order, const uint8_t *sign, int64_t dim, int32_t *inv_order, uint32_t *map,
                                                  const int64_t *ia_ref, int32_t *cnt_new)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < dim; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = order[r];
        inv_order[g] = (int32_t)r;
        map[r] = (uint32_t)g | ((uint32_t)sign[g] << 31);
        cnt_new[g] = (int32_t)(ia_ref[r + 1] - ia_ref[r]);
    }
}

// Does the order a hint describes give the operator the product structure?  Asked of the ORIGINAL arrays through the map --
// one pass over the columns with a 4-byte gather each -- BEFORE the operator is permuted (a full second copy): a wrong hint, or a
// candidate of the library's own search (qbh_opts.basis_detect), costs milliseconds instead of a permutation.  A wavefront per
// 64 rows, lanes striding the row: coalesced column reads.
__global__ __launch_bounds__(256) void k_ref_precheck(const uint32_t *map, int64_t dim, const int64_t *ia, const int32_t *ja, int64_t S, int *flag)
{
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (int64_t r0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; r0 < dim; r0 += (((int64_t)gridDim.x * blockDim.x) >> 6) * 64) {
        if (*reinterpret_cast<volatile int *>(flag)) return;               // somebody found a counter-example already
        const int64_t r1 = r0 + 64 < dim ? r0 + 64 : dim;
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t g = map[r] & 0x7FFFFFFFu, gm = g / S, gd = g - gm * S;
            for (int64_t k = ia[r] + lane; k < ia[r + 1]; k += 64) {
                const int64_t c = map[ja[k]] & 0x7FFFFFFFu, cm = c / S;
                bad = bad || (cm != gm && c - cm * S != gd);
            }
        }
        if (bad) break;
    }
    if (bad) *flag = 1;
}

// QBH_BASIS_SPIN_SECTOR: caller index r = rank of the n_sites-bit pattern with n_dn bits set; internal index = class-major
// (class = bits set among the high sites): rbase[c] + rank(high part) * S[c] + rank(low part)
struct SpinCut {
    int n_sites, n_dn, h, p_min, nc;
    int64_t rbase[kKronMaxClasses + 1], S[kKronMaxClasses];
    const uint64_t *binom;             // [33 * 33]
};
__device__ __forceinline__ int64_t colex_rank(uint32_t bits, const uint64_t *binom)
{
    int64_t r = 0;
    int j = 0;
    while (bits) {
        const int p = __ffs(bits) - 1;
        bits &= bits - 1;
        ++j;
        r += (int64_t)binom[p * 33 + j];
    }
    return r;
}
__global__ __launch_bounds__(256) void k_spin_sector_map(SpinCut a, int64_t dim, int32_t *inv_order, uint32_t *map, const int64_t *ia_ref, int32_t *cnt_new)
{
    RefOrderArgs ua{};
    ua.n_sites = a.n_sites;
    ua.binom = a.binom;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < dim; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t pat = colex_unrank(ua, r, a.n_dn);
        const uint32_t lp = pat & ((1u << a.h) - 1u), hp = pat >> a.h;
        const int c = __popc(hp) - a.p_min;
        const int64_t g = a.rbase[c] + colex_rank(hp, a.binom) * a.S[c] + colex_rank(lp, a.binom);
        inv_order[g] = (int32_t)r;
        map[r] = (uint32_t)g;
        cnt_new[g] = (int32_t)(ia_ref[r + 1] - ia_ref[r]);
    }
}

// vectors between the caller's order (index r) and the internal one (map[r] = internal index | sign << 31)
__global__ __launch_bounds__(256) void k_basis_scatter(const uint32_t *map, const d2 *in, d2 *out, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = map[r];
        const d2 v = in[r];
        out[m & 0x7FFFFFFFu] = (m >> 31) ? -v : v;
    }
}
__global__ __launch_bounds__(256) void k_basis_gather(const uint32_t *map, const d2 *in, d2 *out, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = map[r];
        const d2 v = in[m & 0x7FFFFFFFu];
        out[r] = (m >> 31) ? -v : v;
    }
}

__global__ __launch_bounds__(256) void k_basis_scatter_re(const uint32_t *map, const double *in, double *out, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = map[r];
        const double v = in[r];
        out[m & 0x7FFFFFFFu] = (m >> 31) ? -v : v;
    }
}

}  // namespace

int launch_basis_scatter_re(const uint32_t *map, const double *in, double *out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_basis_scatter_re, dim3(blas_grid(n)), dim3(256), 0, s, map, in, out, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_basis_scatter(const uint32_t *map, const d2 *in, d2 *out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_basis_scatter, dim3(blas_grid(n)), dim3(256), 0, s, map, in, out, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_basis_gather(const uint32_t *map, const d2 *in, d2 *out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_basis_gather, dim3(blas_grid(n)), dim3(256), 0, s, map, in, out, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

}  // namespace qbh

#define RO_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            qbh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
            (void)hipGetLastError();                                                                      \
            return drop(e_ == hipErrorOutOfMemory ? QBH_ENOMEM : QBH_EHIP);                               \
        }                                                                                                 \
    } while (0)

extern "C" int qbh_csr_reference_order(qbh_csr **out, const qbh_csr *A, int kind, int n_sites, int n_up, int n_dn, const qbh_opts *opts)
{
    using namespace qbh;
    if (!out || !A || (kind != 0 && kind != 1) || n_sites < 1 || n_sites > 32 || n_dn < 0 || n_dn > n_sites || n_up < 0 || n_up > n_sites) {
        set_error("qbh_csr_reference_order: invalid argument");
        return QBH_EINVAL;
    }
    if (A->kind != 0 || A->d_val == nullptr || A->has_rem || A->nrows != A->ncols || A->row_offset != 0 || A->kron.active) {
        set_error("qbh_csr_reference_order: needs an unsharded stored CSR with complex128 values (create it with value_dict = 0, kron_split = 0)");
        return QBH_EUNSUPP;
    }
    // everything below (scratch, kernels on A->stream, the new handle) lives on A's device, whatever the caller's current one is
    struct DevGuard {
        int prev = -1;
        explicit DevGuard(int dev)
        {
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
            (void)hipSetDevice(dev);
        }
        ~DevGuard()
        {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } dev_guard(A->device);
    qbh_opts opts_here;
    if (opts) opts_here = *opts;
    else qbh::opts_builtin(&opts_here);
    opts_here.device = A->device;
    opts = &opts_here;
    RefOrderArgs a{};
    a.kind = kind;
    a.n_sites = n_sites;
    a.n_up = n_up;
    a.n_dn = n_dn;
    std::vector<uint64_t> hb(33 * 33, 0);
    for (int p = 0; p <= 32; ++p)
        for (int k = 0; k <= 32; ++k) hb[(size_t)p * 33 + k] = (k == 0) ? 1 : (p == 0 ? 0 : hb[(size_t)(p - 1) * 33 + k - 1] + hb[(size_t)(p - 1) * 33 + k]);
    a.n_minor = (int64_t)hb[(size_t)n_sites * 33 + n_dn];
    a.dim = kind == 0 ? a.n_minor : (int64_t)hb[(size_t)n_sites * 33 + n_up] * a.n_minor;
    if (a.dim != A->nrows) {
        set_error("qbh_csr_reference_order: the operator has %lld rows, the basis described has %lld", (long long)A->nrows, (long long)a.dim);
        return QBH_EINVAL;
    }
    const int64_t dim = a.dim, nnz = A->nnz;
    hipStream_t s = A->stream;
    uint64_t *k0 = nullptr, *k1 = nullptr;
    int32_t *v0 = nullptr, *v1 = nullptr, *cnt = nullptr, *ja_r = nullptr;
    uint8_t *sign = nullptr;
    uint32_t *pos = nullptr;
    void *tmp = nullptr;
    uint64_t *d_binom = nullptr;
    int64_t *ia_r = nullptr;
    d2 *val_r = nullptr;
    auto drop = [&](int code) {
        for (void *q : {(void *)k0, (void *)k1, (void *)v0, (void *)v1, (void *)cnt, (void *)sign, (void *)pos, tmp, (void *)d_binom})
            if (q) (void)hipFree(q);
        if (code != QBH_OK)
            for (void *q : {(void *)ia_r, (void *)ja_r, (void *)val_r})
                if (q) (void)hipFree(q);
        return code;
    };
    RO_HIP(qbh::dev_alloc(&d_binom, hb.size() * 8));
    RO_HIP(hipMemcpy(d_binom, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    a.binom = d_binom;
    RO_HIP(qbh::dev_alloc(&k0, (size_t)dim * 8));
    RO_HIP(qbh::dev_alloc(&k1, (size_t)dim * 8));
    RO_HIP(qbh::dev_alloc(&v0, (size_t)dim * 4));
    RO_HIP(qbh::dev_alloc(&v1, (size_t)dim * 4));
    RO_HIP(qbh::dev_alloc(&sign, (size_t)dim));
    hipLaunchKernelGGL(k_ref_keys, dim3(2048), dim3(256), 0, s, a, k0, v0, sign);
    RO_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    const int key_bits = (kind == 0 ? 1 : 2) * n_sites;
    RO_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0, k1, v0, v1, dim, 0, key_bits, s));
    RO_HIP(qbh::dev_alloc(&tmp, tmp_bytes));
    RO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, dim, 0, key_bits, s));
    RO_HIP(hipStreamSynchronize(s));
    (void)hipFree(tmp);
    tmp = nullptr;
    (void)hipFree(k0);
    k0 = nullptr;
    (void)hipFree(k1);
    k1 = nullptr;
    (void)hipFree(v0);
    v0 = nullptr;
    RO_HIP(qbh::dev_alloc(&pos, (size_t)dim * 4));
    RO_HIP(qbh::dev_alloc(&cnt, (size_t)dim * 4));
    hipLaunchKernelGGL(k_ref_pos, dim3(2048), dim3(256), 0, s, v1, sign, dim, pos, A->d_ia, cnt);
    RO_HIP(hipGetLastError());
    RO_HIP(qbh::dev_alloc(&ia_r, (size_t)(dim + 1) * 8));
    {
        const int rc = exclusive_scan(cnt, dim, ia_r, s);
        if (rc != QBH_OK) return drop(rc);
    }
    RO_HIP(qbh::dev_alloc(&ja_r, (size_t)std::max<int64_t>(nnz, 1) * 4));
    RO_HIP(qbh::dev_alloc(&val_r, (size_t)std::max<int64_t>(nnz, 1) * 16));
    hipLaunchKernelGGL(k_ref_fill, dim3(4096), dim3(256), 0, s, v1, pos, dim, A->d_ia, A->d_ja, A->d_val, ia_r, ja_r, val_r);
    RO_HIP(hipGetLastError());
    RO_HIP(hipStreamSynchronize(s));
    (void)drop(QBH_OK);                       // scratch only; the three result arrays go to the new handle
    const int rc = qbh_csr_create_device(out, dim, dim, 0, nnz, ia_r, ja_r, reinterpret_cast<qbh_z *>(val_r), 1, opts);
    return rc;
}


// qbh_opts.basis_kind = QBH_BASIS_REF_FERMION2: the plain CSR of A is in the reference's order of a two-species fermion basis
// (rows sorted by (sub_b, sub_a), operators ordered by site: what model::generate_Ham_sparse_full hands over,
// src/model.cc:649-679, src/basis.cc:1144-1190).  Re-express it in species-major order -- H_int = D P^T H_ref P D, index =
// up * C(n_sites, n_dn) + down, all up operators before all down operators -- so that the Kronecker split applies, and keep the
// map for the vector seams.  The hint is CHECKED: the permuted operator must have the product structure, else everything is
// left as given (*applied = false).  Needs room for a second copy of the matrix while it runs.
int qbh::basis_to_internal(qbh_csr *A, int kind, int n_sites, int n_up, int n_dn, bool *applied)
{
    using namespace qbh;
    *applied = false;
    if ((kind != QBH_BASIS_REF_FERMION2 && kind != QBH_BASIS_SPIN_SECTOR) || n_sites < 2 || n_sites > 31 || n_up < 0 || n_up > n_sites || n_dn < 0 ||
        n_dn > n_sites)
        return QBH_OK;
    if (A->kind != 0 || !A->d_val || A->d_code || !A->own_arrays || A->has_rem || A->has_comm || A->kron.active || A->nrows != A->ncols ||
        A->row_offset != 0 || A->basis.kind != 0)
        return QBH_OK;
    RefOrderArgs a{};
    a.kind = 1;
    a.n_sites = n_sites;
    a.n_up = n_up;
    a.n_dn = n_dn;
    std::vector<uint64_t> hb(33 * 33, 0);
    for (int p = 0; p <= 32; ++p)
        for (int k = 0; k <= 32; ++k) hb[(size_t)p * 33 + k] = (k == 0) ? 1 : (p == 0 ? 0 : hb[(size_t)(p - 1) * 33 + k - 1] + hb[(size_t)(p - 1) * 33 + k]);
    a.n_minor = (int64_t)hb[(size_t)n_sites * 33 + n_dn];
    a.dim = (int64_t)hb[(size_t)n_sites * 33 + n_up] * a.n_minor;
    SpinCut sc{};
    KronMap classes{};
    if (kind == QBH_BASIS_SPIN_SECTOR) {
        // n_up carries the number of LOW sites of the cut (0: half of them); class = particles among the high sites
        a.dim = a.n_minor;
        const int h = n_up > 0 ? n_up : n_sites / 2;
        const int p_min = std::max(0, n_dn - h), p_max = std::min(n_sites - h, n_dn);
        if (h < 3 || h > n_sites - 1 || n_sites - h > 24 || p_max - p_min + 1 > kKronMaxClasses || p_max <= p_min) return QBH_OK;
        sc.n_sites = n_sites;
        sc.n_dn = n_dn;
        sc.h = h;
        sc.p_min = p_min;
        sc.nc = p_max - p_min + 1;
        classes.nc = sc.nc;
        classes.B = 8;
        classes.sliced = 1;
        classes.U0 = 0;
        for (int c = 0; c < sc.nc; ++c) {
            const int p = p_min + c;
            classes.S[c] = sc.S[c] = (int64_t)hb[(size_t)h * 33 + (n_dn - p)];
            classes.NU[c] = (int64_t)hb[(size_t)(n_sites - h) * 33 + p];
            classes.rbase[c + 1] = sc.rbase[c + 1] = sc.rbase[c] + classes.S[c] * classes.NU[c];
            classes.fbase[c + 1] = classes.fbase[c] + (classes.S[c] / 8) * 8 * classes.NU[c];
            if ((double)classes.NU[c] * 128.0 > 2.5e6) return QBH_OK;          // a band of the class's x must fit an XCD's L2
        }
        if (sc.rbase[sc.nc] != a.dim) return QBH_OK;
    }
    if (a.dim != A->nrows || (kind == QBH_BASIS_REF_FERMION2 && (a.n_minor < 2 || a.n_minor >= a.dim))) return QBH_OK;   // not the basis described: kept as given
    const int64_t dim = a.dim, nnz = A->nnz;
    hipStream_t s = A->stream;
    uint64_t *k0 = nullptr, *k1 = nullptr, *d_binom = nullptr;
    int32_t *v0 = nullptr, *v1 = nullptr, *cnt = nullptr, *ja_n = nullptr, *inv_order = nullptr;
    uint8_t *sign = nullptr;
    uint32_t *map = nullptr;
    void *tmp = nullptr;
    int64_t *ia_n = nullptr;
    d2 *val_n = nullptr;
    auto drop = [&](int code) {                // scratch and (unless adopted) results; out of memory = "kept as given", not an error
        for (void *q : {(void *)k0, (void *)k1, (void *)v0, (void *)v1, (void *)cnt, (void *)sign, tmp, (void *)d_binom, (void *)inv_order, (void *)map,
                        (void *)ia_n, (void *)ja_n, (void *)val_n})
            if (q) (void)hipFree(q);
        return code == QBH_ENOMEM ? QBH_OK : code;
    };
    RO_HIP(qbh::dev_alloc(&d_binom, hb.size() * 8));
    RO_HIP(hipMemcpy(d_binom, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    a.binom = d_binom;
    RO_HIP(qbh::dev_alloc(&inv_order, (size_t)dim * 4));
    RO_HIP(qbh::dev_alloc(&map, (size_t)dim * 4));
    RO_HIP(qbh::dev_alloc(&cnt, (size_t)dim * 4));
    if (kind == QBH_BASIS_REF_FERMION2) {
        RO_HIP(qbh::dev_alloc(&k0, (size_t)dim * 8));
        RO_HIP(qbh::dev_alloc(&k1, (size_t)dim * 8));
        RO_HIP(qbh::dev_alloc(&v0, (size_t)dim * 4));
        RO_HIP(qbh::dev_alloc(&v1, (size_t)dim * 4));
        RO_HIP(qbh::dev_alloc(&sign, (size_t)dim));
        hipLaunchKernelGGL(k_ref_keys, dim3(2048), dim3(256), 0, s, a, k0, v0, sign);
        RO_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        RO_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0, k1, v0, v1, dim, 0, 2 * n_sites, s));
        RO_HIP(qbh::dev_alloc(&tmp, tmp_bytes));
        RO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, dim, 0, 2 * n_sites, s));
        RO_HIP(hipStreamSynchronize(s));
        for (void **q : {(void **)&tmp, (void **)&k0, (void **)&k1, (void **)&v0}) {
            (void)hipFree(*q);
            *q = nullptr;
        }
        hipLaunchKernelGGL(k_inv_maps, dim3(2048), dim3(256), 0, s, v1, sign, dim, inv_order, map, A->d_ia, cnt);
        RO_HIP(hipGetLastError());
        // the hint is checked, not trusted -- and checked BEFORE the operator is permuted
        RO_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_ref_precheck, dim3(4096), dim3(256), 0, s, map, dim, A->d_ia, A->d_ja, a.n_minor, A->d_flag);
        RO_HIP(hipGetLastError());
        int bad = 0;
        RO_HIP(hipMemcpyAsync(&bad, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        RO_HIP(hipStreamSynchronize(s));
        RO_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        if (bad) return drop(QBH_OK);
    } else {
        sc.binom = d_binom;
        hipLaunchKernelGGL(k_spin_sector_map, dim3(2048), dim3(256), 0, s, sc, dim, inv_order, map, A->d_ia, cnt);
        RO_HIP(hipGetLastError());
    }
    RO_HIP(qbh::dev_alloc(&ia_n, (size_t)(dim + 1) * 8));
    {
        const int rc = exclusive_scan(cnt, dim, ia_n, s);
        if (rc != QBH_OK) return drop(rc);
    }
    RO_HIP(qbh::dev_alloc(&ja_n, (size_t)std::max<int64_t>(nnz, 1) * 4));
    RO_HIP(qbh::dev_alloc(&val_n, (size_t)std::max<int64_t>(nnz, 1) * 16));
    // row g of the internal operator = reference row inv_order[g]; column c_ref -> map[c_ref] (index | sign)
    hipLaunchKernelGGL(k_ref_fill, dim3(4096), dim3(256), 0, s, inv_order, map, dim, A->d_ia, A->d_ja, A->d_val, ia_n, ja_n, val_n);
    RO_HIP(hipGetLastError());
    if (kind == QBH_BASIS_REF_FERMION2) {
        // the hint is checked, not trusted: in the order it describes every entry keeps the up or the down configuration
        RO_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        {
            const int rc = launch_kron_check2(ia_n, ja_n, dim, a.n_minor, 0, A->d_flag, s);
            if (rc != QBH_OK) return drop(rc);
        }
        int bad = 0;
        RO_HIP(hipMemcpyAsync(&bad, A->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        RO_HIP(hipStreamSynchronize(s));
        RO_HIP(hipMemsetAsync(A->d_flag, 0, sizeof(int), s));
        if (bad) return drop(QBH_OK);
    } else {
        // nothing to check: whatever the matrix is, entries that keep neither the block nor the position inside it go to the third
        // (unstructured) part of the split, and kron_build gives the split up when that part is most of the operator
        RO_HIP(hipStreamSynchronize(s));
        A->basis.classes = classes;
    }
    (void)hipFree(A->d_ia);
    (void)hipFree(A->d_ja);
    (void)hipFree(A->d_val);
    A->d_ia = ia_n;
    A->d_ja = ja_n;
    A->d_val = val_n;
    ia_n = nullptr;
    ja_n = nullptr;
    val_n = nullptr;
    A->basis.kind = kind;
    A->basis.d_map = map;
    map = nullptr;
    if (kind == QBH_BASIS_REF_FERMION2) A->opts.kron_minor = a.n_minor;
    A->basis.n_sites = n_sites;
    A->basis.n_up = n_up;
    A->basis.n_dn = n_dn;
    *applied = true;
    return drop(QBH_OK);
}

// qbh_opts.basis_detect: host arrays arrive without a word about their basis (the reference's csr_mat(lil_mat&) carries no
// options, src/sparse.cc:202-260; its matrices are in Lin order, src/basis.cc:1144-1190).  Every (n_sites, n_up, n_dn) with
// C(n_sites, n_up) * C(n_sites, n_dn) = dim is tried as QBH_BASIS_REF_FERMION2 through the pre-check above; the first under
// which the operator has the product structure is taken.  Being right about the physics is not required: ANY order with the
// product structure is a valid internal order (vectors are translated with the same map), a matrix of a colliding dimension
// without it stays as given.
int qbh::basis_detect(qbh_csr *A, bool *applied)
{
    *applied = false;
    if (A->kind != 0 || !A->d_val || A->d_code || !A->own_arrays || A->has_rem || A->has_comm || A->kron.active || A->nrows != A->ncols ||
        A->row_offset != 0 || A->basis.kind != 0)
        return QBH_OK;
    const int64_t dim = A->nrows;
    std::vector<uint64_t> hb(33 * 33, 0);
    for (int p = 0; p <= 32; ++p)
        for (int k = 0; k <= 32; ++k) hb[(size_t)p * 33 + k] = (k == 0) ? 1 : (p == 0 ? 0 : hb[(size_t)(p - 1) * 33 + k - 1] + hb[(size_t)(p - 1) * 33 + k]);
    int tried = 0;
    for (int n = 2; n <= 31 && tried < 24; ++n)
        for (int nu = 1; nu < n && tried < 24; ++nu) {
            const uint64_t cu = hb[(size_t)n * 33 + nu];
            if (cu < 2 || (uint64_t)dim % cu != 0) continue;
            const uint64_t want = (uint64_t)dim / cu;
            for (int nd = 1; nd < n && tried < 24; ++nd) {
                if (hb[(size_t)n * 33 + nd] != want || want < 2) continue;
                ++tried;
                QBH_TRY(qbh::basis_to_internal(A, QBH_BASIS_REF_FERMION2, n, nu, nd, applied));
                if (*applied) {
                    A->basis.detected = true;
                    return QBH_OK;
                }
            }
        }
    return QBH_OK;
}
