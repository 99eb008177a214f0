// qbh_mopr.hip -- the operator-apply step of the dynamical correlations: counterpart of
// model<T>::moprXvec_full (src/model.cc:1468-1538), vec_new = A |vec_old>, for the operator families the device
// generators cover, in THEIR basis order:
//   spin-1/2 fixed-N_dn sectors (basis of qbh_gen_heisenberg: colexicographic rank of the down-spin pattern)
//       S^z_q = sum_s c_s S^z_s      same sector
//       S^-_q = sum_s c_s S^-_s      N_dn -> N_dn + 1   (local states of src/basis.cc:52-83: 0 = up, 1 = down)
//       S^+_q = sum_s c_s S^+_s      N_dn -> N_dn - 1
//   two-species fermions (basis of qbh_gen_hubbard: index = rank(up) * C(L, N_dn) + rank(down))
//       A = sum_k w_k c+_{a_k, s_k} c_{b_k, s_k}     any one-body number-conserving operator: densities n_q, S^z_q,
//                                                     hopping / current operators
// The reference scatters (one thread per source state j, a critical section per write, src/model.cc:1528-1532); here
// every TARGET row gathers its contributions, so there are no atomics and the result is deterministic.  The result then
// feeds lanczos(..., "dnmcs") (src/model.cc:1696-1712 measure_full_dynamic) without leaving HBM.
#include <algorithm>
#include <vector>

#include "qbh_internal.hpp"

namespace qbh {
namespace {

constexpr int kMaxSites = 64, kMaxPart = 33;

struct Binom {                      // C(p, k), p <= 64, k <= 33, in device memory (copied to LDS by the kernels)
    uint64_t c[(kMaxSites + 1) * (kMaxPart + 1)];
};

void fill_binom(Binom &b)
{
    for (int p = 0; p <= kMaxSites; ++p)
        for (int k = 0; k <= kMaxPart; ++k) {
            uint64_t v;
            if (k == 0) v = 1;
            else if (p == 0) v = 0;
            else {
                const uint64_t x = b.c[(p - 1) * (kMaxPart + 1) + k - 1], y = b.c[(p - 1) * (kMaxPart + 1) + k];
                v = (x > ~0ULL - y) ? ~0ULL : x + y;       // saturate (never reached for the sizes that fit a GPU)
            }
            b.c[p * (kMaxPart + 1) + k] = v;
        }
}

__device__ __forceinline__ uint64_t bin(const uint64_t *B, int p, int k)
{
    return (k < 0 || k > kMaxPart || k > p) ? 0ULL : B[p * (kMaxPart + 1) + k];
}

// colex unrank: the pattern with n bits set whose rank is r
__device__ __forceinline__ uint64_t unrank(const uint64_t *B, int n_sites, int n, uint64_t r)
{
    uint64_t pat = 0;
    int p = n_sites;
    for (int k = n; k >= 1; --k) {
        --p;
        while (bin(B, p, k) > r) --p;
        pat |= 1ULL << p;
        r -= bin(B, p, k);
    }
    return pat;
}

struct SpinCoef { double re[kMaxSites], im[kMaxSites]; };

// kind 0: S^z_q (same sector).  kind -1: S^-_q, old sector has n_new - 1 down spins.  kind +1: S^+_q, old has n_new + 1.
__global__ __launch_bounds__(256) void k_mopr_spin(int n_sites, int n_new, int kind, SpinCoef cf, const Binom *Bd, const d2 *x_old,
                                                   d2 *y_new, int64_t dim_new)
{
    __shared__ uint64_t B[(kMaxSites + 1) * (kMaxPart + 1)];
    for (int i = threadIdx.x; i < (kMaxSites + 1) * (kMaxPart + 1); i += 256) B[i] = Bd->c[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < dim_new; row += stride) {
        const uint64_t pat = unrank(B, n_sites, n_new, (uint64_t)row);
        d2 acc = {0.0, 0.0};
        if (kind == 0) {
            double sr = 0.0, si = 0.0;
            for (int s = 0; s < n_sites; ++s) {
                const double sz = ((pat >> s) & 1ULL) ? -0.5 : 0.5;
                sr += sz * cf.re[s];
                si += sz * cf.im[s];
            }
            const d2 x = x_old[row];
            acc = d2{sr * x.x - si * x.y, sr * x.y + si * x.x};
        } else {
            // positions of the set bits and the two partial sums that give the rank of the pattern with one bit
            // removed (S^-: the flipped spin was up before) or one bit added (S^+)
            int pos[kMaxPart + 1];
            int n = 0;
            for (int s = 0; s < n_sites; ++s)
                if ((pat >> s) & 1ULL) pos[n++] = s;
            if (kind < 0) {
                // source = pattern without its m-th set bit: sum_{t<m} C(p_t, t+1) + sum_{t>m} C(p_t, t)
                uint64_t hi[kMaxPart + 2];
                hi[n] = 0;
                for (int t = n - 1; t >= 0; --t) hi[t] = hi[t + 1] + bin(B, pos[t], t);
                uint64_t lo = 0;
                for (int m = 0; m < n; ++m) {
                    const uint64_t j = lo + hi[m + 1];
                    const int s = pos[m];
                    const d2 x = x_old[j];
                    acc.x += cf.re[s] * x.x - cf.im[s] * x.y;
                    acc.y += cf.re[s] * x.y + cf.im[s] * x.x;
                    lo += bin(B, pos[m], m + 1);
                }
            } else {
                // source = pattern with an extra bit at s (m set bits below s): sum_{t<m} C(p_t, t+1) + C(s, m+1) + sum_{t>=m} C(p_t, t+2)
                uint64_t hi[kMaxPart + 2];
                hi[n] = 0;
                for (int t = n - 1; t >= 0; --t) hi[t] = hi[t + 1] + bin(B, pos[t], t + 2);
                uint64_t lo = 0;
                int m = 0;
                for (int s = 0; s < n_sites; ++s) {
                    if ((pat >> s) & 1ULL) {
                        lo += bin(B, s, m + 1);
                        ++m;
                        continue;
                    }
                    const uint64_t j = lo + bin(B, s, m + 1) + hi[m];
                    const d2 x = x_old[j];
                    acc.x += cf.re[s] * x.x - cf.im[s] * x.y;
                    acc.y += cf.re[s] * x.y + cf.im[s] * x.x;
                }
            }
        }
        y_new[row] = acc;
    }
}

struct OneBodyTerm { int32_t a, b, spin, pad; double wr, wi; };

__device__ __forceinline__ uint64_t rank_of(const uint64_t *B, uint32_t c)
{
    uint64_t r = 0;
    int t = 0;
    while (c) {
        const int p = __ffs((int)c) - 1;
        ++t;
        r += bin(B, p, t);
        c &= c - 1;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_mopr_onebody(int64_t Nu, int64_t Nd, const uint32_t *cfg_u, const uint32_t *cfg_d, int n_terms,
                                                      const OneBodyTerm *terms, const Binom *Bd, const d2 *x_old, d2 *y_new)
{
    __shared__ uint64_t B[(kMaxSites + 1) * (kMaxPart + 1)];
    for (int i = threadIdx.x; i < (kMaxSites + 1) * (kMaxPart + 1); i += 256) B[i] = Bd->c[i];
    __syncthreads();
    const int64_t dim = Nu * Nd, stride = (int64_t)gridDim.x * 256;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < dim; row += stride) {
        const int64_t u = row / Nd, d = row - u * Nd;
        const uint32_t cu = cfg_u[u], cd = cfg_d[d];
        d2 acc = {0.0, 0.0};
        for (int k = 0; k < n_terms; ++k) {
            const OneBodyTerm t = terms[k];
            const uint32_t c = t.spin ? cd : cu;
            double sign = 1.0;
            int64_t j;
            if (t.a == t.b) {                                  // density n_{a, spin}
                if (!((c >> t.a) & 1u)) continue;
                j = row;
            } else {                                           // <row| c+_a c_b |j>: row has a occupied and b empty
                if (!((c >> t.a) & 1u) || ((c >> t.b) & 1u)) continue;
                const uint32_t cj = (c ^ (1u << t.a)) | (1u << t.b);
                const int lo = t.a < t.b ? t.a : t.b, hi = t.a < t.b ? t.b : t.a;
                const uint32_t between = (uint32_t)(((1ULL << hi) - 1ULL) & ~((1ULL << (lo + 1)) - 1ULL));
                if (__popc(c & between) & 1) sign = -1.0;
                const int64_t rj = (int64_t)rank_of(B, cj);
                j = t.spin ? u * Nd + rj : rj * Nd + d;
            }
            const d2 x = x_old[j];
            acc.x += sign * (t.wr * x.x - t.wi * x.y);
            acc.y += sign * (t.wr * x.y + t.wi * x.x);
        }
        y_new[row] = acc;
    }
}

uint64_t binom_host(int n, int k)
{
    if (k < 0 || k > n) return 0;
    long double r = 1.0L;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (uint64_t)(r + 0.5L);
}

void enumerate(int L, int n, std::vector<uint32_t> &out)          // ascending bit patterns == colex rank order
{
    out.clear();
    if (n == 0) {
        out.push_back(0);
        return;
    }
    uint64_t c = (1ULL << n) - 1ULL;
    const uint64_t lim = 1ULL << L;
    while (c < lim) {
        out.push_back((uint32_t)c);
        const uint64_t t = c | (c - 1);
        c = (t + 1) | (((~t & -~t) - 1) >> (__builtin_ctzll(c) + 1));
    }
}

int upload_binom(Binom **d_out)
{
    static Binom host;
    static bool filled = false;
    if (!filled) {
        fill_binom(host);
        filled = true;
    }
    QBH_HIP(qbh::dev_alloc(d_out, sizeof(Binom)));
    QBH_HIP(hipMemcpy(*d_out, &host, sizeof(Binom), hipMemcpyHostToDevice));
    return QBH_OK;
}

}  // namespace
}  // namespace qbh

extern "C" int qbh_mopr_spin_dev(int n_sites, int n_dn_old, int kind, const qbh_z *coef, const qbh_z *d_vec_old, qbh_z *d_vec_new,
                                 void *stream)
{
    using namespace qbh;
    const int n_new = n_dn_old - kind;                    // S^- adds a down spin, S^+ removes one
    if (!coef || !d_vec_old || !d_vec_new || n_sites <= 0 || n_sites > kMaxSites || kind < -1 || kind > 1 || n_dn_old < 0 ||
        n_dn_old > n_sites || n_new < 0 || n_new > n_sites || n_new > kMaxPart - 1 || n_dn_old > kMaxPart - 1) {
        set_error("qbh_mopr_spin_dev: invalid argument");
        return QBH_EINVAL;
    }
    if (qbh_device_count() <= 0) {
        set_error("no HIP device visible");
        return QBH_ENODEVICE;
    }
    SpinCoef cf{};
    for (int s = 0; s < n_sites; ++s) {
        cf.re[s] = coef[s].re;
        cf.im[s] = coef[s].im;
    }
    Binom *d_b = nullptr;
    QBH_TRY(upload_binom(&d_b));
    const int64_t dim_new = (int64_t)binom_host(n_sites, n_new);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mopr_spin, dim3(blas_grid(dim_new)), dim3(256), 0, s, n_sites, n_new, kind, cf, d_b,
                       reinterpret_cast<const d2 *>(d_vec_old), reinterpret_cast<d2 *>(d_vec_new), dim_new);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_b);
    if (e != hipSuccess) {
        set_error("qbh_mopr_spin_dev: %s", hipGetErrorString(e));
        return QBH_EHIP;
    }
    return QBH_OK;
}

extern "C" int qbh_mopr_onebody_dev(int n_sites, int n_up, int n_dn, int n_terms, const int32_t *a, const int32_t *b, const int32_t *spin,
                                    const qbh_z *w, const qbh_z *d_vec_old, qbh_z *d_vec_new, void *stream)
{
    using namespace qbh;
    if (!a || !b || !spin || !w || !d_vec_old || !d_vec_new || n_sites <= 0 || n_sites > 31 || n_up < 0 || n_dn < 0 || n_up > n_sites ||
        n_dn > n_sites || n_terms <= 0) {
        set_error("qbh_mopr_onebody_dev: invalid argument");
        return QBH_EINVAL;
    }
    for (int k = 0; k < n_terms; ++k)
        if (a[k] < 0 || a[k] >= n_sites || b[k] < 0 || b[k] >= n_sites || (spin[k] != 0 && spin[k] != 1)) {
            set_error("qbh_mopr_onebody_dev: term %d out of range", k);
            return QBH_EINVAL;
        }
    if (qbh_device_count() <= 0) {
        set_error("no HIP device visible");
        return QBH_ENODEVICE;
    }
    std::vector<uint32_t> cu, cd;
    enumerate(n_sites, n_up, cu);
    enumerate(n_sites, n_dn, cd);
    std::vector<OneBodyTerm> terms((size_t)n_terms);
    for (int k = 0; k < n_terms; ++k) terms[(size_t)k] = OneBodyTerm{a[k], b[k], spin[k], 0, w[k].re, w[k].im};
    uint32_t *d_cu = nullptr, *d_cd = nullptr;
    OneBodyTerm *d_t = nullptr;
    Binom *d_b = nullptr;
    int rc = upload_binom(&d_b);
    hipError_t e = hipSuccess;
    if (rc == QBH_OK) {
        e = qbh::dev_alloc(&d_cu, cu.size() * 4);
        if (e == hipSuccess) e = qbh::dev_alloc(&d_cd, cd.size() * 4);
        if (e == hipSuccess) e = qbh::dev_alloc(&d_t, terms.size() * sizeof(OneBodyTerm));
        if (e == hipSuccess) e = hipMemcpy(d_cu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cd, cd.data(), cd.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_t, terms.data(), terms.size() * sizeof(OneBodyTerm), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipStream_t s = (hipStream_t)stream;
            const int64_t dim = (int64_t)cu.size() * (int64_t)cd.size();
            hipLaunchKernelGGL(k_mopr_onebody, dim3(blas_grid(dim)), dim3(256), 0, s, (int64_t)cu.size(), (int64_t)cd.size(), d_cu, d_cd, n_terms,
                               d_t, d_b, reinterpret_cast<const d2 *>(d_vec_old), reinterpret_cast<d2 *>(d_vec_new));
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    for (void *p : {(void *)d_cu, (void *)d_cd, (void *)d_t, (void *)d_b})
        if (p) (void)hipFree(p);
    if (rc != QBH_OK) return rc;
    if (e != hipSuccess) {
        set_error("qbh_mopr_onebody_dev: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? QBH_ENOMEM : QBH_EHIP;
    }
    return QBH_OK;
}
