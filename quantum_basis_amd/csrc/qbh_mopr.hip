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

__device__ __forceinline__ uint64_t rank_of64(const uint64_t *B, uint64_t c)
{
    uint64_t r = 0;
    int t = 0;
    while (c) {
        const int p = __ffsll((long long)c) - 1;
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

// ---- the general form: A = sum_t coef_t O_{t,0} O_{t,1} ... (ordered products of elementary site operators) ----
// Every product maps a basis state to at most one basis state, so the TARGET row finds its source by running the adjoint
// factors over its own pattern, leftmost factor first: (O_0 O_1 ... O_{L-1})^+ |i> = O_{L-1}^+ ... O_1^+ O_0^+ |i> = amp |j>, and
// <i| O_0 ... O_{L-1} |j> = amp (all elementary matrix elements are real).  No atomics, deterministic.
// family 0, spin-1/2 (bit = 1: down):  kind 0 S^z, 1 S^+ (down -> up), 2 S^- (up -> down)
// family 1, two-species fermions, orbital o = site + species * n_sites in the word u | d << n_sites, |w> = prod_{o ascending}
//           c+_o |0> (all up operators left of all down operators, as qbh_gen_hubbard / qbh_mopr_c_hubrepr_dev):
//           kind 0 n_o, 1 c+_o, 2 c_o; c+_o |w> = (-1)^{occupied orbitals below o} |w + o>
struct TermOp { int8_t kind; int8_t orb; };
constexpr int kMaxTermOps = 4096;

template <int FAMILY>
__global__ __launch_bounds__(256) void k_mopr_terms(int n_sites, int na_new, int nb_new, int na_old, int nb_old, int64_t nd_new, int64_t nd_old,
                                                    const uint32_t *cfg_u, const uint32_t *cfg_d, int n_terms, const int32_t *term_ptr,
                                                    const TermOp *ops, const d2 *coef, const Binom *Bd, const d2 *x_old, d2 *y_new, int64_t dim_new)
{
    __shared__ uint64_t B[(kMaxSites + 1) * (kMaxPart + 1)];
    for (int i = threadIdx.x; i < (kMaxSites + 1) * (kMaxPart + 1); i += 256) B[i] = Bd->c[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < dim_new; row += stride) {
        uint64_t w0;
        if (FAMILY == 0) {
            w0 = unrank(B, n_sites, na_new, (uint64_t)row);
        } else {
            const int64_t u = row / nd_new, d = row - u * nd_new;
            w0 = (uint64_t)cfg_u[u] | ((uint64_t)cfg_d[d] << n_sites);
        }
        d2 acc = {0.0, 0.0};
        for (int t = 0; t < n_terms; ++t) {
            uint64_t w = w0;
            double amp = 1.0;
            for (int k = term_ptr[t]; k < term_ptr[t + 1] && amp != 0.0; ++k) {
                const TermOp op = ops[k];
                const uint64_t bit = 1ULL << op.orb;
                const bool set = (w & bit) != 0;
                if (FAMILY == 0) {
                    if (op.kind == 0) amp *= set ? -0.5 : 0.5;
                    else if (op.kind == 1) {                 // (S^+)^+ = S^-: the target has the spin UP here, the source had it down
                        if (set) amp = 0.0;
                        else w |= bit;
                    } else {                                 // (S^-)^+ = S^+
                        if (!set) amp = 0.0;
                        else w &= ~bit;
                    }
                } else {
                    if (op.kind == 0) {
                        if (!set) amp = 0.0;
                    } else {
                        const bool want_set = op.kind == 1;  // (c+_o)^+ = c_o needs the orbital occupied in the target
                        if (set != want_set) amp = 0.0;
                        else {
                            if (__popcll(w & (bit - 1ULL)) & 1) amp = -amp;
                            w ^= bit;
                        }
                    }
                }
            }
            if (amp == 0.0) continue;
            int64_t j;
            if (FAMILY == 0) {
                j = (int64_t)rank_of64(B, w);
            } else {
                const uint32_t uj = (uint32_t)(w & ((1ULL << n_sites) - 1ULL)), dj = (uint32_t)(w >> n_sites);
                j = (int64_t)rank_of(B, uj) * nd_old + (int64_t)rank_of(B, dj);
            }
            const d2 x = x_old[j], c = coef[t];
            acc.x += amp * (c.x * x.x - c.y * x.y);
            acc.y += amp * (c.x * x.y + c.y * x.x);
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


extern "C" int qbh_mopr_terms_dev(int family, int n_sites, int n_a_old, int n_b_old, int n_terms, const int32_t *term_ptr, const int32_t *op_kind,
                                  const int32_t *op_site, const int32_t *op_species, const qbh_z *coef, const qbh_z *d_vec_old, qbh_z *d_vec_new,
                                  int64_t *dim_new_out, void *stream)
{
    using namespace qbh;
    if ((family != 0 && family != 1) || !term_ptr || !op_kind || !op_site || !coef || !d_vec_old || !d_vec_new || n_terms <= 0 || n_sites <= 0 ||
        n_sites > (family == 0 ? kMaxSites : 31) || n_a_old < 0 || n_a_old > n_sites || n_b_old < 0 || n_b_old > n_sites || term_ptr[0] != 0 ||
        (family == 1 && !op_species)) {
        set_error("qbh_mopr_terms_dev: invalid argument");
        return QBH_EINVAL;
    }
    const int n_ops = term_ptr[n_terms];
    if (n_ops <= 0 || n_ops > kMaxTermOps) {
        set_error("qbh_mopr_terms_dev: %d elementary factors (1 .. %d allowed)", n_ops, kMaxTermOps);
        return QBH_EINVAL;
    }
    // every product must change the particle numbers by the same amount: vec_new lives in ONE sector (src/model.cc:1468-1471)
    std::vector<TermOp> ops((size_t)n_ops);
    int da = 0, db = 0;
    for (int t = 0; t < n_terms; ++t) {
        if (term_ptr[t + 1] < term_ptr[t]) {
            set_error("qbh_mopr_terms_dev: term_ptr is not ascending");
            return QBH_EINVAL;
        }
        int ta = 0, tb = 0;
        for (int k = term_ptr[t]; k < term_ptr[t + 1]; ++k) {
            const int sp = family == 1 ? op_species[k] : 0;
            if (op_kind[k] < 0 || op_kind[k] > 2 || op_site[k] < 0 || op_site[k] >= n_sites || sp < 0 || sp > 1) {
                set_error("qbh_mopr_terms_dev: factor %d of term %d out of range", k - term_ptr[t], t);
                return QBH_EINVAL;
            }
            ops[(size_t)k] = TermOp{(int8_t)op_kind[k], (int8_t)(op_site[k] + sp * n_sites)};
            // family 0: S^+ removes a down spin, S^- adds one; family 1: c+ adds a particle of its species, c removes one
            const int delta = op_kind[k] == 0 ? 0 : family == 0 ? (op_kind[k] == 1 ? -1 : 1) : (op_kind[k] == 1 ? 1 : -1);
            (sp ? tb : ta) += delta;
        }
        if (t == 0) da = ta, db = tb;
        else if (ta != da || tb != db) {
            set_error("qbh_mopr_terms_dev: term %d changes the particle numbers by (%d, %d), term 0 by (%d, %d): one target sector only", t, ta, tb,
                      da, db);
            return QBH_EINVAL;
        }
    }
    const int na_new = n_a_old + da, nb_new = family == 1 ? n_b_old + db : 0;
    if (na_new < 0 || na_new > n_sites || nb_new < 0 || nb_new > n_sites || na_new > kMaxPart - 1 || n_a_old > kMaxPart - 1 ||
        (family == 1 && (nb_new > kMaxPart - 1 || n_b_old > kMaxPart - 1))) {
        set_error("qbh_mopr_terms_dev: the target sector (%d, %d) does not exist", na_new, nb_new);
        return QBH_EINVAL;
    }
    if (qbh_device_count() <= 0) {
        set_error("no HIP device visible");
        return QBH_ENODEVICE;
    }
    std::vector<uint32_t> cu, cd;
    int64_t dim_new = 0, nd_new = 1, nd_old = 1;
    if (family == 1) {
        enumerate(n_sites, na_new, cu);
        enumerate(n_sites, nb_new, cd);
        nd_new = (int64_t)cd.size();
        nd_old = (int64_t)binom_host(n_sites, n_b_old);
        dim_new = (int64_t)cu.size() * nd_new;
    } else {
        dim_new = (int64_t)binom_host(n_sites, na_new);
    }
    if (dim_new_out) *dim_new_out = dim_new;
    uint32_t *d_cu = nullptr, *d_cd = nullptr;
    int32_t *d_ptr = nullptr;
    TermOp *d_ops = nullptr;
    d2 *d_coef = nullptr;
    Binom *d_b = nullptr;
    int rc = upload_binom(&d_b);
    hipError_t e = hipSuccess;
    if (rc == QBH_OK) {
        if (family == 1) {
            e = qbh::dev_alloc(&d_cu, cu.size() * 4);
            if (e == hipSuccess) e = qbh::dev_alloc(&d_cd, cd.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(d_cu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_cd, cd.data(), cd.size() * 4, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = qbh::dev_alloc(&d_ptr, (size_t)(n_terms + 1) * 4);
        if (e == hipSuccess) e = qbh::dev_alloc(&d_ops, ops.size() * sizeof(TermOp));
        if (e == hipSuccess) e = qbh::dev_alloc(&d_coef, (size_t)n_terms * sizeof(d2));
        if (e == hipSuccess) e = hipMemcpy(d_ptr, term_ptr, (size_t)(n_terms + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_ops, ops.data(), ops.size() * sizeof(TermOp), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_coef, coef, (size_t)n_terms * sizeof(d2), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipStream_t s = (hipStream_t)stream;
            const dim3 g(blas_grid(dim_new)), b(256);
            if (family == 0)
                hipLaunchKernelGGL((k_mopr_terms<0>), g, b, 0, s, n_sites, na_new, 0, n_a_old, 0, (int64_t)1, (int64_t)1, d_cu, d_cd, n_terms, d_ptr, d_ops,
                                   d_coef, d_b, reinterpret_cast<const d2 *>(d_vec_old), reinterpret_cast<d2 *>(d_vec_new), dim_new);
            else
                hipLaunchKernelGGL((k_mopr_terms<1>), g, b, 0, s, n_sites, na_new, nb_new, n_a_old, n_b_old, nd_new, nd_old, d_cu, d_cd, n_terms, d_ptr,
                                   d_ops, d_coef, d_b, reinterpret_cast<const d2 *>(d_vec_old), reinterpret_cast<d2 *>(d_vec_new), dim_new);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
    }
    for (void *p : {(void *)d_cu, (void *)d_cd, (void *)d_ptr, (void *)d_ops, (void *)d_coef, (void *)d_b})
        if (p) (void)hipFree(p);
    if (rc != QBH_OK) return rc;
    if (e != hipSuccess) {
        set_error("qbh_mopr_terms_dev: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? QBH_ENOMEM : QBH_EHIP;
    }
    return QBH_OK;
}
