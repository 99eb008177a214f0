// The Kronecker split for the library's DEFAULT form of a real operator: dictionary-coded values (1 byte per nonzero) applied
// to packed-double vectors (qbh_opts.real_fast_path).  Same decomposition as qbh_kron.hip -- index = major * S + minor, every
// entry changes the minor index (near) or the major index alone (far) -- but both parts are stored SLICED: groups of 16 rows
// (one band of 16 minor indices of one major index), widths padded to a multiple of 4 entries with a code whose value is zero.
// Inside a group the order is LANE-MAJOR: lane = 16 (k % 4) + j of a wavefront owns entries k, k + 4, ... of row j, and its
// nu = width / 4 columns (2 bytes each) and codes (1 byte each) are contiguous -- ONE 16-byte column load and ONE 8-byte code load
// per lane bring a whole group (up to 32 entries per row), 6 registers per group instead of 12, so a wavefront keeps several
// groups in flight.  The streams stay dense (a group's columns are 64 nu consecutive 2-byte words) and need no row pointers:
//   * far part (gathers from the band-major "tiled" copy of x, KronTile{S, NU, 16}; a far entry keeps the minor index, so its
//     column is stored as the target major index alone, 2 bytes): the 16 rows of a group gather ONE 128-byte line of the
//     tiled x per entry -- 4 lines per wave instruction instead of up to 64 separate requests;
//   * near part (columns relative to the major index's own block, 2 bytes each): the whole block of x -- S doubles, 103 KB at
//     C3 -- is loaded into LDS once per major index and every gather is an LDS read.
// Far pass first (row sums into a buffer in group order), then the near pass with the fused epilogue
// y = alpha (near + far) + beta y + gamma x and the reductions <x, y>, |y|^2 of the finished y.
// The row kernel this replaces is bound by the RATE of 8-byte gathers through the L1 (DESIGN-history 5.0b item 10: 420-490 G/s); here no
// gather is an L1 request of its own.  Two recognitions, each verified entry by entry on the device: a far part whose entries do not
// depend on the row's minor index (T (x) 1) is kept as T alone, a near part whose off-diagonal entries do not depend on the row's
// major index (1 (x) T' + D) as T' and one diagonal code per row -- the two-species models are both, and their passes then stream
// nothing but vectors.  Reference operation: csr_mat<T>::MultMv2 (src/sparse.cc:262-289) on a real operator.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include <mutex>

#include "qbh_internal.hpp"

namespace qbh {
namespace {

constexpr int kGB = 16;                      // rows per group = doubles per 128-byte line

// widths (entries per row, padded) of the near group (maj, b) and of the far group (b, maj): one thread per group
__global__ __launch_bounds__(256) void k_kronc_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int nb, int32_t *wn, int32_t *wf)
{
    const int64_t G = (int64_t)nb * NU;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < G; g += (int64_t)gridDim.x * 256) {
        const int64_t maj = g / nb, b = g - maj * nb;
        const int64_t d0 = b * kGB, wb = S - d0 < kGB ? S - d0 : kGB;
        int mn = 0, mf = 0;
        for (int64_t j = 0; j < wb; ++j) {
            const int64_t r = maj * S + d0 + j;
            int cn = 0, cf = 0;
            for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
                if (ja[q] / S == maj) ++cn;
                else ++cf;
            }
            mn = cn > mn ? cn : mn;
            mf = cf > mf ? cf : mf;
        }
        // widths in multiples of 4 entries: a group is then a whole number of 64-slot wave instructions -- no lane of a pass ever
        // reads another group's slots (nothing to mask), column loads are whole aligned 128-byte lines
        wn[g] = ((mn + 3) & ~3) * kGB;
        wf[b * NU + maj] = ((mf + 3) & ~3) * kGB;
    }
}

// Is the far part T (x) 1 -- the far entries (target major index, code) of row (u, d) the same for every minor index d?  (Two-species
// models in species-major order: a hop of the major species does not see the minor one.)  One thread per row compares with row (u, 0).
__global__ __launch_bounds__(256) void k_kronc_far_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *flag)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int64_t u = r / S;
        if (r == u * S) continue;
        int64_t q = ia[r], q0 = ia[u * S];
        const int64_t e = ia[r + 1], e0 = ia[u * S + 1];
        bool bad = false;
        for (;;) {
            while (q < e && ja[q] / S == u) ++q;
            while (q0 < e0 && ja[q0] / S == u) ++q0;
            if (q >= e || q0 >= e0) break;
            bad = bad || ja[q] / S != ja[q0] / S || code[q] != code[q0];
            ++q;
            ++q0;
        }
        if (bad || q < e || q0 < e0) *flag = 1;
    }
}

// T of a uniform far part: widths (padded to a multiple of 4) and entries of row (u, 0), lane-major over ks = k % 4
__global__ __launch_bounds__(256) void k_kronc_t_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int32_t *wt)
{
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < NU; u += (int64_t)gridDim.x * 256) {
        int cf = 0;
        for (int64_t q = ia[u * S]; q < ia[u * S + 1]; ++q) cf += ja[q] / S != u ? 1 : 0;
        wt[u] = (cf + 3) & ~3;
    }
}
__global__ __launch_bounds__(256) void k_kronc_t_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, uint8_t zcode,
                                                      const int64_t *tp, uint16_t *tcol, uint8_t *tcode)
{
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < NU; u += (int64_t)gridDim.x * 256) {
        const int64_t base = tp[u], w = tp[u + 1] - base, nu = w >> 2;
        int64_t k = 0;
        for (int64_t q = ia[u * S]; q < ia[u * S + 1]; ++q) {
            if (ja[q] / S == u) continue;
            tcol[base + (k & 3) * nu + (k >> 2)] = (uint16_t)(ja[q] / S);
            tcode[base + (k & 3) * nu + (k >> 2)] = code[q];
            ++k;
        }
        for (; k < w; ++k) {
            tcol[base + (k & 3) * nu + (k >> 2)] = (uint16_t)u;
            tcode[base + (k & 3) * nu + (k >> 2)] = zcode;
        }
    }
}

// Is the near part 1 (x) T' + D -- the off-diagonal near entries (column relative to the block, code) of row (u, d) the same for
// every major index u, only the diagonal entry differing?  (Two-species models: a hop of the minor species does not see the
// major one; the diagonal, U times the double occupancy, sees both.)  One thread per row compares with row (0, d).
__global__ __launch_bounds__(256) void k_kronc_near_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *flag)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int64_t u = r / S, d = r - u * S;
        if (u == 0) continue;
        int64_t q = ia[r], q0 = ia[d];
        const int64_t e = ia[r + 1], e0 = ia[d + 1];
        bool bad = false;
        for (;;) {
            while (q < e && (ja[q] / S != u || ja[q] == r)) ++q;
            while (q0 < e0 && (ja[q0] / S != 0 || ja[q0] == d)) ++q0;
            if (q >= e || q0 >= e0) break;
            bad = bad || ja[q] - u * S != ja[q0] || code[q] != code[q0];
            ++q;
            ++q0;
        }
        if (bad || q < e || q0 < e0) *flag = 1;
    }
}

// the shared near pattern T' (rows (0, d) without their diagonal): group widths, entries (lane-major as below), and the code of
// every row's diagonal entry (the padding code where a row has none)
__global__ __launch_bounds__(256) void k_kronc_s_widths(const int64_t *ia, const int32_t *ja, int64_t S, int nb, int32_t *ws)
{
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nb; b += (int64_t)gridDim.x * 256) {
        const int64_t d0 = b * kGB, wb = S - d0 < kGB ? S - d0 : kGB;
        int m = 0;
        for (int64_t j = 0; j < wb; ++j) {
            int c = 0;
            for (int64_t q = ia[d0 + j]; q < ia[d0 + j + 1]; ++q) c += (ja[q] / S == 0 && ja[q] != d0 + j) ? 1 : 0;
            m = c > m ? c : m;
        }
        ws[b] = ((m + 3) & ~3) * kGB;
    }
}
__global__ __launch_bounds__(256) void k_kronc_s_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int nb, uint8_t zcode,
                                                      const int64_t *gs, uint16_t *scol, uint8_t *scode)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)nb * kGB; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i >> 4, j = i & 15, d0 = b * kGB, wb = S - d0 < kGB ? S - d0 : kGB;
        const int64_t base = gs[b], w = (gs[b + 1] - base) >> 4;
        auto pos = [&](int64_t k) { return (16 * (k & 3) + j) * (w >> 2) + (k >> 2); };
        int64_t k = 0;
        if (j < wb)
            for (int64_t q = ia[d0 + j]; q < ia[d0 + j + 1]; ++q) {
                if (ja[q] / S != 0 || ja[q] == d0 + j) continue;
                scol[base + pos(k)] = (uint16_t)ja[q];
                scode[base + pos(k)] = code[q];
                ++k;
            }
        for (; k < w; ++k) {
            scol[base + pos(k)] = (uint16_t)(d0 + (j < wb ? j : 0));
            scode[base + pos(k)] = zcode;
        }
    }
}
__global__ __launch_bounds__(256) void k_kronc_dcode(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t n, uint8_t zcode, uint8_t *dcode)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        uint8_t c = zcode;
        for (int64_t q = ia[r]; q < ia[r + 1]; ++q)
            if (ja[q] == r) c = code[q];
        dcode[r] = c;
    }
}

// one thread per (near group, lane j): scatters the row's entries into the two sliced parts and pads both to the group widths
// (gia_f == nullptr: the far part is kept as T, see above; gia_n == nullptr: the near part as T' and the diagonal codes)
__global__ __launch_bounds__(256) void k_kronc_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, int nb,
                                                    uint8_t zcode, const int64_t *gia_n, uint16_t *ja_n, uint8_t *code_n,
                                                    const int64_t *gia_f, uint16_t *ja_f, uint8_t *code_f)
{
    const int64_t G = (int64_t)nb * NU;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < G * kGB; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i >> 4, j = i & 15;
        const int64_t maj = g / nb, b = g - maj * nb;
        const int64_t d0 = b * kGB, wb = S - d0 < kGB ? S - d0 : kGB;
        const int64_t gf = b * NU + maj;
        const int64_t bn = gia_n ? gia_n[g] : 0, wn = gia_n ? (gia_n[g + 1] - bn) >> 4 : 0;
        const int64_t bf = gia_f ? gia_f[gf] : 0, wf = gia_f ? (gia_f[gf + 1] - bf) >> 4 : 0;
        int64_t kn = 0, kf = 0;
        // entry k of row j of a group of width w: lane 16 (k % 4) + j, its (k / 4)-th word
        auto pos = [&](int64_t k, int64_t w) { return (16 * (k & 3) + j) * (w >> 2) + (k >> 2); };
        const int64_t r0 = maj * S + d0;           // first row of the group
        if (j < wb) {
            const int64_t r = r0 + j;
            for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
                const int64_t c = ja[q];
                if (c / S == maj) {
                    if (!gia_n) continue;
                    ja_n[bn + pos(kn, wn)] = (uint16_t)(c - maj * S);
                    code_n[bn + pos(kn, wn)] = code[q];
                    ++kn;
                } else if (gia_f) {
                    ja_f[bf + pos(kf, wf)] = (uint16_t)(c / S);          // the minor index is the row's own
                    code_f[bf + pos(kf, wf)] = code[q];
                    ++kf;
                }
            }
        }
        const int64_t own = j < wb ? j : 0;
        for (; kn < wn; ++kn) {
            ja_n[bn + pos(kn, wn)] = (uint16_t)(d0 + own);
            code_n[bn + pos(kn, wn)] = zcode;
        }
        for (; kf < wf; ++kf) {
            ja_f[bf + pos(kf, wf)] = (uint16_t)maj;
            code_f[bf + pos(kf, wf)] = zcode;
        }
    }
}

}  // namespace

// ---- the two passes (named in namespace qbh: the profiling tools select kernels by name) ---------------------------------------
// One wavefront works on NG groups at a time; a group's columns and codes of one lane are two loads (unaligned: the lane's words
// start at lane * nu).  NT: non-temporal loads -- the near pass keeps its stream out of the L1, the far pass does not (its
// neighbouring lanes' words share lines).
template <int NG, bool NT>
struct GroupStream {
    uint32_t c[NG][4];          // 8 columns (2 bytes each): entries k = 4 u + ks, u = 0..7
    uint32_t cb[NG][2];         // 8 codes
    int      nu[NG];            // entries per lane = width / 4
    // own: the index of the lane's words inside the group -- the lane itself, or lane >> 4 for a far part kept as T (the 16 rows
    // of a group share their entries: sh = 2, a group is 4 nu words)
    __device__ __forceinline__ void load(int gi, const uint16_t *ja, const uint8_t *code, int64_t base, int64_t end, int own, int sh = 6)
    {
        const int n = (int)((end - base) >> sh);
        nu[gi] = n;
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        const v4 *jp = reinterpret_cast<const v4 *>(ja + base + (int64_t)own * n);
        const v2 *cp = reinterpret_cast<const v2 *>(code + base + (int64_t)own * n);
        const v4 cw = NT ? __builtin_nontemporal_load(jp) : *jp;
        const v2 cd = NT ? __builtin_nontemporal_load(cp) : *cp;
        c[gi][0] = cw.x;
        c[gi][1] = cw.y;
        c[gi][2] = cw.z;
        c[gi][3] = cw.w;
        cb[gi][0] = cd.x;
        cb[gi][1] = cd.y;
    }
    __device__ __forceinline__ int col(int gi, int u) const { return (int)((c[gi][u >> 1] >> (16 * (u & 1))) & 0xFFFFu); }
    __device__ __forceinline__ int cod(int gi, int u) const { return (int)((cb[gi][u >> 2] >> (8 * (u & 3))) & 0xFFu); }
};

__device__ __forceinline__ double quad_sum(double v)        // sum over the four sub-slices ks of a row: lanes 0..15 hold the row sums
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

struct KroncArgs {
    const int64_t *gia_n, *gia_f;
    const uint16_t *ja_n;
    const uint8_t *dcode;       // near part kept as T' (the same groups for every major index) + the code of every row's diagonal entry; else nullptr
    const int64_t *tf_ptr;      // far part kept as T (uniform over the minor index): entries of major index u at [tf_ptr[u], tf_ptr[u + 1]) of ja_f / code_f
    const uint16_t *ja_f;       // far columns: the target MAJOR index (the element is (that major index, the row's minor index))
    const uint8_t *code_n, *code_f;
    const d2 *dict;
    const double *dictr;        // the dictionary's real parts, 256 doubles, entry n_dict = 0 (the padding code)
    int n_dict;
    int64_t S, NU;
    int nb;
    const double *xt, *x;
    double *far, *y;
    double alpha, beta, gamma;
    double *partials;
    unsigned int *ctr;
    unsigned int *fctr;         // far pass: one chunk counter per XCD, 128 bytes apart
    int chunk;                  // far pass: groups per wavefront turn (<= 32)
    int abl;                    // ablation bits (QBH_KRONC_ABL, wrong results by design): 1 no dictionary lookups, 2 no gathers, 4 no row sums, 8 near: no block load, 16 near: no epilogue
};

// UNI: the far part is T (x) 1 -- a group (b, maj) reads the entries of major index maj from T (0.7 MB at C3: always in the L2)
template <int NG, int UN, bool NT, bool UNI>
__global__ __launch_bounds__(256) void k_kronc_far(KroncArgs a)
{
    static_assert(UN == 8, "a lane's loads cover 8 entries");
    __shared__ double dict_s[256];
    dict_s[threadIdx.x] = (int)threadIdx.x < a.n_dict ? a.dict[threadIdx.x].x : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t G = (int64_t)a.nb * a.NU;
    // every XCD sweeps one contiguous eighth of the groups (band-major: a band of the tiled x, 1.6 MB at C3, stays in that XCD's
    // L2) and its wavefronts DRAW chunks of a.chunk groups from the XCD's counter, the next one while the current is processed:
    // with a static round robin the wavefronts drift bands apart over a pass (a few per cent of speed difference over ~1400
    // chunks each) and 42 % of the gather lines missed the L2
    const int xcd = blockIdx.x & 7;
    const int64_t per = (G + 7) >> 3, xb = xcd * per, xe = xb + per < G ? xb + per : G;
    const int CH = a.chunk;
    unsigned int *fc = a.fctr + xcd * 32;
    auto draw = [&]() {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(fc, 1u);
        return (int64_t)__builtin_amdgcn_readfirstlane(v);
    };
    (void)wv;
    for (int64_t cidx = draw(); xb + cidx * CH < xe;) {
        const int64_t cnext = draw();
        const int64_t g0 = xb + cidx * CH;
        cidx = cnext;
        const int ng = (int)(xe - g0 < CH ? xe - g0 : CH);
        int64_t gp, ge = 0;
        if (UNI) {
            const int64_t g = g0 + (lane < ng ? lane : ng - 1);
            const int64_t mj = g - (g / a.NU) * a.NU;
            gp = a.tf_ptr[mj];
            ge = a.tf_ptr[mj + 1];
        } else {
            gp = a.gia_f[g0 + (lane <= ng ? lane : ng)];
        }
        GroupStream<NG, NT> cur;
        int64_t base[NG];
        auto fetch = [&](GroupStream<NG, NT> &st, int64_t (&bs)[NG], int i0) {
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int i = i0 + gi < ng ? i0 + gi : ng - 1;
                bs[gi] = __shfl(gp, i, 64);
                if (UNI) st.load(gi, a.ja_f, a.code_f, bs[gi], __shfl(ge, i, 64), lane >> 4, 2);
                else     st.load(gi, a.ja_f, a.code_f, bs[gi], __shfl(gp, i + 1, 64), lane);
            }
        };
        fetch(cur, base, 0);
        for (int i0 = 0; i0 < ng; i0 += NG) {
            double xv[NG][UN];
            const double *xband[NG];
            int wBs[NG];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                // element (major u', minor 16 b + j) of the tiled x: band base + u' * (width of the band) + j
                const int64_t g = g0 + (i0 + gi < ng ? i0 + gi : ng - 1);
                const int64_t b = g / a.NU;
                wBs[gi] = (int)(a.S - b * kGB < kGB ? a.S - b * kGB : kGB);
                xband[gi] = a.xt + b * kGB * a.NU + (lane & 15);
#pragma unroll
                for (int u = 0; u < UN; ++u) xv[gi][u] = (u < cur.nu[gi] && !(a.abl & 2)) ? xband[gi][cur.col(gi, u) * wBs[gi]] : 0.0;
            }
            // the next groups' streams go out behind the gathers, in front of the arithmetic that waits for them
            GroupStream<NG, NT> nxt;
            int64_t nbase[NG];
            fetch(nxt, nbase, i0 + NG < ng ? i0 + NG : i0);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    if (u < cur.nu[gi]) acc += ((a.abl & 1) ? 1.0 : dict_s[cur.cod(gi, u)]) * xv[gi][u];
                if (cur.nu[gi] > UN) {                                               // rows longer than 32 entries: the lane's further words
                    const uint16_t *jp = a.ja_f + base[gi] + (int64_t)(UNI ? lane >> 4 : lane) * cur.nu[gi];
                    const uint8_t *cp = a.code_f + base[gi] + (int64_t)(UNI ? lane >> 4 : lane) * cur.nu[gi];
                    for (int u = UN; u < cur.nu[gi]; ++u) acc += dict_s[cp[u]] * xband[gi][(int)jp[u] * wBs[gi]];
                }
                if (!(a.abl & 4)) acc = quad_sum(acc);
                if (lane < kGB && i0 + gi < ng) a.far[(g0 + i0 + gi) * kGB + lane] = acc;
            }
            cur = nxt;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) base[gi] = nbase[gi];
        }
    }
}

// One workgroup of 16 wavefronts per major index at a time (drawn from a counter): x[maj * S .. + S) into LDS, then wavefront w
// takes a run of consecutive bands of that major index, four groups at a time, the streams of the next four in flight meanwhile.
// NT: non-temporal stream loads (general form: the stream is read once); the T' form wants its groups kept in the caches
template <int NG, int UN, bool NT>
__global__ __launch_bounds__(1024) void k_kronc_near(KroncArgs a)
{
    static_assert(UN == 8 && NG == 4, "a lane's loads cover 8 entries; a pass is four groups (lane >> 4 picks one in the epilogue)");
    extern __shared__ double lds[];                 // [S] window of x, [256] dictionary, [48] scratch, s_maj
    double *win = lds;
    double *dict = lds + a.S;                       // a small dictionary sits in one row of banks: its lookups do not conflict
    double *red = dict + 256;
    int *s_maj = reinterpret_cast<int *>(red + 48);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 256) dict[tid] = a.dictr[tid];
    double acc3[2] = {0.0, 0.0};                    // <x, y>, |y|^2
    const int nb = a.nb;
    unsigned int turn = 0;
    for (;;) {
        __syncthreads();                            // the previous window is no longer read
        if (tid == 0) *s_maj = a.ctr ? (int)atomicAdd(a.ctr, 1u) : (int)(blockIdx.x + turn * gridDim.x);      // no counter: static turns
        ++turn;
        __syncthreads();
        const int64_t maj = *s_maj;
        if (maj >= a.NU) break;
        const double *xb = a.x + maj * a.S;
        for (int64_t o = 0; o < a.S && !(a.abl & 8); o += 8 * 1024) {          // eight loads per thread in flight at a time
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = o + tid + 1024 * u < a.S ? xb[o + tid + 1024 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (o + tid + 1024 * u < a.S) win[o + tid + 1024 * u] = v[u];
        }
        __syncthreads();
        const int64_t gbase = maj * nb;
        // wavefront wv takes the consecutive bands [wv q, (wv + 1) q), four at a time: after the row sums every lane holds the
        // sum of ITS row j for each of the four groups, so lane l = 16 gi + j takes group gi's -- 64 consecutive rows of y per
        // wave instruction (one 512-byte read of the old y, one 512-byte store) instead of four 128-byte pieces that each
        // straddle a line (S * 8 bytes is not a multiple of 128)
        const int q = (((nb + 15) >> 4) + 3) & ~3;
        for (int r0 = 0; r0 < q; r0 += 60) {             // rounds of up to 60 bands: their pointers sit in one register per lane
            const int bw0 = wv * q + r0;
            const int nround = q - r0 < 60 ? q - r0 : 60;
            const int ng = nb - bw0 < 0 ? 0 : nb - bw0 < nround ? nb - bw0 : nround;
            // T' form (a.dcode): every major index reads the same nb groups -- no stream from HBM, the pattern lives in the L2
            // (ablation bit 32 does the same to an operator in the general form: what the pass costs without its stream)
            const int64_t gp = a.gia_n[((a.abl & 32) || a.dcode ? 0 : gbase) + (bw0 + lane < nb ? bw0 + lane : nb)];
            struct Pass {
                GroupStream<NG, NT> st;
                double yo, fr;
                int dc;
            };
            auto fetch = [&](Pass &P, int i0) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int i = i0 + gi < ng ? i0 + gi : (ng > 0 ? ng - 1 : 0);
                    P.st.load(gi, a.ja_n, a.code_n, __shfl(gp, i, 64), __shfl(gp, i + 1, 64), lane);
                }
                const int b = bw0 + i0 + (lane >> 4);
                const int64_t d = (int64_t)b * kGB + (lane & 15);
                const bool rowok = i0 + (lane >> 4) < ng && d < a.S && !(a.abl & 16);
                P.yo = (rowok && a.beta != 0.0) ? a.y[maj * a.S + d] : 0.0;
                P.fr = rowok ? a.far[((int64_t)b * a.NU + maj) * kGB + (lane & 15)] : 0.0;
                P.dc = (rowok && a.dcode) ? a.dcode[maj * a.S + d] : a.n_dict;          // n_dict = the padding code (value 0)
            };
            Pass cur;
            if (ng > 0) fetch(cur, 0);
            for (int i0 = 0; i0 < ng; i0 += NG) {
                Pass nxt;
                fetch(nxt, i0 + NG < ng ? i0 + NG : i0);
                double mine = 0.0;                       // the row sum this lane finishes: group lane >> 4 of the pass
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    double acc = 0.0;
#pragma unroll
                    for (int u = 0; u < UN; ++u)
                        if (u < cur.st.nu[gi])
                            acc += ((a.abl & 1) ? 1.0 : dict[cur.st.cod(gi, u)]) * ((a.abl & 2) ? 1.0 : win[cur.st.col(gi, u)]);
                    if (cur.st.nu[gi] > UN) {                                        // rows longer than 32 entries
                        const int64_t gb = __shfl(gp, i0 + gi < ng ? i0 + gi : ng - 1, 64) + (int64_t)lane * cur.st.nu[gi];
                        for (int u = UN; u < cur.st.nu[gi]; ++u) acc += dict[a.code_n[gb + u]] * win[a.ja_n[gb + u]];
                    }
                    if (!(a.abl & 4)) acc = quad_sum(acc);
                    if ((lane >> 4) == gi) mine = acc;
                }
                const int b = bw0 + i0 + (lane >> 4);
                const int64_t d = (int64_t)b * kGB + (lane & 15);
                if (i0 + (lane >> 4) < ng && d < a.S && !(a.abl & 16)) {
                    const double xi = win[d];
                    if (a.dcode) mine += dict[cur.dc] * xi;          // the diagonal entry, kept apart from T'
                    const double yn = a.alpha * (mine + cur.fr) + a.beta * cur.yo + a.gamma * xi;
                    a.y[maj * a.S + d] = yn;
                    acc3[0] += xi * yn;
                    acc3[1] += yn * yn;
                }
                cur = nxt;
            }
        }
    }
    if (a.partials != nullptr) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double v = acc3[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[c * 16 + wv] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double s0 = 0.0, s1 = 0.0;
            for (int w = 0; w < 16; ++w) {
                s0 += red[w];
                s1 += red[16 + w];
            }
            a.partials[(size_t)blockIdx.x * 3 + 0] = s0;
            a.partials[(size_t)blockIdx.x * 3 + 1] = 0.0;
            a.partials[(size_t)blockIdx.x * 3 + 2] = s1;
        }
    }
}

int launch_kronc_near_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_near_uniform, dim3(8192), dim3(256), 0, s, ia, ja, code, S, n, d_flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_s_widths(const int64_t *ia, const int32_t *ja, int64_t S, int nb, int32_t *ws, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_s_widths, dim3(64), dim3(256), 0, s, ia, ja, S, nb, ws);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_s_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int nb, int zcode, const int64_t *gs, uint16_t *scol,
                        uint8_t *scode, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_s_fill, dim3(256), dim3(256), 0, s, ia, ja, code, S, nb, (uint8_t)zcode, gs, scol, scode);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_dcode(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t n, int zcode, uint8_t *dcode, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_dcode, dim3(8192), dim3(256), 0, s, ia, ja, code, n, (uint8_t)zcode, dcode);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_far_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_far_uniform, dim3(8192), dim3(256), 0, s, ia, ja, code, S, n, d_flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_t_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int32_t *wt, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_t_widths, dim3(256), dim3(256), 0, s, ia, ja, S, NU, wt);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_kronc_t_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, int zcode, const int64_t *tp, uint16_t *tcol,
                        uint8_t *tcode, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_t_fill, dim3(256), dim3(256), 0, s, ia, ja, code, S, NU, (uint8_t)zcode, tp, tcol, tcode);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kronc_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int nb, int32_t *wn, int32_t *wf, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_widths, dim3(4096), dim3(256), 0, s, ia, ja, S, NU, nb, wn, wf);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kronc_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, int nb, int zcode, const int64_t *gia_n,
                      uint16_t *ja_n, uint8_t *code_n, const int64_t *gia_f, uint16_t *ja_f, uint8_t *code_f, hipStream_t s)
{
    hipLaunchKernelGGL(k_kronc_fill, dim3(8192), dim3(256), 0, s, ia, ja, code, S, NU, nb, (uint8_t)zcode, gia_n, ja_n, code_n, gia_f, ja_f, code_f);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

size_t kronc_near_lds_bytes(int64_t S) { return (size_t)(S + 256 + 48 + 2) * sizeof(double); }

// y <- alpha H x + beta y + gamma x on packed-double vectors; partials (or nullptr): 3 per workgroup of the near launch,
// *nparts_out workgroups.  ctr: 9 x 32 unsigned counters (the near launch's draw of major indices, the far launch's chunk counter
// per XCD); static_near (qbh_opts.deterministic): static turns in the near pass -- the partial sums of the reductions are then
// formed in the same order in every run.
int launch_kronc(const KroncSliced &K, const d2 *dict, int n_dict, const double *xt, const double *x, double *y, double alpha, double beta,
                 double gamma, double *partials, unsigned int *ctr, bool static_near, int *nparts_out, hipStream_t s)
{
    KroncArgs a{};
    a.gia_n = K.gia_n;
    a.gia_f = K.gia_f;
    a.tf_ptr = K.tf_ptr;
    a.dcode = K.near_uni ? K.dcode : nullptr;
    a.ja_n = K.ja_n;
    a.ja_f = K.ja_f;
    a.code_n = K.code_n;
    a.code_f = K.code_f;
    a.dict = dict;
    a.dictr = K.d_dictr;
    a.n_dict = n_dict;
    a.S = K.S;
    a.NU = K.NU;
    a.nb = K.nb;
    a.xt = xt;
    a.x = x;
    a.far = K.d_far;
    a.y = y;
    a.alpha = alpha;
    a.beta = beta;
    a.gamma = gamma;
    a.partials = partials;
    a.ctr = static_near ? nullptr : ctr;      // near pass: draw of major indices (static turns: reductions in a fixed order)
    a.fctr = ctr + 32;                        // far pass: chunk counters (its results do not depend on who computes a group)
    // tuning switches (measurement only): groups a wavefront has in flight per pass
    const int far_ng = debug_sw().kronc_far_ng ? debug_sw().kronc_far_ng : 1;
    const int far_nt = debug_sw().kronc_far_nt;
    auto far_k = K.far_uni ? (far_ng == 1 ? k_kronc_far<1, 8, false, true> : k_kronc_far<2, 8, false, true>)
                 : far_nt  ? (far_ng == 1 ? k_kronc_far<1, 8, true, false> : k_kronc_far<2, 8, true, false>)
                           : (far_ng == 1 ? k_kronc_far<1, 8, false, false> : far_ng == 3 ? k_kronc_far<3, 8, false, false> : k_kronc_far<2, 8, false, false>);
    auto near_k = K.near_uni ? k_kronc_near<4, 8, false> : k_kronc_near<4, 8, true>;
    // hipFuncSetAttribute applies per DEVICE: the bookkeeping is per device as well (a process may drive several GPUs), under a
    // lock (two host threads with one operator each)
    constexpr int kMaxDev = 16;
    static std::mutex mu;
    static int far_occ_dev[kMaxDev] = {0};
    static size_t attr_done_dev[kMaxDev][2] = {{0, 0}};          // per instance of the near kernel
    int dev = 0;
    QBH_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDev) dev = kMaxDev - 1;             // beyond the table: set the attribute every time (below)
    const size_t lds = kronc_near_lds_bytes(K.S);
    int far_occ = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (far_occ_dev[dev] == 0 || dev == kMaxDev - 1) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, far_k, 256, 0) != hipSuccess || n <= 0) n = 4;
            far_occ_dev[dev] = n > 8 ? 8 : n;
        }
        far_occ = far_occ_dev[dev];
        size_t &attr_done = attr_done_dev[dev][K.near_uni ? 1 : 0];
        if (attr_done < lds || dev == kMaxDev - 1) {
            QBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(near_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done = lds;
        }
    }
    {
        const int far_chunk = debug_sw().kronc_far_chunk;
        // the wavefronts of one XCD (32 CUs x far_occ workgroups x 4) together stay inside about one band
        int64_t c = far_chunk > 0 ? far_chunk : K.NU / ((int64_t)32 * far_occ * 4);
        a.chunk = (int)(c < 1 ? 1 : c > 32 ? 32 : c);
    }
    {
        a.abl = debug_sw().kronc_abl;
    }
    QBH_HIP(hipMemsetAsync(ctr, 0, 9 * 32 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(far_k, dim3(256 * far_occ), dim3(256), 0, s, a);
    QBH_HIP(hipGetLastError());
    const int grid_n = (int)(K.NU < 256 ? K.NU : 256);
    hipLaunchKernelGGL(near_k, dim3(grid_n), dim3(1024), lds, s, a);
    QBH_HIP(hipGetLastError());
    if (nparts_out) *nparts_out = grid_n;
    return QBH_OK;
}

}  // namespace qbh
