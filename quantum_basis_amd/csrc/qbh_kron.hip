// qbh_kron.hip -- the Kronecker split made IN PLACE (round 4).
//
// A stored operator on a product basis (index = major * S + minor; every entry keeps the major index -- near -- or keeps the
// minor index and changes the major one -- far; the two-species Hubbard family, src/model.cc:619-685 in species-major order) is
// re-ordered inside its own arrays: [ near entries, rows in natural order | far entries, rows band-major over the minor index,
// interleaved in groups of 8 rows ].  Same values, same int32 columns, same 20 B per nonzero as the CSR it replaces (SURVEY
// 8(d)); the CSR is NOT kept beside it -- qbh_csr_download and kron_restore merge the two parts back row by row (columns
// ascending: the original row, bit for bit).
//
// Row shards: rows [U0 * S, (U0 + NUloc) * S) of the operator, whole major indices.  Near columns are then all locally owned;
// far columns index the gathered x in RANK-MAJOR TILED order (KronCols): every rank contributes the tiled copy of its own
// block, so the all-gather moves contiguous blocks and a band of x is nranks contiguous pieces.
#include "qbh_internal.hpp"

namespace qbh {

namespace {

__device__ __forceinline__ bool edge_row(const KronTile &t, int64_t d) { return d >= (t.S / t.B) * t.B; }

// structure check: every entry of local row r (global major U0 + r / S, minor r % S) keeps the major or the minor index
__global__ __launch_bounds__(kBlock) void k_kron_check2(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int64_t U0, int *flag)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t maj = U0 + r / S, mnr = r % S;
        bool bad = false;
        for (int64_t k = ia[r]; k < ia[r + 1]; ++k) {
            const int64_t c = ja[k], cm = c / S;
            bad = bad || (cm != maj && c - cm * S != mnr);
        }
        if (bad) *flag = 1;
    }
}

// far / near entry counts; f = tiled (band-major) index of the local row.  Rows of the narrow last band (S % B != 0) keep
// their far entries in the NEAR part (the near pass gathers any column from the natural x): the far part then consists of
// whole groups of B rows with one major index each -- no padding for a product operator, whatever S is.
__global__ __launch_bounds__(kBlock) void k_kron_count2(const int64_t *ia, const int32_t *ja, int64_t nrows, KronTile t, int64_t U0,
                                                        int32_t *cnt_near, int32_t *cnt_far)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t.orig(f);
        const int64_t maj = U0 + r / t.S;
        int nf = 0;
        const int64_t s0 = ia[r], e0 = ia[r + 1];
        if (!edge_row(t, r % t.S))
            for (int64_t k = s0; k < e0; ++k) nf += (ja[k] / t.S) != maj;
        cnt_far[f] = nf;
        cnt_near[r] = (int)(e0 - s0) - nf;
    }
}

// far entries of far row f out of the CSR, to slot gia[g] + 8 k + j (sliced: f = 8 g + j, k-th far entry) or ia_f[f] + k.
// COL: the column, in the tiled order of `cols`; otherwise the value.  Padding slots of a sliced group: value 0, column =
// the row's own tiled index (never a genuine far column: a far entry changes the major index).
template <bool COL, bool SLICED>
__global__ __launch_bounds__(kBlock) void k_kron_far_fill(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, KronTile t, int64_t U0,
                                                          KronCols cols, const int64_t *fp, int64_t ngroups, int32_t *out_c, d2 *out_v)
{
    const int64_t nf = SLICED ? ngroups * 8 : nrows;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        int64_t base, stride, w;
        if (SLICED) {
            const int64_t g = f >> 3;
            base = fp[g] + (f & 7);
            stride = 8;
            w = (fp[g + 1] - fp[g]) >> 3;
        } else {
            base = fp[f];
            stride = 1;
            w = fp[f + 1] - fp[f];
        }
        int64_t k = 0;
        int32_t own = 0;
        if (f < nrows) {
            const int64_t r = t.orig(f);
            const int64_t maj = U0 + r / t.S;
            if (COL) own = (int32_t)cols.tile(U0 * t.S + r);
            if (!edge_row(t, r % t.S))
                for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
                    const int32_t c = ja[q];
                    if ((c / t.S) != maj) {
                        if (COL) out_c[base + stride * k] = (int32_t)cols.tile(c);
                        else     out_v[base + stride * k] = val[q];
                        ++k;
                    }
                }
        }
        if (SLICED)
            for (; k < w; ++k) {
                if (COL) out_c[base + stride * k] = own;
                else     out_v[base + stride * k] = d2{0.0, 0.0};
            }
    }
}

// near entries of rows [r0, r1) packed into tmp (row-major, CSR order) -- first half of one step of the in-place compaction
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kron_near_gather(const int64_t *ia, const int32_t *ja, const T *src, int64_t r0, int64_t r1, KronTile t,
                                                             int64_t U0, const int64_t *ia_n, T *tmp)
{
    const int64_t base = ia_n[r0];
    for (int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t maj = U0 + r / t.S;
        const bool edge = edge_row(t, r % t.S);
        int64_t pn = ia_n[r] - base;
        for (int64_t q = ia[r]; q < ia[r + 1]; ++q)
            if (edge || (ja[q] / t.S) == maj) tmp[pn++] = src[q];
    }
}

// rows [r0, r1) of the CSR back out of the two parts: near and far entries merged by ascending column
__global__ __launch_bounds__(kBlock) void k_kron_merge_rows(KronParts p, int64_t r0, int64_t r1, int32_t *out_ja, d2 *out_val, int64_t out_base)
{
    for (int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t pn = p.ia_n[r];
        const int64_t en = p.ia_n[r + 1];
        const int64_t f = p.t.tile(r);
        int64_t fb, fs, fw;
        if (f >= p.nfar_rows) {                 // a row of the narrow last band: everything is in the near part
            fb = 0;
            fs = 1;
            fw = 0;
        } else if (p.sliced) {
            const int64_t g = f >> 3;
            fb = p.fp[g] + (f & 7);
            fs = 8;
            fw = (p.fp[g + 1] - p.fp[g]) >> 3;
        } else {
            fb = p.fp[f];
            fs = 1;
            fw = p.fp[f + 1] - p.fp[f];
        }
        const int32_t own = (int32_t)p.cols.tile(p.U0 * p.t.S + r);
        int64_t k = 0;
        int64_t o = p.ia[r] - out_base;
        const int64_t oe = p.ia[r + 1] - out_base;
        int64_t cf = -1;
        auto next_far = [&]() {
            cf = -1;
            if (k < fw) {
                const int32_t ct = p.ja_f[fb + fs * k];
                if (ct != own) cf = p.cols.orig(ct);
            }
        };
        next_far();
        while (o < oe) {
            const int64_t cn = pn < en ? (int64_t)p.ja_n[pn] : -1;
            if (cf >= 0 && (cn < 0 || cf < cn)) {
                out_ja[o] = (int32_t)cf;
                out_val[o] = p.val_f[fb + fs * k];
                ++k;
                next_far();
            } else if (cn >= 0) {
                out_ja[o] = (int32_t)cn;
                out_val[o] = p.val_n[pn++];
            } else {
                out_ja[o] = -1;                   // the parts do not add up to the row: caught by the caller's checksum
                out_val[o] = d2{0.0, 0.0};
            }
            ++o;
        }
    }
}

// far columns from one tiled order of the gathered x to another (a communicator was attached or detached)
__global__ __launch_bounds__(kBlock) void k_kron_remap_cols(int32_t *ja_f, int64_t n, KronCols from, KronCols to)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ja_f[i] = (int32_t)to.tile(from.orig(ja_f[i]));
}

// y += alpha * far[tile(row)] and the fused reductions <x, y>, |y|^2 of the finished y: closes an SpMV whose near pass ran first
// (under a communicator: it only needs the rank's own block of x and overlaps the all-gather)
__global__ __launch_bounds__(kBlock) void k_kron_combine(const d2 *far, KronTile t, const d2 *xl, d2 *y, int64_t n, double alpha, double *partials)
{
    __shared__ double red[12];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        const d2 v = y[r] + alpha * far[t.tile(r)];
        y[r] = v;
        const d2 xi = xl[r];
        acc[0] += xi.x * v.x + xi.y * v.y;
        acc[1] += xi.x * v.y - xi.y * v.x;
        acc[2] += v.x * v.x + v.y * v.y;
    }
    if (partials != nullptr) {
        for (int c = 0; c < 3; ++c) {
            double v = acc[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[c * 4 + (threadIdx.x >> 6)] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int c = 0; c < 3; ++c) partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4 + 0] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}

}  // namespace

int launch_kron_check2(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int64_t U0, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_check2, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, S, U0, d_flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_count2(const int64_t *ia, const int32_t *ja, int64_t nrows, const KronTile &t, int64_t U0, int32_t *cnt_near, int32_t *cnt_far,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_count2, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, t, U0, cnt_near, cnt_far);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_far_fill(bool col, bool sliced, const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, const KronTile &t, int64_t U0,
                         const KronCols &cols, const int64_t *fp, int64_t ngroups, int32_t *out_c, d2 *out_v, hipStream_t s)
{
    const dim3 g(4096), b(kBlock);
    if (col && sliced)       hipLaunchKernelGGL((k_kron_far_fill<true, true>), g, b, 0, s, ia, ja, val, nrows, t, U0, cols, fp, ngroups, out_c, out_v);
    else if (col)            hipLaunchKernelGGL((k_kron_far_fill<true, false>), g, b, 0, s, ia, ja, val, nrows, t, U0, cols, fp, ngroups, out_c, out_v);
    else if (sliced)         hipLaunchKernelGGL((k_kron_far_fill<false, true>), g, b, 0, s, ia, ja, val, nrows, t, U0, cols, fp, ngroups, out_c, out_v);
    else                     hipLaunchKernelGGL((k_kron_far_fill<false, false>), g, b, 0, s, ia, ja, val, nrows, t, U0, cols, fp, ngroups, out_c, out_v);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_near_gather_cols(const int64_t *ia, const int32_t *ja, int64_t r0, int64_t r1, const KronTile &t, int64_t U0, const int64_t *ia_n,
                                 int32_t *tmp, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_near_gather<int32_t>, dim3(4096), dim3(kBlock), 0, s, ia, ja, ja, r0, r1, t, U0, ia_n, tmp);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_near_gather_vals(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t r0, int64_t r1, const KronTile &t, int64_t U0,
                                 const int64_t *ia_n, d2 *tmp, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_near_gather<d2>, dim3(4096), dim3(kBlock), 0, s, ia, ja, val, r0, r1, t, U0, ia_n, tmp);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_merge_rows(const KronParts &p, int64_t r0, int64_t r1, int32_t *out_ja, d2 *out_val, int64_t out_base, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_merge_rows, dim3(4096), dim3(kBlock), 0, s, p, r0, r1, out_ja, out_val, out_base);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_remap_cols(int32_t *ja_f, int64_t n, const KronCols &from, const KronCols &to, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_remap_cols, dim3(4096), dim3(kBlock), 0, s, ja_f, n, from, to);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_combine(const d2 *far, const KronTile &t, const d2 *xl, d2 *y, int64_t n, double alpha, double *partials, int *nparts, hipStream_t s)
{
    const int g = blas_grid(n);
    hipLaunchKernelGGL(k_kron_combine, dim3(g), dim3(kBlock), 0, s, far, t, xl, y, n, alpha, partials);
    QBH_HIP(hipGetLastError());
    if (nparts) *nparts = g;
    return QBH_OK;
}

}  // namespace qbh
