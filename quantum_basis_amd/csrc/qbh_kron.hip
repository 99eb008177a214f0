// qbh_kron.hip -- the Kronecker split made IN PLACE (round 4).
//
// A stored operator whose rows have a product structure (KronMap, qbh_internal.hpp) is re-ordered inside its own arrays:
//   [ near entries, rows in natural order | far entries, rows band-major inside their class, interleaved in groups of 8 rows |
//     cross entries, row-major ]
// Same values in the same order as the CSR it replaces; columns as int32 (20 B per nonzero) or -- round 5,
// qbh_opts.kron_cols16 -- 2 bytes each relative to a base the wave block's descriptor names (18 B); the CSR is NOT kept beside
// it -- qbh_csr_download and kron_restore merge the parts back row by row (columns ascending: the original row, bit for bit).
//
// One class: the two-species (Hubbard) operators in species-major order, src/model.cc:619-685 -- index = up * S + down; near =
// down hops + diagonal, far = up hops; the cross part is only the far entries of the S % 8 rows per major index that do not fill
// a band.  Row shards of whole major indices: near columns are all locally owned, far / cross columns index the gathered x in
// RANK-MAJOR TILED order (KronCols) -- every rank contributes the tiled copy of its own block.
// Several classes: a single-species (spin-1/2, fixed n_dn) sector with the sites cut into a low and a high half and the rows in
// class-major order (class = particle number of the high half): bonds inside the low half are near, inside the high half far,
// across the cut cross (tools/ragged_kron_analysis.py: 17-21 % of kagome-30's entries).
#include <algorithm>

#include "qbh_internal.hpp"

namespace qbh {

namespace {

__device__ __forceinline__ d2 cmul_h(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// structure check of a ONE-class operator: every entry of local row r (global major U0 + r / S, minor r % S) keeps the major or
// the minor index
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

// entries of every row in the three parts; the far count is filed under the row's far row id
__global__ __launch_bounds__(kBlock) void k_kron_count3(const int64_t *ia, const int32_t *ja, int64_t nrows, KronMap m, int32_t *cnt_near,
                                                        int32_t *cnt_far, int32_t *cnt_x)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        int n[3] = {0, 0, 0};
        for (int64_t k = ia[r]; k < ia[r + 1]; ++k) n[m.kind(r, ja[k])]++;
        cnt_near[r] = n[0];
        cnt_x[r] = n[2];
        const int64_t f = m.frow(r);
        if (f >= 0) cnt_far[f] = n[1];
    }
}

// far entries of far row f out of the CSR, to slot fp[g] + 8 k + j (sliced: f = 8 g + j, k-th far entry) or fp[f] + k.
// COL: the column, in the tiled order of x; otherwise the value.  Padding slots of a sliced group: value 0, column =
// the row's own tiled index (never a genuine far column: a far entry changes the block).
template <bool COL>
__global__ __launch_bounds__(kBlock) void k_kron_far_fill(const int64_t *ia, const int32_t *ja, const d2 *val, KronMap m, const int64_t *fp,
                                                          int64_t ngroups, int32_t *out_c, d2 *out_v)
{
    const int64_t nfr = m.nfar_rows();
    const int64_t nf = m.sliced ? ngroups * 8 : nfr;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        int64_t base, stride, w;
        if (m.sliced) {
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
        if (f < nfr) {
            const int64_t r = m.frow_orig(f);
            if (COL) own = (int32_t)m.xcol((m.nc == 1 ? m.U0 * m.S[0] : 0) + r);
            for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
                const int32_t c = ja[q];
                if (m.kind(r, c) == 1) {
                    if (COL) out_c[base + stride * k] = (int32_t)m.xcol(c);
                    else     out_v[base + stride * k] = val[q];
                    ++k;
                }
            }
        }
        if (m.sliced)
            for (; k < w; ++k) {
                if (COL) out_c[base + stride * k] = own;
                else     out_v[base + stride * k] = d2{0.0, 0.0};
            }
    }
}

// entries of part PART (0 near: natural columns; 2 cross: columns in the tiled order of x) of rows [r0, r1) packed into tmp.
// iap: row pointers of the part; rowidx != nullptr: iap is indexed by the position of the row in that compact row list
// (cross part of a one-class operator), [r0, r1) are then positions
template <typename T, int PART, bool COL>
__global__ __launch_bounds__(kBlock) void k_kron_part_gather(const int64_t *ia, const int32_t *ja, const T *src, int64_t r0, int64_t r1, KronMap m,
                                                             const int64_t *iap, const int32_t *rowidx, T *tmp)
{
    const int64_t base = iap[r0];
    for (int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = rowidx ? (int64_t)rowidx[i] : i;
        int64_t p = iap[i] - base;
        for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
            const int32_t c = ja[q];
            if (m.kind(r, c) == PART) {
                if constexpr (COL) tmp[p++] = (PART == 2) ? (T)m.xcol(c) : (T)c;
                else tmp[p++] = src[q];
            }
        }
    }
}

// compact row list of the rows that have cross entries: pos = exclusive scan of (cnt_x > 0)
__global__ __launch_bounds__(kBlock) void k_kron_flags(const int32_t *cnt, int64_t n, int32_t *flag01)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) flag01[r] = cnt[r] > 0 ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void k_kron_xrows(const int32_t *cnt_x, int64_t nrows, const int64_t *pos, int32_t *xrow, int32_t *cnt_compact)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x)
        if (cnt_x[r] > 0) {
            xrow[pos[r]] = (int32_t)r;
            cnt_compact[pos[r]] = cnt_x[r];
        }
}

// rows [r0, r1) of the CSR back out of the three parts: merged by ascending (original) column
__global__ __launch_bounds__(kBlock) void k_kron_merge_rows(KronParts p, int64_t r0, int64_t r1, int32_t *out_ja, d2 *out_val, int64_t out_base)
{
    const KronMap &m = p.map;
    for (int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t pn = p.ia_n[r];
        const int64_t en = p.ia_n[r + 1];
        // far list
        const int64_t f = m.frow(r);
        int64_t fb = 0, fs = 1, fw = 0;
        if (f >= 0) {
            if (m.sliced) {
                const int64_t g = f >> 3;
                fb = p.fp[g] + (f & 7);
                fs = 8;
                fw = (p.fp[g + 1] - p.fp[g]) >> 3;
            } else {
                fb = p.fp[f];
                fw = p.fp[f + 1] - p.fp[f];
            }
        }
        const int32_t own = (int32_t)m.xcol((m.nc == 1 ? m.U0 * m.S[0] : 0) + r);
        // 2-byte columns (one class): a near column is congruent to the stored value mod S and lies in the row's own block; a far
        // column keeps the row's minor index, its major index is the stored value mod NUg (padding: the row's own major index)
        const int64_t S0 = m.S[0], NUg = m.cols.cu[m.cols.nr], rmaj = m.U0 + r / S0, rmin = r % S0;
        // cross list
        int64_t px = 0, ex = 0;
        if (p.n_xrows > 0) {
            int64_t i = r;
            if (p.xrow) {                          // position of r in the compact row list (or none)
                int64_t lo = 0, hi = p.n_xrows;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (p.xrow[mid] < r) lo = mid + 1;
                    else hi = mid;
                }
                i = (lo < p.n_xrows && p.xrow[lo] == r) ? lo : -1;
            }
            if (i >= 0) {
                px = p.ia_x[i];
                ex = p.ia_x[i + 1];
            }
        }
        int64_t k = 0;
        int64_t o = p.ia[r] - out_base;
        const int64_t oe = p.ia[r + 1] - out_base;
        while (o < oe) {
            int64_t cn = -1, cf = -1, cx = px < ex ? m.xcol_orig(p.ja_x[px]) : -1;
            if (pn < en) cn = p.c16_n ? rmaj * S0 + (int64_t)p.c16_n[pn] % S0 : (int64_t)p.ja_n[pn];
            if (k < fw) {
                if (p.c16_f) {
                    const int64_t um = (int64_t)p.c16_f[fb + fs * k] % NUg;
                    if (um != rmaj) cf = um * S0 + rmin;
                } else {
                    const int32_t ct = p.ja_f[fb + fs * k];
                    if (ct != own) cf = m.xcol_orig(ct);
                }
            }
            int which = -1;
            int64_t best = -1;
            if (cn >= 0) { which = 0; best = cn; }
            if (cf >= 0 && (which < 0 || cf < best)) { which = 1; best = cf; }
            if (cx >= 0 && (which < 0 || cx < best)) { which = 2; best = cx; }
            out_ja[o] = (int32_t)best;               // -1 when the parts do not add up to the row
            out_val[o] = which == 0 ? p.val_n[pn] : which == 1 ? p.val_f[fb + fs * k] : which == 2 ? p.val_x[px] : d2{0.0, 0.0};
            if (which == 0) ++pn;
            else if (which == 1) ++k;
            else if (which == 2) ++px;
            ++o;
        }
    }
}

// far / cross columns from one tiled order of the gathered x to another (a communicator was attached or detached)
__global__ __launch_bounds__(kBlock) void k_kron_remap_cols(int32_t *ja_f, int64_t n, KronCols from, KronCols to)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ja_f[i] = (int32_t)to.tile(from.orig(ja_f[i]));
}

// y += alpha * far[tile(row)] and the fused reductions <x, y>, |y|^2 of the finished y: closes an SpMV whose near pass ran first
// (under a communicator: it only needs the rank's own block of x and overlaps the all-gather)
// coef != nullptr (pipelined three-term step): alpha is read from the device, as the near pass read it
__global__ __launch_bounds__(kBlock) void k_kron_combine(const d2 *far, KronTile t, const d2 *xl, d2 *y, int64_t n, double alpha, double *partials, const double *coef)
{
    __shared__ double red[12];
    double acc[3] = {0.0, 0.0, 0.0};
    if (coef != nullptr) alpha = coef[0];
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

// The cross part of a ONE-class operator is only the far entries of the few rows that do not fill a band (S % 8 rows per major
// index): 8 lanes per listed row, the row sum stored into that row's slot of the far buffer (the far pass never writes it), so
// the pass that adds the far result adds these as well.
__global__ __launch_bounds__(kBlock) void k_kron_cross_rows(const int64_t *ia_x, const int32_t *xrow, int64_t n_xrows, const int32_t *ja_x,
                                                            const d2 *val_x, const d2 *xt, KronTile t, d2 *far)
{
    const int sub = threadIdx.x & 7;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < ((n_xrows + 31) / 32) * 32; i += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        d2 sum = {0.0, 0.0};
        if (i < n_xrows)
            for (int64_t k = ia_x[i] + sub; k < ia_x[i + 1]; k += 8) sum += cmul_h(val_x[k], xt[ja_x[k]]);
        for (int off = 4; off > 0; off >>= 1) {
            sum.x += __shfl_xor(sum.x, off, 64);
            sum.y += __shfl_xor(sum.y, off, 64);
        }
        if (sub == 0 && i < n_xrows) far[t.tile(xrow[i])] = sum;
    }
}

// class of the first row of every near block, for the near pass's far lookup (several classes)
__global__ __launch_bounds__(kBlock) void k_kron_desc_classes(WaveDesc *wd, int64_t n_wb, const KronCls *cls, int nc)
{
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_wb + 2; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = wd[w].r0;
        int c = 0;
        while (c + 1 < nc && r >= cls[c + 1].rbase) ++c;
        wd[w].pad = c;
    }
}

// ---- 2-byte columns (qbh_opts.kron_cols16) ----
__global__ __launch_bounds__(kBlock) void k_kron_desc_c16(WaveDesc *wd, int64_t n_wb, int64_t div, int far, int undo)
{
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_wb + 2; w += (int64_t)gridDim.x * blockDim.x) {
        const int32_t cut = far ? (wd[w].pad & 1) : 0;
        const int64_t base = (undo || w >= n_wb) ? 0 : (int64_t)wd[w].r0 / div;        // the sentinels name element 0
        wd[w].pad = far ? (int32_t)(base << 1) | cut : (int32_t)base;
    }
}

// one wavefront per near block: entries [p0, p0(next)) relative to the block's base
__global__ __launch_bounds__(kBlock) void k_kron_c16_near(const WaveDesc *wd, int64_t n_wb, const int32_t *ja, int64_t S, int64_t col0, uint16_t *out, int *flag)
{
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_wb; w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t p0 = wd[w].p0, p1 = wd[w + 1].p0, base = col0 + (int64_t)wd[w].pad * S;     // col0 = the shard's first (global) column
        for (int64_t k = p0 + lane; k < p1; k += 64) {
            const int64_t c = (int64_t)ja[k] - base;
            bad = bad || c < 0 || c > 65535;
            out[k] = (uint16_t)c;
        }
    }
    if (bad) *flag = 1;
}

// far slot i (block i / 512): tiled column band * 8 NU + u' * 8 + j, j = i % 8 by construction
__global__ __launch_bounds__(kBlock) void k_kron_c16_far(const WaveDesc *wd, const int32_t *ja, int64_t slots, int64_t NU, uint16_t *out, int *flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t band0 = wd[i >> 9].pad >> 1;
        const int64_t col = ja[i], b = col / (8 * NU), rem = col - b * 8 * NU;
        const int64_t c = (rem >> 3) + (b - band0) * NU;
        bad = bad || (rem & 7) != (i & 7) || c < 0 || c > 65535 || wd[i >> 9].p0 != ((i >> 9) << 9);
        out[i] = (uint16_t)c;
    }
    if (bad) *flag = 1;
}

}  // namespace

int launch_kron_desc_c16(WaveDesc *wd, int64_t n_wb, int64_t div, bool far, bool undo, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_desc_c16, dim3(1024), dim3(kBlock), 0, s, wd, n_wb, div, far ? 1 : 0, undo ? 1 : 0);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_c16_near(const WaveDesc *wd, int64_t n_wb, const int32_t *ja, int64_t S, int64_t col0, uint16_t *out, int *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_c16_near, dim3(4096), dim3(kBlock), 0, s, wd, n_wb, ja, S, col0, out, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_c16_far(const WaveDesc *wd, const int32_t *ja, int64_t slots, int64_t NU, uint16_t *out, int *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_c16_far, dim3(4096), dim3(kBlock), 0, s, wd, ja, slots, NU, out, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_check2(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int64_t U0, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_check2, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, S, U0, d_flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_count3(const int64_t *ia, const int32_t *ja, int64_t nrows, const KronMap &map, int32_t *cnt_near, int32_t *cnt_far,
                       int32_t *cnt_x, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_count3, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, map, cnt_near, cnt_far, cnt_x);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_far_fill(bool col, const int64_t *ia, const int32_t *ja, const d2 *val, const KronMap &map, const int64_t *fp, int64_t ngroups,
                         int32_t *out_c, d2 *out_v, hipStream_t s)
{
    const dim3 g(4096), b(kBlock);
    if (col) hipLaunchKernelGGL((k_kron_far_fill<true>), g, b, 0, s, ia, ja, val, map, fp, ngroups, out_c, out_v);
    else     hipLaunchKernelGGL((k_kron_far_fill<false>), g, b, 0, s, ia, ja, val, map, fp, ngroups, out_c, out_v);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_part_gather_cols(int part, const int64_t *ia, const int32_t *ja, int64_t r0, int64_t r1, const KronMap &map, const int64_t *iap,
                                 const int32_t *rowidx, int32_t *tmp, hipStream_t s)
{
    const dim3 g(4096), b(kBlock);
    if (part == 0) hipLaunchKernelGGL((k_kron_part_gather<int32_t, 0, true>), g, b, 0, s, ia, ja, ja, r0, r1, map, iap, rowidx, tmp);
    else           hipLaunchKernelGGL((k_kron_part_gather<int32_t, 2, true>), g, b, 0, s, ia, ja, ja, r0, r1, map, iap, rowidx, tmp);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_part_gather_vals(int part, const int64_t *ia, const int32_t *ja, const d2 *val, int64_t r0, int64_t r1, const KronMap &map,
                                 const int64_t *iap, const int32_t *rowidx, d2 *tmp, hipStream_t s)
{
    const dim3 g(4096), b(kBlock);
    if (part == 0) hipLaunchKernelGGL((k_kron_part_gather<d2, 0, false>), g, b, 0, s, ia, ja, val, r0, r1, map, iap, rowidx, tmp);
    else           hipLaunchKernelGGL((k_kron_part_gather<d2, 2, false>), g, b, 0, s, ia, ja, val, r0, r1, map, iap, rowidx, tmp);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_flags(const int32_t *cnt, int64_t n, int32_t *flag01, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_flags, dim3(2048), dim3(kBlock), 0, s, cnt, n, flag01);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_xrows(const int32_t *cnt_x, int64_t nrows, const int64_t *pos, int32_t *xrow, int32_t *cnt_compact, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_xrows, dim3(2048), dim3(kBlock), 0, s, cnt_x, nrows, pos, xrow, cnt_compact);
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
    if (n <= 0) return QBH_OK;
    hipLaunchKernelGGL(k_kron_remap_cols, dim3(4096), dim3(kBlock), 0, s, ja_f, n, from, to);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_combine(const d2 *far, const KronTile &t, const d2 *xl, d2 *y, int64_t n, double alpha, double *partials, int *nparts, hipStream_t s,
                        const double *coef)
{
    const int g = blas_grid(n);
    hipLaunchKernelGGL(k_kron_combine, dim3(g), dim3(kBlock), 0, s, far, t, xl, y, n, alpha, partials, coef);
    QBH_HIP(hipGetLastError());
    if (nparts) *nparts = g;
    return QBH_OK;
}

int launch_kron_cross_rows(const int64_t *ia_x, const int32_t *xrow, int64_t n_xrows, const int32_t *ja_x, const d2 *val_x, const d2 *xt,
                           const KronTile &t, d2 *far, hipStream_t s)
{
    if (n_xrows <= 0) return QBH_OK;
    const int64_t g = std::min<int64_t>(2048, (n_xrows * 8 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_kron_cross_rows, dim3((unsigned)std::max<int64_t>(g, 1)), dim3(kBlock), 0, s, ia_x, xrow, n_xrows, ja_x, val_x, xt, t, far);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_desc_classes(WaveDesc *wd, int64_t n_wb, const KronCls *cls, int nc, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_desc_classes, dim3(1024), dim3(kBlock), 0, s, wd, n_wb, cls, nc);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

}  // namespace qbh
