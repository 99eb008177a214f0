// qbh_internal.hpp -- private definitions shared by the translation units of libqbhip.so
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "qbhip.h"

namespace qbh {

typedef double d2 __attribute__((ext_vector_type(2)));   // one complex128 (re, im)

constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr int kDictLds = 1024;          // largest value dictionary the row kernel keeps in LDS (2-byte codes)
constexpr int kMaxRedBlocks = 2048;  // grid of the BLAS-1 kernels == number of partial sums
constexpr int kRowCap = 1024;        // row offsets staged in LDS per row block

void set_error(const char *fmt, ...);

// Measurement, tuning and tracing knobs from the single environment variable QBH_DEBUG, read when an operator is CREATED (the handle
// keeps a snapshot, qbh_csr::dbg: nothing on a hot path touches the environment, and a change of the variable in the middle of a
// solve changes nothing) ("key=value,key=value";
// a bare integer n means flags=n).  None of them selects a form of the computation a caller could rely on -- those are fields
// of qbh_opts (include/qbhip.h).  0 / -1 = the library's own choice.
struct DebugSw {
    int flags = 0;            // bit 0: mask the gather columns so that they stay in cache (results wrong by design)
    int colmask = 0;          // the mask for flags & 1 (0: 1023)
    int tpr = 0, unroll = 0, grid = 0, wave_tpr = 0, chunk_mult = 0;      // launch-geometry overrides
    int trace_create = 0, trace_tune = 0, trace_dict = 0, print_ptrs = 0;
    int sec_walk = -1, sec_grid = 0, sec_unroll = 0, sec_tile = 0;        // matrix-free sector kernel
    int wave_pipelined = 0;   // the pipelined wave kernel on an unsplit operator
    long long create_chunk = 0;                                           // staging chunk of qbh_csr_create (nonzeros)
    int force_ragged = 0;     // native communicator: the send/recv all-gather-v even for uniform cuts
    int mf_row = -1, mf_chunk = 0, mf_window = 0;                         // matrix-free Hubbard kernel
    int kronc_abl = 0, kronc_far_chunk = 0, kronc_far_ng = 0, kronc_far_nt = 0;
    int no_far_align = 0;     // in-place split: the far part directly behind the near part (unaligned)
    int no_defer = 0;         // Lanczos: read <u, w> back every step instead of keeping it on the device
    int host_delay_us = 0;    // Lanczos: the host spins this long after every read-back of a step's scalars (models a slow / descheduled host)
    int pipe_nospec = 0;      // pipelined Lanczos loop: 1 = never enqueue a step before the one before it has been read back (debugging)
    int side_noprio = 0;      // native communicator: RCCL's side stream at the default priority (A/B of the priority, tools/solo_rank.py)
    int comm_reserve = 0;     // split shard under a communicator: overrides qbh_opts.comm_reserve (> 0: that many workgroups, < 0: none; A/B only)
    int comm_far_cap = 0;     // ... and the cap on the far pass's workgroups per CU (> 0: that many, < 0: none; A/B only)
};
const DebugSw &debug_sw();
void opts_builtin(qbh_opts *o);          // the built-in defaults, whatever qbh_opts_set_default says
inline void opts_generated(const qbh_opts *opts, qbh_opts *out)      // options of an operator the library generates itself
{
    if (opts) *out = *opts;
    else opts_builtin(out);
    out->basis_detect = 0;                 // its rows are in the generator's own order: no basis to look for
    if (out->basis_kind == QBH_BASIS_REF_FERMION2) out->basis_kind = QBH_BASIS_NONE;      // a host's hint about ITS arrays does not describe them
}

#define QBH_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            qbh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                      \
            (void)hipGetLastError(); /* the error is reported here: do not leave it sticky for the next call */ \
            return (_e == hipErrorOutOfMemory) ? QBH_ENOMEM : QBH_EHIP;                    \
        }                                                                                  \
    } while (0)

#define QBH_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != QBH_OK) return _rc; \
    } while (0)

// descriptor of one wave block of k_spmv_wave: the whole rows [r0, r0(next)) holding the nonzeros [p0, p0(next));
// entry n_wb is a sentinel {nnz, nrows}
constexpr int kWctrRegions = 3 + 8;

struct WaveDesc {
    int64_t p0;
    int32_t r0;
    int32_t pad;
};

// arguments of the SpMV kernels (passed by value)
struct SpmvArgs {
    const int64_t *ia;     // [nrows+1] local row pointers
    const int32_t *ja;     // [nnz] global columns
    const d2      *val;    // [nnz] values (or nullptr when dictionary-coded)
    const uint8_t *code;   // [nnz] dictionary codes (value_dict): 1 byte each, or 2 bytes when dict_mode >= 2
    const d2      *dict;   // dictionary (>= kDictLds entries allocated)
    int            dict_mode;   // 1: <= 256 values | 2: <= kDictLds values, 2-byte codes | 3: <= 65536, 2-byte codes
    const int32_t *rb;     // [n_blocks+1] first row of each row block (stream kernel)
    const int64_t *bp;     // [n_blocks+1] first nonzero of each row block (= ia[rb[.]])
    int64_t        n_blocks;
    int64_t        nrows;
    const d2      *xg;     // gather source (full-length x)
    const double  *xr;     // row kernel, real fast path: packed real parts of x (nullptr = complex gather)
    const d2      *xl;     // shard-local x (xg + row_offset without a communicator)
    d2            *y;
    double         alpha, beta, gamma;
    double        *partials;   // [grid*3] or nullptr
    int            swizzle;
    int            chunk_mult; // xcd_swizzle 2: rounds of an XCD's resident workgroups per contiguous chunk of row blocks
    int            unroll;     // k_spmv_rows: gathers in flight per lane and loop trip
    int            colmask;    // -1; QBH_DEBUG=1 sets 1023 so the gather stays in cache (timing experiments only)
    // all-real operation (real operator, real vectors, one GPU): y_re is the in/out vector stored as doubles, xl_re
    // the shard-local x as doubles (xr then is both the gather source and x_local); y / xl are unused
    double        *y_re;
    const double  *xl_re;
    // wave kernel (k_spmv_wave): one wavefront per block of whole rows with <= 512 nonzeros
    const WaveDesc *wd;
    int64_t        n_wb;
    unsigned long long *wctr;  // swizzle 3 (dynamic ordered walk): one counter per XCD, 16 apart, zeroed before the launch
    // Kronecker split (H = H_near + H_far, see KronSplit): the far pass stores plain row sums, the near pass adds them
    // back, reading the far result at the TILED index of its row
    const d2      *far;        // near pass: far-part row sums in tiled order (nullptr: none)
    int64_t        kS, kNU;    // minor size / major count of the product basis
    int            kB;         // band width of the tiling
    int            rowmap;     // row kernel, coded Kronecker split: rows are in TILED order; y / x_local are addressed at orig(row)
    const struct KronCls *kcls; // near pass of an operator with several classes (OPS 4): class table, nc + 1 entries
    // 2-byte columns of a part of the Kronecker split (k_spmv_wave2<.., C16 = true>; ja is then unused): relative to a base the
    // block's descriptor names (WaveDesc::pad) -- near part: column - pad * kS (pad = major index of the block's first row);
    // sliced far part: target major index + (band - (pad >> 1)) * kNU, the x element is (pad >> 1) * 8 kNU + 8 * that + lane % 8
    const uint16_t *ja16;
    // k_spmv_wave2 under the dynamic walk, passes with the fused epilogue: one slot of three sums per chunk of blocks
    // (wave2_chunk_slots(n_wb) slots; never nullptr for those launches), added up in a fixed order by launch_reduce_chunks
    double *chunk_red;
    // pipelined three-term step (lanczos_core, round 6): the beta term reads yin (nullptr = y: in place); with coef_mode != 0 the
    // coefficients come from device memory, left there by the step before (k_lanczos_tail) -- 1: alpha = coef[0], beta = coef[1];
    // 2: alpha = coef[0] only (a later pass accumulating onto y with its own beta)
    const d2      *yin;
    const double  *coef;
    int            coef_mode;
};
// first statement of every kernel that takes SpmvArgs: the coefficients a previous kernel of the stream left on the device
__device__ __forceinline__ void spmv_args_resolve(SpmvArgs &a)
{
    if (a.coef_mode != 0) {
        a.alpha = a.coef[0];
        if (a.coef_mode == 1) a.beta = a.coef[1];
    }
}

// element (u, d) of the product basis <-> its position in the band-major ("tiled") order (band, u, d % B): the B minor
// indices of one band are contiguous for every major index, so 8 consecutive far rows gather one 128-byte line per entry
struct KronTile {
    int64_t S, NU;
    int     B;
    __host__ __device__ int64_t tile(int64_t r) const
    {
        const int64_t u = r / S, d = r - u * S;
        const int64_t b = d / B, j = d - b * B;
        const int64_t wB = (S - b * B) < B ? (S - b * B) : B;
        return b * B * NU + u * wB + j;
    }
    __host__ __device__ int64_t orig(int64_t f) const
    {
        const int64_t full = (int64_t)B * NU;
        const int64_t b = f / full;
        const int64_t wB = (S - b * B) < B ? (S - b * B) : B;
        const int64_t rem = f - b * full;
        const int64_t u = rem / wB, j = rem - u * wB;
        return u * S + b * B + j;
    }
};

// Column order of the far part = order of the gathered x it reads: rank q (major indices [cu[q], cu[q+1])) contributes the
// tiled copy (KronTile{S, cu[q+1] - cu[q], B}) of its own block, blocks in rank order.  One rank (cu = {0, NU}): the tiled
// order of the whole vector.
constexpr int kKronMaxRanks = 16;
struct KronCols {
    int64_t S;
    int     B, nr;
    int64_t cu[kKronMaxRanks + 1];
    __host__ __device__ int64_t tile(int64_t c) const
    {
        const int64_t u = c / S;
        int q = 0;
        while (q + 1 < nr && u >= cu[q + 1]) ++q;
        return cu[q] * S + KronTile{S, cu[q + 1] - cu[q], B}.tile(c - cu[q] * S);
    }
    __host__ __device__ int64_t orig(int64_t i) const
    {
        int q = 0;
        while (q + 1 < nr && i >= cu[q + 1] * S) ++q;
        return cu[q] * S + KronTile{S, cu[q + 1] - cu[q], B}.orig(i - cu[q] * S);
    }
};
// Product structure of the rows of an operator split in place (qbh_kron.hip).
// One class (nc == 1): index = major * S + minor over the whole (local) row range -- the two-species operators; a row shard
// starts at major index U0 and its far columns follow `cols` (rank-major tiled order of the gathered x).
// Several classes (a single-species sector whose sites are cut into a low and a high half, rows in class-major order,
// unsharded): class c holds NU[c] blocks of S[c] rows from row rbase[c] on.
// An entry is NEAR when it stays inside the row's block, FAR when it stays inside the class and keeps the position inside the
// block, CROSS otherwise (and for the far entries of the rows of a narrow last band, S % B != 0: the far part consists of whole
// groups of B rows with one major index each -- no padding for a product operator, whatever S is).
constexpr int kKronMaxClasses = 24;
struct KronCls {                        // what the near pass needs of one class (device table, nc + 1 entries; the last is a sentinel)
    int64_t rbase, S, NU, fbase;
};
struct KronMap {
    int      nc, B, sliced;
    int      cross_near;                // several classes: the unstructured entries stay in the near part (two passes instead of three)
    int64_t  U0;
    int64_t  rbase[kKronMaxClasses + 1], S[kKronMaxClasses], NU[kKronMaxClasses];
    int64_t  fbase[kKronMaxClasses + 1];        // sliced: far rows of class c are numbered fbase[c] + tiled index (full bands only)
    KronCols cols;
    __host__ __device__ int cls(int64_t r) const
    {
        int c = 0;
        while (c + 1 < nc && r >= rbase[c + 1]) ++c;
        return c;
    }
    __host__ __device__ bool edge(int c, int64_t d) const { return d >= (S[c] / B) * B; }
    // 0 near, 1 far, 2 cross; r = local row, col = global column
    __host__ __device__ int kind(int64_t r, int64_t col) const
    {
        const int c = cls(r);
        const int64_t local = r - rbase[c], u = local / S[c], d = local - u * S[c];
        if (nc == 1) {
            const int64_t cm = col / S[0];
            if (cm == U0 + u) return 0;
            if (col - cm * S[0] == d) return edge(0, d) ? 2 : 1;
            return 2;
        }
        const int64_t lo = rbase[c] + u * S[c];
        if (col >= lo && col < lo + S[c]) return 0;
        if (col >= rbase[c] && col < rbase[c + 1] && (col - rbase[c]) % S[c] == d && !edge(c, d)) return 1;
        return cross_near ? 0 : 2;
    }
    // position of (global) column col in the tiled x the far and cross parts gather from, and back
    __host__ __device__ int64_t xcol(int64_t col) const
    {
        if (nc == 1) return cols.tile(col);
        const int c = cls(col);
        return rbase[c] + KronTile{S[c], NU[c], B}.tile(col - rbase[c]);
    }
    __host__ __device__ int64_t xcol_orig(int64_t i) const
    {
        if (nc == 1) return cols.orig(i);
        const int c = cls(i);
        return rbase[c] + KronTile{S[c], NU[c], B}.orig(i - rbase[c]);
    }
    // far row id of local row r (-1: a row of a narrow last band when sliced), and back
    __host__ __device__ int64_t frow(int64_t r) const
    {
        const int c = cls(r);
        const int64_t local = r - rbase[c];
        if (sliced && edge(c, local % S[c])) return -1;
        return (sliced ? fbase[c] : rbase[c]) + KronTile{S[c], NU[c], B}.tile(local);
    }
    __host__ __device__ int64_t frow_orig(int64_t f) const
    {
        int c = 0;
        if (sliced) while (c + 1 < nc && f >= fbase[c + 1]) ++c;
        else c = cls(f);
        return rbase[c] + KronTile{S[c], NU[c], B}.orig(f - (sliced ? fbase[c] : rbase[c]));
    }
    __host__ __device__ int64_t nfar_rows() const { return sliced ? fbase[nc] : rbase[nc]; }
};
// the three parts of an operator split in place (kernel argument of the merge)
struct KronParts {
    const int64_t *ia, *ia_n, *fp;      // CSR row pointers; near row pointers; far row pointers (or group pointers when sliced)
    const int32_t *ja_n, *ja_f;
    const uint16_t *c16_n, *c16_f;      // 2-byte forms of the near / far columns (nullptr: the int32 arrays hold them); see SpmvArgs::ja16
    const d2      *val_n, *val_f;
    const int64_t *ia_x;                // cross part: row pointers over xrow (compact row list) or over all rows (xrow == nullptr)
    const int32_t *xrow;
    int64_t        n_xrows;
    const int32_t *ja_x;                // columns in the tiled order of x (KronMap::xcol)
    const d2      *val_x;
    KronMap        map;
};
int launch_kron_count3(const int64_t *ia, const int32_t *ja, int64_t nrows, const KronMap &map, int32_t *cnt_near, int32_t *cnt_far,
                       int32_t *cnt_x, hipStream_t s);
int launch_kron_far_fill(bool col, const int64_t *ia, const int32_t *ja, const d2 *val, const KronMap &map, const int64_t *fp, int64_t ngroups,
                         int32_t *out_c, d2 *out_v, hipStream_t s);
// part 0 (near, natural columns) or 2 (cross, tiled columns) of rows [r0, r1) packed behind one another into tmp
int launch_kron_part_gather_cols(int part, const int64_t *ia, const int32_t *ja, int64_t r0, int64_t r1, const KronMap &map, const int64_t *iap,
                                 const int32_t *rowidx, int32_t *tmp, hipStream_t s);
int launch_kron_part_gather_vals(int part, const int64_t *ia, const int32_t *ja, const d2 *val, int64_t r0, int64_t r1, const KronMap &map,
                                 const int64_t *iap, const int32_t *rowidx, d2 *tmp, hipStream_t s);
int launch_kron_xrows(const int32_t *cnt_x, int64_t nrows, const int64_t *pos, int32_t *xrow, int32_t *cnt_compact, hipStream_t s);
int launch_kron_flags(const int32_t *cnt, int64_t n, int32_t *flag01, hipStream_t s);
int launch_kron_merge_rows(const KronParts &p, int64_t r0, int64_t r1, int32_t *out_ja, d2 *out_val, int64_t out_base, hipStream_t s);
int launch_kron_remap_cols(int32_t *ja_f, int64_t n, const KronCols &from, const KronCols &to, hipStream_t s);
int launch_kron_combine(const d2 *far, const KronTile &t, const d2 *xl, d2 *y, int64_t n, double alpha, double *partials, int *nparts, hipStream_t s, const double *coef = nullptr);
// sparse cross part (nc == 1: the far entries of the rows of the narrow last band): row sums into the far buffer's slots of those rows
int launch_kron_cross_rows(const int64_t *ia_x, const int32_t *xrow, int64_t n_xrows, const int32_t *ja_x, const d2 *val_x, const d2 *xt,
                           const KronTile &t, d2 *far, hipStream_t s);
int launch_kron_desc_classes(WaveDesc *wd, int64_t n_wb, const KronCls *cls, int nc, hipStream_t s);
// 2-byte columns: the base every block's columns are relative to goes into its descriptor (far = false: pad = r0 / div, the major
// index of the block's first row; far = true: pad = cut flag | band of the block's first group << 1), undo restores the plain form
int launch_kron_desc_c16(WaveDesc *wd, int64_t n_wb, int64_t div, bool far, bool undo, hipStream_t s);
// near columns of every block relative to its base / far columns as target major index relative to the block's band; *flag is
// raised when a value does not fit 16 bits (or a far column is not what the sliced layout promises)
int launch_kron_c16_near(const WaveDesc *wd, int64_t n_wb, const int32_t *ja, int64_t S, int64_t col0, uint16_t *out, int *flag, hipStream_t s);
int launch_kron_c16_far(const WaveDesc *wd, const int32_t *ja, int64_t slots, int64_t NU, uint16_t *out, int *flag, hipStream_t s);
int launch_kron_check2(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int64_t U0, int *d_flag, hipStream_t s);
// qbh_opts.basis_kind (qbh_reorder.hip): re-express the plain CSR of A in the library's internal order, keep the vector map
int basis_to_internal(qbh_csr *A, int kind, int n_sites, int n_up, int n_dn, bool *applied);
int basis_detect(qbh_csr *A, bool *applied);     // qbh_opts.basis_detect: try every two-species basis of the operator's dimension
int launch_basis_scatter(const uint32_t *map, const d2 *in, d2 *out, int64_t n, hipStream_t s);
int launch_basis_gather(const uint32_t *map, const d2 *in, d2 *out, int64_t n, hipStream_t s);
int launch_basis_scatter_re(const uint32_t *map, const double *in, double *out, int64_t n, hipStream_t s);

// hipMalloc that releases live Kronecker splits (second copies of a matrix: acceleration structures, qbh_api.cpp) before it
// reports out of memory.  Every allocation of the library except the splits' own goes through it.
hipError_t device_alloc(void **p, size_t bytes);
template <typename T>
inline hipError_t dev_alloc(T **p, size_t bytes)
{
    return device_alloc(reinterpret_cast<void **>(p), bytes);
}

// launchers implemented in qbh_kernels.hip (all asynchronous on `s`)
int launch_spmv(const SpmvArgs &a, int kernel, int npb, int tpr, int grid, hipStream_t s);
int spmv_grid(int kernel, int64_t n_blocks, int64_t nrows, int tpr);
int rows_kernel_occupancy(int npb, int tpr, int un, int dict_mode);
int launch_spmv_wave(const SpmvArgs &a, int tpr, int grid, hipStream_t s);
int launch_spmv_wave2(const SpmvArgs &a, int tpr, int ops, int grid, hipStream_t s);   // pipelined; ops 0 plain store, 2 epilogue + far addend
int wave2_kernel_occupancy(int tpr, int ops);
int64_t wave2_chunk_slots(int64_t n_wb);
int launch_reduce_chunks(const double *slots, int64_t n_slots, double *partials, int *nparts_out, hipStream_t s);
int launch_kron_tile(const d2 *x, d2 *xt, int64_t n, const KronTile &t, hipStream_t s, int xt_real = 0, int *flag = nullptr);
int launch_kron_check(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int *d_flag, hipStream_t s);
int launch_kron_count(const int64_t *ia, const int32_t *ja, int64_t nrows, const KronTile &t, int32_t *cnt_near, int32_t *cnt_far, hipStream_t s);
int launch_build_slotdesc(const int64_t *gia, int64_t ngroups, int64_t slots, WaveDesc *wd, int64_t n_wb, int64_t shift, hipStream_t s);
int launch_zero_cut_groups(const WaveDesc *wd, int64_t n_wb, int64_t nrows, d2 *far, hipStream_t s);
int launch_kron_group_width(const int32_t *cnt_far, int64_t nrows, int64_t ngroups, int32_t *gw, hipStream_t s);
int launch_kron_fill_sliced(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, const KronTile &t, const int64_t *ia_n,
                            int32_t *ja_n, d2 *val_n, const int64_t *gia, int64_t ngroups, int32_t *ja_f, d2 *val_f, hipStream_t s);
// the same split for dictionary-coded values (cw bytes per code) and the tiled copy of a packed-double x
int launch_kron_fill_codes(const int64_t *ia, const int32_t *ja, const uint8_t *code, int cw, int64_t nrows, const KronTile &t, const int64_t *ia_n,
                           int32_t *ja_n, uint8_t *code_n, const int64_t *ia_f, int32_t *ja_f, uint8_t *code_f, hipStream_t s);
int launch_kron_tile_re(const double *x, double *xt, int64_t n, const KronTile &t, hipStream_t s);
int launch_kron_fill(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, const KronTile &t, const int64_t *ia_n, int32_t *ja_n,
                     d2 *val_n, const int64_t *ia_f, int32_t *ja_f, d2 *val_f, hipStream_t s);
int wave_kernel_occupancy(int tpr);
int launch_build_wavedesc(const int64_t *d_ia, int64_t nrows, int64_t window, WaveDesc *d_wd, int64_t n_wb, hipStream_t s);
int vector_kernel_occupancy(int tpr, int un, bool dict);
int launch_build_rowblocks(const int64_t *d_ia, int64_t nrows, int64_t window, int32_t *d_rb,
                           int64_t *d_bp, int64_t n_blocks, hipStream_t s);
int launch_reduce_partials(const double *partials, int nparts, int ncomp, double *out, hipStream_t s);
int launch_dotc(const d2 *x, const d2 *y, int64_t n, double *partials, hipStream_t s);
int launch_axpy_norm(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, int64_t n, double *partials, double *yr, int *flag,
                     hipStream_t s, const double *scale_dev = nullptr);
// the same passes writing the TILED copy of the updated y as well (Kronecker split, band 8: the next SpMV's far-pass gather source)
int launch_axpy_norm_tile(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, d2 *yt, int64_t n, const KronTile &t, double *partials,
                          hipStream_t s, const double *scale_dev = nullptr, int yt_real = 0, int *flag = nullptr);
// the tiled blocks of the ranks, as gathered (rank after rank; complex or packed real parts), moved into the tiled order of the
// whole vector the far part of every shard indexes: elements [off[q], off[q] + len[q]) of rank q's block
struct KronPlace {
    const d2 *src;                   // d_xfull (or, real != 0, d_xfull_r viewed as doubles)
    d2       *dst;                   // the handle's tiled x (ncols elements)
    int       real, nr, B;
    int64_t   S, NUg, nfb;           // minor size, major indices of the whole operator, full bands
    int64_t   cu[kKronMaxRanks + 1]; // major-index cuts of the ranks
    int64_t   base[kKronMaxRanks];   // first element of rank q's block in src
    int64_t   off[kKronMaxRanks], len[kKronMaxRanks];
    // only the major indices this shard's far / cross entries read (k_kron_need): list[lo[q] .. lo[q + 1]) = the needed LOCAL major
    // indices of rank q, ascending; band0 .. band1 = the bands of this piece (band1 == nfb + 1: the narrow edge band as well).
    // list == nullptr: everything (the element ranges off / len above)
    const int32_t *list;
    int64_t   lo[kKronMaxRanks + 1];
    int64_t   band0, band1;
    // compact != 0 (personalised exchange): rank q's piece in src holds ONLY the listed major indices, band-major
    // (band b, i-th listed major, j) at base[q] + b B m_q + i B + j with m_q = lo[q + 1] - lo[q] -- what k_kron_pack wrote on rank q
    int       compact;
    int       skip;                  // compact: this rank's range of the list is not in src (the own rank: placed from its tiled block); -1: none
};
int launch_kron_place(const KronPlace &a, hipStream_t s);
// personalised exchange, sender: the listed major indices of the rank's own tiled block, packed per destination
// (dest p: [b][i][j] at base[p], i over list[lo[p] .. lo[p + 1])); real != 0: 8-byte elements
struct KronPack {
    const d2 *src;                   // the tiled copy of the own block (d_xsend; complex or, real != 0, packed real parts)
    d2       *dst;                   // d_vsend
    int       real, nr, B;
    int64_t   S, NUq, nfb;           // minor size, major indices of THIS rank, full bands
    const int32_t *list;
    int64_t   lo[kKronMaxRanks + 1];
    int64_t   base[kKronMaxRanks];
};
int launch_kron_pack(const KronPack &a, hipStream_t s);
// need[u] = 1 for every major index u of the WHOLE operator that a far (2-byte or int32 columns) or cross entry of this shard reads
int launch_kron_need(const uint16_t *c16_f, const int32_t *ja_f, int64_t far_slots, const int32_t *ja_x, int64_t nnz_x, int64_t S, int64_t NUg, int B,
                     uint8_t *need, hipStream_t s);
// tail of a pipelined Lanczos step: |w'|^2 from the axpy's partial sums, a = sc_x * <u, w>, b = sqrt(|w'|^2), the next step's
// coefficients into state[0..3], {<u,w>, |w'|^2, a, b} into log_slot (host-visible)
int launch_lanczos_tail(const double *partials, int nparts, const double *dot, double *state, double *log_slot, double sc_x_host, int use_host,
                        hipStream_t s, const double *sq_ready = nullptr);
int launch_xpby_tile(const d2 *x, double b, d2 *y, d2 *yt, int64_t n, const KronTile &t, hipStream_t s, int yt_real = 0, int *flag = nullptr);
int launch_nrm2sq(const d2 *x, int64_t n, double *partials, hipStream_t s);
int launch_scal(double a, d2 *x, int64_t n, hipStream_t s);
int launch_scal_to(double a, const d2 *x, d2 *y, int64_t n, hipStream_t s);
int launch_axpy_norm_re(double alpha, const double *alpha_dev, const double *x, double *y, int64_t n, double *partials, hipStream_t s,
                        double *yt = nullptr, const KronTile &t = KronTile{1, 1, 1});
int launch_cg_update_re(double alpha, const double *p, const double *pp, double *v, double *r, int64_t n, double *partials, hipStream_t s);
int launch_xpby_re(const double *x, double b, double *y, int64_t n, hipStream_t s);
int launch_dot_re(const double *x, const double *y, int64_t n, double *partials, hipStream_t s);
int launch_nrm2sq_re(const double *x, int64_t n, double *partials, hipStream_t s);
int launch_scal_re(double a, double *x, int64_t n, hipStream_t s);
int launch_xpby(const d2 *x, double b, d2 *y, int64_t n, double *yr, int *flag, hipStream_t s);            // y = x + b*y
int launch_cg_update(d2 alpha, const d2 *p, const d2 *pp, d2 *v, d2 *r, int64_t n, double *partials,
                     hipStream_t s, const double *delta_dev = nullptr, double accu2 = 0.0);                                               // v+=a p; r-=a pp; |r|^2
int launch_randomize(d2 *x, double *xr, int64_t n, int64_t global_offset, uint32_t seed, double *partials, hipStream_t s, const int32_t *major_inv = nullptr, int64_t S = 0);
int launch_fill_const(d2 *x, int64_t n, double re, hipStream_t s);
int launch_max_rowlen(const int64_t *d_ia, int64_t nrows, int64_t *d_out, hipStream_t s);
int blas_grid(int64_t n);

// sliced form of the coded Kronecker split (qbh_kronc.hip): groups of 16 rows, entry k of row j at gia[g] + 16 k + j
struct KroncSliced {
    bool      active = false;
    int64_t   S = 0, NU = 0;
    int       nb = 0;                   // bands of 16 minor indices (the last may be narrower)
    int64_t   slots_n = 0, slots_f = 0; // stored entries (nonzeros + padding)
    int64_t  *gia_n = nullptr, *gia_f = nullptr;      // [nb * NU + 1] group pointers: near groups (maj, b), far groups (b, maj)
    uint16_t *ja_n = nullptr;           // near columns relative to the major index's block
    uint16_t *ja_f = nullptr;           // far columns: the target major index (gathered from the tiled x, KronTile{S, NU, 16})
    bool      near_uni = false;         // the near part is 1 (x) T' + D: gia_n / ja_n / code_n hold the nb groups of T' (shared by all major indices), dcode the diagonal codes
    uint8_t  *dcode = nullptr;
    bool      far_uni = false;          // the far part is T (x) 1: ja_f / code_f hold T (entries of major index u at tf_ptr[u] ..), no far groups
    int64_t  *tf_ptr = nullptr;
    uint8_t  *code_n = nullptr, *code_f = nullptr;
    double   *d_far = nullptr;          // [nb * NU * 16] far row sums in far-group order
    double   *d_dictr = nullptr;        // [256] real parts of the value dictionary, zero from entry n_dict on (the padding code)
};
int launch_kronc_near_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *d_flag, hipStream_t s);
int launch_kronc_s_widths(const int64_t *ia, const int32_t *ja, int64_t S, int nb, int32_t *ws, hipStream_t s);
int launch_kronc_s_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int nb, int zcode, const int64_t *gs, uint16_t *scol,
                        uint8_t *scode, hipStream_t s);
int launch_kronc_dcode(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t n, int zcode, uint8_t *dcode, hipStream_t s);
int launch_kronc_far_uniform(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t n, int *d_flag, hipStream_t s);
int launch_kronc_t_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int32_t *wt, hipStream_t s);
int launch_kronc_t_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, int zcode, const int64_t *tp, uint16_t *tcol,
                        uint8_t *tcode, hipStream_t s);
int launch_kronc_widths(const int64_t *ia, const int32_t *ja, int64_t S, int64_t NU, int nb, int32_t *wn, int32_t *wf, hipStream_t s);
int launch_kronc_fill(const int64_t *ia, const int32_t *ja, const uint8_t *code, int64_t S, int64_t NU, int nb, int zcode, const int64_t *gia_n,
                      uint16_t *ja_n, uint8_t *code_n, const int64_t *gia_f, uint16_t *ja_f, uint8_t *code_f, hipStream_t s);
size_t kronc_near_lds_bytes(int64_t S);
int launch_kronc(const KroncSliced &K, const d2 *dict, int n_dict, const double *xt, const double *x, double *y, double alpha, double beta,
                 double gamma, double *partials, unsigned int *ctr, bool static_near, int *nparts_out, hipStream_t s);
int launch_pack_real(const d2 *x, double *out, int64_t n, int *flag, hipStream_t s);
int launch_unpack_real(const double *in, d2 *out, int64_t n, hipStream_t s);
int launch_imag_norm(const d2 *x, int64_t n, double *partials, hipStream_t s);
int exclusive_scan(const int32_t *d_cnt, int64_t n, int64_t *d_ia, hipStream_t s);
int launch_split_count(const int64_t *ia, const int32_t *ja, int64_t nrows, int32_t lo, int32_t hi, int32_t *cnt0, hipStream_t s);
int launch_split_fill(const int64_t *ia, const int32_t *ja, const d2 *val, const uint8_t *code, int64_t nrows, int32_t lo,
                      int32_t hi, const int64_t *ia0, int32_t *ja0, d2 *val0, uint8_t *code0, int64_t *ia1, int32_t *ja1,
                      d2 *val1, uint8_t *code1, int code_w, hipStream_t s);
struct Coef8 { double v[16]; };   // up to 8 complex coefficients passed by value
int launch_multi_dot8(const d2 *V, int64_t ldv, const d2 *w, int64_t n, int nv, double *partials, hipStream_t s);
int launch_multi_axpy8(const d2 *V, int64_t ldv, const Coef8 &c, int nv, d2 *w, int64_t n, double *partials, hipStream_t s);
int launch_basis_rotate(d2 *V, int64_t ldv, int64_t n, int m, int keep, const double *d_S, hipStream_t s);
int symmetric_eigen_jacobi(int m, double *a, double *w, double *z);
int build_value_dict(const d2 *d_val, int64_t nnz, int cap, uint8_t **d_code_out, d2 **d_dict_out, int *n_out, hipStream_t s);

// host CSR -> device row shard (qbh_build.hip)
int host_threads();
int validate_host_csr(int64_t dim, int64_t nnz, int sym_upper, const int64_t *ia, const int64_t *ja);
int check_hermitian_host(int64_t dim, const int64_t *ia, const int64_t *ja, const d2 *hv);
int build_shard_from_host(int64_t dim, int64_t nnz, int sym, const int64_t *ia, const int64_t *ja, const d2 *val, int64_t r0,
                          int64_t r1, hipStream_t s, int64_t **d_ia_out, int32_t **d_ja_out, d2 **d_val_out, int64_t *nnz_out,
                          double *ms_out);
int balanced_row_cuts(int64_t dim, int64_t nnz, int sym, const int64_t *ia, const int64_t *ja, int nranks, int64_t *cuts);

// native RCCL communicator (qbh_comm.cpp)
void release_native_comm(qbh_csr *A);
void harvest_native_comm(qbh_csr *A);      // adds the event-timed duration of the last gather to stats.ms_gather

// host tridiagonal solver (qbh_hess.cpp)
int tridiag_eigen_full(int64_t m, const double *a, const double *b1, double *w, double *z);
int tridiag_eigen_lastrow(int64_t m, const double *a, const double *b1, double *w, double *zlast);
int tridiag_lowest(int64_t m, const double *a, const double *b1, int nev, double *ritz, double *zlast0);

}  // namespace qbh

namespace qbh {
// matrix-free two-species (Hubbard) operator: H = T_up (x) 1 + 1 (x) T_dn + U * (double occupancy), applied from the
// two small hop tables instead of a stored CSR (counterpart of the matrix-free model<T>::MultMv2, src/model.cc:941-1109)
struct MfHubbard {
    int64_t Nu = 0, Nd = 0;
    int     wu = 0, wd = 0;          // padded hops per configuration (ELL width)
    double  U = 0.0;
    uint32_t *cfg_u = nullptr, *cfg_d = nullptr;
    // ELL tables, entry k of configuration c at [k*N + c]; padding = (c itself, amplitude 0).  Kept small so that they
    // stay L2-resident next to the x window: targets as uint32 (N < 2^24 configurations per species), amplitudes as
    // 1-byte codes into amp[] (<= 16 distinct hopping amplitudes: +-t times the bond multiplicity, and 0)
    uint32_t *tgt_u = nullptr, *tgt_d = nullptr;
    uint8_t  *val_u = nullptr, *val_d = nullptr;
    // down-species table once more as packed words {target | code << 24}, hops 4j..4j+3 of configuration c in the
    // 16 bytes at [(j * N + c) * 4] (one coalesced load per four hops in the row-staged kernel)
    uint32_t *pk_d = nullptr;
    double    amp[16] = {0};
};

// Matrix-free part of a Hubbard momentum-sector operator (qbh_mf_hubbard_repr).  Representatives are ordered by the down
// pattern first; a "regular" down block (trivially stabilised down pattern) holds EVERY up pattern, its up hops are the
// full-basis up-hop table applied inside the block and each allowed down hop is (target block, translation, coefficient).
// Rows in or next to stabilised blocks are a small stored CSR remainder (the handle's ordinary CSR arrays).
struct MfSecBlock {
    int64_t  row0;               // first row of the block
    uint32_t d;                  // down pattern
    int32_t  hop0, nhop;         // its regular down hops in hop[]
    int32_t  nrows;              // C(N, n_up) for a regular block
    int32_t  regular;
};
struct MfSecHop {
    int64_t off;                 // first row of the target block
    int32_t g, pad;              // canonicalising translation
    double  cr, ci;              // amplitude * hop sign * sign(g, down part) * conj(chi(g))
};
struct MfSec {
    int      n_sites = 0, n_up = 0, n_dn = 0, n_trans = 0, w_up = 0, n_pairs = 0, tile = 1024;
    int64_t  dim = 0, cu = 0, n_blocks = 0, n_items = 0;
    double   U = 0.0;
    MfSecBlock *blk = nullptr;   // [n_blocks], ascending down pattern
    MfSecHop   *hop = nullptr;
    int64_t    *item = nullptr;  // [n_items] work items: block << 20 | tile (1024 rows per tile)
    uint32_t   *ucfg = nullptr;  // [cu] up patterns
    uint32_t   *upell = nullptr; // [w_up][cu]: code << 24 | target rank, 0xFFFFFFFF = none
    uint32_t   *prank = nullptr; // [n_trans][cu]: parity << 31 | rank of the translated up pattern
    double      updict[256] = {0};
    // ORBIT ORDER of the rows of a regular block (qbh_opts.sector_orbit): position p holds the up pattern image(e, u0) of an
    // orbit {image(g, u0)} of the translation group, the orbits one after the other (ascending smallest member u0), their
    // members in ascending order of the first group element e that produces them.  A translated pattern is then a member of
    // the SAME orbit -- a down hop reads its target block at the same positions, permuted inside runs of <= n_trans rows --
    // and the up hops of a member are the images of the hops of u0, so their table is per orbit, not per row:
    //   down hop (block B', translation g):  x[B'.row0 + p - kidx[kind][e] + kidx[kind][comp[g][e]]] * sign(bit g of tpar[p])
    //   up hop, slot k of the orbit (target orbit at base', kind', group element s with hop_k(u0) = image(s, u0')):
    //                                        x[B.row0 + base' + kidx[kind'][comp[e][s]]] * updict[code] * sign(bit k of usgn[p])
    // comp[a][b] = the group element "b, then a"; kidx[kind][a] = position of image(a, u0) inside an orbit whose stabiliser
    // is of that kind (kind 0: trivial stabiliser, kidx = identity).
    int         orbit = 0, w_orb = 0, n_kinds = 0;
    int64_t     n_orb = 0;
    uint32_t   *oid = nullptr;   // [cu] orbit of position p
    uint16_t   *oek = nullptr;   // [cu] e | kind << 6
    uint64_t   *tpar = nullptr;  // [cu] bit g: parity of translation g on the up pattern at p
    uint64_t   *usgn = nullptr;  // [cu] bit k: fermion sign of up-hop slot k for the pattern at p
    uint32_t   *utab = nullptr;  // [w_orb][n_orb]: base' | s << 24 | ext << 30 | valid << 31
    uint16_t   *uext = nullptr;  // [w_orb][n_orb]: code | kind' << 8, read only where ext is set (else code 0, kind' 0)
    uint8_t     comp[64 * 64] = {0};
    uint8_t     kidx[16 * 64] = {0};
    double      nup[32] = {0}, ndn[32] = {0};          // number-operator terms per site
    int8_t      pi[128] = {0}, pj[128] = {0};
    double      pv[128][4] = {{0}};
    bool        all_real = true, has_number_terms = false;
    // stored remainder, compact: only the rows that have entries
    int64_t     n_rrows = 0, rnnz = 0;
    int32_t    *rrow = nullptr;  // [n_rrows] row index
    int64_t    *ria = nullptr;   // [n_rrows + 1]
    int32_t    *rja = nullptr;
    d2         *rval = nullptr;
};
struct MfSecArgs {
    const MfSec *t;              // device copy
    int64_t n_items, dim, n_rrows;
    const int32_t *rrow;
    const int64_t *ria;
    const int32_t *rja;
    const d2 *rval;
    const d2 *xg, *xl;
    const double *xr, *xl_re;
    d2 *y;
    double *y_re;
    double alpha, beta, gamma;
    double *partials;            // [nparts * 3] or nullptr
    unsigned int *ctr;           // ordered walk: 8 zeroed counters, 128 bytes apart (nullptr: static assignment)
    int orbit;                   // the tables are in orbit order (MfSec::orbit): k_mf_sector_orb
    // the orbit-order tables once more as kernel arguments: pointers read out of *t are generic (flat loads), these are global
    const uint32_t *ucfg, *oid, *utab;
    const uint16_t *oek, *uext;
    const uint64_t *tpar, *usgn;
    int64_t n_orb;
    int w_orb, tile;
};
// y <- alpha H x + beta y + gamma x in three launches (block tables, remainder rows, reductions); *nparts_out = partial sums
int launch_mf_sector(const MfSecArgs &a, hipStream_t s, int *nparts_out);
int adopt_mf_sector(qbh_csr **out, MfSec *host_tables, MfSec *dev_tables, int64_t dim, int64_t nnz_equiv, const qbh_opts *opts);

struct MfArgs {
    MfHubbard t;
    int64_t row_begin, nrows;
    const d2 *xg, *xl;
    const double *xr;
    d2 *y;
    double alpha, beta, gamma;
    double *partials;
    double *y_re;                // all-real operation: y stored as doubles (x_local = xr)
    // the diagonal as one value code per row + the real parts of the value dictionary (an operator RECOGNISED as T (x) 1 + 1 (x) T' + D,
    // qbh_split.cpp kronc_table_route) instead of U * double occupancy from the two configuration lists (t.cfg_u / t.cfg_d unused then)
    const uint8_t *dcode = nullptr;
    const double  *ddict = nullptr;
};
int launch_mf_hubbard(const MfArgs &a, int grid, hipStream_t s, int *nparts_out);

// matrix-free spin-1/2 Heisenberg operator in a fixed-n_dn sector, basis = colexicographic rank of the down-spin bit
// pattern (the basis of qbh_gen_heisenberg): H = sum_bonds J_b (S+S- + S-S+)/2 + J_b SzSz applied on the fly --
// unrank the row, flip every bond with opposite spins, re-rank the flipped pattern through 6-bit chunk tables in LDS.
struct MfHeis {
    int       n_sites = 0, n_dn = 0, n_bonds = 0, n_chunks = 0;   // n_bonds padded to a multiple of 8 (mask 0)
    uint64_t *binom = nullptr;     // [(n_sites+1) * (n_dn+1)]  C(p, k)
    uint64_t *chunk = nullptr;     // [n_chunks][n_dn+1][64]: colex-rank contribution of the bits of chunk c when j bits lie below it
    uint64_t *mask = nullptr;      // [n_bonds] (1 << a) | (1 << b)
    double   *offd = nullptr;      // [n_bonds] J_b / 2
    double   *diag = nullptr;      // [n_bonds] J_b / 4
    int       n_real = 0;          // bonds before padding
    int       uniform = 0;         // 1: every bond has the same weight (offd0, diag0): no per-bond amplitude reads
    double    offd0 = 0.0, diag0 = 0.0;
};
struct MfHeisArgs {
    MfHeis t;
    int64_t row_begin, nrows;
    const d2 *xg, *xl;
    const double *xr;
    d2 *y;
    double alpha, beta, gamma;
    double *partials;
    double *y_re;
};
int launch_mf_heis(const MfHeisArgs &a, hipStream_t s, int *nparts_out);
int adopt_mf_heis(qbh_csr **out, const MfHeis &t, int64_t nrows, int64_t ncols, int64_t row_offset, int64_t nnz_equiv,
                  const qbh_opts *opts);
// adopt a matrix-free operator (tables already in HBM) behind a qbh_csr handle (qbh_api.cpp)
int adopt_mf_hubbard(qbh_csr **out, const MfHubbard &t, int64_t nrows, int64_t ncols, int64_t row_offset,
                     int64_t nnz_equiv, const qbh_opts *opts);
// adopt a CSR whose value stream was generated directly in dictionary-coded form (the handle owns every array;
// d_code holds nnz + 16 bytes, d_dict 256 entries)
int adopt_coded_csr(qbh_csr **out, int64_t nrows, int64_t ncols, int64_t row_offset, int64_t nnz, int64_t *d_ia,
                    int32_t *d_ja, uint8_t *d_code, d2 *d_dict, int n_dict, const qbh_opts *opts);

}  // namespace qbh

// second part of a split row shard: the entries whose column is NOT owned by this shard
struct CsrPart {
    int64_t  nnz = 0;
    int64_t *d_ia = nullptr;
    int32_t *d_ja = nullptr;
    qbh::d2 *d_val = nullptr;
    uint8_t *d_code = nullptr;
    int      npb = 2048, tpr = 1, unroll = 4, grid = 0;
    int64_t  window = 0, n_blocks = 0;
    int32_t *d_rb = nullptr;
    int64_t *d_bp = nullptr;
    qbh::WaveDesc *d_wd = nullptr;   // wave kernel geometry (uncoded complex128 values)
    int64_t  n_wb = 0;
    int      wtpr = 2, wgrid = 0;
};

struct qbh_csr {
    int          device = 0;
    hipStream_t  stream = nullptr;
    bool         own_stream = false;
    qbh_opts     opts{};

    int64_t nrows = 0, ncols = 0, row_offset = 0, nnz = 0;
    int64_t *d_ia = nullptr;
    int32_t *d_ja = nullptr;
    qbh::d2 *d_val = nullptr;
    uint8_t *d_code = nullptr;
    qbh::d2 *d_dict = nullptr;
    int      n_dict = 0;
    int      code_w = 1;            // bytes per value code (2 when n_dict > 256)
    int      dict_mode = 0;         // k_spmv_rows' DICT
    bool     own_arrays = true;

    // streaming kernel geometry
    int      kernel = QBH_KERNEL_STREAM;
    int      npb = 2048;       // LDS product slots per workgroup
    int      unroll = 4;
    int      tpr = 4;          // threads cooperating on one row in the reduce phase
    int64_t  window = 0;       // nnz window that defines a row block
    int64_t  n_blocks = 0;
    int32_t *d_rb = nullptr;
    int64_t *d_bp = nullptr;
    int      grid = 0;
    int      chunk_mult = 1;   // see BlockWalk (xcd_swizzle 2)
    // Kronecker split of a product-basis operator H = T_major (x) 1 + 1 (x) T_minor + D (two-species Hubbard in the
    // generator's order): "far" = entries that change the major index (same minor index), stored band-major over the
    // minor index with TILED columns and applied first from a tiled copy of x; "near" = the rest in the original row order
    // the split of a dictionary-coded REAL operator applied to packed-double vectors (the library's default form for real
    // operators): two parts for the row kernel -- near (natural order) with the full epilogue, then far (rows and columns in
    // tiled order, x = tiled copy) accumulating onto it at orig(row).  Band = 16 doubles = one 128-byte line.
    struct KronCoded {
        bool     active = false;
        qbh::KronTile t{0, 0, 0};
        CsrPart  near_p, far_p;
        double  *d_xt = nullptr;
        qbh::KroncSliced sl;            // the sliced form (both parts; near gathers from LDS): near_p / far_p are then empty
        // both parts recognised (T (x) 1 + 1 (x) T' + D): T and T' once more as the hop tables of the row-staged table kernel
        // (k_mf_hubbard_row), which then applies the operator instead of the two sliced passes (qbh_opts.kron_uniform bit 2)
        bool     table_route = false;
        qbh::MfHubbard tables;
        const void *xt_of = nullptr;    // the packed vector whose tiled copy d_xt holds (written by the pass that produced it); consumed by one SpMV
    } kronc;
    struct KronSplit {
        bool     active = false;
        bool     inplace = false;       // the handle's d_ja / d_val hold [near | far | cross]: there is no CSR beside the split
        bool     own_far = false;       // padded far groups: ja_f / val_f are allocations of their own
        bool     own_x = false;         // one class: the small cross part lives in its own arrays (its room aligns the far part)
        qbh::KronMap map{};             // product structure of the rows: one class (two-species) or several (cut single-species sector)
        qbh::KronTile t{0, 0, 8};       // one class: tiled order of the LOCAL rows (NU = major indices of this shard)
        int64_t  U0 = 0, NUg = 0;       // one class: first major index of the shard, major indices of the whole operator
        qbh::KronCols cols{};           // one class: order of the gathered x the far / cross columns index (one rank: KronTile{S, NUg, B})
        bool     comm_tiled = false;    // a communicator is attached and every rank exchanges the tiled copy of its block
        int      n_ranks = 1;           // ... of that communicator, and the major-index cuts of its ranks: the gathered blocks are moved
        int64_t  rank_cu[qbh::kKronMaxRanks + 1] = {0};     // (k_kron_place) into the tiled order of the WHOLE vector, which `cols` keeps describing
        int32_t *d_need = nullptr;      // the needed local major indices of every rank, rank after rank (k_kron_need at attach): only those are moved
        int64_t  need_lo[qbh::kKronMaxRanks + 1] = {0};
        double   need_frac = 1.0;       // needed / all major indices of the peers (what a sparse exchange would have to carry)
        // personalised exchange (qbh_opts.sparse_gather + qbh_comm.exchange_v): what every peer reads of THIS rank's major indices
        bool     sparse = false;
        int32_t *d_send_list = nullptr; // local major indices per destination, destination after destination
        int64_t  send_lo[qbh::kKronMaxRanks + 1] = {0};
        qbh::d2 *d_vsend = nullptr, *d_vrecv = nullptr;      // packed pieces out / in (complex128 capacity)
        int64_t  vsend_cap = 0, vrecv_cap = 0;
        int64_t  nnz_n = 0, nnz_f = 0, nnz_x = 0;
        bool     sliced = false;        // far part interleaved inside groups of 8 rows (ia_f = group pointers, n_groups + 1 entries)
        int64_t  n_groups = 0, far_slots = 0;   // far_slots = entries stored in the far arrays (nnz_f + padding)
        int64_t *ia_n = nullptr, *ia_f = nullptr;
        int32_t *ja_n = nullptr, *ja_f = nullptr;
        uint16_t *c16_n = nullptr, *c16_f = nullptr;    // 2-byte columns (qbh_opts.kron_cols16): allocations of their own; the int32 form of that part is gone
        qbh::d2 *val_n = nullptr, *val_f = nullptr;
        qbh::WaveDesc *wd_n = nullptr, *wd_f = nullptr;
        int64_t  nwb_n = 0, nwb_f = 0;
        int      tpr_n = 2, tpr_f = 2, grid_n = 0, grid_f = 0;
        // cross part.  One class: the far entries of the rows of the narrow last band, compact row list (xrow), applied by
        // k_kron_cross_rows into the far buffer.  Several classes: row pointers over ALL rows, applied by k_spmv_wave as a third pass.
        int64_t *ia_x = nullptr;
        int32_t *xrow = nullptr, *ja_x = nullptr;
        qbh::d2 *val_x = nullptr;
        int64_t  n_xrows = 0;
        qbh::WaveDesc *wd_x = nullptr;
        int64_t  nwb_x = 0;
        int      tpr_x = 2, grid_x = 0;
        qbh::KronCls *d_cls = nullptr;  // several classes: device table for the near pass
        double  *d_chunk_red = nullptr; // near pass under the dynamic walk: three sums per chunk of blocks (reproducible reductions)
        int64_t  n_chunk_slots = 0;
        qbh::d2 *d_xt = nullptr, *d_far = nullptr;      // tiled copy of x (xt_cap elements, made on first use), far-part row sums
        int64_t  xt_cap = 0;
        const void *xt_of = nullptr;    // the vector whose tiled copy d_xt holds (written by the pass that produced it); consumed by one SpMV
        bool     fold = false;          // set by a driver for the duration of a solve: its BLAS-1 passes write the tiled copy of the next x
        // the gather in parts (comm_tiled, sliced far part, a communicator with allgather_part_begin): part k = bands
        // [k nfb / n_parts, (k + 1) nfb / n_parts) of every rank's tiled block (the last part takes the narrow edge band along);
        // the far pass of part k = blocks [part_blk[k], part_blk[k + 1]) and starts when that part has arrived
        int      n_parts = 1;
        int64_t  part_blk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int64_t  part_band[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};    // first band of every part (part_band[n_parts] = number of full bands)
        std::vector<int64_t> part_off_len;      // [n_parts][2 * nranks]: offset, length (elements) inside each rank's block
    } kron;
    // wave kernel geometry (uncoded complex128 values; QBH_KERNEL_WAVE)
    bool     use_wave = false;
    bool     broken = false;         // a failed call left the arrays inconsistent: every SpMV is refused
    bool     kron_off = false;       // the split was merged back into a CSR (kron_restore): stay unsplit
    unsigned long long *d_wctr = nullptr;   // [kWctrRegions * 128] work counters of the dynamic walk (main / far / near launch, far launches of a gather in parts)
    int      tuned = -1;             // kernel timed best at creation: -1 not timed, 0 row kernel, 1 wave kernel (kept across rebuilds)
    double   tune_ms[2] = {0.0, 0.0}; // the two times (row kernel, wave kernel) when it was
    qbh::WaveDesc *d_wd = nullptr;
    int64_t  n_wb = 0;
    int      wtpr = 2, wgrid = 0;

    // workspace
    double  *d_partials = nullptr;   // [max(grid, kMaxRedBlocks) * 16]: up to 16 partial sums per workgroup (k_multi_dot<8>)
    double  *d_scal = nullptr;       // [16] reduction results (library-owned unless comm)
    double  *h_scal = nullptr;       // pinned mirror
    qbh::d2 *d_stage_x = nullptr, *d_stage_y = nullptr;   // host-vector seam staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // matrix-free operator (kind 1) instead of CSR arrays (kind 0)
    int      kind = 0;               // 0 stored CSR | 1 matrix-free Hubbard | 2 matrix-free Heisenberg | 3 matrix-free Hubbard momentum sector
    qbh::MfSec *mfsec = nullptr;     // kind 3: block tables + compact remainder (host copy of the descriptor, device arrays)
    qbh::MfSec *d_mfsec = nullptr;   // its device copy (kernel argument)
    qbh::MfHubbard mf;
    qbh::MfHeis    mfh;

    // split shard: the arrays above hold the locally-owned columns, `rem` the remote ones
    bool     has_rem = false;
    int64_t  nnz_total = 0;
    CsrPart  rem;
    hipEvent_t ev2 = nullptr, ev3 = nullptr;
    bool     ev_pending2 = false;

    // real wire format of the x exchange (see k_pack_real)
    bool     values_real = false;   // every stored value has a zero imaginary part
    bool     real_wire = false;     // enabled by a driver for the duration of one solve
    bool     real_mode = false;     // row kernel gathers packed real parts (same conditions, any rank count)
    double  *d_xr = nullptr;        // [ncols] packed Re(x) when there is no communicator
    const void *xr_of = nullptr;    // the vector whose real parts the packed buffer currently holds
    int     *d_flag = nullptr;      // raised by k_pack_real on a non-zero imaginary part

    // the caller's basis when it is not the order the operator is held in (qbh_opts.basis_kind)
    struct BasisMap {
        int       kind = 0;              // QBH_BASIS_*; 0: the operator is held in the caller's order
        int       n_sites = 0, n_up = 0, n_dn = 0;       // the basis described (or found)
        bool      detected = false;      // found by the library itself (qbh_opts.basis_detect), not named by the caller
        qbh::KronMap classes{};          // QBH_BASIS_SPIN_SECTOR: the class table of the internal (class-major) order
        uint32_t *d_map = nullptr;       // [nrows] caller index r -> internal index | sign << 31
        qbh::d2  *d_stage = nullptr;     // [nrows] staging of one vector in the caller's order
    } basis;

    // communicator
    bool     has_comm = false;
    qbh_comm comm{};
    std::vector<int64_t> comm_cuts;          // copy of comm.row_cuts (ragged partition) or empty
    int64_t  comm_full = 0;                  // elements of d_xfull: nranks * nblk (uniform) or ncols (ragged)
    // qbh_opts.major_partition (qbh_gen_hubbard): the generator's major index of every major index of this operator, [major_n]
    int32_t *d_major_inv = nullptr;
    int64_t  major_n = 0, major_S = 0;
    int      major_parts = 0;
    int      wire_bytes_last = 0;            // bytes per element the last gather carried (qbh_csr_info.wire_element_bytes)
    struct qbh_native_comm *native = nullptr; // RCCL communicator owned by the handle (qbh_comm_create_rccl)

    // creation from host arrays: wall ms of the whole qbh_csr_create call / of the upload + expansion, host bytes read
    double    create_ms = 0.0, create_ms_upload = 0.0, detect_ms = 0.0;
    int64_t   create_bytes_in = 0;

    // stats
    qbh_stats stats{};
    bool      ev_pending = false;
    int       debug = 0;
    int ncu = 0;                     // CUs of the handle's device (filled where a launch needs it)
    qbh::DebugSw dbg;                // QBH_DEBUG as it was when the handle was made (new_handle): set the variable before creating the operator
    const double *ovr_xr = nullptr;  // all-real operation requested by a driver for the next spmv_run: x and ...
    double       *ovr_yr = nullptr;  // ... y as packed doubles (the complex pointer arguments are ignored)
    bool      defer_red = false;     // spmv_run leaves its three reduced scalars in d_scal[0..2] (no copy, no sync)

    // Pipelined three-term recurrence (lanczos_core, round 6): step m + 1 is enqueued before step m's scalars have been read
    // back.  ovr_yin / ovr_coef apply to the next spmv_run only (SpmvArgs::yin / coef).
    const qbh::d2 *ovr_yin = nullptr;
    const double  *ovr_coef = nullptr;
    struct LzPipe {
        qbh::d2 *d_buf = nullptr;    // third vector: a step writes v_m where v_{m-3} was, so a discarded speculative step destroys nothing
        int64_t  cap = 0;            // its capacity in elements
        double  *d_state = nullptr;  // [8] alpha, beta, axpy scale of the NEXT step, scale of the newest vector (k_lanczos_tail)
        double  *h_log = nullptr;    // pinned, device-visible: kLzRing slots of 4 doubles {<u,w>, |w'|^2, a, b}, written by the tail kernel itself
        double  *d_log = nullptr;    // the device address of h_log
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};    // one per slot: its tail kernel has finished
    } lz;
    // SpMV timing events of the steps still in flight behind the current set ev0..ev3 (profile = 1 under a pipelined driver)
    struct EvSet {
        hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
        bool p = false, p2 = false, drop = false;
    };
    EvSet ev_old[3];
    int   n_ev_old = 0;
    bool  ev_keep = false;           // set by a pipelined driver: next_event_set() queues the current set instead of waiting for it
    bool  ev_drop = false;           // the current set times a discarded speculative SpMV: not counted
};
constexpr int kLzRing = 8;
