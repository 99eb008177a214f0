// qbh_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the CSR x vector hot path.
//
// Everything here is HBM-bound complex128 / int32 work: no MFMA.  What matters is
// (1) the val/col streams are read once, fully coalesced, non-temporal (so that the
// gathered x keeps the L2 / Infinity Cache), (2) many independent loads in flight per lane,
// (3) workgroup -> row-block mapping that lets each XCD's private L2 see a contiguous range
// of rows (x-gather locality), (4) BLAS-1 work fused into the SpMV epilogue so vectors are
// not re-read.
//
// Replaces mkl_sparse_z_mv (src/sparse.cc:287) and the cblas_z* level-1 calls of the
// Lanczos / CG loops (src/lanczos.cc:195-214, 296-337).
#include <algorithm>
#include <cstring>
#include <vector>

#include "qbh_internal.hpp"
#include "qbh_dict.hpp"

namespace qbh {

// ------------------------------------------------------------------ helpers ----
__device__ __forceinline__ d2 cmul(d2 a, d2 b)
{
    d2 r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}

template <typename T>
__device__ __forceinline__ T ntload(const T *p)
{
    return __builtin_nontemporal_load(p);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum NC doubles per thread over the workgroup; result valid in thread 0.
// `scratch` must hold NC*4 doubles of LDS.  Deterministic (fixed tree).
template <int NC>
__device__ __forceinline__ void block_sum(double (&v)[NC], double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) scratch[c * 4 + wave] = v[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            v[c] = (scratch[c * 4 + 0] + scratch[c * 4 + 1]) + (scratch[c * 4 + 2] + scratch[c * 4 + 3]);
    }
}

// XCD-aware walk over row blocks: workgroup w runs on XCD w % 8 (observed dispatch
// order, used for speed only).  With swizzle each XCD walks one contiguous eighth of the
// row blocks so the x windows it gathers stay in its own 4 MiB L2.
struct BlockWalk {
    int64_t per_xcd, xcd, slot, nslot, nb, chunk;
    int swz;
    __device__ BlockWalk(int64_t n_blocks, int swizzle, int chunk_mult = 1)
    {
        nb = n_blocks;
        per_xcd = (n_blocks + 7) >> 3;
        xcd = blockIdx.x & 7;
        slot = blockIdx.x >> 3;
        nslot = gridDim.x >> 3;
        swz = swizzle;
        // mode 2 walks chunk-wise (a chunk = chunk_mult rounds of the XCD's resident workgroups); round the per-XCD
        // count up to whole chunks so every block is visited
        chunk = nslot * (chunk_mult > 0 ? chunk_mult : 1);
        if (swz == 2) per_xcd = ((per_xcd + chunk - 1) / chunk) * chunk;
    }
    // lb = this XCD's local sequence number (slot, slot + nslot, ...).
    //  0: interleaved           block = lb*8 + xcd        (all XCDs sweep the same region)
    //  1: contiguous eighths    block = xcd*per_xcd + lb  (each XCD owns one eighth of the rows)
    //  2: chunked               the concurrently resident workgroups of an XCD (nslot of them) take
    //                           one contiguous chunk (chunk_mult rounds long); the 8 XCDs take 8 neighbouring chunks
    __device__ int64_t block(int64_t lb) const
    {
        if (swz == 1) return xcd * per_xcd + lb;
        if (swz == 2) return ((lb / chunk) * 8 + xcd) * chunk + (lb % chunk);
        return lb * 8 + xcd;
    }
};

// Dynamic ORDERED walk of the wave kernels (xcd_swizzle 3).  A persistent static walk lets the workgroups of an XCD drift apart
// (nothing synchronises them over thousands of blocks): on C3 their union of x windows then no longer fits the L2 -- 45 % of
// the gathers of the band-major far part missed a 1.65 MB window that the L2 holds perfectly in isolation, and the count did
// not depend on the band width (narrower bands, proportionally more of them in flight).  Here every XCD owns one contiguous
// eighth of the wave blocks and ALL its wavefronts draw chunks of kDynChunk consecutive blocks from one counter, so the eighth
// is consumed in order and the blocks in flight on an XCD are always neighbours.  blockIdx % 8 names the counter (the observed
// dispatch order puts those workgroups on one XCD; if it did not, only the locality would suffer).
#ifndef QBH_DYN_CHUNK
#define QBH_DYN_CHUNK 4
#endif
constexpr int kDynChunk = QBH_DYN_CHUNK;
static_assert(kDynChunk >= 3, "the walk looks two turns ahead and learns the next chunk in the first turn of the current one");
struct DynWalk {
    // cur: first block of the chunk the wavefront is in (turns n0 .. n0 + kDynChunk - 1); nxt: of the chunk after it.  The
    // counter is asked (ask) in the first turn of a chunk BEFORE that turn's gathers and read (take) after they have been
    // waited for: results return in order, so the reply costs no wait of its own and never drains the stream loads issued
    // behind it.  (An atomic add would be rewritten by the compiler into a wave reduction that reads the reply at once;
    // the wrapping increment is left alone, and the counter counts chunks.)
    int64_t xbase, xend, n_wb, cur, nxt, n0;
    int region;
    unsigned int *ctr;
    __device__ unsigned int ask(int lane) const
    {
        unsigned int c = 0;
        if (lane == 0) c = atomicInc(ctr, 0xFFFFFFFFu);
        return c;
    }
    __device__ int64_t take(unsigned int c) const { return xbase + (int64_t)__builtin_amdgcn_readfirstlane(c) * kDynChunk; }
    // region: whose eighth of the blocks (0..7); a wavefront starts on its own XCD's and, when that is exhausted, joins the
    // queues of the others one after the other (k_spmv_wave2's hop loop): the XCDs do not run at the same speed
    __device__ void init(int64_t n_blocks, unsigned long long *counters, int lane, int region = -1)
    {
        n_wb = n_blocks;
        const int xcd = region < 0 ? (int)(blockIdx.x & 7) : region;
        this->region = xcd;
        const int64_t per = (n_blocks + 7) >> 3;
        xbase = xcd * per;
        xend = xbase + per < n_blocks ? xbase + per : n_blocks;
        ctr = reinterpret_cast<unsigned int *>(counters + xcd * 16);
        cur = take(ask(lane));
        nxt = xend;
        n0 = 0;
    }
    __device__ bool asks(int64_t n) const { return n == n0; }        // the turn that draws the next chunk
    // number of the CURRENT chunk among all chunks of the launch (8 regions x chunks per region): where its reduction partials go
    __device__ int64_t chunk_slot() const
    {
        const int64_t per = (n_wb + 7) >> 3, cpx = (per + kDynChunk - 1) / kDynChunk;
        return (int64_t)region * cpx + (cur - xbase) / kDynChunk;
    }
    __device__ int64_t at(int64_t n) const { return (n < n0 + kDynChunk ? cur : nxt) + n % kDynChunk; }
    // wave block of this wavefront's n-th turn (n = the current turn .. two turns ahead); n_wb = the sentinel (past the end)
    __device__ int64_t block(int64_t n) const
    {
        const int64_t w = at(n);
        return w < xend ? w : n_wb;
    }
    __device__ bool live(int64_t n) const { return at(n) < xend; }
    // after moving on to turn n
    __device__ void advance(int64_t n)
    {
        if (n % kDynChunk == 0) {
            cur = nxt;
            n0 = n;
        }
    }
};

// fused epilogue of one row: y <- alpha*(Hx) + beta*y + gamma*x_local, and the running
// partial sums of <x,y> and |y|^2 (K3, K4 and the CG shift folded into K1).
// yo / xi are the old y[row] and x_local[row], loaded by the caller (so that the loads can
// be issued long before the row sum is ready).
// rowmap (far part of a coded Kronecker split): the kernel's rows are in tiled order, the vectors are not
__device__ __forceinline__ int64_t out_row(const SpmvArgs &a, int64_t row)
{
    return a.rowmap ? KronTile{a.kS, a.kNU, a.kB}.orig(row) : row;
}
__device__ __forceinline__ d2 load_y_old(const SpmvArgs &a, int64_t row)
{
    row = out_row(a, row);
    return a.y_re != nullptr ? d2{a.y_re[row], 0.0} : a.yin[row];
}
__device__ __forceinline__ d2 load_x_local(const SpmvArgs &a, int64_t row)
{
    row = out_row(a, row);
    return a.y_re != nullptr ? d2{a.xl_re[row], 0.0} : a.xl[row];
}

__device__ __forceinline__ void row_epilogue2(const SpmvArgs &a, int64_t row, d2 sum, d2 yo, d2 xi,
                                              double (&acc)[3])
{
    d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
    row = out_row(a, row);
    if (a.y_re != nullptr) a.y_re[row] = yn.x;
    else                   a.y[row] = yn;
    acc[0] += xi.x * yn.x + xi.y * yn.y;
    acc[1] += xi.x * yn.y - xi.y * yn.x;
    acc[2] += yn.x * yn.x + yn.y * yn.y;
}

__device__ __forceinline__ void row_epilogue(const SpmvArgs &a, int64_t row, d2 sum, double (&acc)[3])
{
    d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
    if (a.beta != 0.0) yo = load_y_old(a, row);
    if (a.gamma != 0.0 || a.partials != nullptr) xi = load_x_local(a, row);
    row_epilogue2(a, row, sum, yo, xi, acc);
}

// ------------------------------------------------- streaming SpMV (default) ----
// One workgroup per row block of <= NPB nonzeros.  Phase 1: all 256 lanes stream the
// block's col/val ranges (perfectly coalesced, NPB/256 independent 4 B + 16 B + gathered
// 16 B loads in flight per lane) and park val*x products in LDS.  Phase 2: TPR lanes per
// row sum the row's LDS segment, shuffle-reduce, run the fused epilogue.
// The block descriptors (first row rb[], first nonzero bp[]) of the NEXT block are fetched
// while the current one is processed, and the epilogue operands (old y, local x) are
// requested before the stream loads, so a block's critical path is col -> x -> LDS only.
template <int NPB, int TPR, bool DICT>
__global__ __launch_bounds__(kBlock) void k_spmv_stream(SpmvArgs a)
{
    spmv_args_resolve(a);
    __shared__ d2 prod[NPB];
    __shared__ int rowoff[kRowCap + 1];
    __shared__ double red[12];
    __shared__ d2 dict_s[DICT ? 256 : 1];

    constexpr int U = NPB / kBlock;          // independent load chains per lane
    constexpr int G = kBlock / TPR;          // rows reduced per pass
    const int tid = threadIdx.x;
    const int g = tid / TPR, sub = tid % TPR;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0;
    const bool need_x = a.gamma != 0.0 || a.partials != nullptr;

    if (DICT) {
        dict_s[tid] = a.dict[tid];
        __syncthreads();
    }

    BlockWalk walk(a.n_blocks, a.swizzle);
    int64_t lb = walk.slot;
    int64_t b = walk.block(lb);
    bool live = lb < walk.per_xcd && b < a.n_blocks;
    int r0 = 0, r1 = 0;
    int64_t p0 = 0, p1 = 0;
    if (live) {
        r0 = a.rb[b]; r1 = a.rb[b + 1];
        p0 = a.bp[b]; p1 = a.bp[b + 1];
    }
    while (lb < walk.per_xcd) {
        // descriptors of the next block this workgroup will take (uniform -> scalar loads)
        const int64_t lb_n = lb + walk.nslot;
        const int64_t b_n = walk.block(lb_n);
        const bool live_n = lb_n < walk.per_xcd && b_n < a.n_blocks;
        int r0_n = 0, r1_n = 0;
        int64_t p0_n = 0, p1_n = 0;
        if (live_n) {
            r0_n = a.rb[b_n]; r1_n = a.rb[b_n + 1];
            p0_n = a.bp[b_n]; p1_n = a.bp[b_n + 1];
        }
        const int nr = r1 - r0;
        const int64_t nlong = p1 - p0;
        if (live && nr > 0) {
            if (nlong <= NPB && nr <= kRowCap) {
                const int n = (int)nlong;
                // row offsets and epilogue operands: requested first, consumed last
                const int ro = (int)(a.ia[r0 + (tid <= nr ? tid : 0)] - p0);
                d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
                const bool mine = sub == 0 && g < nr;
                if (mine && need_y) yo = load_y_old(a, r0 + g);
                if (mine && need_x) xi = load_x_local(a, r0 + g);
                if (n > 0) {
                    // All U loads of each stream are issued back to back with a clamped index
                    // (no per-element branch): U col + U val + U gathered-x loads in flight
                    // per lane -- the memory-level parallelism a branchy loop does not have.
                    const int32_t *jp = a.ja + p0;
                    const int nm1 = n - 1;
                    int c[U];
                    d2 v[U], xv[U];
                    uint8_t cb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = tid + u * kBlock;
                        c[u] = ntload(jp + (i < n ? i : nm1)) & a.colmask;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = tid + u * kBlock;
                        const int ii = i < n ? i : nm1;
                        if (DICT) cb[u] = ntload(a.code + p0 + ii);
                        else      v[u] = ntload(a.val + p0 + ii);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) xv[u] = a.xg[c[u]];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = tid + u * kBlock;
                        if (DICT) v[u] = dict_s[cb[u]];
                        if (i < n) prod[i] = cmul(v[u], xv[u]);
                    }
                }
                if (tid <= nr) rowoff[tid] = ro;
                for (int i = tid + kBlock; i <= nr; i += kBlock) rowoff[i] = (int)(a.ia[r0 + i] - p0);
                __syncthreads();
                for (int r = g; r < nr; r += G) {
                    const int e = rowoff[r + 1];
                    d2 sum = {0.0, 0.0};
                    for (int q = rowoff[r] + sub; q < e; q += TPR) sum += prod[q];
#pragma unroll
                    for (int off = TPR / 2; off > 0; off >>= 1) {
                        sum.x += __shfl_xor(sum.x, off, 64);
                        sum.y += __shfl_xor(sum.y, off, 64);
                    }
                    if (sub == 0) {
                        if (r == g) row_epilogue2(a, (int64_t)r0 + r, sum, yo, xi, acc);
                        else        row_epilogue(a, (int64_t)r0 + r, sum, acc);
                    }
                }
                __syncthreads();
            } else {
                // oversized block (a row longer than the LDS tile, or > kRowCap very short
                // rows): row by row, whole workgroup per row.  Correctness path, not tuned.
                for (int r = 0; r < nr; ++r) {
                    const int64_t s = a.ia[r0 + r], e = a.ia[r0 + r + 1];
                    double part[2] = {0.0, 0.0};
                    for (int64_t q = s + tid; q < e; q += kBlock) {
                        d2 v;
                        if (DICT) v = dict_s[a.code[q]];
                        else      v = a.val[q];
                        const d2 t = cmul(v, a.xg[a.ja[q] & a.colmask]);
                        part[0] += t.x;
                        part[1] += t.y;
                    }
                    block_sum<2>(part, red);
                    if (tid == 0) {
                        d2 sum = {part[0], part[1]};
                        row_epilogue(a, (int64_t)r0 + r, sum, acc);
                    }
                    __syncthreads();
                }
            }
        }
        lb = lb_n; b = b_n; live = live_n;
        r0 = r0_n; r1 = r1_n; p0 = p0_n; p1 = p1_n;
    }
    if (a.partials != nullptr) {
        block_sum<3>(acc, red);
        if (tid == 0) {
            a.partials[(size_t)blockIdx.x * 3 + 0] = acc[0];
            a.partials[(size_t)blockIdx.x * 3 + 1] = acc[1];
            a.partials[(size_t)blockIdx.x * 3 + 2] = acc[2];
        }
    }
}

// ------------------------------------------------ row-aligned SpMV -------------
// Same row blocks as k_spmv_stream, different use of LDS.  Phase 1 stages only the block's
// (col, value-or-code) ranges into LDS with coalesced non-temporal loads.  Phase 2 maps LANES
// TO ROWS: sub-slice s of P handles steps k = s, s+P, ... of each of R = 256/P consecutive
// rows, so at every step the lanes of a wavefront gather the k-th entries of consecutive rows.
// For Kronecker-structured Hamiltonians (H = T_up (x) 1 + 1 (x) T_dn + D) the leading and the
// trailing entries of consecutive rows point at CONSECUTIVE x elements, so stepping alternately
// from the front and from the back of the row turns those gathers into full-line coalesced
// loads.  Row sums stay in registers (no 16-byte products through LDS, no shuffle tree);
// with the value dictionary the LDS footprint is 5 B/nnz and 8 workgroups fit a CU.
// DICT: 0 complex128 values | 1 one-byte codes, dictionary (<= 256) in LDS | 2 two-byte codes, dictionary
// (<= kDictLds) in LDS | 3 two-byte codes, dictionary (<= 65536) read through the caches
template <int NPB, int P, int UN, int DICT, bool REALX>
__global__ __launch_bounds__(kBlock) void k_spmv_rows(SpmvArgs a)
{
    spmv_args_resolve(a);
    constexpr int R = kBlock / P;            // rows per pass
    constexpr int U = NPB / kBlock;          // staged cols per lane
    constexpr int CPW = DICT >= 2 ? 4 : 8;   // codes per 8-byte word
    constexpr int UC = (NPB / CPW + kBlock - 1) / kBlock;   // 8-byte code words per lane
    __shared__ int scol[NPB];
    __shared__ d2 sval[DICT ? 1 : NPB];
    __shared__ unsigned long long scode8[DICT ? NPB / CPW : 1];
    __shared__ d2 dict_s[DICT == 1 ? 256 : DICT == 2 ? kDictLds : 1];
    __shared__ int rowoff[kRowCap + 1];
    __shared__ d2 part[P > 1 ? kBlock : 1];
    __shared__ double red[12];

    const int tid = threadIdx.x;
    const int sub = tid / R, rloc = tid % R;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0;
    const bool need_x = a.gamma != 0.0 || a.partials != nullptr;
    const uint8_t *scode = reinterpret_cast<const uint8_t *>(scode8);
    const uint16_t *scode16 = reinterpret_cast<const uint16_t *>(scode8);
    const uint16_t *gcode16 = reinterpret_cast<const uint16_t *>(a.code);
    auto coded_value = [&](int i) -> d2 {          // value of staged element i
        if (DICT == 1) return dict_s[scode[i]];
        if (DICT == 2) return dict_s[scode16[i]];
        return a.dict[scode16[i]];
    };

    if (DICT == 1) {
        dict_s[tid] = a.dict[tid];
        __syncthreads();
    }
    if (DICT == 2) {
        for (int i = tid; i < kDictLds; i += kBlock) dict_s[i] = a.dict[i];
        __syncthreads();
    }

    BlockWalk walk(a.n_blocks, a.swizzle, a.chunk_mult);
    int64_t lb = walk.slot;
    int64_t b = walk.block(lb);
    bool live = lb < walk.per_xcd && b < a.n_blocks;
    int r0 = 0, r1 = 0;
    int64_t p0 = 0, p1 = 0;
    if (live) {
        r0 = a.rb[b]; r1 = a.rb[b + 1];
        p0 = a.bp[b]; p1 = a.bp[b + 1];
    }
    while (lb < walk.per_xcd) {
        const int64_t lb_n = lb + walk.nslot;
        const int64_t b_n = walk.block(lb_n);
        const bool live_n = lb_n < walk.per_xcd && b_n < a.n_blocks;
        int r0_n = 0, r1_n = 0;
        int64_t p0_n = 0, p1_n = 0;
        if (live_n) {
            r0_n = a.rb[b_n]; r1_n = a.rb[b_n + 1];
            p0_n = a.bp[b_n]; p1_n = a.bp[b_n + 1];
        }
        const int nr = r1 - r0;
        const int64_t nlong = p1 - p0;
        if (live && nr > 0) {
            if (nlong <= NPB) {
                const int n = (int)nlong;
                // ---- phase 1: stage the block's index / value streams ----
                // (a block of very short rows can hold more than kRowCap of them: their offsets are staged and their sums
                // formed kRowCap rows at a time, nrg = rows of the current group)
                const int nrg0 = nr < kRowCap ? nr : kRowCap;
                const int ro = (int)(a.ia[r0 + (tid <= nrg0 ? tid : 0)] - p0);
                d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
                const bool mine = sub == 0 && rloc < nr;
                if (mine && need_y) yo = load_y_old(a, r0 + rloc);
                if (mine && need_x) xi = load_x_local(a, r0 + rloc);
                if (n > 0) {
                    const int32_t *jp = a.ja + p0;
                    const int nm1 = n - 1;
                    int c[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = tid + u * kBlock;
                        c[u] = ntload(jp + (i < n ? i : nm1)) & a.colmask;
                    }
                    if (DICT) {
                        unsigned long long w[UC];
                        const int nw = (n + CPW - 1) / CPW;          // 8-byte words covering the codes
#pragma unroll
                        for (int u = 0; u < UC; ++u) {
                            const int i = tid + u * kBlock;
                            // unaligned 8-byte global load; the last word may read up to 7 bytes past the
                            // block's range but never past the code array (padded by 16 bytes at build)
                            w[u] = ntload(reinterpret_cast<const unsigned long long *>(a.code + p0 * (8 / CPW)) + (i < nw ? i : 0));
                        }
#pragma unroll
                        for (int u = 0; u < UC; ++u) {
                            const int i = tid + u * kBlock;
                            if (i < NPB / CPW) scode8[i] = w[u];
                        }
                    } else {
                        d2 v[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int i = tid + u * kBlock;
                            v[u] = ntload(a.val + p0 + (i < n ? i : nm1));
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) sval[tid + u * kBlock] = v[u];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) scol[tid + u * kBlock] = c[u];
                }
                if (tid <= nrg0) rowoff[tid] = ro;
                for (int i = tid + kBlock; i <= nrg0; i += kBlock) rowoff[i] = (int)(a.ia[r0 + i] - p0);
                __syncthreads();
                for (int rg = 0; rg < nr; rg += kRowCap) {
                const int nrg = nr - rg < kRowCap ? nr - rg : kRowCap;
                if (rg > 0) {                                          // next group of rows of the same staged block
                    __syncthreads();
                    for (int i = tid; i <= nrg; i += kBlock) rowoff[i] = (int)(a.ia[r0 + rg + i] - p0);
                    __syncthreads();
                }
                // ---- phase 2: lanes <-> rows ----
                for (int rbase = 0; rbase < nrg; rbase += R) {
                    const int row = rbase + rloc;
                    const bool rowok = row < nrg;
                    const int base = rowok ? rowoff[row] : 0;
                    const int len = rowok ? rowoff[row + 1] - base : 0;
                    int wmax = len;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const int o = __shfl_xor(wmax, off, 64);
                        wmax = o > wmax ? o : wmax;
                    }
                    d2 sum = {0.0, 0.0};
                    for (int k0 = sub; k0 < wmax; k0 += UN * P) {
                        int cc[UN];
                        int ix[UN];
                        bool ok[UN];
#pragma unroll
                        for (int j = 0; j < UN; ++j) {
                            const int k = k0 + j * P;
                            ok[j] = k < len;
                            const int f = (k & 1) ? len - 1 - (k >> 1) : (k >> 1);   // front / back alternation
                            ix[j] = base + (ok[j] ? f : 0);
                            cc[j] = scol[ix[j]];
                        }
                        if (REALX) {
                            // real operator applied to a real vector: gather 8-byte real parts from the packed
                            // copy of x; (a+0i)(b+0i) = ab+0i exactly, so the result is bit-identical
                            double xr[UN], vr[UN];
#pragma unroll
                            for (int j = 0; j < UN; ++j) xr[j] = a.xr[cc[j]];
#pragma unroll
                            for (int j = 0; j < UN; ++j) {
                                if (DICT) vr[j] = coded_value(ix[j]).x;
                                else      vr[j] = sval[ix[j]].x;
                            }
#pragma unroll
                            for (int j = 0; j < UN; ++j)
                                if (ok[j]) sum.x += vr[j] * xr[j];
                        } else {
                            d2 xv[UN], vv[UN];
#pragma unroll
                            for (int j = 0; j < UN; ++j) xv[j] = a.xg[cc[j]];
#pragma unroll
                            for (int j = 0; j < UN; ++j) {
                                if (DICT) vv[j] = coded_value(ix[j]);
                                else      vv[j] = sval[ix[j]];
                            }
#pragma unroll
                            for (int j = 0; j < UN; ++j)
                                if (ok[j]) sum += cmul(vv[j], xv[j]);
                        }
                    }
                    if (P > 1) {
                        part[tid] = sum;
                        __syncthreads();
                        if (sub == 0) {
#pragma unroll
                            for (int s2 = 1; s2 < P; ++s2) sum += part[s2 * R + rloc];
                        }
                    }
                    if (sub == 0 && rowok) {
                        if (rbase == 0 && rg == 0) row_epilogue2(a, (int64_t)r0 + row, sum, yo, xi, acc);
                        else                       row_epilogue(a, (int64_t)r0 + rg + row, sum, acc);
                    }
                    if (P > 1 && rbase + R < nrg) __syncthreads();     // part[] is reused by the next pass
                }
                }
                if (P == 1) __syncthreads();                           // scol is rewritten by the next block
            } else {
                for (int r = 0; r < nr; ++r) {
                    const int64_t s = a.ia[r0 + r], e = a.ia[r0 + r + 1];
                    double pr[2] = {0.0, 0.0};
                    for (int64_t q = s + tid; q < e; q += kBlock) {
                        d2 v;
                        if (DICT == 1)      v = dict_s[a.code[q]];
                        else if (DICT == 2) v = dict_s[gcode16[q]];
                        else if (DICT == 3) v = a.dict[gcode16[q]];
                        else                v = a.val[q];
                        const int cq = a.ja[q] & a.colmask;
                        const d2 xq = REALX ? d2{a.xr[cq], 0.0} : a.xg[cq];
                        const d2 t = cmul(v, xq);
                        pr[0] += t.x;
                        pr[1] += t.y;
                    }
                    block_sum<2>(pr, red);
                    if (tid == 0) {
                        d2 sum = {pr[0], pr[1]};
                        row_epilogue(a, (int64_t)r0 + r, sum, acc);
                    }
                    __syncthreads();
                }
            }
        }
        lb = lb_n; b = b_n; live = live_n;
        r0 = r0_n; r1 = r1_n; p0 = p0_n; p1 = p1_n;
    }
    if (a.partials != nullptr) {
        block_sum<3>(acc, red);
        if (tid == 0) {
            a.partials[(size_t)blockIdx.x * 3 + 0] = acc[0];
            a.partials[(size_t)blockIdx.x * 3 + 1] = acc[1];
            a.partials[(size_t)blockIdx.x * 3 + 2] = acc[2];
        }
    }
}


// ------------------------------------------------ wave-granular SpMV (uncoded) ---
// One WAVEFRONT per block of whole rows holding <= 512 nonzeros; no workgroup barrier anywhere.  Lane l takes the
// entries l, l+64, ... of the block: 8 column + 8 value loads (coalesced, non-temporal), then the 8 gathers, all in
// flight together; the products go to a wave-private 8 KB LDS tile and TPR lanes per row sum them, shuffle-reduce
// and run the fused epilogue.  Against the workgroup-granular kernels above this keeps 16 independent load / gather
// / reduce pipelines per CU instead of 3 lock-stepped ones, which is what the cache-friendly operators were limited
// by (DESIGN-history 5.0 item 4): chain L = 26 goes from 1.06 to 0.72 ms.  Descriptors (first row / first nonzero of the
// block and of the next one) are fetched one block ahead.
// Complex128 values, complex vectors only: the coded / real-gather formats stay on k_spmv_rows.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int TPR, bool DYN>
__global__ __launch_bounds__(kBlock) void k_spmv_wave(SpmvArgs a)
{
    spmv_args_resolve(a);
    constexpr int U = 8, NW = 64 * U, RP = 64 / TPR;
    __shared__ d2 prod_s[4 * NW];
    __shared__ double red[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    d2 *prod = prod_s + wv * NW;
    const int sub = lane % TPR, rloc = lane / TPR;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0;
    const bool need_x = a.gamma != 0.0 || a.partials != nullptr;

    // unit = 4 consecutive wave blocks (one per wavefront of the workgroup), walked XCD-aware like the row blocks
    BlockWalk walk((a.n_wb + 3) >> 2, a.swizzle, a.chunk_mult);
    // The descriptor pair (this block, next block) is read one trip ahead as ONE 32-byte vector load -- lane j holds
    // dword j -- and broadcast with readlane when the trip starts: the compiler tracks it like any other load (an
    // explicit s_load would be faster still, but nothing stops the register allocator from copying its destination
    // SGPRs while the load is in flight).  A block past the end reads the sentinel pair (n_wb, n_wb + 1): zero rows.
    constexpr bool dyn = DYN;            // compile-time: the atomic of the dynamic walk must not leak into the static kernel's waits
    DynWalk dw;
    if constexpr (dyn) dw.init(a.n_wb, a.wctr, lane);
    auto load_desc = [&](int64_t lb) -> int {
        int64_t w = a.n_wb;
        if (dyn) {
            w = dw.block(lb);
        } else if (lb < walk.per_xcd) {
            w = walk.block(lb) * 4 + wv;
            if (w > a.n_wb) w = a.n_wb;
        }
        return reinterpret_cast<const int *>(a.wd + w)[lane & 7];
    };
    const int64_t step = dyn ? 1 : walk.nslot;
    int64_t lb = dyn ? 0 : walk.slot;
    int dq = load_desc(lb);
    while (dyn ? dw.live(lb) : (lb < walk.per_xcd)) {
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane(dq, 0), q1 = (uint32_t)__builtin_amdgcn_readlane(dq, 1);
        const uint32_t q4 = (uint32_t)__builtin_amdgcn_readlane(dq, 4), q5 = (uint32_t)__builtin_amdgcn_readlane(dq, 5);
        const int r0 = __builtin_amdgcn_readlane(dq, 2), nr = __builtin_amdgcn_readlane(dq, 6) - r0;
        const int64_t p0 = (int64_t)(((uint64_t)q1 << 32) | q0);
        const int64_t p1 = (int64_t)(((uint64_t)q5 << 32) | q4);
        if constexpr (dyn) {
            if (dw.asks(lb)) dw.nxt = dw.take(dw.ask(lane));
        }
        lb += step;
        if constexpr (dyn) dw.advance(lb);
        dq = load_desc(lb);                                                 // next block's descriptors, used one trip later
        if (nr <= 0) continue;
        // the stream is read from the 128-byte boundary below the block's first value (8 entries): every 1 KB value load then
        // covers exactly 8 lines instead of 9, the tile holds sh + n <= 512 entries (the builder leaves the room)
        const int sh = (int)(p0 & 7);
        const int64_t base = p0 - sh;
        const int64_t nlong = p1 - base;
        if (nlong <= NW) {
            const int n = (int)nlong;
            // first pass row offsets + epilogue operands: requested before the streams, consumed last
            int s0 = 0, e0 = 0;
            d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
            if (rloc < nr) {
                s0 = (int)(a.ia[r0 + rloc] - base);
                e0 = (int)(a.ia[r0 + rloc + 1] - base);
                if (sub == 0) {
                    if (need_y) yo = a.yin[r0 + rloc];
                    if (need_x) xi = a.xl[r0 + rloc];
                }
            }
            if (n > 0) {
                const int nm1 = n - 1;
                int c[U];
                d2 v[U], xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = lane + u * 64;
                    c[u] = ntload(a.ja + base + (i < n ? i : nm1)) & a.colmask;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = lane + u * 64;
                    v[u] = ntload(a.val + base + (i < n ? i : nm1));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = a.xg[c[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) prod[lane + u * 64] = cmul(v[u], xv[u]);
            }
            wave_lds_fence();
            for (int rbase = 0; rbase < nr; rbase += RP) {
                const int row = rbase + rloc;
                int s = s0, e = e0;
                if (rbase > 0) {
                    s = e = 0;
                    if (row < nr) {
                        s = (int)(a.ia[r0 + row] - base);
                        e = (int)(a.ia[r0 + row + 1] - base);
                    }
                }
                d2 sum = {0.0, 0.0};
                for (int k = s + sub; k < e; k += TPR) sum += prod[k];
#pragma unroll
                for (int off = TPR / 2; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (sub == 0 && row < nr) {
                    if (rbase == 0) row_epilogue2(a, (int64_t)r0 + row, sum, yo, xi, acc);
                    else            row_epilogue(a, (int64_t)r0 + row, sum, acc);
                }
            }
            wave_lds_fence();                      // the tile is rewritten by the next block
        } else {
            // a row longer than the wave tile: the wavefront walks the block's rows one at a time (correctness path)
            for (int r = 0; r < nr; ++r) {
                const int64_t s = a.ia[r0 + r], e = a.ia[r0 + r + 1];
                d2 sum = {0.0, 0.0};
                for (int64_t k = s + lane; k < e; k += 64) sum += cmul(a.val[k], a.xg[a.ja[k] & a.colmask]);
                sum.x = wave_sum(sum.x);
                sum.y = wave_sum(sum.y);
                if (lane == 0) row_epilogue(a, (int64_t)r0 + r, sum, acc);
            }
        }
    }
    if (a.partials != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c * 4 + wv] = acc[c];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4 + 0] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
        }
    }
}

int launch_spmv_wave(const SpmvArgs &a_in, int tpr, int grid, hipStream_t s)
{
    SpmvArgs a = a_in;
    if (a.yin == nullptr) a.yin = a.y;           // the beta term reads y itself unless a driver names another vector
    if (a.swizzle == 3) {
        switch (tpr) {
        case 2:  hipLaunchKernelGGL((k_spmv_wave<2, true>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 4:  hipLaunchKernelGGL((k_spmv_wave<4, true>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 8:  hipLaunchKernelGGL((k_spmv_wave<8, true>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((k_spmv_wave<16, true>), dim3(grid), dim3(kBlock), 0, s, a); break;
        }
    } else {
        switch (tpr) {
        case 2:  hipLaunchKernelGGL((k_spmv_wave<2, false>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 4:  hipLaunchKernelGGL((k_spmv_wave<4, false>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 8:  hipLaunchKernelGGL((k_spmv_wave<8, false>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((k_spmv_wave<16, false>), dim3(grid), dim3(kBlock), 0, s, a); break;
        }
    }
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int wave_kernel_occupancy(int tpr)
{
    int occ = 0;
    hipError_t e;
    switch (tpr) {
    case 2:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave<2, false>, kBlock, 0); break;
    case 4:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave<4, false>, kBlock, 0); break;
    case 8:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave<8, false>, kBlock, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave<16, false>, kBlock, 0); break;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return occ;
}

// wave block w = the rows whose first nonzero lies in [w*window, (w+1)*window); entry n_wb and n_wb + 1 = sentinels
__global__ void k_build_wavedesc(const int64_t *ia, int64_t nrows, int64_t window, WaveDesc *wd, int64_t n_wb)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > n_wb + 1) return;
    WaveDesc d;
    d.pad = 0;
    if (w >= n_wb) {
        d.p0 = ia[nrows];
        d.r0 = (int32_t)nrows;
    } else {
        const int64_t target = w * window;
        int64_t lo = 0, hi = nrows;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ia[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        d.p0 = ia[lo];
        d.r0 = (int32_t)lo;
    }
    wd[w] = d;
}

int launch_build_wavedesc(const int64_t *d_ia, int64_t nrows, int64_t window, WaveDesc *d_wd, int64_t n_wb, hipStream_t s)
{
    const int64_t n = n_wb + 2;
    hipLaunchKernelGGL(k_build_wavedesc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ia, nrows, window, d_wd, n_wb);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}


// ------------------------------------ pipelined wave kernel (Kronecker split) ----
// The same decomposition as k_spmv_wave, software-pipelined per wavefront for the two passes of a split operator, whose
// gathers hit the L2: the gathers of block i are issued, THEN the 16 stream loads of block i+1, and only the gathers are
// waited for (vector-memory results return in order, so the order of issue is what keeps a block's stream in flight
// while the previous block is reduced).  Every load of the steady state is unconditional (clamped addresses): the
// compiler's s_waitcnt counts are then exact.  Descriptors are fetched two blocks ahead.
// OPS 0: plain store of the row sums (far pass; y is the far buffer, rows are far rows)
// OPS 3: the same for the far part stored SLICED: inside every group of 8 consecutive far rows the entries are interleaved
//        (entry k of rows 8g..8g+7 contiguous, the group padded to its longest row -- no padding where the 8 rows are the
//        8 minor indices of one major index), so the coalesced stream ALREADY has consecutive lanes on consecutive rows with
//        the same entry number: every gather instruction reads full 128-byte lines of the tiled x.  ia holds the group
//        pointers, the descriptor's row fields count groups.
// OPS 2: fused epilogue, the far result of the row added first (read at the row's tiled index)
// wavefronts per SIMD of the near pass: 2 = 204 VGPRs, no spill; 3 = 168 VGPRs with 17 spilled (measured: see DESIGN-history 4.1c)
#ifndef QBH_NEAR_WAVES
#define QBH_NEAR_WAVES 2
#endif
#ifndef QBH_FAR_WAVES
#define QBH_FAR_WAVES 3
#endif
#ifdef QBH_NT_ROW_STORE
#define QBH_ROW_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define QBH_ROW_STORE(p, v) (*(p) = (v))
#endif
// C16: the part's columns are 2 bytes each (a.ja16), relative to a base named by the block's descriptor (SpmvArgs::ja16): 8 lines
// of column stream per block instead of 16 -- the passes are bound by line requests, not bytes (DESIGN-history 5.0b)
template <int TPR, int OPS, bool DYN, bool C16 = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((OPS == 0 || OPS == 3) ? QBH_FAR_WAVES : QBH_NEAR_WAVES, (OPS == 0 || OPS == 3) ? QBH_FAR_WAVES : QBH_NEAR_WAVES))) void k_spmv_wave2(SpmvArgs a)
{
    spmv_args_resolve(a);
    static_assert(!C16 || OPS == 1 || OPS == 2 || OPS == 3, "2-byte columns: the one-class near passes and the sliced far pass");
    constexpr int NW = 512, RP = 64 / TPR;
    constexpr bool EPI = OPS == 1 || OPS == 2 || OPS == 4, FAR = OPS == 2 || OPS == 4;      // OPS 1: the fused epilogue WITHOUT a far addend
    // The fused reductions under the ordered dynamic walk: which wavefront takes which chunk depends on the run, so per-wavefront
    // partial sums would make <x, y> and |y|^2 differ in the last bits from run to run.  Every CHUNK (kDynChunk consecutive blocks,
    // always taken whole by one wavefront, rows in order) has a slot of its own instead: its three sums are stored when the
    // wavefront moves on and k_reduce_chunks adds the slots in a fixed order -- bit-reproducible a_j / b_j at the dynamic walk's speed.
    constexpr bool CHUNKRED = DYN && (OPS == 1 || OPS == 2 || OPS == 4);
    constexpr bool MULTI = OPS == 4;             // OPS 4 = OPS 2 for an operator with several classes (KronMap): the far result of a row sits at its
                                                 // compact far row id, looked up through the class table a.kcls; the block's descriptor names its class
    __shared__ d2 prod_s[4 * NW];
    __shared__ double red[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    d2 *prod = prod_s + wv * NW;
    const int sub = lane % TPR, rloc = lane / TPR;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0;

    BlockWalk walk((a.n_wb + 3) >> 2, a.swizzle, a.chunk_mult);
    constexpr bool dyn = DYN;
    DynWalk dw;
    auto load_desc = [&](int64_t lb) -> int {
        int64_t w = a.n_wb;
        if (dyn) {
            w = dw.block(lb);
        } else if (lb < walk.per_xcd) {
            w = walk.block(lb) * 4 + wv;
            if (w > a.n_wb) w = a.n_wb;
        }
        return reinterpret_cast<const int *>(a.wd + w)[lane & 7];
    };
    const int64_t step = dyn ? 1 : walk.nslot;
    struct Blk {
        int64_t p0;              // the 128-byte boundary below the block's first value (the stream is read from there: 8 lines per
                                 // 1 KB value load instead of 9); row offsets are taken relative to it
        int r0, nr, n;           // n = entries from p0 to the block's end; -1: a row longer than the tile (row-at-a-time path)
        bool cont0, cont1;       // OPS 3: the first group began in the block before / the last group goes on in the block after
        int cls;                 // OPS 4: class of the block's first row
        int64_t xb;              // C16: element of the gather source that column value 0 of this block names
        int lead;                // entries between p0 and the block's first entry (they belong to the block before: loaded, never used)
    };
    // far result of one row (OPS 2 / 4)
    // (rows and minor sizes are below 2^31 -- int32 columns -- so the index arithmetic of a row's far slot is 32-bit: a 64-bit
    // division by a run-time divisor is ~100 instructions on the critical path of every block's epilogue operands)
    const uint32_t kS32 = (uint32_t)a.kS, kB32 = (uint32_t)a.kB;
    const int kLB = 31 - __builtin_clz(kB32 | 1u);                   // band widths are powers of two
    auto far_at = [&](int64_t row, int c0) -> d2 {
        if constexpr (MULTI) {
            int c = c0;
            while (row >= a.kcls[c + 1].rbase) ++c;                 // a block rarely straddles two classes
            const KronCls k = a.kcls[c];
            const uint32_t local = (uint32_t)(row - k.rbase), S32 = (uint32_t)k.S, u = local / S32, d = local - u * S32;
            if (d >= ((S32 >> 3) << 3)) return d2{0.0, 0.0};         // a row of the class's narrow last band: no far part
            return a.far[k.fbase + (int64_t)(d >> 3) * 8 * k.NU + (int64_t)u * 8 + (d & 7)];
        } else {
            const uint32_t r = (uint32_t)row, u = r / kS32, d = r - u * kS32;
            const uint32_t b = d >> kLB, j = d & (kB32 - 1u), rem = kS32 - (b << kLB), wB = rem < kB32 ? rem : kB32;
            return a.far[(int64_t)b * (a.kNU << kLB) + (int64_t)u * wB + j];
        }
    };
    // OPS 3: a block is 512 consecutive SLOTS of the sliced stream whatever the groups are (descriptor: first slot, first
    // group that overlaps, pad = 1 when that group began in the previous block); a group cut by a block boundary gets its
    // row sums from both blocks by atomic add into rows zeroed before the pass (two addends: the result does not depend
    // on their order), every other row a plain store
    auto decode = [&](int dq) -> Blk {
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane(dq, 0), q1 = (uint32_t)__builtin_amdgcn_readlane(dq, 1);
        const uint32_t q4 = (uint32_t)__builtin_amdgcn_readlane(dq, 4), q5 = (uint32_t)__builtin_amdgcn_readlane(dq, 5);
        Blk b;
        const int64_t pfirst = (int64_t)(((uint64_t)q1 << 32) | q0);
        // OPS 3: blocks are exact runs of slots, cut so that they start on 128-byte boundaries of the arrays (k_build_slotdesc's shift)
        b.p0 = OPS == 3 ? pfirst : pfirst - (pfirst & 7);
        b.lead = OPS == 3 ? 0 : (int)(pfirst & 7);
        const int64_t p1 = (int64_t)(((uint64_t)q5 << 32) | q4);
        b.r0 = __builtin_amdgcn_readlane(dq, 2);
        b.nr = __builtin_amdgcn_readlane(dq, 6) - b.r0;
        b.cls = MULTI ? __builtin_amdgcn_readlane(dq, 3) : 0;
        b.xb = 0;
        if constexpr (C16) {
            const int64_t pad = (uint32_t)__builtin_amdgcn_readlane(dq, 3);
            b.xb = OPS == 3 ? (pad >> 1) * 8 * a.kNU : pad * a.kS;
        }
        b.cont0 = OPS == 3 && (__builtin_amdgcn_readlane(dq, 3) & 1);
        b.cont1 = OPS == 3 && (__builtin_amdgcn_readlane(dq, 7) & 1);
        if (b.cont1) b.nr += 1;
        b.n = (p1 - b.p0) <= NW ? (int)(p1 - b.p0) : -1;
        return b;
    };
    struct Ops {
        int s, e;                // OPS 3: s = the group pointer of group `lane` of the block, relative to the block's first slot
        d2 yo, xi, fr;
    };
    // OPS 3: a block overlaps at most 64 groups (a group holds 8 slots or more), so ONE load per lane, issued with the block's
    // stream, brings every group pointer the block needs; the reduction passes read them by cross-lane moves.  (A load
    // issued later would have to be waited for with the whole next stream in front of it: results return in order.)
    auto group_range = [&](const Blk &b, int gp, int row, int &s_, int &e_, bool &clip) {     // row = 8 * group + j inside the block
        const int gi = row >> 3;
        const int gs = __shfl(gp, gi & 63, 64);
        int ge = __shfl(gp, (gi + 1) & 63, 64);
        if (gi + 1 >= 64) ge = b.n;
        clip = gs < 0 || ge > b.n;
        s_ = (gs < 0 ? 0 : gs) + (row & 7);
        e_ = ge > b.n ? b.n : ge;
    };
    // stream + first-pass operands of a block; an empty / oversized block reads entry 0 of its range (clamped)
    auto issue = [&](const Blk &b, int (&c)[8], d2 (&v)[8], Ops &o) {
        const int nn = b.n > 0 ? b.n : 1;
        const int nm1 = nn - 1;
        // p0 of the sentinel is nnz: clamp the base so that even an empty block loads inside the arrays
        const int64_t base = b.n > 0 ? b.p0 : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            if constexpr (C16) c[u] = ntload(a.ja16 + base + (i < nn ? i : nm1));
            else               c[u] = ntload(a.ja + base + (i < nn ? i : nm1)) & a.colmask;
        }
        // The up to 7 entries in front of the block's first one are the tail of the block BEFORE: their 2-byte columns are relative
        // to THAT block's base, and decoded with this block's they can point up to two major indices ahead -- past the end of x
        // for the last blocks of an operator or shard (found by the 4-rank C3 rehearsal, round 5: a memory access fault on the ranks
        // whose vectors ended at an allocation boundary).  Their products are never used: gather element 0 of the block's base.
        if constexpr (C16 && OPS != 3) {
            if (b.n > 0 && lane < b.lead) c[0] = 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            v[u] = ntload(a.val + base + (i < nn ? i : nm1));
        }
        if (OPS == 3) {
            const int64_t ng = (a.nrows + 7) >> 3;
            int64_t g = (int64_t)b.r0 + lane;
            g = g < ng ? g : ng;
            int64_t rel = a.ia[g] - base;
            rel = rel < -4096 ? -4096 : rel > 4096 ? 4096 : rel;
            o.s = (int)rel;
            o.e = 0;
        } else {
            const bool mine = rloc < b.nr;
            const int64_t row = mine ? (int64_t)b.r0 + rloc : 0;
            o.s = (int)(a.ia[row] - base);
            o.e = (int)(a.ia[row + 1] - base);
            if (!mine) o.s = o.e = 0;
        }
    };
    // epilogue operands of the CURRENT block's first pass: issued after its gathers and before the next block's stream (they
    // are not carried across a block as a second register set; the wait for the gathers still leaves them in flight)
    auto issue_ops = [&](const Blk &b, Ops &o) {
        if (EPI) {
            int64_t row = rloc < b.nr ? (int64_t)b.r0 + rloc : 0;
            if (MULTI && rloc >= b.nr) row = b.r0 < a.nrows ? b.r0 : a.nrows - 1;      // idle lanes: a row of the block's own class (the class search starts there)
            o.yo = a.yin[row];
            o.xi = a.xl[row];
            o.fr = FAR ? far_at(row, b.cls) : d2{0.0, 0.0};
            if (!need_y) o.yo = d2{0.0, 0.0};
        }
    };
    // Finished rows of the ordered walk are not stored row by row: a wavefront's consecutive blocks hold consecutive rows, so
    // the results wait in a wave-private LDS buffer and leave as full 1 KB stores when the run of rows ends (chunk change) or
    // the buffer is full.  (tools/lab/region_probe: one 512-byte store per 8 KB block costs a stream 17 %, the same bytes in
    // 1 KB stores every fourth block 4 %.)  OPS 3: a group cut between two blocks of the SAME run is summed in the buffer;
    // only the groups cut at the ends of a run are added atomically.
#ifndef QBH_BUF_MODE
#define QBH_BUF_MODE 1                           // 0 never | 1 the sliced far pass | 2 every pass of the ordered walk
#endif
    constexpr bool BUF = DYN && (QBH_BUF_MODE == 2 || (QBH_BUF_MODE == 1 && OPS == 3));
    constexpr int CAP = 256;
    __shared__ d2 rbuf_s[BUF ? 4 * CAP : 1];
    d2 *rbuf = rbuf_s + (BUF ? wv * CAP : 0);
    int64_t buf_r0 = 0;
    int buf_n = 0, buf_nb = 0;                   // rows buffered; rows buffered before the current block
    bool at_first = false, at_last = false, merge_first = false, direct = !BUF;
    auto emit = [&](int64_t row, d2 v, bool atomic) {
        if (OPS == 3) {
            if (row < a.nrows) {
                if (atomic) {
                    double *yp = reinterpret_cast<double *>(a.y + row);
                    unsafeAtomicAdd(yp, v.x);
                    unsafeAtomicAdd(yp + 1, v.y);
                } else {
                    QBH_ROW_STORE(a.y + row, v);
                }
            }
        } else {
            QBH_ROW_STORE(a.y + row, v);
        }
    };
    auto flush = [&]() {
        if (BUF && buf_n > 0) {
            wave_lds_fence();
            for (int i = lane; i < buf_n; i += 64) emit(buf_r0 + i, rbuf[i], OPS == 3 && ((i < 8 && at_first) || (i >= buf_n - 8 && at_last)));
            wave_lds_fence();
            buf_n = 0;
        }
    };
    // before the rows of a block: first row, row count, "first group continues the buffered one"
    auto open_block = [&](int64_t first, int nrows_blk, bool c0, bool fits) {
        if (BUF) {
            if (buf_n > 0 && (!fits || first != buf_r0 + buf_n - (c0 ? 8 : 0) || first + nrows_blk - buf_r0 > CAP)) flush();
            direct = !fits || nrows_blk > CAP;
            if (!direct) {
                if (buf_n == 0) {
                    buf_r0 = first;
                    at_first = c0;
                    merge_first = false;
                } else {
                    merge_first = c0;
                }
                buf_nb = buf_n;
            }
        }
    };
    auto close_block = [&](int64_t first, int nrows_blk, bool c1) {
        if (BUF && !direct) {
            buf_n = (int)(first + nrows_blk - buf_r0);
            at_last = c1;
        }
    };
    auto finish_row = [&](int64_t row, d2 sum, d2 yo, d2 xi, d2 fr, bool clip) {
        d2 v = sum;
        if (EPI) {
            if (FAR) sum += fr;
            v = a.alpha * sum + a.beta * yo + a.gamma * xi;
            acc[0] += xi.x * v.x + xi.y * v.y;
            acc[1] += xi.x * v.y - xi.y * v.x;
            acc[2] += v.x * v.x + v.y * v.y;
        }
        if (BUF && !direct) {
            const int idx = (int)(row - buf_r0);
            if (OPS == 3 && merge_first && idx < buf_nb) v += rbuf[idx];
            rbuf[idx] = v;
        } else {
            emit(row, v, clip);
        }
    };

    // Ordered dynamic walk: the XCDs do not run at the same speed (measured with QBH_XCD_TIMING on C3: three of the eight finish
    // their eighth of the near pass 1.5-1.9 ms before the pass ends, one its eighth of the far pass 1.3 ms early), so a wavefront
    // whose own XCD's region is exhausted joins the queue of the next XCD's region, and so on round the ring.  The hop is OUTSIDE
    // the pipelined loop (a pipeline drain and refill per hop, at most 7 per wavefront): the loop itself has no new branch.
    constexpr int NHOP =
#ifdef QBH_NO_XCD_STEAL
        1;
#else
        dyn ? 8 : 1;
#endif
#ifdef QBH_WAVE_TIMING
    unsigned long long tm[4] = {0, 0, 0, 0}, nblk = 0;
#endif
    int64_t red_slot = -1;                       // CHUNKRED: slot of the chunk whose rows acc[] is collecting
    auto red_flush = [&]() {
        if constexpr (CHUNKRED) {
            if (red_slot >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
                if (lane == 0) {
                    double *slot = a.chunk_red + red_slot * 3;
                    slot[0] = acc[0];
                    slot[1] = acc[1];
                    slot[2] = acc[2];
                }
                acc[0] = acc[1] = acc[2] = 0.0;
            }
        }
    };
    for (int hop = 0; hop < NHOP; ++hop) {
    if constexpr (dyn) dw.init(a.n_wb, a.wctr, lane, (int)((blockIdx.x + hop) & 7));
    int64_t lb = dyn ? 0 : walk.slot;
    int dq0 = load_desc(lb), dq1 = load_desc(lb + step);
    Blk b0 = decode(dq0), b1 = decode(dq1);
    int cA[8];
    d2 vA[8];
    Ops oA;
    issue(b0, cA, vA, oA);
#ifdef QBH_WAVE_TIMING
#define QBH_TICK(i, t_from) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm[i] += t_ - (t_from); t_from = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long t_mark = __builtin_amdgcn_s_memtime();
#else
#define QBH_TICK(i, t_from) do { } while (0)
#endif
    while (dyn ? dw.live(lb) : (lb < walk.per_xcd)) {
        const int dq2 = load_desc(lb + 2 * step);
        unsigned int reply = 0;
        bool asking = false;
        if constexpr (dyn) {
            asking = dw.asks(lb);
            if (asking) {
                if constexpr (CHUNKRED) {            // first turn of a chunk: the sums of the chunk before go to its slot
                    red_flush();
                    red_slot = dw.chunk_slot();
                }
                reply = dw.ask(lane);                // in front of the gathers: answered by the time they are
            }
        }
        d2 xv[8];
        if constexpr (C16) {
            // sliced far part: slot i of a block belongs to far row 8 g + i % 8 (groups and blocks start at multiples of 8 slots)
            const d2 *xq = a.xg + b0.xb + (OPS == 3 ? (lane & 7) : 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = xq[OPS == 3 ? (cA[u] << 3) : cA[u]];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = a.xg[cA[u]];
        }
        issue_ops(b0, oA);
        __builtin_amdgcn_sched_barrier(0);      // the gathers go out BEFORE the next block's stream (in-order return)
        int cB[8];
        d2 vB[8];
        Ops oB;
        issue(b1, cB, vB, oB);
        __builtin_amdgcn_sched_barrier(0);
        QBH_TICK(0, t_mark);                   // top of the turn .. gathers and next stream issued (waits for this block's columns)
        if (b0.n >= 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) prod[lane + u * 64] = cmul(vA[u], xv[u]);
            wave_lds_fence();
            QBH_TICK(1, t_mark);               // .. gathers arrived, products in LDS
            constexpr int RSTRIDE = OPS == 3 ? 8 : 1;                  // distance of a row's consecutive entries in the tile
            const int nrows_blk = OPS == 3 ? 8 * b0.nr : b0.nr;
            const int64_t first_row = OPS == 3 ? (int64_t)b0.r0 * 8 : (int64_t)b0.r0;
            open_block(first_row, nrows_blk, b0.cont0, true);
            for (int rbase = 0; rbase < nrows_blk; rbase += RP) {
                const int row = rbase + rloc;
                int s_ = oA.s, e_ = oA.e;
                bool clip = false;
                d2 yo = oA.yo, xi = oA.xi, fr = oA.fr;
                if (OPS == 3) {
                    group_range(b0, oA.s, row, s_, e_, clip);        // every lane takes part in the cross-lane moves
                    if (row >= nrows_blk) s_ = e_ = 0;
                } else if (rbase > 0) {
                    s_ = e_ = 0;
                    if (row < nrows_blk) {
                        s_ = (int)(a.ia[b0.r0 + row] - b0.p0);
                        e_ = (int)(a.ia[b0.r0 + row + 1] - b0.p0);
                        if (EPI && sub == 0) {
                            yo = need_y ? a.yin[b0.r0 + row] : d2{0.0, 0.0};
                            xi = a.xl[b0.r0 + row];
                            if (FAR) fr = far_at((int64_t)b0.r0 + row, b0.cls);
                        }
                    }
                }
                d2 sum = {0.0, 0.0};
                for (int k = s_ + sub * RSTRIDE; k < e_; k += TPR * RSTRIDE) sum += prod[k];
#pragma unroll
                for (int off = TPR / 2; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (sub == 0 && row < nrows_blk) finish_row(first_row + row, sum, yo, xi, fr, clip);
            }
            close_block(first_row, nrows_blk, b0.cont1);
            QBH_TICK(2, t_mark);               // .. rows reduced and finished
            wave_lds_fence();
        } else {
            open_block(0, 0, false, false);
            for (int r = 0; r < (OPS == 3 ? 0 : b0.nr); ++r) {     // a row longer than the tile: row at a time (correctness path; sliced blocks never exceed the tile)
                const int64_t row = (int64_t)b0.r0 + r;
                const int64_t s_ = a.ia[row], e_ = a.ia[row + 1];
                d2 sum = {0.0, 0.0};
                for (int64_t k = s_ + lane; k < e_; k += 64) sum += cmul(a.val[k], C16 ? a.xg[b0.xb + a.ja16[k]] : a.xg[a.ja[k] & a.colmask]);
                sum.x = wave_sum(sum.x);
                sum.y = wave_sum(sum.y);
                if (lane == 0) {
                    d2 yo = {0.0, 0.0}, xi = {0.0, 0.0}, fr = {0.0, 0.0};
                    if (EPI) {
                        if (need_y) yo = a.yin[row];
                        xi = a.xl[row];
                        if (FAR) fr = far_at(row, b0.cls);
                    }
                    finish_row(row, sum, yo, xi, fr, false);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cA[u] = cB[u];
            vA[u] = vB[u];
        }
        oA.s = oB.s;
        oA.e = oB.e;
        b0 = b1;
        b1 = decode(dq2);
        if constexpr (dyn) {
            if (asking) dw.nxt = dw.take(reply);
        }
        lb += step;
        if constexpr (dyn) dw.advance(lb);
#ifdef QBH_WAVE_TIMING
        // the copy of the next block's registers needs its VALUES: the wait for the rest of the stream lands here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QBH_TICK(3, t_mark);
        ++nblk;
#endif
    }
    flush();
    red_flush();
    red_slot = -1;
    }       // hop
#ifdef QBH_XCD_TIMING            // debug build: when does each XCD run out of blocks?  (s_memtime ticks; slots 1 / 2 behind every XCD's counter)
    if (dyn && lane == 0) {
        const unsigned long long t_end = wall_clock64();       // the device-wide constant-rate counter (s_memtime is per XCD)
        atomicMax(a.wctr + (blockIdx.x & 7) * 16 + 1, t_end);
        atomicMin(a.wctr + (blockIdx.x & 7) * 16 + 2, t_end);
    }
#endif
#ifdef QBH_WAVE_TIMING
    if (lane == 0) {
        unsigned long long *dbg = a.wctr + 128 - 8;             // last 8 words of this pass's counter block
        for (int i = 0; i < 4; ++i) atomicAdd(dbg + i, tm[i]);
        atomicAdd(dbg + 4, nblk);
    }
#endif
    if (!CHUNKRED && a.partials != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c * 4 + wv] = acc[c];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4 + 0] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
        }
    }
}

// slots of the chunk partials -> 256 x 3 partial sums in a fixed order (block b: slots [b * per, (b + 1) * per), lanes striding,
// fixed tree): what finish_reduction / k_reduce_partials then add up.  34 MB at C3, once per SpMV: ~10 us.
__global__ __launch_bounds__(256) void k_reduce_chunks(const double *slots, int64_t n_slots, double *partials)
{
    __shared__ double sm[12];
    const int64_t per = (n_slots + gridDim.x - 1) / gridDim.x, s0 = (int64_t)blockIdx.x * per, s1 = s0 + per < n_slots ? s0 + per : n_slots;
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = s0 + threadIdx.x; i < s1; i += 256) {
        v[0] += slots[i * 3 + 0];
        v[1] += slots[i * 3 + 1];
        v[2] += slots[i * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = wave_sum(v[c]);
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) sm[c * 4 + (threadIdx.x >> 6)] = v[c];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) partials[(size_t)blockIdx.x * 3 + c] = (sm[c * 4 + 0] + sm[c * 4 + 1]) + (sm[c * 4 + 2] + sm[c * 4 + 3]);
}
int launch_reduce_chunks(const double *slots, int64_t n_slots, double *partials, int *nparts_out, hipStream_t s)
{
    const int g = 256;
    hipLaunchKernelGGL(k_reduce_chunks, dim3(g), dim3(256), 0, s, slots, n_slots, partials);
    QBH_HIP(hipGetLastError());
    if (nparts_out) *nparts_out = g;
    return QBH_OK;
}
int64_t wave2_chunk_slots(int64_t n_wb)
{
    const int64_t per = (n_wb + 7) >> 3;
    return 8 * ((per + kDynChunk - 1) / kDynChunk);
}

template <int OPS, bool C16 = false>
static void launch_wave2_tpr(const SpmvArgs &a, int tpr, int grid, hipStream_t s)
{
    if (a.swizzle == 3) {
        switch (tpr) {
        case 2:  hipLaunchKernelGGL((k_spmv_wave2<2, OPS, true, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 4:  hipLaunchKernelGGL((k_spmv_wave2<4, OPS, true, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((k_spmv_wave2<8, OPS, true, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        }
    } else {
        switch (tpr) {
        case 2:  hipLaunchKernelGGL((k_spmv_wave2<2, OPS, false, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        case 4:  hipLaunchKernelGGL((k_spmv_wave2<4, OPS, false, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((k_spmv_wave2<8, OPS, false, C16>),  dim3(grid), dim3(kBlock), 0, s, a); break;
        }
    }
}

int launch_spmv_wave2(const SpmvArgs &a_in, int tpr, int ops, int grid, hipStream_t s)
{
    SpmvArgs a = a_in;
    if (a.yin == nullptr) a.yin = a.y;           // the beta term reads y itself unless a driver names another vector
    if (a.swizzle == 3 && (ops == 1 || ops == 2 || ops == 4) && a.chunk_red == nullptr) {
        set_error("launch_spmv_wave2: the dynamic walk of an epilogue pass needs its chunk-partial slots");
        return QBH_EINVAL;
    }
    if (a.ja16 != nullptr) {                 // 2-byte columns: the one-class near passes and the sliced far pass
        if (ops == 3)      launch_wave2_tpr<3, true>(a, tpr, grid, s);
        else if (ops == 1) launch_wave2_tpr<1, true>(a, tpr, grid, s);
        else if (ops == 2) launch_wave2_tpr<2, true>(a, tpr, grid, s);
        else {
            set_error("launch_spmv_wave2: 2-byte columns with pass form %d", ops);
            return QBH_EINVAL;
        }
    }
    else if (ops == 0) launch_wave2_tpr<0>(a, tpr, grid, s);
    else if (ops == 3) launch_wave2_tpr<3>(a, tpr, grid, s);
    else if (ops == 1) launch_wave2_tpr<1>(a, tpr, grid, s);
    else if (ops == 4) launch_wave2_tpr<4>(a, tpr, grid, s);
    else               launch_wave2_tpr<2>(a, tpr, grid, s);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

template <int OPS>
static int occ_wave2(int tpr)
{
    int occ = 0;
    hipError_t e;
    switch (tpr) {
    case 2:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave2<2, OPS, true>, kBlock, 0); break;
    case 4:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave2<4, OPS, true>, kBlock, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_spmv_wave2<8, OPS, true>, kBlock, 0); break;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return occ;
}
int wave2_kernel_occupancy(int tpr, int ops)
{
    return ops == 0 ? occ_wave2<0>(tpr) : ops == 3 ? occ_wave2<3>(tpr) : ops == 1 ? occ_wave2<1>(tpr) : ops == 4 ? occ_wave2<4>(tpr) : occ_wave2<2>(tpr);
}

// ---- Kronecker split: tiled copy of x, structure check, count / fill of the two parts ----
// Tiled copy of x for B = 8, through LDS: a workgroup moves 32 major indices x 8 bands; it reads 1 KB runs of x (64 minor indices
// of one major index) and writes 4 KB runs of the tiled copy (32 major indices of one band) -- row stores in 128-byte pieces cost
// several times their share of the bytes (tools/lab/region_probe).  The last, narrower band (S % 8 != 0) and other band widths
// take the element-wise kernel.
// xt_real (real wire of a split shard, qbh_opts.real_wire): the tiled copy is written as packed REAL PARTS (8 bytes per element);
// a non-zero imaginary part raises *flag (finish_real_wire turns that into a loud error)
__global__ __launch_bounds__(kBlock) void k_kron_tile_edge(const d2 *x, d2 *xt, KronTile t, int64_t band0, int xt_real, int *flag)
{
    // elements of bands >= band0
    const int64_t d0 = band0 * t.B, w = t.S - d0;
    const int64_t cnt = t.NU * w;
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = e / w, r = u * t.S + d0 + (e - u * w);
        const d2 v = x[r];
        if (xt_real) {
            reinterpret_cast<double *>(xt)[t.tile(r)] = v.x;
            bad |= v.y != 0.0;
        } else {
            xt[t.tile(r)] = v;
        }
    }
    if (bad) *flag = 1;
}
__global__ __launch_bounds__(kBlock) void k_kron_tile8(const d2 *x, d2 *xt, KronTile t, int64_t nfb, int xt_real, int *flag)
{
    constexpr int TU = 32, TB = 8, LD = TB * 8 + 1;          // +1: the band-major read of the tile walks rows of the LDS array
    __shared__ d2 tilebuf[TU * LD];
    const int64_t tiles_u = (t.NU + TU - 1) / TU, tiles_b = (nfb + TB - 1) / TB;
    bool bad = false;
    for (int64_t w = blockIdx.x; w < tiles_u * tiles_b; w += gridDim.x) {
        const int64_t tb = w / tiles_u, tu = w - tb * tiles_u;     // consecutive workgroups: the same bands, consecutive major indices
        const int64_t u0 = tu * TU, b0 = tb * TB;
        const int nu = (int)(t.NU - u0 < TU ? t.NU - u0 : TU), nb = (int)(nfb - b0 < TB ? nfb - b0 : TB);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TU * TB * 8 / kBlock; ++i) {
            const int idx = threadIdx.x + i * kBlock, ul = idx >> 6, dl = idx & 63;
            if (ul < nu && dl < nb * 8) tilebuf[ul * LD + dl] = __builtin_nontemporal_load(x + (u0 + ul) * t.S + b0 * 8 + dl);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TU * TB * 8 / kBlock; ++i) {
            const int idx = threadIdx.x + i * kBlock, bl = idx >> 8, rest = idx & 255, ul = rest >> 3, j = rest & 7;
            if (bl < nb && ul < nu) {
                const d2 v = tilebuf[ul * LD + bl * 8 + j];
                const int64_t o = (b0 + bl) * 8 * t.NU + (u0 + ul) * 8 + j;
                if (xt_real) {
                    reinterpret_cast<double *>(xt)[o] = v.x;
                    bad |= v.y != 0.0;
                } else {
                    xt[o] = v;
                }
            }
        }
    }
    if (bad) *flag = 1;
}
int launch_kron_tile(const d2 *x, d2 *xt, int64_t n, const KronTile &t, hipStream_t s, int xt_real, int *flag)
{
    (void)n;
    if (xt_real && !flag) return QBH_EINVAL;
    const int64_t nfb = t.B == 8 ? t.S / 8 : 0;                  // full bands through the LDS kernel
    if (nfb > 0) hipLaunchKernelGGL(k_kron_tile8, dim3(4096), dim3(kBlock), 0, s, x, xt, t, nfb, xt_real, flag);
    if (nfb * t.B < t.S) hipLaunchKernelGGL(k_kron_tile_edge, dim3(nfb > 0 ? 256 : 2048), dim3(kBlock), 0, s, x, xt, t, nfb, xt_real, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// Under a communicator the tiled blocks of the ranks arrive rank after rank (d_xfull, or d_xfull_r as packed real parts); the far
// part of EVERY shard indexes the tiled order of the WHOLE vector (KronTile{S, NUg, B}) -- the same 2-byte columns as the
// one-GPU operator, relative to the block's band -- so the pieces are moved to their place: band b of rank q's block (NU_q
// consecutive major indices, one contiguous run) becomes the run behind major index cu[q] of band b.  One launch per gather
// part, blockIdx.y = source rank; real wire: the 8-byte elements are expanded on the way (zero imaginary part).
__global__ __launch_bounds__(kBlock) void k_kron_place(KronPlace a)
{
    const int q = blockIdx.y;
    const int64_t nuq = a.cu[q + 1] - a.cu[q], full = a.nfb * a.B * nuq, wE = a.S - a.nfb * a.B;     // elements of rank q's full bands; width of the edge band
    const double *sr = reinterpret_cast<const double *>(a.src);
    if (a.list != nullptr) {
        // needed major indices only: work item = (band, listed major), B elements each (one 128-byte line of complex128)
        if (a.compact && q == a.skip) return;
        const int64_t nl = a.lo[q + 1] - a.lo[q];
        const int32_t *lst = a.list + a.lo[q];
        const int64_t nb = a.band1 - a.band0;
        for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < nb * nl * a.B; w += (int64_t)gridDim.x * kBlock) {
            const int64_t j = w % a.B, t = w / a.B, i = t % nl, b = a.band0 + t / nl;
            const int64_t ul = lst[i];
            int64_t e, o;
            const int64_t um = a.compact ? i : ul, nm = a.compact ? nl : nuq;      // position and count of the major indices in the source piece
            if (b < a.nfb) {
                e = b * a.B * nm + um * a.B + j;
                o = b * a.B * a.NUg + (a.cu[q] + ul) * a.B + j;
            } else {                                         // the narrow edge band: wE elements per major index
                if (j >= wE) continue;
                e = a.nfb * a.B * nm + um * wE + j;
                o = a.nfb * a.B * a.NUg + (a.cu[q] + ul) * wE + j;
            }
            a.dst[o] = a.real ? d2{sr[a.base[q] + e], 0.0} : a.src[a.base[q] + e];
        }
        return;
    }
    const int64_t e0 = a.off[q], e1 = e0 + a.len[q];
    for (int64_t e = e0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; e < e1; e += (int64_t)gridDim.x * kBlock) {
        int64_t o;
        if (e < full) {
            const int64_t b = e / (a.B * nuq), r = e - b * a.B * nuq;
            o = b * a.B * a.NUg + a.cu[q] * a.B + r;
        } else {
            o = a.nfb * a.B * a.NUg + a.cu[q] * wE + (e - full);
        }
        a.dst[o] = a.real ? d2{sr[a.base[q] + e], 0.0} : a.src[a.base[q] + e];
    }
}
int launch_kron_place(const KronPlace &a, hipStream_t s)
{
    int64_t longest = 0;
    for (int q = 0; q < a.nr; ++q) {
        const int64_t n = a.list ? ((a.compact && q == a.skip) ? 0 : (a.band1 - a.band0) * (a.lo[q + 1] - a.lo[q]) * a.B) : a.len[q];
        longest = n > longest ? n : longest;
    }
    if (longest <= 0 || a.nr <= 0) return QBH_OK;
    const int64_t gx = std::min<int64_t>(2048, (longest + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_kron_place, dim3((unsigned)gx, (unsigned)a.nr), dim3(kBlock), 0, s, a);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// personalised exchange, sender side: for destination p (blockIdx.y) the listed major indices of the own tiled block, band-major
__global__ __launch_bounds__(kBlock) void k_kron_pack(KronPack a)
{
    const int p = blockIdx.y;
    const int64_t nl = a.lo[p + 1] - a.lo[p], wE = a.S - a.nfb * a.B, nb = a.nfb + (wE > 0 ? 1 : 0);
    const int32_t *lst = a.list + a.lo[p];
    const double *sr = reinterpret_cast<const double *>(a.src);
    double *dr = reinterpret_cast<double *>(a.dst);
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < nb * nl * a.B; w += (int64_t)gridDim.x * kBlock) {
        const int64_t j = w % a.B, t = w / a.B, i = t % nl, b = t / nl;
        const int64_t ul = lst[i];
        int64_t e, o;
        if (b < a.nfb) {
            e = b * a.B * a.NUq + ul * a.B + j;
            o = b * a.B * nl + i * a.B + j;
        } else {
            if (j >= wE) continue;
            e = a.nfb * a.B * a.NUq + ul * wE + j;
            o = a.nfb * a.B * nl + i * wE + j;
        }
        if (a.real) dr[a.base[p] + o] = sr[e];
        else        a.dst[a.base[p] + o] = a.src[e];
    }
}
int launch_kron_pack(const KronPack &a, hipStream_t s)
{
    int64_t longest = 0;
    const int64_t nb = a.nfb + ((a.S - a.nfb * a.B) > 0 ? 1 : 0);
    for (int p = 0; p < a.nr; ++p) longest = std::max<int64_t>(longest, nb * (a.lo[p + 1] - a.lo[p]) * a.B);
    if (longest <= 0 || a.nr <= 0) return QBH_OK;
    const int64_t gx = std::min<int64_t>(2048, (longest + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_kron_pack, dim3((unsigned)gx, (unsigned)a.nr), dim3(kBlock), 0, s, a);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// which major indices of the whole operator does this shard read through its far and cross parts?  (Columns are positions in the
// tiled order of the whole vector: full band b, major u, j -> b B NUg + u B + j; edge band -> nfb B NUg + u wE + j.)
__global__ __launch_bounds__(kBlock) void k_kron_need(const uint16_t *c16_f, const int32_t *ja_f, int64_t far_slots, const int32_t *ja_x, int64_t nnz_x,
                                                      int64_t S, int64_t NUg, int B, uint8_t *need)
{
    const int64_t nfb = S / B, fullx = nfb * B * NUg, wE = S - nfb * B;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < far_slots; i += stride) {
        int64_t u;
        if (c16_f != nullptr) u = (int64_t)c16_f[i] % NUg;
        else {
            const int64_t c = ja_f[i];
            u = c < fullx ? (c % (B * NUg)) / B : (c - fullx) / (wE > 0 ? wE : 1);
        }
        if (u >= 0 && u < NUg) need[u] = 1;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz_x; i += stride) {
        const int64_t c = ja_x[i];
        const int64_t u = c < fullx ? (c % (B * NUg)) / B : (c - fullx) / (wE > 0 ? wE : 1);
        if (u >= 0 && u < NUg) need[u] = 1;
    }
}
int launch_kron_need(const uint16_t *c16_f, const int32_t *ja_f, int64_t far_slots, const int32_t *ja_x, int64_t nnz_x, int64_t S, int64_t NUg, int B,
                     uint8_t *need, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_need, dim3(2048), dim3(kBlock), 0, s, c16_f, ja_f, far_slots, ja_x, nnz_x, S, NUg, B, need);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// coded values: the same two parts with cw-byte codes instead of complex128 values
__global__ __launch_bounds__(kBlock) void k_kron_fill_codes(const int64_t *ia, const int32_t *ja, const uint8_t *code, int cw, int64_t nrows, KronTile t,
                                                            const int64_t *ia_n, int32_t *ja_n, uint8_t *code_n, const int64_t *ia_f, int32_t *ja_f,
                                                            uint8_t *code_f)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t.orig(f);
        const int64_t maj = r / t.S;
        int64_t pn = ia_n[r], pf = ia_f[f];
        for (int64_t k = ia[r]; k < ia[r + 1]; ++k) {
            const int32_t c = ja[k];
            if ((c / t.S) != maj) {
                ja_f[pf] = (int32_t)t.tile(c);
                for (int b = 0; b < cw; ++b) code_f[pf * cw + b] = code[k * cw + b];
                ++pf;
            } else {
                ja_n[pn] = c;
                for (int b = 0; b < cw; ++b) code_n[pn * cw + b] = code[k * cw + b];
                ++pn;
            }
        }
    }
}
int launch_kron_fill_codes(const int64_t *ia, const int32_t *ja, const uint8_t *code, int cw, int64_t nrows, const KronTile &t, const int64_t *ia_n,
                           int32_t *ja_n, uint8_t *code_n, const int64_t *ia_f, int32_t *ja_f, uint8_t *code_f, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_fill_codes, dim3(4096), dim3(kBlock), 0, s, ia, ja, code, cw, nrows, t, ia_n, ja_n, code_n, ia_f, ja_f, code_f);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
__global__ __launch_bounds__(kBlock) void k_kron_tile_re(const double *x, double *xt, int64_t n, KronTile t)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) xt[t.tile(r)] = x[r];
}
int launch_kron_tile_re(const double *x, double *xt, int64_t n, const KronTile &t, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_tile_re, dim3(2048), dim3(kBlock), 0, s, x, xt, n, t);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// product structure with minor size S: every entry keeps the major index (near) or keeps the minor index (far)
__global__ __launch_bounds__(kBlock) void k_kron_check(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int *flag)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t maj = r / S, mnr = r - maj * S;
        bool bad = false;
        for (int64_t k = ia[r]; k < ia[r + 1]; ++k) {
            const int64_t c = ja[k], cm = c / S;
            bad = bad || (cm != maj && c - cm * S != mnr);
        }
        if (bad) *flag = 1;
    }
}
int launch_kron_check(const int64_t *ia, const int32_t *ja, int64_t nrows, int64_t S, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_check, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, S, d_flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_kron_count(const int64_t *ia, const int32_t *ja, int64_t nrows, KronTile t, int32_t *cnt_near,
                                                       int32_t *cnt_far)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t.orig(f);
        const int64_t maj = r / t.S;
        int nf = 0;
        const int64_t s0 = ia[r], e0 = ia[r + 1];
        for (int64_t k = s0; k < e0; ++k) nf += (ja[k] / t.S) != maj;
        cnt_far[f] = nf;
        cnt_near[r] = (int)(e0 - s0) - nf;
    }
}
int launch_kron_count(const int64_t *ia, const int32_t *ja, int64_t nrows, const KronTile &t, int32_t *cnt_near, int32_t *cnt_far, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_count, dim3(4096), dim3(kBlock), 0, s, ia, ja, nrows, t, cnt_near, cnt_far);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// near entries keep their row and column; far entries go to the tiled row with the tiled column (ascending in both)
__global__ __launch_bounds__(kBlock) void k_kron_fill(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, KronTile t,
                                                      const int64_t *ia_n, int32_t *ja_n, d2 *val_n, const int64_t *ia_f, int32_t *ja_f, d2 *val_f)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t.orig(f);
        const int64_t maj = r / t.S;
        int64_t pn = ia_n[r], pf = ia_f[f];
        for (int64_t k = ia[r]; k < ia[r + 1]; ++k) {
            const int32_t c = ja[k];
            if ((c / t.S) != maj) {
                ja_f[pf] = (int32_t)t.tile(c);
                val_f[pf++] = val[k];
            } else {
                ja_n[pn] = c;
                val_n[pn++] = val[k];
            }
        }
    }
}
// sliced far part: slots of a group (8 consecutive far rows) = 8 * (longest far row of the group)
__global__ __launch_bounds__(kBlock) void k_kron_group_width(const int32_t *cnt_far, int64_t nrows, int64_t ngroups, int32_t *gw)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
        int mx = 0;
        for (int j = 0; j < 8; ++j) {
            const int64_t f = g * 8 + j;
            const int c = f < nrows ? cnt_far[f] : 0;
            mx = c > mx ? c : mx;
        }
        gw[g] = 8 * (mx > 0 ? mx : 1);       // an empty group keeps one (padding) slot per row: every row belongs to a block
    }
}
int launch_kron_group_width(const int32_t *cnt_far, int64_t nrows, int64_t ngroups, int32_t *gw, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_group_width, dim3(2048), dim3(kBlock), 0, s, cnt_far, nrows, ngroups, gw);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// blocks of the sliced far part: block i = slots [512 i, 512 (i + 1)); r0 = the group that holds its first slot
// shift: the arrays of the far part start `shift` entries behind a 128-byte boundary (they follow the near part inside the
// operator's own arrays): the blocks are cut `shift` slots early so that every block starts ON a boundary (8 lines per 1 KB
// value load instead of 9, 2 per 256-byte column load instead of 3); the first block is that much shorter
__global__ __launch_bounds__(kBlock) void k_build_slotdesc(const int64_t *gia, int64_t ngroups, int64_t slots, WaveDesc *wd, int64_t n_wb, int64_t shift)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_wb + 2; i += (int64_t)gridDim.x * blockDim.x) {
        WaveDesc d;
        if (i >= n_wb) {
            d.p0 = slots;
            d.r0 = (int32_t)ngroups;
            d.pad = 0;
        } else {
            const int64_t P = i * 512 > shift ? i * 512 - shift : 0;
            int64_t lo = 0, hi = ngroups;            // first group with gia[g] > P
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (gia[mid] > P) hi = mid;
                else lo = mid + 1;
            }
            d.p0 = P;
            d.r0 = (int32_t)(lo - 1);
            d.pad = gia[lo - 1] < P ? 1 : 0;
        }
        wd[i] = d;
    }
}
int launch_build_slotdesc(const int64_t *gia, int64_t ngroups, int64_t slots, WaveDesc *wd, int64_t n_wb, int64_t shift, hipStream_t s)
{
    hipLaunchKernelGGL(k_build_slotdesc, dim3(2048), dim3(kBlock), 0, s, gia, ngroups, slots, wd, n_wb, shift);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// rows of the groups that a block boundary cuts: zero before the far pass adds both parts
__global__ __launch_bounds__(kBlock) void k_zero_cut_groups(const WaveDesc *wd, int64_t n_wb, int64_t nrows, d2 *far)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_wb * 8; t += (int64_t)gridDim.x * blockDim.x) {
        const WaveDesc d = wd[t >> 3];
        const int64_t row = (int64_t)d.r0 * 8 + (t & 7);
        if ((d.pad & 1) && row < nrows) far[row] = d2{0.0, 0.0};
    }
}
int launch_zero_cut_groups(const WaveDesc *wd, int64_t n_wb, int64_t nrows, d2 *far, hipStream_t s)
{
    hipLaunchKernelGGL(k_zero_cut_groups, dim3(2048), dim3(kBlock), 0, s, wd, n_wb, nrows, far);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// near entries as in k_kron_fill; far entries of far row f = 8g + j go to gia[g] + 8k + j (k-th far entry of the row), the
// rest of the group's slots are padding: value 0, column = the row's own tiled index (always a valid element of the tiled x)
__global__ __launch_bounds__(kBlock) void k_kron_fill_sliced(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, KronTile t,
                                                             const int64_t *ia_n, int32_t *ja_n, d2 *val_n, const int64_t *gia, int64_t ngroups,
                                                             int32_t *ja_f, d2 *val_f)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < ngroups * 8; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = f >> 3, jj = f & 7;
        const int64_t gb = gia[g], w = (gia[g + 1] - gb) >> 3;
        int64_t k = 0;
        if (f < nrows) {
            const int64_t r = t.orig(f);
            const int64_t maj = r / t.S;
            int64_t pn = ia_n[r];
            for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
                const int32_t c = ja[q];
                if ((c / t.S) != maj) {
                    ja_f[gb + 8 * k + jj] = (int32_t)t.tile(c);
                    val_f[gb + 8 * k + jj] = val[q];
                    ++k;
                } else {
                    ja_n[pn] = c;
                    val_n[pn++] = val[q];
                }
            }
        }
        for (; k < w; ++k) {
            ja_f[gb + 8 * k + jj] = (int32_t)(f < nrows ? f : 0);
            val_f[gb + 8 * k + jj] = d2{0.0, 0.0};
        }
    }
}
int launch_kron_fill_sliced(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, const KronTile &t, const int64_t *ia_n,
                            int32_t *ja_n, d2 *val_n, const int64_t *gia, int64_t ngroups, int32_t *ja_f, d2 *val_f, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_fill_sliced, dim3(4096), dim3(kBlock), 0, s, ia, ja, val, nrows, t, ia_n, ja_n, val_n, gia, ngroups, ja_f, val_f);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_kron_fill(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, const KronTile &t, const int64_t *ia_n, int32_t *ja_n,
                     d2 *val_n, const int64_t *ia_f, int32_t *ja_f, d2 *val_f, hipStream_t s)
{
    hipLaunchKernelGGL(k_kron_fill, dim3(4096), dim3(kBlock), 0, s, ia, ja, val, nrows, t, ia_n, ja_n, val_n, ia_f, ja_f, val_f);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// ------------------------------------------- sub-wavefront-per-row SpMV --------
// G lanes per row, NO LDS staging and no workgroup barrier: a wavefront owns 64/G consecutive rows and every lane walks
// its row in strides of G, UN entries at a time -- UN column loads, UN value loads and then UN x gathers in flight per
// lane.  Nothing limits occupancy but registers (the row kernel's 20 B/nnz of LDS hold it at 3 workgroups per CU), so
// the chain row pointer -> (column, value) -> x -> FMA of one wavefront hides behind up to 8 wavefronts per SIMD.
// The matrix stream is read straight from global memory: the G lanes of a row take G consecutive entries (64 B of
// values at G = 4), consecutive steps touch the same 128-byte lines again while they are still in L1, so HBM sees each
// line once.  Entries are taken alternately from the front and the back of the row, as in k_spmv_rows: for the
// Kronecker-structured Hamiltonians the k-th entries of consecutive rows then point at consecutive x elements.
template <int G, int UN, bool DICT>
__global__ __launch_bounds__(kBlock) void k_spmv_vector(SpmvArgs a)
{
    spmv_args_resolve(a);
    __shared__ double red[12];
    __shared__ d2 dict_s[DICT ? 256 : 1];
    const int tid = threadIdx.x;
    constexpr int RPB = kBlock / G;            // rows per workgroup pass
    const int g = tid / G, sub = tid % G;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0;
    const bool need_x = a.gamma != 0.0 || a.partials != nullptr;
    if (DICT) {
        dict_s[tid] = a.dict[tid];
        __syncthreads();
    }
    const int64_t n_chunks = (a.nrows + RPB - 1) / RPB;
    const int64_t last = a.ia[a.nrows] - 1;    // clamp for the masked lanes (nnz > 0)
    BlockWalk walk(n_chunks, a.swizzle, a.chunk_mult);
    for (int64_t lb = walk.slot; lb < walk.per_xcd; lb += walk.nslot) {
        const int64_t b = walk.block(lb);
        if (b >= n_chunks) continue;
        const int64_t row = b * RPB + g;
        const bool rowok = row < a.nrows;
        const int64_t s = rowok ? a.ia[row] : 0;
        const int len = rowok ? (int)(a.ia[row + 1] - s) : 0;
        d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
        const bool mine = sub == 0 && rowok;
        if (mine && need_y) yo = load_y_old(a, row);
        if (mine && need_x) xi = load_x_local(a, row);
        int wmax = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(wmax, off, 64);
            wmax = o > wmax ? o : wmax;
        }
        d2 sum = {0.0, 0.0};
        for (int k0 = sub; k0 < wmax; k0 += G * UN) {
            int c[UN];
            bool ok[UN];
            int64_t q[UN];
            d2 v[UN], xv[UN];
            uint8_t cb[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int k = k0 + j * G;
                ok[j] = k < len;
                const int f = (k & 1) ? len - 1 - (k >> 1) : (k >> 1);     // front / back alternation
                q[j] = ok[j] ? s + f : last;
                c[j] = ntload(a.ja + q[j]) & a.colmask;
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                if (DICT) cb[j] = ntload(a.code + q[j]);
                else      v[j] = ntload(a.val + q[j]);
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) xv[j] = a.xg[c[j]];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                if (DICT) v[j] = dict_s[cb[j]];
                if (ok[j]) sum += cmul(v[j], xv[j]);
            }
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
            sum.x += __shfl_xor(sum.x, off, 64);
            sum.y += __shfl_xor(sum.y, off, 64);
        }
        if (mine) row_epilogue2(a, row, sum, yo, xi, acc);
    }
    if (a.partials != nullptr) {
        block_sum<3>(acc, red);
        if (tid == 0) {
            a.partials[(size_t)blockIdx.x * 3 + 0] = acc[0];
            a.partials[(size_t)blockIdx.x * 3 + 1] = acc[1];
            a.partials[(size_t)blockIdx.x * 3 + 2] = acc[2];
        }
    }
}

template <int G, bool DICT>
static int occ_vector_un(int un)
{
    int n = 0;
    hipError_t e = (un == 8) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmv_vector<G, 8, DICT>, kBlock, 0)
                 : (un == 2) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmv_vector<G, 2, DICT>, kBlock, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmv_vector<G, 4, DICT>, kBlock, 0);
    return e == hipSuccess ? n : 0;
}

// workgroups of the lanes-per-row kernel resident per CU (0 if unknown)
int vector_kernel_occupancy(int tpr, int un, bool dict)
{
    switch (tpr) {
    case 2:  return dict ? occ_vector_un<2, true>(un)  : occ_vector_un<2, false>(un);
    case 4:  return dict ? occ_vector_un<4, true>(un)  : occ_vector_un<4, false>(un);
    case 8:  return dict ? occ_vector_un<8, true>(un)  : occ_vector_un<8, false>(un);
    case 16: return dict ? occ_vector_un<16, true>(un) : occ_vector_un<16, false>(un);
    case 32: return dict ? occ_vector_un<32, true>(un) : occ_vector_un<32, false>(un);
    case 64: return dict ? occ_vector_un<64, true>(un) : occ_vector_un<64, false>(un);
    default: return 0;
    }
}

template <int G, bool DICT>
static void launch_vector_un(const SpmvArgs &a, int grid, hipStream_t s)
{
    if (a.unroll == 8)      hipLaunchKernelGGL((k_spmv_vector<G, 8, DICT>), dim3(grid), dim3(kBlock), 0, s, a);
    else if (a.unroll == 2) hipLaunchKernelGGL((k_spmv_vector<G, 2, DICT>), dim3(grid), dim3(kBlock), 0, s, a);
    else                    hipLaunchKernelGGL((k_spmv_vector<G, 4, DICT>), dim3(grid), dim3(kBlock), 0, s, a);
}

int spmv_grid(int kernel, int64_t n_blocks, int64_t nrows, int tpr)
{
    // 256 CUs; the streaming kernel fits 4 workgroups per CU (LDS), the vector kernel 8.
    int64_t units = n_blocks;
    int64_t cap = 256 * 4 * 4;
    if (kernel == QBH_KERNEL_ROWS) cap = 256 * 8 * 2;
    if (kernel == QBH_KERNEL_VECTOR) {
        const int rpb = kBlock / tpr;
        units = (nrows + rpb - 1) / rpb;
        cap = 256 * 8 * 2;
    }
    int64_t g = units < cap ? units : cap;
    g = ((g + 7) / 8) * 8;
    if (g < 8) g = 8;
    return (int)g;
}

template <int NPB, bool DICT>
static int launch_stream_tpr(const SpmvArgs &a, int tpr, int grid, hipStream_t s)
{
    switch (tpr) {
    case 1:  hipLaunchKernelGGL((k_spmv_stream<NPB, 1, DICT>),  dim3(grid), dim3(kBlock), 0, s, a); break;
    case 2:  hipLaunchKernelGGL((k_spmv_stream<NPB, 2, DICT>),  dim3(grid), dim3(kBlock), 0, s, a); break;
    case 4:  hipLaunchKernelGGL((k_spmv_stream<NPB, 4, DICT>),  dim3(grid), dim3(kBlock), 0, s, a); break;
    case 8:  hipLaunchKernelGGL((k_spmv_stream<NPB, 8, DICT>),  dim3(grid), dim3(kBlock), 0, s, a); break;
    case 16: hipLaunchKernelGGL((k_spmv_stream<NPB, 16, DICT>), dim3(grid), dim3(kBlock), 0, s, a); break;
    default: set_error("unsupported threads-per-row %d", tpr); return QBH_EINVAL;
    }
    return QBH_OK;
}

template <int NPB, int PP, int DICT>
static int occ_rows_un(int un)
{
    int n = 0;
    hipError_t e = (un == 8)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmv_rows<NPB, PP, 8, DICT, false>, kBlock, 0)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmv_rows<NPB, PP, 4, DICT, false>, kBlock, 0);
    return e == hipSuccess ? n : 0;
}

template <int NPB, int DICT>
static int occ_rows_p(int tpr, int un)
{
    switch (tpr) {
    case 1: return occ_rows_un<NPB, 1, DICT>(un);
    case 2: return occ_rows_un<NPB, 2, DICT>(un);
    case 4: return occ_rows_un<NPB, 4, DICT>(un);
    case 8: return occ_rows_un<NPB, 8, DICT>(un);
    default: return 0;
    }
}

template <int DICT>
static int occ_rows(int npb, int tpr, int un)
{
    switch (npb) {
    case 1024: return occ_rows_p<1024, DICT>(tpr, un);
    case 2048: return occ_rows_p<2048, DICT>(tpr, un);
    case 4096: return occ_rows_p<4096, DICT>(tpr, un);
    case 8192:
        if constexpr (DICT == 1) return occ_rows_p<8192, DICT>(tpr, un);
        return 0;
    default: return 0;
    }
}

// workgroups of the row kernel that are resident per CU (0 if unknown); dict_mode as k_spmv_rows' DICT
int rows_kernel_occupancy(int npb, int tpr, int un, int dict_mode)
{
    switch (dict_mode) {
    case 0: return occ_rows<0>(npb, tpr, un);
    case 1: return occ_rows<1>(npb, tpr, un);
    case 2: return occ_rows<2>(npb, tpr, un);
    default: return occ_rows<3>(npb, tpr, un);
    }
}

template <int NPB, int PP, int DICT>
static int launch_rows_un(const SpmvArgs &a, int un, int grid, hipStream_t s)
{
    if (a.xr != nullptr) {
        if (un == 8) hipLaunchKernelGGL((k_spmv_rows<NPB, PP, 8, DICT, true>), dim3(grid), dim3(kBlock), 0, s, a);
        else         hipLaunchKernelGGL((k_spmv_rows<NPB, PP, 4, DICT, true>), dim3(grid), dim3(kBlock), 0, s, a);
    } else {
        if (un == 8) hipLaunchKernelGGL((k_spmv_rows<NPB, PP, 8, DICT, false>), dim3(grid), dim3(kBlock), 0, s, a);
        else         hipLaunchKernelGGL((k_spmv_rows<NPB, PP, 4, DICT, false>), dim3(grid), dim3(kBlock), 0, s, a);
    }
    return QBH_OK;
}

template <int NPB, int DICT>
static int launch_rows_p(const SpmvArgs &a, int tpr, int un, int grid, hipStream_t s)
{
    switch (tpr) {
    case 1: return launch_rows_un<NPB, 1, DICT>(a, un, grid, s);
    case 2: return launch_rows_un<NPB, 2, DICT>(a, un, grid, s);
    case 4: return launch_rows_un<NPB, 4, DICT>(a, un, grid, s);
    case 8: return launch_rows_un<NPB, 8, DICT>(a, un, grid, s);
    default: set_error("k_spmv_rows: unsupported lanes-per-row %d (1, 2, 4, 8)", tpr); return QBH_EINVAL;
    }
}

template <int DICT>
static int launch_rows(const SpmvArgs &a, int npb, int tpr, int un, int grid, hipStream_t s)
{
    switch (npb) {
    case 1024: return launch_rows_p<1024, DICT>(a, tpr, un, grid, s);
    case 2048: return launch_rows_p<2048, DICT>(a, tpr, un, grid, s);
    case 4096: return launch_rows_p<4096, DICT>(a, tpr, un, grid, s);
    case 8192:
        if constexpr (DICT == 1) return launch_rows_p<8192, DICT>(a, tpr, un, grid, s);
        set_error("nnz_per_block 8192 needs the one-byte value dictionary");
        return QBH_EINVAL;
    default: set_error("k_spmv_rows: unsupported nnz_per_block %d", npb); return QBH_EINVAL;
    }
}

template <bool DICT>
static int launch_spmv_t(const SpmvArgs &a, int kernel, int npb, int tpr, int grid, hipStream_t s)
{
    if (kernel == QBH_KERNEL_VECTOR) {
        switch (tpr) {
        case 2:  launch_vector_un<2, DICT>(a, grid, s); break;
        case 4:  launch_vector_un<4, DICT>(a, grid, s); break;
        case 8:  launch_vector_un<8, DICT>(a, grid, s); break;
        case 16: launch_vector_un<16, DICT>(a, grid, s); break;
        case 32: launch_vector_un<32, DICT>(a, grid, s); break;
        case 64: launch_vector_un<64, DICT>(a, grid, s); break;
        default: set_error("unsupported lanes-per-row %d", tpr); return QBH_EINVAL;
        }
        return QBH_OK;
    }
    switch (npb) {
    case 1024: return launch_stream_tpr<1024, DICT>(a, tpr, grid, s);
    case 2048: return launch_stream_tpr<2048, DICT>(a, tpr, grid, s);
    case 4096: return launch_stream_tpr<4096, DICT>(a, tpr, grid, s);
    default: set_error("unsupported nnz_per_block %d (1024, 2048 or 4096)", npb); return QBH_EINVAL;
    }
}

int launch_spmv(const SpmvArgs &a_in, int kernel, int npb, int tpr, int grid, hipStream_t s)
{
    SpmvArgs a = a_in;
    if (a.yin == nullptr) a.yin = a.y;           // the beta term reads y itself unless a driver names another vector
    int rc;
    if (kernel == QBH_KERNEL_ROWS) {
        switch (a.code == nullptr ? 0 : a.dict_mode) {
        case 0: rc = launch_rows<0>(a, npb, tpr, a.unroll, grid, s); break;
        case 1: rc = launch_rows<1>(a, npb, tpr, a.unroll, grid, s); break;
        case 2: rc = launch_rows<2>(a, npb, tpr, a.unroll, grid, s); break;
        default: rc = launch_rows<3>(a, npb, tpr, a.unroll, grid, s); break;
        }
    } else if (a.code != nullptr && a.dict_mode != 1) {
        set_error("two-byte value codes need the row kernel");
        return QBH_EUNSUPP;
    } else {
        rc = (a.code != nullptr) ? launch_spmv_t<true>(a, kernel, npb, tpr, grid, s)
                                 : launch_spmv_t<false>(a, kernel, npb, tpr, grid, s);
    }
    if (rc != QBH_OK) return rc;
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// ------------------------------------------------------ row-block builder ------
// Row block w = the rows whose first nonzero falls in the nnz window
// [w*window, (w+1)*window): rb[w] = lower_bound(ia, w*window).  Embarrassingly parallel
// and nnz-balanced; block nnz < window + (longest row).
__global__ void k_build_rowblocks(const int64_t *ia, int64_t nrows, int64_t window, int32_t *rb,
                                  int64_t *bp, int64_t n_blocks)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > n_blocks) return;
    if (w == n_blocks) {
        rb[w] = (int32_t)nrows;
        bp[w] = ia[nrows];
        return;
    }
    const int64_t target = w * window;
    int64_t lo = 0, hi = nrows;          // first r in [0, nrows] with ia[r] >= target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ia[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    rb[w] = (int32_t)lo;
    bp[w] = ia[lo];
}

int launch_build_rowblocks(const int64_t *d_ia, int64_t nrows, int64_t window, int32_t *d_rb,
                           int64_t *d_bp, int64_t n_blocks, hipStream_t s)
{
    const int64_t n = n_blocks + 1;
    hipLaunchKernelGGL(k_build_rowblocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ia, nrows,
                       window, d_rb, d_bp, n_blocks);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ void k_max_rowlen(const int64_t *ia, int64_t nrows, unsigned long long *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long mx = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += stride) {
        const unsigned long long len = (unsigned long long)(ia[r + 1] - ia[r]);
        mx = len > mx ? len : mx;
    }
    if (mx) atomicMax(out, mx);
}

int launch_max_rowlen(const int64_t *d_ia, int64_t nrows, int64_t *d_out, hipStream_t s)
{
    QBH_HIP(hipMemsetAsync(d_out, 0, sizeof(int64_t), s));
    hipLaunchKernelGGL(k_max_rowlen, dim3(blas_grid(nrows)), dim3(kBlock), 0, s, d_ia, nrows,
                       (unsigned long long *)d_out);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// ------------------------------------------------- value dictionary ------------
// Lossless coding of the value stream: when a matrix holds at most 256 distinct complex128
// values (every full-basis Hamiltonian of the reference's model families does: hopping
// amplitudes, exchange constants and a handful of diagonal sums), each value is replaced by
// a 1-byte index into a dictionary that lives in LDS during SpMV; up to 65536 distinct values
// (momentum sectors) by a 2-byte index.  The stream shrinks from 20 to 5 or 6 bytes per
// nonzero; products are computed from the exact original doubles.
// (device helpers: qbh_dict.hpp)
__global__ __launch_bounds__(kBlock) void k_dict_collect(const d2 *val, int64_t nnz, DictTab T)
{
    __shared__ DictCollect D;
    dict_collect_init(D);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += stride)
        if (!dict_collect_insert(D, T, val[i])) break;
}

template <typename CT>
__global__ __launch_bounds__(kBlock) void k_dict_encode(const d2 *val, int64_t nnz, DictTab T, const d2 *dict, CT *code)
{
    __shared__ DictEncode E;
    dict_encode_init(E);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += stride)
        code[i] = (CT)dict_encode_one(E, T, dict, val[i]);
}

int dict_build_begin(DictBuild *b, int cap, hipStream_t s)
{
    DictTab &T = b->tab;
    T.cap = cap < kDictMax ? cap : kDictMax;
    QBH_HIP(qbh::dev_alloc(&T.fp, (size_t)kDictSlots * sizeof(unsigned long long)));
    QBH_HIP(qbh::dev_alloc(&T.val, (size_t)kDictSlots * sizeof(d2)));
    QBH_HIP(qbh::dev_alloc(&T.code, (size_t)kDictSlots * sizeof(uint32_t)));
    QBH_HIP(qbh::dev_alloc(&T.flags, 4 * sizeof(int)));
    QBH_HIP(hipMemsetAsync(T.fp, 0, (size_t)kDictSlots * sizeof(unsigned long long), s));
    QBH_HIP(hipMemsetAsync(T.flags, 0, 4 * sizeof(int), s));
    return QBH_OK;
}

int dict_build_finalize(DictBuild *b, d2 **d_dict_out, int *n_out, hipStream_t s)
{
    DictTab &T = b->tab;
    *n_out = 0;
    *d_dict_out = nullptr;
    int h[4] = {0, 0, 0, 0};
    QBH_HIP(hipMemcpyAsync(h, T.flags, sizeof(h), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    if (debug_sw().trace_dict) fprintf(stderr, "dict: overflow %d claimed %d cap %d\n", h[0], h[1], T.cap);
    if (h[0] || h[1] <= 0 || h[1] > T.cap) return QBH_OK;
    std::vector<unsigned long long> fp((size_t)kDictSlots);
    std::vector<d2> val((size_t)kDictSlots);
    QBH_HIP(hipMemcpy(fp.data(), T.fp, fp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    QBH_HIP(hipMemcpy(val.data(), T.val, val.size() * sizeof(d2), hipMemcpyDeviceToHost));
    struct Ent {
        unsigned long long a, b;
        int slot;
    };
    std::vector<Ent> ents;
    ents.reserve((size_t)h[1]);
    for (int sl = 0; sl < kDictSlots; ++sl)
        if (fp[(size_t)sl] != 0ULL) {
            Ent e;
            const double vx = val[(size_t)sl].x, vy = val[(size_t)sl].y;
            memcpy(&e.a, &vx, 8);
            memcpy(&e.b, &vy, 8);
            e.slot = sl;
            ents.push_back(e);
        }
    // order by bit pattern: the codes do not depend on which workgroup won an atomic
    std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    // two different values with one fingerprint would share a slot: the count would not add up
    const int n = (int)ents.size();
    if (debug_sw().trace_dict) fprintf(stderr, "dict: %d entries\n", n);
    if (n != h[1] || n > T.cap) return QBH_OK;
    const size_t n_alloc = (size_t)std::max(n, kDictLds);
    std::vector<d2> dict(n_alloc, d2{0.0, 0.0});
    std::vector<uint32_t> code((size_t)kDictSlots, 0u);
    for (int c = 0; c < n; ++c) {
        dict[(size_t)c] = val[(size_t)ents[(size_t)c].slot];
        code[(size_t)ents[(size_t)c].slot] = (uint32_t)c;
    }
    d2 *d_dict = nullptr;
    QBH_HIP(qbh::dev_alloc(&d_dict, n_alloc * sizeof(d2)));
    hipError_t e1 = hipMemcpy(d_dict, dict.data(), n_alloc * sizeof(d2), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(T.code, code.data(), code.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        (void)hipFree(d_dict);
        set_error("value dictionary upload failed");
        return QBH_EHIP;
    }
    *d_dict_out = d_dict;
    *n_out = n;
    return QBH_OK;
}

int dict_build_mismatch(DictBuild *b, int *bad, hipStream_t s)
{
    int h[4] = {0, 0, 0, 0};
    QBH_HIP(hipMemcpyAsync(h, b->tab.flags, sizeof(h), hipMemcpyDeviceToHost, s));
    QBH_HIP(hipStreamSynchronize(s));
    *bad = h[3];
    return QBH_OK;
}

void dict_build_end(DictBuild *b)
{
    DictTab &T = b->tab;
    if (T.fp) (void)hipFree(T.fp);
    if (T.val) (void)hipFree(T.val);
    if (T.code) (void)hipFree(T.code);
    if (T.flags) (void)hipFree(T.flags);
    T = DictTab{nullptr, nullptr, nullptr, nullptr, 0};
}

// Codes the value stream when it holds at most `cap` distinct values: *d_code_out (nnz * width + 16 bytes),
// *d_dict_out and *n_out > 0 on success; n = 0 (nothing allocated) when there are too many distinct values.
int build_value_dict(const d2 *d_val, int64_t nnz, int cap, uint8_t **d_code_out, d2 **d_dict_out, int *n_out, hipStream_t s)
{
    *n_out = 0;
    *d_code_out = nullptr;
    *d_dict_out = nullptr;
    DictBuild b;
    int rc = dict_build_begin(&b, cap, s);
    uint8_t *code = nullptr;
    d2 *dict = nullptr;
    if (rc == QBH_OK) {
        const int grid = blas_grid(nnz);
        hipLaunchKernelGGL(k_dict_collect, dim3(grid), dim3(kBlock), 0, s, d_val, nnz, b.tab);
        int n = 0;
        rc = dict_build_finalize(&b, &dict, &n, s);
        if (rc == QBH_OK && n > 0) {
            const int w = dict_code_width(n);
            if (qbh::dev_alloc(&code, (size_t)nnz * w + 16) != hipSuccess) {
                (void)hipGetLastError();
                n = 0;                                   // no room for the codes: stay uncoded
            } else {
                (void)hipMemsetAsync(code + (size_t)nnz * w, 0, 16, s);
                if (w == 1) hipLaunchKernelGGL(k_dict_encode<uint8_t>, dim3(grid), dim3(kBlock), 0, s, d_val, nnz, b.tab, dict, code);
                else hipLaunchKernelGGL(k_dict_encode<uint16_t>, dim3(grid), dim3(kBlock), 0, s, d_val, nnz, b.tab, dict,
                                        reinterpret_cast<uint16_t *>(code));
                int bad = 0;
                rc = dict_build_mismatch(&b, &bad, s);
                if (rc != QBH_OK || bad) n = 0;
            }
        }
        if (n > 0) {
            *n_out = n;
            *d_code_out = code;
            *d_dict_out = dict;
        } else {
            if (code) (void)hipFree(code);
            if (dict) (void)hipFree(dict);
        }
    }
    dict_build_end(&b);
    return rc;
}

// -------------------------------------------------------------- BLAS-1 ---------
int blas_grid(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > kMaxRedBlocks) g = kMaxRedBlocks;
    if (g < 1) g = 1;
    return (int)g;
}

// second stage of every reduction: one workgroup sums `nparts` partials of `ncomp`
// components in a fixed order (run-to-run reproducible, no atomics).
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *partials, int nparts, int ncomp,
                                                          double *out)
{
    __shared__ double sm[16];
    for (int c = 0; c < ncomp; ++c) {
        double v = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 1024) v += partials[(size_t)i * ncomp + c];
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += sm[w];
            out[c] = t;
        }
    }
}

int launch_reduce_partials(const double *partials, int nparts, int ncomp, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, s, partials, nparts, ncomp, out);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// Tail of a pipelined Lanczos step (lanczos_core): the second stage of the axpy's |w'|^2 (the same summation order as
// k_reduce_partials, so b_m is the number the unpipelined step returns), then the scalars of src/lanczos.cc:200-214 in the
// arithmetic the host used to do -- a = sc_x * <u, w>, b = sqrt(|w'|^2), sc_new = 1 / b -- and the NEXT step's coefficients
// (alpha = sc_new, beta = -b * sc_x, axpy scale = -sc_new^2) left in state[] for the kernels of step m + 1, which the host has
// already enqueued.  The four numbers of this step go to a pinned host slot directly: no copy engine in the stream.
// sq_ready != nullptr (under a communicator): |w'|^2 has been reduced and all-reduced already (partials unused).
__global__ __launch_bounds__(1024) void k_lanczos_tail(const double *partials, int nparts, const double *dot, double *state, double *log_slot,
                                                       double sc_x_host, int use_host, const double *sq_ready)
{
    __shared__ double sm[16];
    double v = 0.0;
    if (sq_ready == nullptr)
        for (int i = threadIdx.x; i < nparts; i += 1024) v += partials[i];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sq = 0.0;
        for (int w = 0; w < 16; ++w) sq += sm[w];
        if (sq_ready != nullptr) sq = sq_ready[0];
        const double sc_x = use_host ? sc_x_host : state[3];
        const double d = dot[0];
        const double a = sc_x * d;
        const double b = sqrt(sq);
        const double sc_new = 1.0 / b;
        state[0] = sc_new;
        state[1] = -b * sc_x;
        state[2] = -sc_new * sc_new;
        state[3] = sc_new;
        log_slot[0] = d;
        log_slot[1] = sq;
        log_slot[2] = a;
        log_slot[3] = b;
        __threadfence_system();
    }
}
int launch_lanczos_tail(const double *partials, int nparts, const double *dot, double *state, double *log_slot, double sc_x_host, int use_host,
                        hipStream_t s, const double *sq_ready)
{
    hipLaunchKernelGGL(k_lanczos_tail, dim3(1), dim3(1024), 0, s, partials, nparts, dot, state, log_slot, sc_x_host, use_host, sq_ready);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_dotc(const d2 *x, const d2 *y, int64_t n, double *partials)
{
    __shared__ double red[8];
    double acc[2] = {0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const d2 a = x[i], b = y[i];
        acc[0] += a.x * b.x + a.y * b.y;
        acc[1] += a.x * b.y - a.y * b.x;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

int launch_dotc(const d2 *x, const d2 *y, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_dotc, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, y, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// y += alpha*x ; partial |y|^2   (cblas_zaxpy + cblas_dznrm2 in one pass: K5+K6)
// yr (optional): packed real parts of the updated y -- the next SpMV's gather source in the real fast
// path, produced here instead of by a separate k_pack_real pass; flag as in k_pack_real.
// alpha_dev != nullptr: the coefficient is alpha.x * alpha_dev[0] (a scalar a previous kernel of the same stream left
// on the device -- the Lanczos step then needs one host synchronisation instead of two)
// scale_dev != nullptr (pipelined Lanczos step): alpha.x itself is read from the device as well
__global__ __launch_bounds__(kBlock) void k_axpy_norm(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, int64_t n,
                                                      double *partials, double *yr, int *flag, const double *scale_dev)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    bool bad = false;
    if (scale_dev != nullptr) alpha.x = scale_dev[0];
    if (alpha_dev != nullptr) alpha = d2{alpha.x * alpha_dev[0], 0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        d2 v = y[i] + cmul(alpha, x[i]);
        y[i] = v;
        if (yr != nullptr) {
            yr[i] = v.x;
            bad |= (v.y != 0.0);
        }
        acc[0] += v.x * v.x + v.y * v.y;
    }
    if (bad) *flag = 1;
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_axpy_norm(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, int64_t n, double *partials, double *yr, int *flag,
                     hipStream_t s, const double *scale_dev)
{
    hipLaunchKernelGGL(k_axpy_norm, dim3(blas_grid(n)), dim3(kBlock), 0, s, alpha, alpha_dev, x, y, n, partials, yr, flag, scale_dev);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// The same update for an operator with a Kronecker split (band 8): y is the next SpMV's x in every driver, and the far pass
// gathers from its TILED copy (KronTile) -- written here, by the pass that produces y, instead of by a k_kron_tile launch
// in front of the SpMV (one read of y and one launch less per step).  Work item = 32 major indices x 8 bands through LDS as in
// k_kron_tile8: 1 KB runs of x / y in, 1 KB runs of y and 4 KB runs of the tiled copy out; the narrow last band (S % 8 != 0)
// element-wise.  MODE 0: y += alpha x (alpha_dev as in k_axpy_norm), partial |y|^2.  MODE 1: y = x + alpha.x * y (k_xpby), no sum.
// Static assignment of the items to workgroups: the partial sums are run-to-run reproducible.
// item = TU major indices x TB bands: reads runs of TB * 128 bytes of x and y (TB * 8 minor indices of one major index), writes
// the updated y in natural order (the same runs) AND through LDS in tiled order (runs of TU * 128 bytes: TU major indices of one
// band).  TU * TB = 256 keeps the tile at 33 KB.  QBH_TILE_TU / QBH_TILE_TB: build-time tuning (round 4: 32 x 8; round 5, measured through whole bench lines: 8 x 32 saves 0.15-0.25 ms of the 2.2 ms pass).
#ifndef QBH_TILE_TU
#define QBH_TILE_TU 8
#endif
#ifndef QBH_TILE_TB
#define QBH_TILE_TB 32
#endif
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_axpy_norm_tile8(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, d2 *yt, KronTile t,
                                                            int64_t nfb, double *partials, const double *scale_dev, int yt_real, int *flag)
{
    bool bad = false;           // yt_real (real wire of a split shard): the tiled copy as packed real parts, *flag on a non-zero imaginary part
    constexpr int TU = QBH_TILE_TU, TB = QBH_TILE_TB, RW = TB * 8, LD = RW + 1;      // RW: elements of one major index in the item
    static_assert(TU * TB == 256 && (TU & (TU - 1)) == 0 && (TB & (TB - 1)) == 0, "TU x TB = 256, powers of two");
    __shared__ d2 tilebuf[TU * LD];
    __shared__ double red[4];
    double acc[1] = {0.0};
    if (MODE == 0 && scale_dev != nullptr) alpha.x = scale_dev[0];
    if (MODE == 0 && alpha_dev != nullptr) alpha = d2{alpha.x * alpha_dev[0], 0.0};
    auto upd = [&](d2 xv, d2 yv) -> d2 {
        if (MODE == 0) {
            const d2 v = yv + cmul(alpha, xv);
            acc[0] += v.x * v.x + v.y * v.y;
            return v;
        }
        return xv + alpha.x * yv;
    };
    const int64_t tiles_u = (t.NU + TU - 1) / TU, tiles_b = (nfb + TB - 1) / TB;
    for (int64_t w = blockIdx.x; w < tiles_u * tiles_b; w += gridDim.x) {
        const int64_t tb = w / tiles_u, tu = w - tb * tiles_u;
        const int64_t u0 = tu * TU, b0 = tb * TB;
        const int nu = (int)(t.NU - u0 < TU ? t.NU - u0 : TU), nb = (int)(nfb - b0 < TB ? nfb - b0 : TB);
        d2 xv[TU * TB * 8 / kBlock], yv[TU * TB * 8 / kBlock];
#pragma unroll
        for (int i = 0; i < TU * TB * 8 / kBlock; ++i) {
            const int idx = threadIdx.x + i * kBlock, ul = idx / RW, dl = idx % RW;
            const bool in = ul < nu && dl < nb * 8;
            const int64_t r = in ? (u0 + ul) * t.S + b0 * 8 + dl : 0;
            xv[i] = __builtin_nontemporal_load(x + r);
            yv[i] = __builtin_nontemporal_load(y + r);
        }
        __syncthreads();                               // the previous item's tile has been read
#pragma unroll
        for (int i = 0; i < TU * TB * 8 / kBlock; ++i) {
            const int idx = threadIdx.x + i * kBlock, ul = idx / RW, dl = idx % RW;
            if (ul < nu && dl < nb * 8) {
                const d2 v = upd(xv[i], yv[i]);
                y[(u0 + ul) * t.S + b0 * 8 + dl] = v;
                tilebuf[ul * LD + dl] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TU * TB * 8 / kBlock; ++i) {
            const int idx = threadIdx.x + i * kBlock, bl = idx / (TU * 8), rest = idx % (TU * 8), ul = rest >> 3, j = rest & 7;
            if (bl < nb && ul < nu) {
                const d2 v = tilebuf[ul * LD + bl * 8 + j];
                const int64_t o = (b0 + bl) * 8 * t.NU + (u0 + ul) * 8 + j;
                if (yt_real) {
                    reinterpret_cast<double *>(yt)[o] = v.x;
                    bad |= v.y != 0.0;
                } else {
                    yt[o] = v;
                }
            }
        }
    }
    const int64_t d0 = nfb * 8, we = t.S - d0;             // the narrow last band
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < t.NU * we; e += (int64_t)gridDim.x * kBlock) {
        const int64_t u = e / we, r = u * t.S + d0 + (e - u * we);
        const d2 v = upd(x[r], y[r]);
        y[r] = v;
        if (yt_real) {
            reinterpret_cast<double *>(yt)[t.tile(r)] = v.x;
            bad |= v.y != 0.0;
        } else {
            yt[t.tile(r)] = v;
        }
    }
    if (bad) *flag = 1;
    if (MODE == 0) {
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
    }
}

// grid = blas_grid(n): the partial sums are reduced by the same second stage as k_axpy_norm's
int launch_axpy_norm_tile(d2 alpha, const double *alpha_dev, const d2 *x, d2 *y, d2 *yt, int64_t n, const KronTile &t, double *partials,
                          hipStream_t s, const double *scale_dev, int yt_real, int *flag)
{
    if (t.B != 8 || t.S < 8 || (yt_real && !flag)) return QBH_EINVAL;
    hipLaunchKernelGGL(k_axpy_norm_tile8<0>, dim3(blas_grid(n)), dim3(kBlock), 0, s, alpha, alpha_dev, x, y, yt, t, t.S / 8, partials, scale_dev, yt_real, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_xpby_tile(const d2 *x, double b, d2 *y, d2 *yt, int64_t n, const KronTile &t, hipStream_t s, int yt_real, int *flag)
{
    if (t.B != 8 || t.S < 8 || (yt_real && !flag)) return QBH_EINVAL;
    hipLaunchKernelGGL(k_axpy_norm_tile8<1>, dim3(blas_grid(n)), dim3(kBlock), 0, s, d2{b, 0.0}, (const double *)nullptr, x, y, yt, t, t.S / 8,
                       (double *)nullptr, (const double *)nullptr, yt_real, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_nrm2sq(const d2 *x, int64_t n, double *partials)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const d2 v = x[i];
        acc[0] += v.x * v.x + v.y * v.y;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_nrm2sq(const d2 *x, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_nrm2sq, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_scal(double a, d2 *x, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] = a * x[i];
}

// y = a * x (out of place: the exit of the pipelined Lanczos driver moves a vector into the caller's slot and normalises it in one pass)
__global__ __launch_bounds__(kBlock) void k_scal_to(double a, const d2 *x, d2 *y, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = a * x[i];
}
int launch_scal_to(double a, const d2 *x, d2 *y, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scal_to, dim3(blas_grid(n)), dim3(kBlock), 0, s, a, x, y, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_scal(double a, d2 *x, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scal, dim3(blas_grid(n)), dim3(kBlock), 0, s, a, x, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// y = x + b*y   (CG direction update p = r + beta^2 p, src/lanczos.cc:327-328)
// all-real Lanczos: y += alpha * x on vectors stored as doubles, partial |y|^2 (alpha_dev as in k_axpy_norm)
// yt != nullptr: the result also in the tiled order t (16 consecutive minor indices = one aligned 128-byte line): the next SpMV of
// a coded Kronecker split (qbh_kronc.hip) gathers its far part from it and needs no k_kron_tile_re
__global__ __launch_bounds__(kBlock) void k_axpy_norm_re(double alpha, const double *alpha_dev, const double *x, double *y,
                                                         int64_t n, double *partials, double *yt, KronTile t)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    if (alpha_dev != nullptr) alpha *= alpha_dev[0];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double v = y[i] + alpha * x[i];
        y[i] = v;
        if (yt != nullptr) {
            if (n < 2147483647LL && t.B == 16) {          // 32-bit index arithmetic (the 64-bit divisions of tile() cost more than the store)
                const uint32_t S32 = (uint32_t)t.S, u = (uint32_t)i / S32, d = (uint32_t)i - u * S32, b = d >> 4;
                const uint32_t wB = S32 - (b << 4) < 16u ? S32 - (b << 4) : 16u;
                yt[(int64_t)b * 16 * t.NU + (int64_t)(u * wB + (d & 15u))] = v;
            } else {
                yt[t.tile(i)] = v;
            }
        }
        acc[0] += v * v;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_axpy_norm_re(double alpha, const double *alpha_dev, const double *x, double *y, int64_t n, double *partials, hipStream_t s, double *yt,
                        const KronTile &t)
{
    hipLaunchKernelGGL(k_axpy_norm_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, alpha, alpha_dev, x, y, n, partials, yt, t);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// yr != nullptr: also the packed real parts of the result (it is the next SpMV's gather source in the real fast path)
__global__ __launch_bounds__(kBlock) void k_xpby(const d2 *x, double b, d2 *y, int64_t n, double *yr, int *flag)
{
    bool bad = false;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const d2 v = x[i] + b * y[i];
        y[i] = v;
        if (yr != nullptr) {
            yr[i] = v.x;
            bad |= (v.y != 0.0);
        }
    }
    if (bad) *flag = 1;
}

int launch_xpby(const d2 *x, double b, d2 *y, int64_t n, double *yr, int *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_xpby, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, b, y, n, yr, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// v += alpha*p ; r -= alpha*pp ; partial |r|^2   (src/lanczos.cc:324-326 in one pass)
// delta_dev != nullptr: alpha = accu2 / delta with delta = <p, pp> left on the device by the SpMV that produced pp (no host
// round-trip between the SpMV and this pass); the arithmetic of the host expression, operation by operation (no contraction)
__global__ __launch_bounds__(kBlock) void k_cg_update(d2 alpha, const d2 *p, const d2 *pp, d2 *v, d2 *r,
                                                      int64_t n, double *partials, const double *delta_dev, double accu2)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    if (delta_dev != nullptr) {
        const double re = delta_dev[0], im = delta_dev[1];
        const double den = __dadd_rn(__dmul_rn(re, re), __dmul_rn(im, im));
        alpha = d2{__ddiv_rn(__dmul_rn(accu2, re), den), -__ddiv_rn(__dmul_rn(accu2, im), den)};
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        v[i] = v[i] + cmul(alpha, p[i]);
        const d2 rr = r[i] - cmul(alpha, pp[i]);
        r[i] = rr;
        acc[0] += rr.x * rr.x + rr.y * rr.y;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_cg_update(d2 alpha, const d2 *p, const d2 *pp, d2 *v, d2 *r, int64_t n, double *partials,
                     hipStream_t s, const double *delta_dev, double accu2)
{
    hipLaunchKernelGGL(k_cg_update, dim3(blas_grid(n)), dim3(kBlock), 0, s, alpha, p, pp, v, r, n, partials, delta_dev, accu2);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// ---- the same passes on vectors stored as doubles (all-real CG, see qbh_eigenvec_cg_dev) ----
__global__ __launch_bounds__(kBlock) void k_cg_update_re(double alpha, const double *p, const double *pp, double *v, double *r,
                                                         int64_t n, double *partials)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        v[i] = v[i] + alpha * p[i];
        const double rr = r[i] - alpha * pp[i];
        r[i] = rr;
        acc[0] += rr * rr;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(kBlock) void k_xpby_re(const double *x, double b, double *y, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = x[i] + b * y[i];
}

__global__ __launch_bounds__(kBlock) void k_dot_re(const double *x, const double *y, int64_t n, double *partials)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc[0] += x[i] * y[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_dot_re(const double *x, const double *y, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_dot_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, y, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_nrm2sq_re(const double *x, int64_t n, double *partials)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc[0] += x[i] * x[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(kBlock) void k_scal_re(double a, double *x, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] *= a;
}

int launch_cg_update_re(double alpha, const double *p, const double *pp, double *v, double *r, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_cg_update_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, alpha, p, pp, v, r, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_xpby_re(const double *x, double b, double *y, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_xpby_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, b, y, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_nrm2sq_re(const double *x, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_nrm2sq_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}
int launch_scal_re(double a, double *x, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scal_re, dim3(blas_grid(n)), dim3(kBlock), 0, s, a, x, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// ------------------------------------------------------ start vector -----------
// vec_randomize (src/miscellaneous.cc:371-386): std::minstd_rand0 is the Lehmer
// generator s <- 16807 s mod (2^31-1); element j takes draw j+1.  Each lane jumps ahead
// with a modular power and then walks a short run, so the device vector is bit-identical
// to the host one before normalisation.
constexpr int kRandRun = 16;

__device__ __forceinline__ uint64_t lehmer_pow(uint64_t e)
{
    const uint64_t M = 2147483647ULL;
    uint64_t base = 16807ULL, r = 1ULL;
    while (e) {
        if (e & 1ULL) r = (r * base) % M;
        base = (base * base) % M;
        e >>= 1;
    }
    return r;
}

// xr != nullptr: the vector is stored as packed doubles (qbh_vec_randomize_real), same stream of numbers
// major_inv != nullptr (qbh_opts.major_partition): local element j = (local major j / S, minor j % S) is drawn at position
// major_inv[j / S] * S + j % S of the stream -- the same physical vector whatever order the major indices are held in
__global__ __launch_bounds__(kBlock) void k_randomize(d2 *x, double *xr, int64_t n, int64_t global_offset, uint32_t seed,
                                                      double *partials, const int32_t *major_inv, int64_t S)
{
#pragma clang fp contract(off)
    __shared__ double red[4];
    const uint64_t M = 2147483647ULL;
    double acc[1] = {0.0};
    const int64_t nruns = (n + kRandRun - 1) / kRandRun;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    uint64_t s0 = (uint64_t)seed % M;
    if (s0 == 0) s0 = 1;
    for (int64_t run = (int64_t)blockIdx.x * kBlock + threadIdx.x; run < nruns; run += stride) {
        const int64_t j0 = run * kRandRun;
        auto pos = [&](int64_t j) -> uint64_t { return major_inv ? (uint64_t)((int64_t)major_inv[j / S] * S + j % S) : (uint64_t)(global_offset + j); };
        uint64_t state = (s0 * lehmer_pow(pos(j0))) % M;   // state before the draw of element j0
        const int64_t j1 = (j0 + kRandRun < n) ? j0 + kRandRun : n;
        for (int64_t j = j0; j < j1; ++j) {
            if (major_inv != nullptr && j > j0 && j % S == 0) state = (s0 * lehmer_pow(pos(j))) % M;      // a new major index: another stretch of the stream
            state = (state * 16807ULL) % M;
            const double t = (double)state * (1.0 / 2147483647.0);
            d2 v;
            v.x = t - 0.5;
            v.y = 0.0;
            if (xr != nullptr) xr[j] = v.x;
            else               x[j] = v;
            acc[0] += v.x * v.x;
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_randomize(d2 *x, double *xr, int64_t n, int64_t global_offset, uint32_t seed, double *partials, hipStream_t s, const int32_t *major_inv,
                     int64_t S)
{
    const int64_t nruns = (n + kRandRun - 1) / kRandRun;
    hipLaunchKernelGGL(k_randomize, dim3(blas_grid(nruns)), dim3(kBlock), 0, s, x, xr, n, global_offset, seed,
                       partials, major_inv, S);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

__global__ __launch_bounds__(kBlock) void k_fill_const(d2 *x, int64_t n, double re)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    d2 v = {re, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] = v;
}

int launch_fill_const(d2 *x, int64_t n, double re, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill_const, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, n, re);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}


// --------------------------------------------- Krylov-basis kernels (qbh_iram) --
// h_i = <V_i, w> for NV basis vectors in ONE pass over w (full re-orthogonalisation of the
// thick-restart Lanczos basis); partials[(block*NV + i)*2 + {re,im}].
template <int NV>
__global__ __launch_bounds__(kBlock) void k_multi_dot(const d2 *V, int64_t ldv, const d2 *w, int64_t n, int nv,
                                                      double *partials)
{
    __shared__ double red[2 * NV * 4];
    double acc[2 * NV];
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) acc[i] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        const d2 wv = w[e];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i < nv) {
                const d2 vi = V[(size_t)i * ldv + e];
                acc[2 * i] += vi.x * wv.x + vi.y * wv.y;
                acc[2 * i + 1] += vi.x * wv.y - vi.y * wv.x;
            }
        }
    }
    block_sum<2 * NV>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) partials[(size_t)blockIdx.x * 2 * NV + i] = acc[i];
    }
}

int launch_multi_dot8(const d2 *V, int64_t ldv, const d2 *w, int64_t n, int nv, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL((k_multi_dot<8>), dim3(blas_grid(n)), dim3(kBlock), 0, s, V, ldv, w, n, nv, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// w -= sum_i c_i V_i  (c complex), one pass; optionally the partial sums of |w|^2 of the result
__global__ __launch_bounds__(kBlock) void k_multi_axpy(const d2 *V, int64_t ldv, Coef8 c, int nv, d2 *w, int64_t n,
                                                       double *partials)
{
    __shared__ double red[4];
    double nrm[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        d2 acc = w[e];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nv) {
                const d2 ci = {c.v[2 * i], c.v[2 * i + 1]};
                acc -= cmul(ci, V[(size_t)i * ldv + e]);
            }
        }
        w[e] = acc;
        nrm[0] += acc.x * acc.x + acc.y * acc.y;
    }
    if (partials != nullptr) {
        block_sum<1>(nrm, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = nrm[0];
    }
}

int launch_multi_axpy8(const d2 *V, int64_t ldv, const Coef8 &c, int nv, d2 *w, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_multi_axpy, dim3(blas_grid(n)), dim3(kBlock), 0, s, V, ldv, c, nv, w, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// Restart rotation, in place: V[:, c] <- sum_i S[i + c*m] V[:, i]  for c < keep (S real, m <= 32).
// S is real, so the rotation acts on the real and imaginary parts independently: the basis is treated as vectors of
// doubles (2n per complex vector; n for the packed-real basis).  MMAX = 32 or 64 basis vectors are held in registers.
template <int MMAX>
__global__ __launch_bounds__(kBlock) void k_basis_rotate(double *V, int64_t ldv, int64_t n, int m, int keep, const double *S)
{
    __shared__ double Ss[MMAX * MMAX];
    for (int i = threadIdx.x; i < m * keep; i += kBlock) Ss[i] = S[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        double x[MMAX];
#pragma unroll
        for (int i = 0; i < MMAX; ++i)
            if (i < m) x[i] = V[(size_t)i * ldv + e];
        for (int c = 0; c < keep; ++c) {
            double y = 0.0;
#pragma unroll
            for (int i = 0; i < MMAX; ++i)
                if (i < m) y += Ss[i + c * m] * x[i];
            V[(size_t)c * ldv + e] = y;
        }
    }
}

// V (complex view: leading dimension ldv and length n in complex elements) <- V S[:, 0..keep), m <= 64
int launch_basis_rotate(d2 *V, int64_t ldv, int64_t n, int m, int keep, const double *d_S, hipStream_t s)
{
    double *Vd = reinterpret_cast<double *>(V);
    if (m <= 32) hipLaunchKernelGGL(k_basis_rotate<32>, dim3(blas_grid(2 * n)), dim3(kBlock), 0, s, Vd, 2 * ldv, 2 * n, m, keep, d_S);
    else         hipLaunchKernelGGL(k_basis_rotate<64>, dim3(blas_grid(2 * n)), dim3(kBlock), 0, s, Vd, 2 * ldv, 2 * n, m, keep, d_S);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}


// ------------------------------------------------------- shard column split -----
// A row shard is split once, at creation, into the entries whose column lies inside the shard's own
// row range [lo, hi) and the rest.  The first part needs only the locally owned block of x, so it can
// run while the all-gather of x is still in flight; the second part runs after it and accumulates.
__global__ __launch_bounds__(kBlock) void k_split_count(const int64_t *ia, const int32_t *ja, int64_t nrows, int32_t lo,
                                                        int32_t hi, int32_t *cnt0)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < nrows; r += stride) {
        int c = 0;
        for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
            const int32_t col = ja[p];
            c += (col >= lo && col < hi) ? 1 : 0;
        }
        cnt0[r] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k_split_fill(const int64_t *ia, const int32_t *ja, const d2 *val, const uint8_t *code,
                                                       int64_t nrows, int32_t lo, int32_t hi, const int64_t *ia0, int32_t *ja0,
                                                       d2 *val0, uint8_t *code0, int64_t *ia1, int32_t *ja1, d2 *val1,
                                                       uint8_t *code1, int code_w)
{
    const uint16_t *wcode = reinterpret_cast<const uint16_t *>(code);
    uint16_t *wcode0 = reinterpret_cast<uint16_t *>(code0), *wcode1 = reinterpret_cast<uint16_t *>(code1);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= nrows; r += stride) {
        ia1[r] = ia[r] - ia0[r];
        if (r == nrows) break;
        int64_t q0 = ia0[r], q1 = ia[r] - ia0[r];
        for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
            const int32_t col = ja[p];
            if (col >= lo && col < hi) {
                ja0[q0] = col;
                if (code && code_w == 2) wcode0[q0] = wcode[p];
                else if (code) code0[q0] = code[p];
                else      val0[q0] = val[p];
                ++q0;
            } else {
                ja1[q1] = col;
                if (code && code_w == 2) wcode1[q1] = wcode[p];
                else if (code) code1[q1] = code[p];
                else      val1[q1] = val[p];
                ++q1;
            }
        }
    }
}

int launch_split_count(const int64_t *ia, const int32_t *ja, int64_t nrows, int32_t lo, int32_t hi, int32_t *cnt0, hipStream_t s)
{
    hipLaunchKernelGGL(k_split_count, dim3(blas_grid(nrows)), dim3(kBlock), 0, s, ia, ja, nrows, lo, hi, cnt0);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_split_fill(const int64_t *ia, const int32_t *ja, const d2 *val, const uint8_t *code, int64_t nrows, int32_t lo,
                      int32_t hi, const int64_t *ia0, int32_t *ja0, d2 *val0, uint8_t *code0, int64_t *ia1, int32_t *ja1,
                      d2 *val1, uint8_t *code1, int code_w, hipStream_t s)
{
    hipLaunchKernelGGL(k_split_fill, dim3(blas_grid(nrows + 1)), dim3(kBlock), 0, s, ia, ja, val, code, nrows, lo, hi, ia0, ja0,
                       val0, code0, ia1, ja1, val1, code1, code_w);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

// exclusive scan int32 counts -> int64 offsets (three small kernels; one-time setup work)
constexpr int kScanChunk = 2048;

__global__ __launch_bounds__(256) void k_scan_chunksum(const int32_t *cnt, int64_t n, int64_t *chunk_sum)
{
    __shared__ double red_dummy;   // keep LDS layout trivial
    (void)red_dummy;
    __shared__ long long sm[4];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk;
    long long s = 0;
    for (int i = threadIdx.x; i < kScanChunk; i += 256)
        if (base + i < n) s += cnt[base + i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void k_scan_chunks_serial(int64_t *chunk_sum, int64_t nchunks)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t run = 0;
        for (int64_t i = 0; i < nchunks; ++i) {
            const int64_t t = chunk_sum[i];
            chunk_sum[i] = run;
            run += t;
        }
        chunk_sum[nchunks] = run;
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(const int32_t *cnt, int64_t n, const int64_t *chunk_off,
                                                    int64_t *ia)
{
    // one workgroup per chunk; thread t scans 8 consecutive elements, wave/LDS scan of the sums
    __shared__ long long wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 8;
    long long loc[8], tot = 0;
    for (int i = 0; i < 8; ++i) {
        loc[i] = tot;
        if (base + i < n) tot += cnt[base + i];
    }
    long long incl = tot;                                // inclusive scan across the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const long long excl = chunk_off[blockIdx.x] + woff + incl - tot;
    for (int i = 0; i < 8; ++i)
        if (base + i < n) ia[base + i] = excl + loc[i];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) ia[n] = chunk_off[gridDim.x];
}

int exclusive_scan(const int32_t *d_cnt, int64_t n, int64_t *d_ia, hipStream_t s)
{
    if (n <= 0) {                                // an empty part (a cut sector's class without rows of that kind): ia = {0}; a launch of no
        QBH_HIP(hipMemsetAsync(d_ia, 0, sizeof(int64_t), s));       // workgroups is an error that would stay behind as the "last error"
        QBH_HIP(hipStreamSynchronize(s));
        return QBH_OK;
    }
    const int64_t nchunks = (n + kScanChunk - 1) / kScanChunk;
    int64_t *d_chunk = nullptr;
    QBH_HIP(qbh::dev_alloc(&d_chunk, (size_t)(nchunks + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(k_scan_chunksum, dim3((unsigned)nchunks), dim3(256), 0, s, d_cnt, n, d_chunk);
    hipLaunchKernelGGL(k_scan_chunks_serial, dim3(1), dim3(64), 0, s, d_chunk, nchunks);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nchunks), dim3(256), 0, s, d_cnt, n, d_chunk, d_ia);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(d_chunk);
    if (e != hipSuccess) {
        set_error("scan failed: %s", hipGetErrorString(e));
        return QBH_EHIP;
    }
    return QBH_OK;
}



// ------------------------------------------------- real wire format -------------
// For a real Hamiltonian and real start vector every Lanczos / CG vector has an exactly zero imaginary
// part, so the all-gather of x can carry 8 instead of 16 bytes per element (lossless).  pack also raises
// *flag if it ever meets a non-zero imaginary part (checked by the drivers: never silently wrong).
__global__ __launch_bounds__(kBlock) void k_pack_real(const d2 *x, double *out, int64_t n, int *flag)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const d2 v = x[i];
        out[i] = v.x;
        bad |= (v.y != 0.0);
    }
    if (bad) *flag = 1;
}

__global__ __launch_bounds__(kBlock) void k_unpack_real(const double *in, d2 *out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = d2{in[i], 0.0};
}

// partial sums of |Im x|^2 (entry check of the drivers)
__global__ __launch_bounds__(kBlock) void k_imag_norm(const d2 *x, int64_t n, double *partials)
{
    __shared__ double red[4];
    double acc[1] = {0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc[0] += x[i].y * x[i].y;
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

int launch_pack_real(const d2 *x, double *out, int64_t n, int *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_real, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, out, n, flag);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_unpack_real(const double *in, d2 *out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_unpack_real, dim3(blas_grid(n)), dim3(kBlock), 0, s, in, out, n);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}

int launch_imag_norm(const d2 *x, int64_t n, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_imag_norm, dim3(blas_grid(n)), dim3(kBlock), 0, s, x, n, partials);
    QBH_HIP(hipGetLastError());
    return QBH_OK;
}


// -------------------------------------------- matrix-free two-species operator --
// y <- alpha*(H x) + beta*y + gamma*x_local with H = T_up (x) 1 + 1 (x) T_dn + U*D applied from the hop tables
// (ELL layout, coalesced).  One lane per row (u, d); 256 consecutive rows per workgroup pass.  The up-species
// hops of consecutive rows read consecutive x elements (full-line coalesced), the down-species hops stay inside
// the 16*N_dn-byte window of the row's own u.  Same fused epilogue and partial sums as the CSR kernels.
template <bool REALX>
__global__ __launch_bounds__(kBlock) void k_mf_hubbard(MfArgs a)
{
    __shared__ double red[12];
    __shared__ double amp_s[16];
    double acc[3] = {0.0, 0.0, 0.0};
    const MfHubbard &t = a.t;
    if (threadIdx.x < 16) amp_s[threadIdx.x] = t.amp[threadIdx.x];
    __syncthreads();
    const int64_t n_chunks = (a.nrows + kBlock - 1) / kBlock;
    const bool need_x = a.gamma != 0.0 || a.partials != nullptr;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t lrow = chunk * kBlock + threadIdx.x;
        if (lrow < a.nrows) {
            const int64_t grow = a.row_begin + lrow;
            const int64_t u = grow / t.Nd, d = grow - u * t.Nd;
            d2 sum = {0.0, 0.0};
            // diagonal: U * number of doubly occupied sites
            const double diag = t.U * (double)__popc(t.cfg_u[u] & t.cfg_d[d]);
            if (REALX) sum.x = diag * a.xr[grow];
            else       sum = diag * a.xg[grow];
            // The tables are padded to a multiple of 8 hops with (target = the configuration itself, amplitude 0),
            // so each group of 8 table reads and 8 gathers is issued without a branch (8 loads in flight per lane).
            // up-species hops: x[u' * Nd + d], consecutive lanes -> consecutive addresses
            for (int k0 = 0; k0 < t.wu; k0 += 8) {
                int64_t c[8];
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c[j] = (int64_t)t.tgt_u[(size_t)(k0 + j) * t.Nu + u] * t.Nd + d;
                    v[j] = amp_s[t.val_u[(size_t)(k0 + j) * t.Nu + u]];
                }
                if (REALX) {
                    double xr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xr[j] = a.xr[c[j]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum.x += v[j] * xr[j];
                } else {
                    d2 xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = a.xg[c[j]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += v[j] * xv[j];
                }
            }
            // down-species hops: x[u * Nd + d'] inside the row's own window
            const int64_t base = u * t.Nd;
            for (int k0 = 0; k0 < t.wd; k0 += 8) {
                int64_t c[8];
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c[j] = base + t.tgt_d[(size_t)(k0 + j) * t.Nd + d];
                    v[j] = amp_s[t.val_d[(size_t)(k0 + j) * t.Nd + d]];
                }
                if (REALX) {
                    double xr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xr[j] = a.xr[c[j]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum.x += v[j] * xr[j];
                } else {
                    d2 xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = a.xg[c[j]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += v[j] * xv[j];
                }
            }
            d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
            if (a.y_re != nullptr) {                  // all-real operation (REALX): y and x_local as doubles
                if (a.beta != 0.0) yo.x = a.y_re[lrow];
                if (need_x) xi.x = a.xr[grow];
            } else {
                if (a.beta != 0.0) yo = a.y[lrow];
                if (need_x) xi = a.xl[lrow];
            }
            const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
            if (a.y_re != nullptr) a.y_re[lrow] = yn.x;
            else                   a.y[lrow] = yn;
            acc[0] += xi.x * yn.x + xi.y * yn.y;
            acc[1] += xi.x * yn.y - xi.y * yn.x;
            acc[2] += yn.x * yn.x + yn.y * yn.y;
        }
    }
    if (a.partials != nullptr) {
        block_sum<3>(acc, red);
        if (threadIdx.x == 0) {
            a.partials[(size_t)blockIdx.x * 3 + 0] = acc[0];
            a.partials[(size_t)blockIdx.x * 3 + 1] = acc[1];
            a.partials[(size_t)blockIdx.x * 3 + 2] = acc[2];
        }
    }
}

// Row-staged variant for real vectors: X is the N_up x N_dn matrix x[u * N_dn + d].  One 1024-lane workgroup owns one
// up-configuration u at a time and keeps the whole row X[u][:] (8 * N_dn bytes) in LDS, so
//   * the down-species hops  Y[u][d] += sum_k a_k X[u][d'_k]  gather from LDS instead of through the texture path
//     (scattered 8-byte gathers are what bounds the lane-per-row kernel: one cache line per lane per cycle),
//   * the up-species hops  Y[u][:] += sum_j a_j X[u'_j][:]  are row AXPYs: fully coalesced 512-byte wave loads, the
//     ~17 neighbour rows shared through L2 / Infinity Cache with the workgroups working on nearby u,
//   * x_local of the fused epilogue comes from the staged row for free.
// The down-hop table is read as packed {target:24 | amplitude code:8} words, four hops per 16-byte load.
constexpr int kMfRowBlock = 1024;
constexpr int kMfMaxUp = 64;

// WINDOWED = false: the whole row X[u][:] is staged (8 * N_dn <= LDS).  WINDOWED = true (longer rows): a work item is
// (u, chunk of `chunk` consecutive d); LDS holds the window of `wcap` elements of the row centred on the chunk.  Hops
// in the colex order mostly move a configuration's rank a little (4x5 lattice, 6 particles: 84 % of the hops stay
// within +-8192), so most down-hop gathers still come from LDS; the rest read the row through L2.
template <bool WINDOWED>
__global__ __launch_bounds__(kMfRowBlock) void k_mf_hubbard_row(MfArgs a, int chunk, int wcap)
{
    extern __shared__ double xs[];                 // [Nd] or [wcap]
    __shared__ double amp_s[16];
    __shared__ double dd_s[256];                   // a.dcode: the value dictionary's real parts (the diagonal is dd_s[dcode[row]])
    __shared__ long long up_off[kMfMaxUp + 8];
    __shared__ double up_amp[kMfMaxUp + 8];
    __shared__ int up_n;
    __shared__ double red[3 * (kMfRowBlock / 64)];
    const MfHubbard &t = a.t;
    const int tid = threadIdx.x;
    const int64_t Nd = t.Nd;
    double acc[3] = {0.0, 0.0, 0.0};
    if (tid < 16) amp_s[tid] = t.amp[tid];
    if (a.dcode != nullptr && tid < 256) dd_s[tid] = a.ddict[tid];
    const int64_t u_first = a.row_begin / Nd, u_last = (a.row_begin + a.nrows - 1) / Nd;
    const bool need_y = a.beta != 0.0;
    const uint4 *pk = reinterpret_cast<const uint4 *>(t.pk_d);
    const int nk4 = t.wd / 4;
    const int64_t n_chunks = WINDOWED ? (Nd + chunk - 1) / chunk : 1;
    const int64_t n_items = (u_last - u_first + 1) * n_chunks;
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t u = u_first + item / n_chunks;
        const int64_t c_lo = WINDOWED ? (item % n_chunks) * chunk : 0;
        const int64_t c_hi = WINDOWED ? (c_lo + chunk < Nd ? c_lo + chunk : Nd) : Nd;
        // window [w_lo, w_hi) of the row kept in LDS
        int64_t w_lo = 0, w_hi = Nd;
        if (WINDOWED) {
            w_lo = c_lo - (wcap - (c_hi - c_lo)) / 2;
            if (w_lo < 0) w_lo = 0;
            w_hi = w_lo + wcap;
            if (w_hi > Nd) {
                w_hi = Nd;
                w_lo = w_hi - wcap > 0 ? w_hi - wcap : 0;
            }
        }
        const double *xrow = a.xr + u * Nd;
        for (int64_t d = w_lo + tid; d < w_hi; d += kMfRowBlock) xs[d - w_lo] = xrow[d];
        if (tid < 64) {
            // the up-neighbours of u with a non-zero amplitude, compacted by one wavefront and padded to a group of 8
            // with (u itself, amplitude 0) so that the row loop below is branch-free
            const bool in = tid < t.wu;
            const int code = in ? t.val_u[(size_t)tid * t.Nu + u] : 0;
            const bool live = in && code != 0;
            const unsigned long long mask = __ballot(live);
            const int pos = __popcll(mask & ((1ULL << tid) - 1ULL));
            const int n = __popcll(mask);
            if (live) {
                up_off[pos] = (long long)t.tgt_u[(size_t)tid * t.Nu + u] * Nd;
                up_amp[pos] = amp_s[code];
            }
            const int npad = (n + 7) & ~7;
            if (tid >= n && tid < npad) {
                up_off[tid] = (long long)u * Nd;
                up_amp[tid] = 0.0;
            }
            if (tid == 0) up_n = npad;
        }
        __syncthreads();
        int64_t d_lo = a.row_begin > u * Nd ? a.row_begin - u * Nd : 0;
        int64_t d_hi = (a.row_begin + a.nrows - u * Nd) < Nd ? (a.row_begin + a.nrows - u * Nd) : Nd;
        if (d_lo < c_lo) d_lo = c_lo;
        if (d_hi > c_hi) d_hi = c_hi;
        const uint32_t cu = a.dcode != nullptr ? 0u : t.cfg_u[u];
        const int nu = up_n;
        for (int64_t d = d_lo + tid; d < d_hi; d += kMfRowBlock) {
            const double xd = xs[d - w_lo];
            double sum = (a.dcode != nullptr ? dd_s[a.dcode[u * Nd + d]] : t.U * (double)__popc(cu & t.cfg_d[d])) * xd;
            // up-species hops first (global, longest latency): 8 coalesced row loads in flight
            for (int j0 = 0; j0 < nu; j0 += 8) {
                double xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = a.xr[up_off[j0 + j] + d];
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += up_amp[j0 + j] * xv[j];
            }
            // down-species hops from the staged row
            for (int k4 = 0; k4 < nk4; k4 += 2) {
                const uint4 e0 = pk[(size_t)k4 * Nd + d], e1 = pk[(size_t)(k4 + 1) * Nd + d];
                const uint32_t w[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                double xv[8];
                if (WINDOWED) {
                    bool miss[8];
                    bool any_miss = false;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int64_t tg = (int64_t)(w[j] & 0xFFFFFFu);
                        miss[j] = tg < w_lo || tg >= w_hi;
                        any_miss = any_miss || miss[j];
                        xv[j] = xs[miss[j] ? 0 : tg - w_lo];
                    }
                    if (any_miss) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (miss[j]) xv[j] = xrow[w[j] & 0xFFFFFFu];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = xs[w[j] & 0xFFFFFFu];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += amp_s[w[j] >> 24] * xv[j];
            }
            const int64_t lrow = u * Nd + d - a.row_begin;
            d2 yo = {0.0, 0.0};
            if (need_y) {
                if (a.y_re != nullptr) yo.x = a.y_re[lrow];
                else                   yo = a.y[lrow];
            }
            d2 yn;
            yn.x = a.alpha * sum + a.beta * yo.x + a.gamma * xd;
            yn.y = a.beta * yo.y;
            if (a.y_re != nullptr) a.y_re[lrow] = yn.x;
            else                   a.y[lrow] = yn;
            acc[0] += xd * yn.x;
            acc[1] += xd * yn.y;
            acc[2] += yn.x * yn.x + yn.y * yn.y;
        }
        __syncthreads();                           // the row and the neighbour list are rewritten next
    }
    if (a.partials != nullptr) {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c * (kMfRowBlock / 64) + wave] = acc[c];
        }
        __syncthreads();
        if (tid == 0) {
            for (int c = 0; c < 3; ++c) {
                double v = 0.0;
                for (int w2 = 0; w2 < kMfRowBlock / 64; ++w2) v += red[c * (kMfRowBlock / 64) + w2];
                a.partials[(size_t)blockIdx.x * 3 + c] = v;
            }
        }
    }
}

// -------------------------------------------- matrix-free Heisenberg operator --
// One lane per row.  LDS holds the binomials (unranking), the chunk tables (re-ranking a flipped pattern costs one
// lookup per 6 bits) and the bond list; the only global traffic is the x gather, y and the epilogue operands.
constexpr int kMfHeisBlock = 512;

// NCH > 0: the number of chunks as a compile-time constant (re-ranking loop fully unrolled, 32-bit index arithmetic)
template <bool REALX, int NCH>
__global__ __launch_bounds__(kMfHeisBlock) void k_mf_heis(MfHeisArgs a)
{
    extern __shared__ unsigned long long lds_u64[];
    __shared__ double red[3 * (kMfHeisBlock / 64)];
    const MfHeis &t = a.t;
    const int nk = t.n_dn + 1;
    unsigned long long *binom = lds_u64;                                  // [(n_sites+1) * nk]
    unsigned long long *chunk = binom + (size_t)(t.n_sites + 1) * nk;     // [n_chunks * nk * 64]
    unsigned long long *mask = chunk + (size_t)t.n_chunks * nk * 64;      // [n_bonds]
    double *offd = reinterpret_cast<double *>(mask + t.n_bonds);          // [n_bonds]
    double *diag = offd + t.n_bonds;                                      // [n_bonds]
    const int tid = threadIdx.x;
    for (int i = tid; i < (t.n_sites + 1) * nk; i += kMfHeisBlock) binom[i] = t.binom[i];
    for (int i = tid; i < t.n_chunks * nk * 64; i += kMfHeisBlock) chunk[i] = t.chunk[i];
    for (int i = tid; i < t.n_bonds; i += kMfHeisBlock) {
        mask[i] = t.mask[i];
        offd[i] = t.offd[i];
        diag[i] = t.diag[i];
    }
    __syncthreads();
    double acc[3] = {0.0, 0.0, 0.0};
    const bool uniform = t.uniform != 0;
    const double offd0 = t.offd0, diag0 = t.diag0;
    const int64_t stride = (int64_t)gridDim.x * kMfHeisBlock;
    for (int64_t lrow = (int64_t)blockIdx.x * kMfHeisBlock + tid; lrow < a.nrows; lrow += stride) {
        const int64_t grow = a.row_begin + lrow;
        // unrank (colexicographic): largest p with C(p, k) <= r, for k = n_dn .. 1
        unsigned long long s = 0, r = (unsigned long long)grow;
        int p = t.n_sites - 1;
        for (int k = t.n_dn; k >= 1; --k) {
            while (binom[p * nk + k] > r) --p;
            s |= 1ULL << p;
            r -= binom[p * nk + k];
            --p;
        }
        // (a per-lane walk over only the flipping bonds was measured: 7 % faster at 39 % flipping bonds, 9 % slower at
        // Sz = 0 where half of them flip -- the uniform loop stays)
        double dg = 0.0;
        int ndiff = 0;
        d2 sum = {0.0, 0.0};
        for (int b0 = 0; b0 < t.n_bonds; b0 += 8) {
            long long idx[8];
            double amp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long m = mask[b0 + j];
                const bool differ = __popcll(s & m) == 1;
                if (uniform) {
                    ndiff += differ ? 1 : 0;
                    amp[j] = differ ? offd0 : 0.0;
                } else {
                    dg += differ ? -diag[b0 + j] : diag[b0 + j];
                    amp[j] = differ ? offd[b0 + j] : 0.0;
                }
                long long q = grow;
                if (differ) {
                    const unsigned long long f = s ^ m;
                    unsigned long long rk = 0;
                    if (NCH > 0) {
                        const uint32_t flo = (uint32_t)f, fhi = (uint32_t)(f >> 30);      // chunks 0-4 | chunks 5-9
                        int below64 = 0;                                               // 64 * (particles below)
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int bits = (int)(((c < 5 ? flo >> (6 * c) : fhi >> (6 * (c - 5)))) & 63u);
                            rk += chunk[c * nk * 64 + below64 + bits];
                            below64 += __popc(bits) << 6;
                        }
                    } else {
                        int below = 0;
                        for (int c = 0; c < t.n_chunks; ++c) {
                            const int bits = (int)((f >> (6 * c)) & 63ULL);
                            rk += chunk[((size_t)c * nk + below) * 64 + bits];
                            below += __popc(bits);
                        }
                    }
                    q = (long long)rk;
                }
                idx[j] = q;
            }
            if (REALX) {
                double xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = a.xr[idx[j]];
#pragma unroll
                for (int j = 0; j < 8; ++j) sum.x += amp[j] * xv[j];
            } else {
                d2 xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = a.xg[idx[j]];
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += amp[j] * xv[j];
            }
        }
        d2 yo = {0.0, 0.0}, xi = {0.0, 0.0};
        if (a.y_re != nullptr) {
            if (a.beta != 0.0) yo.x = a.y_re[lrow];
            xi.x = a.xr[grow];
        } else {
            if (a.beta != 0.0) yo = a.y[lrow];
            if (REALX) xi.x = a.xr[grow];
            else       xi = a.xg[grow];
        }
        if (uniform) dg = diag0 * (double)(t.n_real - 2 * ndiff);
        sum += dg * xi;                                // diagonal: sum_b +-J_b/4
        const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
        if (a.y_re != nullptr) a.y_re[lrow] = yn.x;
        else                   a.y[lrow] = yn;
        acc[0] += xi.x * yn.x + xi.y * yn.y;
        acc[1] += xi.x * yn.y - xi.y * yn.x;
        acc[2] += yn.x * yn.x + yn.y * yn.y;
    }
    if (a.partials != nullptr) {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c * (kMfHeisBlock / 64) + wave] = acc[c];
        }
        __syncthreads();
        if (tid == 0) {
            for (int c = 0; c < 3; ++c) {
                double v = 0.0;
                for (int w2 = 0; w2 < kMfHeisBlock / 64; ++w2) v += red[c * (kMfHeisBlock / 64) + w2];
                a.partials[(size_t)blockIdx.x * 3 + c] = v;
            }
        }
    }
}

int launch_mf_heis(const MfHeisArgs &a, hipStream_t s, int *nparts_out)
{
    static int ncu = 0;
    if (ncu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const MfHeis &t = a.t;
    const size_t nk = (size_t)t.n_dn + 1;
    const size_t lds = ((size_t)(t.n_sites + 1) * nk + (size_t)t.n_chunks * nk * 64 + (size_t)t.n_bonds) * 8 + (size_t)t.n_bonds * 16;
    if (lds > (size_t)150 * 1024) {
        set_error("qbh_mf_heisenberg: tables (%zu bytes) do not fit LDS", lds);
        return QBH_EUNSUPP;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, ((size_t)158 * 1024) / (lds + 1024)));
    const int64_t nblk = (a.nrows + kMfHeisBlock - 1) / kMfHeisBlock;
    const int g = (int)std::min<int64_t>(nblk, (int64_t)ncu * per_cu);
#define QBH_HEIS_LAUNCH(RX, NC)                                                                                                    \
    do {                                                                                                                          \
        QBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mf_heis<RX, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                                                   \
        hipLaunchKernelGGL((k_mf_heis<RX, NC>), dim3(g), dim3(kMfHeisBlock), lds, s, a);                                           \
    } while (0)
    const bool rx = a.xr != nullptr;
    switch (t.n_chunks) {                                   // 24..36 sites get the unrolled forms
    case 4: if (rx) QBH_HEIS_LAUNCH(true, 4); else QBH_HEIS_LAUNCH(false, 4); break;
    case 5: if (rx) QBH_HEIS_LAUNCH(true, 5); else QBH_HEIS_LAUNCH(false, 5); break;
    case 6: if (rx) QBH_HEIS_LAUNCH(true, 6); else QBH_HEIS_LAUNCH(false, 6); break;
    default: if (rx) QBH_HEIS_LAUNCH(true, 0); else QBH_HEIS_LAUNCH(false, 0); break;
    }
#undef QBH_HEIS_LAUNCH
    QBH_HIP(hipGetLastError());
    if (nparts_out) *nparts_out = g;
    return QBH_OK;
}

// true when a row-staged kernel applies: real vectors, the neighbour list fits one wavefront
bool mf_row_kernel_ok(const MfArgs &a)
{
    if (a.xr == nullptr || a.t.pk_d == nullptr) return false;
    if (debug_sw().mf_row == 0) return false;
    return a.t.Nd >= 256 && a.t.Nd < (1 << 24) && a.t.wu <= kMfMaxUp && (a.t.wd % 8) == 0;
}

// *nparts_out = number of partial-sum triples written (workgroups launched)
int launch_mf_hubbard(const MfArgs &a, int grid, hipStream_t s, int *nparts_out)
{
    if (mf_row_kernel_ok(a)) {
        static int ncu = 0;
        if (ncu == 0) {
            hipDeviceProp_t prop;
            int dev = 0;
            ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
        }
        const size_t lds_cap = (size_t)150 * 1024;
        const bool windowed = (size_t)a.t.Nd * sizeof(double) > lds_cap;
        const int64_t n_u = (a.row_begin + a.nrows - 1) / a.t.Nd - a.row_begin / a.t.Nd + 1;
        if (!windowed) {
            const size_t lds = (size_t)a.t.Nd * sizeof(double);
            QBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mf_hubbard_row<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            // one workgroup per CU when the row takes most of the LDS, more when several rows fit
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, lds_cap / (lds + 2048)));
            const int g = (int)std::min<int64_t>(n_u, (int64_t)ncu * per_cu);
            hipLaunchKernelGGL(k_mf_hubbard_row<false>, dim3(g), dim3(kMfRowBlock), lds, s, a, 0, 0);
            QBH_HIP(hipGetLastError());
            if (nparts_out) *nparts_out = g;
            return QBH_OK;
        }
        int chunk = 8192, wcap = 18432;                // 144 KB window around an 8192-element chunk (measured: 2048..8192 within 5 %; bound by the up-row reads)
        if (debug_sw().mf_chunk) chunk = std::max(1024, debug_sw().mf_chunk);
        if (debug_sw().mf_window) wcap = std::max(chunk, std::min(18432, debug_sw().mf_window));
        const size_t lds = (size_t)wcap * sizeof(double);
        QBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mf_hubbard_row<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t n_items = n_u * ((a.t.Nd + chunk - 1) / chunk);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, lds_cap / (lds + 2048)));
        const int g = (int)std::min<int64_t>(n_items, (int64_t)ncu * per_cu);
        hipLaunchKernelGGL(k_mf_hubbard_row<true>, dim3(g), dim3(kMfRowBlock), lds, s, a, chunk, wcap);
        QBH_HIP(hipGetLastError());
        if (nparts_out) *nparts_out = g;
        return QBH_OK;
    }
    if (a.xr != nullptr) hipLaunchKernelGGL((k_mf_hubbard<true>), dim3(grid), dim3(kBlock), 0, s, a);
    else                 hipLaunchKernelGGL((k_mf_hubbard<false>), dim3(grid), dim3(kBlock), 0, s, a);
    QBH_HIP(hipGetLastError());
    if (nparts_out) *nparts_out = grid;
    return QBH_OK;
}

}  // namespace qbh
