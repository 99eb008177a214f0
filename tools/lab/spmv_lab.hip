// spmv_lab.hip -- kernel laboratory for the complex128 CSR SpMV (measurement tool, not product code).
//
// Builds one benchmark operator through the C ABI (north-star format: complex128 values, int32 columns), takes the
// library's own launch as the reference result and time, and runs candidate kernels / layouts against it:
//   wave      one wavefront per block of whole rows (<= 64*U nonzeros), no workgroup barrier
//   split     the same kernel on H = H_near + H_far: H_far (entries whose column lies in another major block of
//             `stride` columns) is stored band-major over the minor index and swept first into a scratch vector
// usage: spmv_lab.bin <workload> [reps]      workload: chain26 | chain24 | c3 | h4x3
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qbh_internal.hpp"

using qbh::d2;

#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)
#define QB(call)                                                                         \
    do {                                                                                 \
        int r_ = (call);                                                                 \
        if (r_ != 0) {                                                                   \
            fprintf(stderr, "%s -> %d %s\n", #call, r_, qbh_last_error());               \
            exit(3);                                                                     \
        }                                                                                \
    } while (0)

struct Bands {
    int64_t S, NU, B;   // minor size (stride), major count, band width
    __host__ __device__ int64_t orig(int64_t f) const
    {
        const int64_t full = B * NU;
        const int64_t b = f / full;
        const int64_t wB = (S - b * B) < B ? (S - b * B) : B;
        const int64_t rem = f - b * full;
        const int64_t u = rem / wB, j = rem % wB;
        return u * S + b * B + j;
    }
    __host__ __device__ int64_t tile(int64_t r) const   // inverse of orig
    {
        const int64_t u = r / S, d = r - u * S;
        const int64_t b = d / B, j = d - b * B;
        const int64_t wB = (S - b * B) < B ? (S - b * B) : B;
        return b * B * NU + u * wB + j;
    }
};

struct WDesc {
    int64_t p0;     // first nonzero
    int32_t r0;     // first row
    int32_t nrn;    // rows << 16 | nonzeros (0xFFFF: a row longer than the tile, slow path)
};

struct WArgs {
    const int64_t *ia;
    const int32_t *ja;
    const d2 *val;
    const int32_t *wb;   // [nwb+1] first row of each wave block
    const struct WDesc *wd;   // [nwb] packed descriptors (k_wave2)
    int64_t nwb;
    const d2 *xg, *xl;
    d2 *y;
    const int32_t *perm;   // output row of row f (nullptr = f)
    const d2 *tmp;         // addend per row (nullptr = none)
    int plain;             // 1: y[perm[row]] = sum, no epilogue
    double alpha, beta, gamma;
    double *partials;
    int swizzle, chunk;      // swizzle 3 (k_wave2): dynamic, ordered per XCD -- chunks of 4 wave blocks from a counter per XCD
    unsigned long long *ctr; // [8 * 16] one counter per XCD, 128 bytes apart, zero before the launch
    int ntstore;           // plain pass stores non-temporally
    int tiled;             // tmp is indexed by the tiled row (bands)
    Bands bd;
};

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
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// unit walk: unit = 4 consecutive wave blocks (one per wavefront of the workgroup)
__device__ __forceinline__ int64_t unit_of(int64_t lb, int64_t per_xcd, int64_t chunk, int xcd, int swz)
{
    if (swz == 1) return xcd * per_xcd + lb;
    if (swz == 2) return ((lb / chunk) * 8 + xcd) * chunk + (lb % chunk);
    return lb * 8 + xcd;
}

// TAG only gives the passes distinct kernel names in profiles: 0 unsplit, 1 far pass, 2 near pass
// POL: cache policy of the matrix stream loads: -2 global_load nt (builtin) | -1 plain global_load | >= 0 buffer_load with
// that aux field (1 sc0, 2 nt, 16 sc1 and sums)
template <int POL>
__device__ __forceinline__ void stream_loads8(const int32_t *ja, const d2 *val, int64_t p0, int n, int lane, int (&c)[8], d2 (&v)[8])
{
    const int nm1 = n - 1;
    if constexpr (POL < 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            const int ii = i < n ? i : nm1;
            c[u] = POL == -2 ? ntload(ja + p0 + ii) : ja[p0 + ii];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            const int ii = i < n ? i : nm1;
            v[u] = POL == -2 ? ntload(val + p0 + ii) : val[p0 + ii];
        }
    } else {
        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(ja + p0), 0, n * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<d2 *>(val + p0), 0, n * 16, 0x00020000);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            const int ii = i < n ? i : nm1;
            c[u] = __builtin_amdgcn_raw_buffer_load_b32(rj, ii * 4, 0, POL);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + u * 64;
            const int ii = i < n ? i : nm1;
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rv, ii * 16, 0, POL);
            v[u] = __builtin_bit_cast(d2, w);
        }
    }
}

template <int U, int TPR, int TAG = 0, int POL = -2>
__global__ __launch_bounds__(256) void k_wave(WArgs a)
{
    constexpr int NW = 64 * U;
    constexpr int RP = 64 / TPR;
    __shared__ d2 prod_s[4 * NW];
    __shared__ double red[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    d2 *prod = prod_s + wv * NW;
    const int sub = lane % TPR, rloc = lane / TPR;
    double acc[3] = {0.0, 0.0, 0.0};
    const bool need_y = a.beta != 0.0 && !a.plain;
    const bool need_x = (a.gamma != 0.0 || a.partials != nullptr) && !a.plain;

    const int64_t nunits = (a.nwb + 3) >> 2;
    int64_t per_xcd = (nunits + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int64_t nslot = gridDim.x >> 3;
    const int64_t chunk = nslot * (a.chunk > 0 ? a.chunk : 1);
    if (a.swizzle == 2) per_xcd = ((per_xcd + chunk - 1) / chunk) * chunk;

    for (int64_t lb = blockIdx.x >> 3; lb < per_xcd; lb += nslot) {
        const int64_t unit = unit_of(lb, per_xcd, chunk, xcd, a.swizzle);
        const int64_t wbi = __builtin_amdgcn_readfirstlane((int)(unit * 4 + wv));
        if (wbi >= a.nwb) continue;
        const int r0 = a.wb[wbi], r1 = a.wb[wbi + 1];
        const int nr = r1 - r0;
        if (nr <= 0) continue;
        const int64_t p0 = a.ia[r0], p1 = a.ia[r1];
        const int64_t nlong = p1 - p0;
        if (nlong <= NW) {
            const int n = (int)nlong;
            // first pass row offsets + epilogue operands, requested before the streams
            int s0 = 0, e0 = 0;
            d2 yo = {0.0, 0.0}, xi = {0.0, 0.0}, tp = {0.0, 0.0};
            int orow = 0;
            if (rloc < nr) {
                s0 = (int)(a.ia[r0 + rloc] - p0);
                e0 = (int)(a.ia[r0 + rloc + 1] - p0);
                if (sub == 0) {
                    orow = a.perm ? a.perm[r0 + rloc] : r0 + rloc;
                    if (need_y) yo = a.y[orow];
                    if (need_x) xi = a.xl[orow];
                    if (a.tmp) tp = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                }
            }
            if (n > 0) {
                const int nm1 = n - 1;
                int c[U];
                d2 v[U], xv[U];
                if constexpr (U == 8) {
                    stream_loads8<POL>(a.ja, a.val, p0, n, lane, c, v);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = lane + u * 64;
                        c[u] = ntload(a.ja + p0 + (i < n ? i : nm1));
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = lane + u * 64;
                        v[u] = ntload(a.val + p0 + (i < n ? i : nm1));
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = a.xg[c[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) prod[lane + u * 64] = cmul(v[u], xv[u]);
            }
            wave_fence();
            for (int rbase = 0; rbase < nr; rbase += RP) {
                const int row = rbase + rloc;
                int s = s0, e = e0;
                if (rbase > 0) {
                    s = e = 0;
                    if (row < nr) {
                        s = (int)(a.ia[r0 + row] - p0);
                        e = (int)(a.ia[r0 + row + 1] - p0);
                    }
                }
                d2 sum = {0.0, 0.0};
                for (int q = s + sub; q < e; q += TPR) sum += prod[q];
#pragma unroll
                for (int off = TPR / 2; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (sub == 0 && row < nr) {
                    if (rbase > 0) {
                        orow = a.perm ? a.perm[r0 + row] : r0 + row;
                        yo = need_y ? a.y[orow] : d2{0.0, 0.0};
                        xi = need_x ? a.xl[orow] : d2{0.0, 0.0};
                        tp = a.tmp ? a.tmp[a.tiled ? a.bd.tile(orow) : orow] : d2{0.0, 0.0};
                    }
                    if (a.plain) {
                        if (a.ntstore) __builtin_nontemporal_store(sum, a.y + orow);
                        else           a.y[orow] = sum;
                    } else {
                        sum += tp;
                        const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
                        a.y[orow] = yn;
                        acc[0] += xi.x * yn.x + xi.y * yn.y;
                        acc[1] += xi.x * yn.y - xi.y * yn.x;
                        acc[2] += yn.x * yn.x + yn.y * yn.y;
                    }
                }
            }
            wave_fence();
        } else {
            // a row longer than the wave tile: the wavefront walks the rows one at a time
            for (int r = 0; r < nr; ++r) {
                const int64_t s = a.ia[r0 + r], e = a.ia[r0 + r + 1];
                d2 sum = {0.0, 0.0};
                for (int64_t q = s + lane; q < e; q += 64) sum += cmul(a.val[q], a.xg[a.ja[q]]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (lane == 0) {
                    const int orow = a.perm ? a.perm[r0 + r] : r0 + r;
                    if (a.plain) {
                        a.y[orow] = sum;
                    } else {
                        if (a.tmp) sum += a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                        const d2 yo = need_y ? a.y[orow] : d2{0.0, 0.0};
                        const d2 xi = need_x ? a.xl[orow] : d2{0.0, 0.0};
                        const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
                        a.y[orow] = yn;
                        acc[0] += xi.x * yn.x + xi.y * yn.y;
                        acc[1] += xi.x * yn.y - xi.y * yn.x;
                        acc[2] += yn.x * yn.x + yn.y * yn.y;
                    }
                }
            }
        }
    }
    if (a.partials != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (lane == 0)
            for (int c = 0; c < 3; ++c) red[c * 4 + wv] = acc[c];
        __syncthreads();
        if (tid == 0)
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}


// ---------------------------------------------------------------- pipelined wave kernel ----
// Same decomposition as k_wave (U = 8), software-pipelined per wavefront: the stream loads of block i+1 are issued right
// after the gathers of block i, descriptors are fetched two blocks ahead (scalar loads), so a wavefront always has one
// block's 9.6 KB of matrix stream in flight.  Every load of the steady state is unconditional (clamped addresses) so
// that the compiler's s_waitcnt counts are exact: waiting for the gathers of block i never waits for block i+1's stream.
// OPS: 0 plain store of the row sums (far pass) | 1 epilogue with y_old and x_local | 2 the same plus the far addend
// ABL (ablation, results wrong by design): 1 no gathers | 2 no gathers, no LDS / row sums (one store per wavefront) |
// 3 like 2 and the column stream is not read either
template <int TPR, int OPS, int POL, int ABL = 0, int GPOL = -1>
__global__ __launch_bounds__(256) void k_wave2(WArgs a)
{
    constexpr int NW = 512;
    constexpr int RP = 64 / TPR;
    __shared__ d2 prod_s[4 * NW];
    __shared__ double red[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    d2 *prod = prod_s + wv * NW;
    const int sub = lane % TPR, rloc = lane / TPR;
    double acc[3] = {0.0, 0.0, 0.0};

    const int64_t nunits = (a.nwb + 3) >> 2;
    int64_t per_xcd = (nunits + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int64_t nslot = gridDim.x >> 3;
    const int64_t chunk = nslot * (a.chunk > 0 ? a.chunk : 1);
    if (a.swizzle == 2) per_xcd = ((per_xcd + chunk - 1) / chunk) * chunk;

    // dynamic walk: the wavefronts of an XCD draw chunks of CH consecutive wave blocks from that XCD's counter, so the XCD's
    // eighth of the blocks is consumed IN ORDER by all its wavefronts (a static persistent walk lets workgroups drift many
    // bands apart, and the union of their windows no longer fits the L2)
    const bool dyn = a.swizzle == 3;
    constexpr int CH = 4;
    const int64_t bpx = (a.nwb + 7) >> 3;
    const int64_t xbase = (int64_t)xcd * bpx, xend = (xbase + bpx) < a.nwb ? (xbase + bpx) : a.nwb;
    int64_t ci0 = 0, ck0 = a.nwb, ck1 = a.nwb;
    auto grab = [&]() -> int64_t {
        unsigned long long c = 0;
        if (lane == 0) c = atomicAdd(a.ctr + xcd * 16, (unsigned long long)CH);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)c), hi = __builtin_amdgcn_readfirstlane((uint32_t)(c >> 32));
        return xbase + (int64_t)(((uint64_t)hi << 32) | lo);
    };
    if (dyn) {
        ck0 = grab();
        ck1 = grab();
    }
    // descriptors come through the scalar cache (explicit s_load: the compiler would use a vector load + full wait because
    // the kernel also stores); an out-of-range block reads descriptor nwb, an empty block appended by the builder
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto load_desc = [&](int64_t lb) -> v4i {
        int64_t w = a.nwb;
        if (dyn) {
            const int64_t st = (lb / CH) == ci0 ? ck0 : ck1;
            const int64_t wd_ = st + lb % CH;
            w = wd_ < xend ? wd_ : a.nwb;
        } else if (lb < per_xcd) {
            const int64_t unit = unit_of(lb, per_xcd, chunk, xcd, a.swizzle);
            w = unit * 4 + wv;
            if (w > a.nwb) w = a.nwb;
        }
        const WDesc *p = a.wd + __builtin_amdgcn_readfirstlane((int)w);
        v4i r;
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p) : "memory");
        return r;
    };
    auto desc_of = [&](v4i r) -> WDesc {
        WDesc d;
        d.p0 = (int64_t)(((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x);
        d.r0 = r.z;
        d.nrn = r.w;
        return d;
    };
    struct Ops {
        int s, e, orow;
        d2 yo, xi, tp;
    };
    // stream + first-pass operands of one block; unconditional loads (an empty block reads entry 0 / row 0)
    auto issue = [&](const WDesc &d, int (&c)[8], d2 (&v)[8], Ops &o) {
        const int n = d.nrn & 0xFFFF, nr = d.nrn >> 16;
        const int nn = (n == 0 || n == 0xFFFF) ? 1 : n;
        if (ABL != 4) stream_loads8<POL>(a.ja, a.val, d.p0, nn, lane, c, v);
        if (ABL == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = 0;
        }
        if (ABL == 4) {          // gathers kept, value stream not read: 4 B/nnz instead of 20
            const int nm1 = nn - 1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + u * 64;
                c[u] = ntload(a.ja + d.p0 + (i < nn ? i : nm1));
                v[u] = d2{1.0, 0.25};
            }
        }
        const int rr = rloc < nr ? rloc : 0;
        o.s = (int)(a.ia[d.r0 + rr] - d.p0);
        o.e = (int)(a.ia[d.r0 + rr + 1] - d.p0);
        if (rloc >= nr) o.s = o.e = 0;
        o.orow = d.r0 + rr;
        if (OPS >= 1) {
            o.yo = a.y[o.orow];
            o.xi = a.xl[o.orow];
        }
        if (OPS == 2) o.tp = a.tmp[a.tiled ? a.bd.tile(o.orow) : o.orow];
    };
    auto finish_row = [&](int orow, d2 sum, d2 yo, d2 xi, d2 tp) {
        if (OPS == 0) {
            if (a.ntstore) __builtin_nontemporal_store(sum, a.y + orow);
            else           a.y[orow] = sum;
        } else {
            if (OPS == 2) sum += tp;
            const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
            a.y[orow] = yn;
            acc[0] += xi.x * yn.x + xi.y * yn.y;
            acc[1] += xi.x * yn.y - xi.y * yn.x;
            acc[2] += yn.x * yn.x + yn.y * yn.y;
        }
    };

    int64_t lb = dyn ? 0 : (blockIdx.x >> 3);
    const int64_t step = dyn ? 1 : nslot;
    v4i r0 = load_desc(lb), r1 = load_desc(lb + step);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1) : : "memory");   // ties the uses of r0/r1 behind the wait
    WDesc d0 = desc_of(r0), d1 = desc_of(r1);
    int cA[8];
    d2 vA[8];
    Ops oA;
    issue(d0, cA, vA, oA);
    while (dyn ? (ck0 + lb % CH < xend) : (lb < per_xcd)) {
        v4i r2 = load_desc(lb + 2 * step);
        const int n = d0.nrn & 0xFFFF, nr = d0.nrn >> 16;
        d2 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ABL == 0 || ABL == 4) {
                if constexpr (GPOL < 0) {
                    xv[u] = a.xg[cA[u]];
                } else {   // gather through a buffer resource over x (< 4 GB) with an explicit cache policy
                    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<d2 *>(a.xg), 0, 0xFFFFFFF0u, 0x00020000);
                    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                    const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rx, cA[u] * 16, 0, GPOL);
                    xv[u] = __builtin_bit_cast(d2, w);
                }
            } else {
                xv[u] = d2{1.0 + cA[u], 0.5};
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // the gathers go out BEFORE the next block's stream (in-order return)
        int cB[8];
        d2 vB[8];
        Ops oB;
        issue(d1, cB, vB, oB);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL >= 2) {
            d2 t = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 8; ++u) t += cmul(vA[u], xv[u]);
            if (t.x == 1.2345 || lane == 0) a.y[d0.r0] = t;
        } else if (n != 0xFFFF) {
#pragma unroll
            for (int u = 0; u < 8; ++u) prod[lane + u * 64] = cmul(vA[u], xv[u]);
            wave_fence();
            for (int rbase = 0; rbase < nr; rbase += RP) {
                const int row = rbase + rloc;
                int s = oA.s, e = oA.e, orow = oA.orow;
                d2 yo = oA.yo, xi = oA.xi, tp = oA.tp;
                if (rbase > 0) {
                    s = e = 0;
                    if (row < nr) {
                        s = (int)(a.ia[d0.r0 + row] - d0.p0);
                        e = (int)(a.ia[d0.r0 + row + 1] - d0.p0);
                        orow = d0.r0 + row;
                        if (OPS >= 1) {
                            yo = a.y[orow];
                            xi = a.xl[orow];
                        }
                        if (OPS == 2) tp = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                    }
                }
                d2 sum = {0.0, 0.0};
                for (int q = s + sub; q < e; q += TPR) sum += prod[q];
#pragma unroll
                for (int off = TPR / 2; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (sub == 0 && row < nr) finish_row(orow, sum, yo, xi, tp);
            }
            wave_fence();
        } else {
            for (int r = 0; r < nr; ++r) {
                const int64_t s = a.ia[d0.r0 + r], e = a.ia[d0.r0 + r + 1];
                d2 sum = {0.0, 0.0};
                for (int64_t q = s + lane; q < e; q += 64) sum += cmul(a.val[q], a.xg[a.ja[q]]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                }
                if (lane == 0) {
                    const int orow = d0.r0 + r;
                    d2 yo = {0.0, 0.0}, xi = {0.0, 0.0}, tp = {0.0, 0.0};
                    if (OPS >= 1) {
                        yo = a.y[orow];
                        xi = a.xl[orow];
                    }
                    if (OPS == 2) tp = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                    finish_row(orow, sum, yo, xi, tp);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cA[u] = cB[u];
            vA[u] = vB[u];
        }
        oA = oB;
        d0 = d1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r2) : : "memory");
        d1 = desc_of(r2);
        lb += step;
        if (dyn && lb % CH == 0) {          // entering the next chunk: draw the one after it (used two blocks from now at the earliest)
            ci0 = lb / CH;
            ck0 = ck1;
            ck1 = grab();
        }
    }
    if (a.partials != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        if (lane == 0)
            for (int c = 0; c < 3; ++c) red[c * 4 + wv] = acc[c];
        __syncthreads();
        if (tid == 0)
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}

__global__ void k_build_wd(const int64_t *ia, const int32_t *wb, int64_t nwb, WDesc *wd)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwb) return;
    if (w == nwb) {
        wd[w] = WDesc{0, 0, 0};
        return;
    }
    const int r0 = wb[w], r1 = wb[w + 1];
    const int64_t p0 = ia[r0], n = ia[r1] - p0;
    WDesc d;
    d.p0 = p0;
    d.r0 = r0;
    d.nrn = ((r1 - r0) << 16) | (n > 512 ? 0xFFFF : (int)n);
    wd[w] = d;
}

// ------------------------------------------------ pipelined wave kernel, row-group gathers ----
// Stream and pipeline as k_wave2, but the block's (column, value) pairs are staged in wave-private LDS and the gathers
// are issued with the lanes arranged as 8 consecutive rows x 8 consecutive entries: lanes 0..7 hold entry k of rows
// 8g..8g+7, so when those rows point at consecutive x elements (far part in band-major order: always) one 128-byte
// line serves 8 lanes.  Descriptor word: n | nr << 10 | maxlen << 16, bit 31 = slow path (a long row or > 63 rows).
template <int OPS, int POL, int ABL = 0>
__global__ __launch_bounds__(256) void k_wave3(WArgs a)
{
    constexpr int NW = 512;
    __shared__ int scol_s[4 * NW];
    __shared__ d2 sval_s[4 * NW];
    __shared__ int srow_s[4 * 64];
    __shared__ double red[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int *scol = scol_s + wv * NW;
    d2 *sval = sval_s + wv * NW;
    int *srow = srow_s + wv * 64;
    const int j = lane & 7, kk = lane >> 3;
    double acc3[3] = {0.0, 0.0, 0.0};

    const int64_t nunits = (a.nwb + 3) >> 2;
    int64_t per_xcd = (nunits + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int64_t nslot = gridDim.x >> 3;
    const int64_t chunk = nslot * (a.chunk > 0 ? a.chunk : 1);
    if (a.swizzle == 2) per_xcd = ((per_xcd + chunk - 1) / chunk) * chunk;

    const bool dyn = a.swizzle == 3;          // dynamic ordered walk per XCD (see k_wave2)
    constexpr int CH = 4;
    const int64_t bpx = (a.nwb + 7) >> 3;
    const int64_t xbase = (int64_t)xcd * bpx, xend = (xbase + bpx) < a.nwb ? (xbase + bpx) : a.nwb;
    int64_t ci0 = 0, ck0 = a.nwb, ck1 = a.nwb;
    auto grab = [&]() -> int64_t {
        unsigned long long c = 0;
        if (lane == 0) c = atomicAdd(a.ctr + xcd * 16, (unsigned long long)CH);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)c), hi = __builtin_amdgcn_readfirstlane((uint32_t)(c >> 32));
        return xbase + (int64_t)(((uint64_t)hi << 32) | lo);
    };
    if (dyn) {
        ck0 = grab();
        ck1 = grab();
    }
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto load_desc = [&](int64_t lb) -> v4i {
        int64_t w = a.nwb;
        if (dyn) {
            const int64_t st = (lb / CH) == ci0 ? ck0 : ck1;
            const int64_t wd_ = st + lb % CH;
            w = wd_ < xend ? wd_ : a.nwb;
        } else if (lb < per_xcd) {
            const int64_t unit = unit_of(lb, per_xcd, chunk, xcd, a.swizzle);
            w = unit * 4 + wv;
            if (w > a.nwb) w = a.nwb;
        }
        const WDesc *p = a.wd + __builtin_amdgcn_readfirstlane((int)w);
        v4i r;
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p) : "memory");
        return r;
    };
    auto desc_of = [&](v4i r) -> WDesc {
        WDesc d;
        d.p0 = (int64_t)(((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x);
        d.r0 = r.z;
        d.nrn = r.w;
        return d;
    };
    auto issue = [&](const WDesc &d, int (&c)[8], d2 (&v)[8], int &ro) {
        const bool slow = d.nrn < 0;
        const int n = d.nrn & 1023, nr = (d.nrn >> 10) & 63;
        const int nn = (n == 0 || slow) ? 1 : n;
        stream_loads8<POL>(a.ja, a.val, d.p0, nn, lane, c, v);
        ro = (int)(a.ia[d.r0 + (lane <= nr ? lane : 0)] - d.p0);
    };
    auto finish_row = [&](int orow, d2 sum, d2 yo, d2 xi, d2 tp) {
        if (OPS == 0) {
            if (a.ntstore) __builtin_nontemporal_store(sum, a.y + orow);
            else           a.y[orow] = sum;
        } else {
            if (OPS == 2) sum += tp;
            const d2 yn = a.alpha * sum + a.beta * yo + a.gamma * xi;
            a.y[orow] = yn;
            acc3[0] += xi.x * yn.x + xi.y * yn.y;
            acc3[1] += xi.x * yn.y - xi.y * yn.x;
            acc3[2] += yn.x * yn.x + yn.y * yn.y;
        }
    };

    int64_t lb = dyn ? 0 : (blockIdx.x >> 3);
    const int64_t step = dyn ? 1 : nslot;
    v4i q0 = load_desc(lb), q1 = load_desc(lb + step);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1) : : "memory");
    WDesc d0 = desc_of(q0), d1 = desc_of(q1);
    int cA[8];
    d2 vA[8];
    int roA;
    issue(d0, cA, vA, roA);
    while (dyn ? (ck0 + lb % CH < xend) : (lb < per_xcd)) {
        v4i q2 = load_desc(lb + 2 * step);
        const bool slow = d0.nrn < 0;
        const int nr = slow ? 0 : (d0.nrn >> 10) & 63;
        const int maxlen = (d0.nrn >> 16) & 1023;
        // stage this block
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            scol[lane + u * 64] = cA[u];
            sval[lane + u * 64] = vA[u];
        }
        srow[lane] = roA;
        wave_fence();
        const int g_first = d0.r0 >> 3;
        const int ngr = nr > 0 ? ((d0.r0 + nr - 1) >> 3) - g_first + 1 : 0;
        const int nst = (maxlen + 7) >> 3;
        // ---- chunk 0 (groups 0..3, steps 0..2), straight-line: its gathers go out before the next block's stream ----
        int grow[4];
        int s0[4], e0[4];
        int slot[4][3];
        bool ok[4][3];
        d2 xv[4][3];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            grow[gg] = ((g_first + gg) << 3) + j - d0.r0;
            const bool valid = grow[gg] >= 0 && grow[gg] < nr;
            s0[gg] = valid ? srow[grow[gg]] : 0;
            e0[gg] = valid ? srow[grow[gg] + 1] : 0;
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int sl = s0[gg] + 8 * st + kk;
                ok[gg][st] = sl < e0[gg];
                slot[gg][st] = ok[gg][st] ? sl : 0;
                if (ABL == 0) xv[gg][st] = a.xg[scol[slot[gg][st]]];
                else          xv[gg][st] = d2{1.0 + scol[slot[gg][st]], 0.5};
            }
        }
        d2 yo[4], xi[4], tp[4];
        if (OPS >= 1) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const bool mine = kk == 0 && grow[gg] >= 0 && grow[gg] < nr;
                const int orow = d0.r0 + (mine ? grow[gg] : 0);
                yo[gg] = a.y[orow];
                xi[gg] = a.xl[orow];
                if (OPS == 2) tp[gg] = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        int cB[8];
        d2 vB[8];
        int roB;
        issue(d1, cB, vB, roB);
        __builtin_amdgcn_sched_barrier(0);
        d2 sum[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            sum[gg] = d2{0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const d2 t = cmul(sval[slot[gg][st]], xv[gg][st]);
                if (ok[gg][st]) sum[gg] += t;
            }
        }
        if (nst > 3) {   // rows longer than 24 entries: the remaining steps, not pipelined
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                for (int st = 3; st < nst; ++st) {
                    const int sl = s0[gg] + 8 * st + kk;
                    if (sl < e0[gg]) sum[gg] += cmul(sval[sl], a.xg[scol[sl]]);
                }
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                sum[gg].x += __shfl_xor(sum[gg].x, off, 64);
                sum[gg].y += __shfl_xor(sum[gg].y, off, 64);
            }
            if (kk == 0 && grow[gg] >= 0 && grow[gg] < nr) finish_row(d0.r0 + grow[gg], sum[gg], yo[gg], xi[gg], tp[gg]);
        }
        // ---- further groups of the block (short rows): plain loop ----
        for (int gc = 4; gc < ngr; ++gc) {
            const int gr = ((g_first + gc) << 3) + j - d0.r0;
            const bool valid = gr >= 0 && gr < nr;
            const int s = valid ? srow[gr] : 0, e = valid ? srow[gr + 1] : 0;
            d2 sm = {0.0, 0.0};
            for (int st = 0; st < nst; ++st) {
                const int sl = s + 8 * st + kk;
                if (sl < e) sm += cmul(sval[sl], a.xg[scol[sl]]);
            }
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                sm.x += __shfl_xor(sm.x, off, 64);
                sm.y += __shfl_xor(sm.y, off, 64);
            }
            if (kk == 0 && valid) {
                const int orow = d0.r0 + gr;
                d2 y1 = {0.0, 0.0}, x1 = {0.0, 0.0}, t1 = {0.0, 0.0};
                if (OPS >= 1) {
                    y1 = a.y[orow];
                    x1 = a.xl[orow];
                }
                if (OPS == 2) t1 = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                finish_row(orow, sm, y1, x1, t1);
            }
        }
        wave_fence();
        if (slow) {
            const int nrs = (d0.nrn >> 10) & 0x1FFFFF;   // slow path: rows in bits 10..30
            for (int r = 0; r < nrs; ++r) {
                const int64_t s = a.ia[d0.r0 + r], e = a.ia[d0.r0 + r + 1];
                d2 sm = {0.0, 0.0};
                for (int64_t q = s + lane; q < e; q += 64) sm += cmul(a.val[q], a.xg[a.ja[q]]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    sm.x += __shfl_xor(sm.x, off, 64);
                    sm.y += __shfl_xor(sm.y, off, 64);
                }
                if (lane == 0) {
                    const int orow = d0.r0 + r;
                    d2 y1 = {0.0, 0.0}, x1 = {0.0, 0.0}, t1 = {0.0, 0.0};
                    if (OPS >= 1) {
                        y1 = a.y[orow];
                        x1 = a.xl[orow];
                    }
                    if (OPS == 2) t1 = a.tmp[a.tiled ? a.bd.tile(orow) : orow];
                    finish_row(orow, sm, y1, x1, t1);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cA[u] = cB[u];
            vA[u] = vB[u];
        }
        roA = roB;
        d0 = d1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q2) : : "memory");
        d1 = desc_of(q2);
        lb += step;
        if (dyn && lb % CH == 0) {
            ci0 = lb / CH;
            ck0 = ck1;
            ck1 = grab();
        }
    }
    if (a.partials != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int off = 32; off > 0; off >>= 1) acc3[c] += __shfl_xor(acc3[c], off, 64);
        if (lane == 0)
            for (int c = 0; c < 3; ++c) red[c * 4 + wv] = acc3[c];
        __syncthreads();
        if (tid == 0)
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 3 + c] = (red[c * 4] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}

// descriptors of k_wave3
__global__ void k_build_wd3(const int64_t *ia, const int32_t *wb, int64_t nwb, WDesc *wd)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwb) return;
    if (w == nwb) {
        wd[w] = WDesc{0, 0, 0};
        return;
    }
    const int r0 = wb[w], r1 = wb[w + 1];
    const int64_t p0 = ia[r0], n = ia[r1] - p0;
    int64_t mx = 0;
    for (int r = r0; r < r1; ++r) mx = ia[r + 1] - ia[r] > mx ? ia[r + 1] - ia[r] : mx;
    WDesc d;
    d.p0 = p0;
    d.r0 = r0;
    const int nr = r1 - r0;
    if (n > 512 || nr > 63) d.nrn = (int)(0x80000000u | ((unsigned)nr << 10));
    else                    d.nrn = (int)n | (nr << 10) | ((int)mx << 16);
    wd[w] = d;
}

// ------------------------------------------------------------------ layout builders ----
__global__ void k_build_wb(const int64_t *ia, int64_t nrows, int64_t window, int32_t *wb, int64_t nwb)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwb) return;
    if (w == nwb) {
        wb[w] = (int32_t)nrows;
        return;
    }
    const int64_t target = w * window;
    int64_t lo = 0, hi = nrows;   // first row with ia[row] >= target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ia[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    wb[w] = (int32_t)lo;
}
__global__ void k_maxlen(const int64_t *ia, int64_t nrows, unsigned long long *out)
{
    unsigned long long mx = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long len = ia[r + 1] - ia[r];
        mx = len > mx ? len : mx;
    }
    if (mx) atomicMax(out, mx);
}

// far = column in another major block
__global__ void k_split_count(const int64_t *ia, const int32_t *ja, int64_t nrows, Bands bd, int32_t *cnt_near, int32_t *cnt_far,
                              int32_t *perm)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = bd.orig(f);
        perm[f] = (int32_t)r;
        const int64_t maj = r / bd.S;
        int nf = 0;
        const int64_t s = ia[r], e = ia[r + 1];
        for (int64_t q = s; q < e; ++q) nf += (ja[q] / bd.S) != maj;
        cnt_far[f] = nf;
        cnt_near[r] = (int)(e - s) - nf;
    }
}
__global__ void k_split_fill(const int64_t *ia, const int32_t *ja, const d2 *val, int64_t nrows, Bands bd, const int64_t *ia_n,
                             int32_t *ja_n, d2 *val_n, const int64_t *ia_f, int32_t *ja_f, d2 *val_f, int tiled_cols, int colmask)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nrows; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = bd.orig(f);
        const int64_t maj = r / bd.S;
        int64_t pn = ia_n[r], pf = ia_f[f];
        for (int64_t q = ia[r]; q < ia[r + 1]; ++q) {
            const int32_t c = ja[q];
            if ((c / bd.S) != maj) {
                ja_f[pf] = (tiled_cols ? (int32_t)bd.tile(c) : c) & colmask;
                val_f[pf++] = val[q];
            } else {
                ja_n[pn] = c;
                val_n[pn++] = val[q];
            }
        }
    }
}
__global__ void k_tile(const d2 *x, d2 *xt, int64_t n, Bands bd)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        xt[bd.tile(r)] = x[r];
}
__global__ void k_xcc(int *out)
{
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int)(v & 0xf);
    }
}
__global__ void k_fill_rand(d2 *x, int64_t n, uint32_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i].x = (double)(z & 0xFFFFFF) / 16777216.0 - 0.5;
        x[i].y = (double)((z >> 24) & 0xFFFFFF) / 16777216.0 - 0.5;
    }
}
__global__ void k_maxdiff(const d2 *a, const d2 *b, int64_t n, double *out)
{
    double md = 0.0, mv = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double dx = fabs(a[i].x - b[i].x), dy = fabs(a[i].y - b[i].y);
        md = fmax(md, fmax(dx, dy));
        mv = fmax(mv, fmax(fabs(b[i].x), fabs(b[i].y)));
    }
    // doubles >= 0 compare like their bit patterns
    atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(md));
    atomicMax((unsigned long long *)(out + 1), (unsigned long long)__double_as_longlong(mv));
}

static void exscan(const int32_t *d_cnt, int64_t n, int64_t *d_ia)
{
    // ia[0..n]: exclusive sum with the total at ia[n]
    void *tmp = nullptr;
    size_t bytes = 0;
    struct Cast {
        __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
    };
    hipcub::TransformInputIterator<int64_t, Cast, const int32_t *> it(d_cnt, Cast());
    CK(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, it, d_ia, n));
    CK(hipMalloc(&tmp, bytes));
    CK(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, it, d_ia, n));
    CK(hipDeviceSynchronize());
    CK(hipFree(tmp));
    int64_t last_ia = 0;
    int32_t last_cnt = 0;
    CK(hipMemcpy(&last_ia, d_ia + n - 1, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&last_cnt, d_cnt + n - 1, 4, hipMemcpyDeviceToHost));
    last_ia += last_cnt;
    CK(hipMemcpy(d_ia + n, &last_ia, 8, hipMemcpyHostToDevice));
}

struct WaveLayout {
    int32_t *wb = nullptr;
    WDesc *wd = nullptr, *wd3 = nullptr;
    int64_t nwb = 0;
};
static WaveLayout build_wave_blocks(const int64_t *d_ia, int64_t nrows, int64_t nnz, int NW)
{
    unsigned long long *d_mx;
    CK(hipMalloc(&d_mx, 8));
    CK(hipMemset(d_mx, 0, 8));
    k_maxlen<<<1024, 256>>>(d_ia, nrows, d_mx);
    unsigned long long mx = 0;
    CK(hipMemcpy(&mx, d_mx, 8, hipMemcpyDeviceToHost));
    CK(hipFree(d_mx));
    int64_t window = NW - (int64_t)mx + 1;
    if (window < 1) window = 1;
    WaveLayout L;
    L.nwb = (nnz + window - 1) / window;
    if (L.nwb < 1) L.nwb = 1;
    CK(hipMalloc(&L.wb, (L.nwb + 1) * 4));
    k_build_wb<<<(unsigned)((L.nwb + 256) / 256), 256>>>(d_ia, nrows, window, L.wb, L.nwb);
    CK(hipMalloc(&L.wd, (L.nwb + 1) * sizeof(WDesc)));
    k_build_wd<<<(unsigned)((L.nwb + 256) / 256), 256>>>(d_ia, L.wb, L.nwb, L.wd);
    CK(hipMalloc(&L.wd3, (L.nwb + 1) * sizeof(WDesc)));
    k_build_wd3<<<(unsigned)((L.nwb + 256) / 256), 256>>>(d_ia, L.wb, L.nwb, L.wd3);
    CK(hipDeviceSynchronize());
    printf("    wave blocks: NW %d maxrow %llu window %lld -> %lld blocks\n", NW, mx, (long long)window, (long long)L.nwb);
    return L;
}

template <int U, int TPR, int TAG = 0, int POL = -2>
static float run_wave(WArgs a, int reps, int wgs_per_cu_cap, const char *tag)
{
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_wave<U, TPR, TAG, POL>, 256, 0));
    if (wgs_per_cu_cap > 0 && occ > wgs_per_cu_cap) occ = wgs_per_cu_cap;
    const int grid = 256 * occ;
    double *d_part = nullptr;
    CK(hipMalloc(&d_part, (size_t)grid * 3 * 8));
    if (a.partials) a.partials = d_part;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_wave<U, TPR, TAG, POL><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_wave<U, TPR, TAG, POL><<<grid, 256>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(d_part));
    printf("    %-28s U%d TPR%d occ %d grid %d swz %d: %.3f ms\n", tag, U, TPR, occ, grid, a.swizzle, ms / reps);
    fflush(stdout);
    return ms / reps;
}


template <int TPR, int OPS, int POL, int ABL = 0, int GPOL = -1>
static float run_wave2(WArgs a, int reps, const char *tag)
{
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_wave2<TPR, OPS, POL, ABL, GPOL>, 256, 0));
    const int grid = 256 * occ;
    double *d_part = nullptr;
    CK(hipMalloc(&d_part, (size_t)grid * 3 * 8));
    if (a.partials) a.partials = d_part;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long *ctr = nullptr;
    CK(hipMalloc(&ctr, 8 * 16 * 8));
    a.ctr = ctr;
    CK(hipMemsetAsync(ctr, 0, 8 * 16 * 8));
    k_wave2<TPR, OPS, POL, ABL, GPOL><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
        if (a.swizzle == 3) CK(hipMemsetAsync(ctr, 0, 8 * 16 * 8));
        k_wave2<TPR, OPS, POL, ABL, GPOL><<<grid, 256>>>(a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(d_part));
    CK(hipFree(ctr));
    printf("    %-28s pipelined TPR%d occ %d grid %d swz %d: %.3f ms\n", tag, TPR, occ, grid, a.swizzle, ms / reps);
    fflush(stdout);
    return ms / reps;
}

template <int OPS, int POL, int ABL = 0>
static float run_wave3(WArgs a, int reps, const char *tag)
{
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_wave3<OPS, POL, ABL>, 256, 0));
    const int grid = 256 * occ;
    double *d_part = nullptr;
    CK(hipMalloc(&d_part, (size_t)grid * 3 * 8));
    if (a.partials) a.partials = d_part;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long *ctr = nullptr;
    CK(hipMalloc(&ctr, 8 * 16 * 8));
    a.ctr = ctr;
    CK(hipMemsetAsync(ctr, 0, 8 * 16 * 8));
    k_wave3<OPS, POL, ABL><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
        if (a.swizzle == 3) CK(hipMemsetAsync(ctr, 0, 8 * 16 * 8));
        k_wave3<OPS, POL, ABL><<<grid, 256>>>(a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(d_part));
    printf("    %-28s row-group gathers occ %d grid %d swz %d: %.3f ms\n", tag, occ, grid, a.swizzle, ms / reps);
    fflush(stdout);
    return ms / reps;
}

static double check(const d2 *y, const d2 *yref, int64_t n)
{
    double *d_o, h[2];
    CK(hipMalloc(&d_o, 16));
    CK(hipMemset(d_o, 0, 16));
    k_maxdiff<<<1024, 256>>>(y, yref, n, d_o);
    CK(hipMemcpy(h, d_o, 16, hipMemcpyDeviceToHost));
    CK(hipFree(d_o));
    return h[0] / (h[1] > 0 ? h[1] : 1.0);
}

static std::vector<int32_t> square_bonds(int Lx, int Ly)
{
    std::vector<int32_t> b;
    for (int x = 0; x < Lx; ++x)
        for (int y = 0; y < Ly; ++y) {
            const int s = x + Lx * y;
            b.push_back(s);
            b.push_back((x + 1) % Lx + Lx * y);
            b.push_back(s);
            b.push_back(x + Lx * ((y + 1) % Ly));
        }
    return b;
}
static int64_t binom(int n, int k)
{
    double r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (int64_t)llround(r);
}

int main(int argc, char **argv)
{
    const std::string wl = argc > 1 ? argv[1] : "chain24";
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const char *only = getenv("LAB_ONLY");   // substring filter on variant tags
    auto want = [&](const char *tag) {   // LAB_ONLY: comma-separated list of exact tags ("wave", "split", policy names)
        if (only == nullptr) return true;
        const std::string o = std::string(",") + only + ",";
        return o.find(std::string(",") + tag + ",") != std::string::npos;
    };

    qbh_opts o;
    qbh_opts_default(&o);
    o.value_dict = 0;
    o.real_fast_path = 0;
    o.profile = 1;
    qbh_csr *A = nullptr;
    int64_t stride = 0, nmajor = 0;
    if (wl == "c3" || wl == "h4x3" || wl == "h4x2") {
        const int Ly = wl == "c3" ? 4 : wl == "h4x3" ? 3 : 2;
        const int ns = 4 * Ly, np = ns / 2;
        auto b = square_bonds(4, Ly);
        const int64_t dim = binom(ns, np) * binom(ns, np);
        QB(qbh_gen_hubbard(&A, ns, np, np, (int)b.size() / 2, b.data(), 1.0, 1.1, 0, dim, &o));
        stride = binom(ns, np);
        nmajor = stride;
    } else {
        const int L = wl == "chain26" ? 26 : wl == "chain22" ? 22 : 24;
        std::vector<int32_t> b;
        for (int x = 0; x < L; ++x) {
            b.push_back(x);
            b.push_back((x + 1) % L);
        }
        const int64_t dim = binom(L, L / 2);
        QB(qbh_gen_heisenberg(&A, L, L / 2, L, b.data(), 1.0, 0, dim, &o));
    }
    {
        int *d_x, h[64];
        CK(hipMalloc(&d_x, 64 * 4));
        k_xcc<<<64, 256>>>(d_x);
        CK(hipMemcpy(h, d_x, 64 * 4, hipMemcpyDeviceToHost));
        printf("XCC_ID of workgroups 0..31:");
        for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
        printf("\n");
        CK(hipFree(d_x));
    }
    const int64_t n = A->nrows, nnz = A->nnz;
    const double balg = (double)nnz * 20 + (double)(n + 1) * 8 + (double)n * 32;
    printf("workload %s: dim %lld nnz %lld  algorithmic bytes %.3f GB\n", wl.c_str(), (long long)n, (long long)nnz, balg / 1e9);

    d2 *x, *y, *yref, *tmp;
    CK(hipMalloc(&x, n * 16));
    CK(hipMalloc(&y, n * 16));
    CK(hipMalloc(&yref, n * 16));
    CK(hipMalloc(&tmp, n * 16));
    k_fill_rand<<<1024, 256>>>(x, n, 1);
    k_fill_rand<<<1024, 256>>>(yref, n, 2);
    CK(hipDeviceSynchronize());
    const double alpha = 0.7, beta = -0.3, gamma = 0.0;
    double red[3];

    // library reference (result + time)
    QB(qbh_spmv_dev(A, (qbh_z *)x, (qbh_z *)yref, alpha, beta, gamma, red));   // yref = alpha H x + beta y0
    {
        CK(hipMemcpy(y, yref, n * 16, hipMemcpyDeviceToDevice));
        qbh_stats st;
        QB(qbh_get_stats(A, &st, 1));
        for (int i = 0; i < reps; ++i) QB(qbh_spmv_dev(A, (qbh_z *)x, (qbh_z *)y, alpha, beta, gamma, red));
        QB(qbh_sync(A));
        QB(qbh_get_stats(A, &st, 0));
        const double ms = st.ms_spmv / st.n_spmv;
        printf("  library kernel: %.3f ms  -> %.0f GB/s, frac %.3f\n", ms, balg / ms / 1e6, balg / ms / 1e6 / 8000);
        fflush(stdout);
    }
    auto report = [&](const char *tag, float ms, const d2 *yy) {
        (void)yy;
        printf("  == %-34s %.3f ms  -> %.0f GB/s, frac %.3f\n", tag, ms, balg / ms / 1e6, balg / ms / 1e6 / 8000);
        fflush(stdout);
    };
    auto reset_y = [&]() {
        k_fill_rand<<<1024, 256>>>(y, n, 2);   // y0 again
        CK(hipDeviceSynchronize());
    };

    WArgs a{};
    a.ia = A->d_ia;
    a.ja = A->d_ja;
    a.val = A->d_val;
    a.xg = x;
    a.xl = x;
    a.y = y;
    a.alpha = alpha;
    a.beta = 0.0;   // timing runs overwrite y (beta = 0 reads no y); the checked run uses beta
    a.gamma = gamma;
    a.partials = (double *)1;
    a.chunk = 1;

    // ---------------- unsplit, wave kernel ----------------
    auto parity = [&](auto kern, WArgs c) {
        reset_y();
        c.partials = nullptr;
        c.beta = beta;
        kern<<<1024, 256>>>(c);
        CK(hipDeviceSynchronize());
        return check(y, yref, n);
    };
    if (want("wave")) {
        printf("  unsplit operator, wave kernel\n");
        for (int U : {8, 4}) {
            if (const char *e = getenv("LAB_U")) {
                if (atoi(e) != U) continue;
            }
            WaveLayout L = build_wave_blocks(A->d_ia, n, nnz, 64 * U);
            a.wb = L.wb;
            a.nwb = L.nwb;
            a.beta = beta;
            for (int swz : {2, 1}) {
                a.swizzle = swz;
                if (U == 8) {
                    printf("     parity U8: %.2e\n", parity(k_wave<8, 4>, a));
                    report("wave U8 TPR4", run_wave<8, 4>(a, reps, 0, "timed"), yref);
                    report("wave U8 TPR2", run_wave<8, 2>(a, reps, 0, "timed"), yref);
                    report("wave U8 TPR8", run_wave<8, 8>(a, reps, 0, "timed"), yref);
                    a.wd = L.wd;
                    printf("     parity pipelined: %.2e\n", parity(k_wave2<4, 1, -2>, a));
                    report("wave2 TPR4 nt", run_wave2<4, 1, -2>(a, reps, "timed"), yref);
                    report("wave2 TPR2 nt", run_wave2<2, 1, -2>(a, reps, "timed"), yref);
                    report("wave2 TPR4 bufnt", run_wave2<4, 1, 2>(a, reps, "timed"), yref);
                } else {
                    printf("     parity U4: %.2e\n", parity(k_wave<4, 4>, a));
                    report("wave U4 TPR4", run_wave<4, 4>(a, reps, 0, "timed"), yref);
                    report("wave U4 TPR2", run_wave<4, 2>(a, reps, 0, "timed"), yref);
                }
            }
            CK(hipFree(L.wb));
            CK(hipFree(L.wd));
            CK(hipFree(L.wd3));
        }
    }

    // ---------------- split band-major ----------------
    if (stride > 0 && want("split")) {
        std::vector<int> Bs = {8, 4};
        if (const char *e = getenv("LAB_B")) {
            Bs.clear();
            for (const char *q = e; *q;) {
                Bs.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        for (int B : Bs) {
            const int tiled = getenv("LAB_UNTILED") ? 0 : 1;
            const int ntstore = getenv("LAB_NTSTORE") ? 1 : 0;
            printf("  split layout: stride %lld, band width %d, %s, %s stores of the far result\n", (long long)stride, B,
                   tiled ? "tiled x copy" : "gathers from x", ntstore ? "nt" : "plain");
            d2 *xt = nullptr;
            if (tiled) CK(hipMalloc(&xt, n * 16));
            Bands bd{stride, nmajor, B};
            int32_t *cn, *cf, *perm;
            CK(hipMalloc(&cn, n * 4));
            CK(hipMalloc(&cf, n * 4));
            CK(hipMalloc(&perm, n * 4));
            k_split_count<<<4096, 256>>>(A->d_ia, A->d_ja, n, bd, cn, cf, perm);
            CK(hipDeviceSynchronize());
            int64_t *ia_n, *ia_f;
            CK(hipMalloc(&ia_n, (n + 1) * 8));
            CK(hipMalloc(&ia_f, (n + 1) * 8));
            exscan(cn, n, ia_n);
            exscan(cf, n, ia_f);
            CK(hipFree(cn));
            CK(hipFree(cf));
            int64_t nnz_n, nnz_f;
            CK(hipMemcpy(&nnz_n, ia_n + n, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&nnz_f, ia_f + n, 8, hipMemcpyDeviceToHost));
            printf("    near nnz %lld, far nnz %lld\n", (long long)nnz_n, (long long)nnz_f);
            int32_t *ja_n, *ja_f;
            d2 *val_n, *val_f;
            CK(hipMalloc(&ja_n, nnz_n * 4 + 64));
            CK(hipMalloc(&ja_f, nnz_f * 4 + 64));
            CK(hipMalloc(&val_n, nnz_n * 16 + 64));
            CK(hipMalloc(&val_f, nnz_f * 16 + 64));
            k_split_fill<<<4096, 256>>>(A->d_ia, A->d_ja, A->d_val, n, bd, ia_n, ja_n, val_n, ia_f, ja_f, val_f, tiled, getenv("LAB_COLMASK") ? atoi(getenv("LAB_COLMASK")) : -1);
            CK(hipDeviceSynchronize());

            WaveLayout Ln = build_wave_blocks(ia_n, n, nnz_n, 512);
            WaveLayout Lf = build_wave_blocks(ia_f, n, nnz_f, 512);
            WArgs f{};   // far pass: tmp[f] = sum (tiled) or tmp[perm[f]] = sum
            f.ia = ia_f;
            f.ja = ja_f;
            f.val = val_f;
            f.wb = Lf.wb;
            f.wd = Lf.wd;
            f.nwb = Lf.nwb;
            f.xg = tiled ? xt : x;
            f.xl = x;
            f.y = tmp;
            f.perm = tiled ? nullptr : perm;
            f.plain = 1;
            f.ntstore = ntstore;
            f.swizzle = 1;
            f.chunk = 1;
            WArgs nr = a;   // near pass
            nr.ia = ia_n;
            nr.ja = ja_n;
            nr.val = val_n;
            nr.wb = Ln.wb;
            nr.wd = Ln.wd;
            nr.nwb = Ln.nwb;
            nr.tmp = tmp;
            nr.tiled = tiled;
            nr.bd = bd;
            nr.beta = beta;
            nr.swizzle = 2;
            auto tile_ms = [&]() {
                float mt = 0;
                if (!tiled) return mt;
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) k_tile<<<2048, 256>>>(x, xt, n, bd);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&mt, e0, e1));
                return mt / reps;
            };
            auto one = [&](auto kf, auto kn, auto rf, auto rn, const char *pol) {
                reset_y();
                WArgs nc = nr;
                nc.partials = nullptr;
                if (tiled) k_tile<<<2048, 256>>>(x, xt, n, bd);
                CK(hipDeviceSynchronize());
                fprintf(stderr, "[%s] far...\n", pol);
                unsigned long long *ctr2 = nullptr;
                CK(hipMalloc(&ctr2, 2 * 8 * 16 * 8));
                CK(hipMemset(ctr2, 0, 2 * 8 * 16 * 8));
                f.ctr = ctr2;
                nc.ctr = ctr2 + 8 * 16;
                kf<<<1024, 256>>>(f);
                CK(hipDeviceSynchronize());
                fprintf(stderr, "[%s] near...\n", pol);
                kn<<<1024, 256>>>(nc);
                CK(hipDeviceSynchronize());
                const double err = check(y, yref, n);
                const float mt = tile_ms();
                const float mf = rf(f, reps, 0, "far pass");
                const float mn = rn(nr, reps, 0, "near pass");
                printf("  == split B%d stream policy %-12s tile %.3f + far %.3f + near %.3f = %.3f ms -> frac %.3f, err %.2e\n", B, pol, mt, mf,
                       mn, mt + mf + mn, balg / (mt + mf + mn) / 1e6 / 8000, err);
                fflush(stdout);
            };
#define POLICY(P, NAME)                                                                                                  \
    if (want(NAME)) one(k_wave<8, 4, 1, P>, k_wave<8, 4, 2, P>, run_wave<8, 4, 1, P>, run_wave<8, 4, 2, P>, NAME)
#define POLICY2(P, T, NAME)                                                                                              \
    if (want(NAME)) one(k_wave2<T, 0, P>, k_wave2<T, 2, P>, [&](WArgs q, int r, int, const char *t) { return run_wave2<T, 0, P>(q, r, t); }, \
                        [&](WArgs q, int r, int, const char *t) { return run_wave2<T, 2, P>(q, r, t); }, NAME)
            auto with_wd3 = [&](WArgs q, WDesc *w3) {
                q.wd = w3;
                return q;
            };
            if (want("g3far")) {     // far: row-group gathers; near: pipelined lane-per-entry
                const WArgs f3 = with_wd3(f, Lf.wd3);
                const WArgs fsave = f;
                f = f3;
                one(k_wave3<0, -2>, k_wave2<2, 2, -2>, [&](WArgs q, int r, int, const char *t) { return run_wave3<0, -2>(q, r, t); },
                    [&](WArgs q, int r, int, const char *t) { return run_wave2<2, 2, -2>(q, r, t); }, "g3far");
                f = fsave;
            }
            if (want("g3both")) {    // both passes with row-group gathers
                const WArgs fsave = f, nsave = nr;
                f = with_wd3(f, Lf.wd3);
                nr = with_wd3(nr, Ln.wd3);
                one(k_wave3<0, -2>, k_wave3<2, -2>, [&](WArgs q, int r, int, const char *t) { return run_wave3<0, -2>(q, r, t); },
                    [&](WArgs q, int r, int, const char *t) { return run_wave3<2, -2>(q, r, t); }, "g3both");
                f = fsave;
                nr = nsave;
            }
            if (want("gpol")) {
                run_wave2<2, 0, -2, 4, -1>(f, reps, "far ABL4 columns + gathers, no values");
                run_wave2<2, 0, -2, 0, -1>(f, reps, "far gathers plain");
                run_wave2<2, 0, -2, 0, 0>(f, reps, "far gathers buffer");
                run_wave2<2, 0, -2, 0, 1>(f, reps, "far gathers buffer sc0");
                run_wave2<2, 0, -2, 0, 16>(f, reps, "far gathers buffer sc1");
                run_wave2<2, 0, -2, 0, 17>(f, reps, "far gathers buffer sc0 sc1");
                run_wave2<2, 0, -2, 0, 2>(f, reps, "far gathers buffer nt");
            }
            if (want("abl3")) {
                const WArgs f3 = with_wd3(f, Lf.wd3);
                run_wave3<0, -2, 0>(f3, reps, "far g3 full");
                run_wave3<0, -2, 1>(f3, reps, "far g3 no gathers");
                WArgs f4 = f3;
                f4.swizzle = 2;
                run_wave3<0, -2, 0>(f4, reps, "far g3 full swz2");
            }
            if (want("abldyn")) {      // the same ablation under the dynamic ordered walk (gather misses out of the way)
                WArgs fd = f;
                fd.swizzle = 3;
                run_wave2<2, 0, -2, 0>(fd, reps, "far dyn ABL0 full");
                run_wave2<2, 0, -2, 1>(fd, reps, "far dyn ABL1 no gathers");
                run_wave2<2, 0, -2, 2>(fd, reps, "far dyn ABL2 stream only");
                run_wave2<2, 0, -2, 3>(fd, reps, "far dyn ABL3 values only");
                run_wave2<2, 0, -2, 4>(fd, reps, "far dyn ABL4 columns + gathers");
                WArgs nd = nr;
                nd.swizzle = 3;
                run_wave2<2, 2, -2, 0>(nd, reps, "near dyn ABL0 full");
                run_wave2<2, 2, -2, 1>(nd, reps, "near dyn ABL1 no gathers");
                run_wave2<2, 2, -2, 2>(nd, reps, "near dyn ABL2 stream only");
            }
            if (want("abl")) {
                run_wave2<2, 0, -2, 0>(f, reps, "far ABL0 full");
                run_wave2<2, 0, -2, 1>(f, reps, "far ABL1 no gathers");
                run_wave2<2, 0, -2, 2>(f, reps, "far ABL2 stream only");
                run_wave2<2, 0, -2, 3>(f, reps, "far ABL3 values only");
                run_wave2<2, 0, -1, 2>(f, reps, "far ABL2 plain loads");
                run_wave2<2, 0, -1, 3>(f, reps, "far ABL3 plain loads");
                run_wave2<2, 0, 2, 3>(f, reps, "far ABL3 buffer nt");
                WArgs f2 = f;
                f2.swizzle = 2;
                run_wave2<2, 0, -2, 2>(f2, reps, "far ABL2 swz2");
                run_wave2<2, 0, -2, 3>(f2, reps, "far ABL3 swz2");
                f2.swizzle = 0;
                run_wave2<2, 0, -2, 3>(f2, reps, "far ABL3 swz0");
            }
            if (want("g3dyn")) {
                const WArgs fsave = f, nsave = nr;
                f = with_wd3(f, Lf.wd3);
                f.swizzle = 3;
                nr.swizzle = 3;
                one(k_wave3<0, -2>, k_wave2<2, 2, -2>, [&](WArgs q, int r, int, const char *t) { return run_wave3<0, -2>(q, r, t); },
                    [&](WArgs q, int r, int, const char *t) { return run_wave2<2, 2, -2>(q, r, t); }, "g3dyn far");
                nr = with_wd3(nr, Ln.wd3);
                nr.swizzle = 3;
                one(k_wave3<0, -2>, k_wave3<2, -2>, [&](WArgs q, int r, int, const char *t) { return run_wave3<0, -2>(q, r, t); },
                    [&](WArgs q, int r, int, const char *t) { return run_wave3<2, -2>(q, r, t); }, "g3dyn both");
                f = fsave;
                nr = nsave;
            }
            if (want("dyn")) {
                const WArgs fsave = f, nsave = nr;
                f.swizzle = 3;
                nr.swizzle = 3;
                POLICY2(-2, 2, "dyn");
                nr = nsave;
                nr.swizzle = 2;
                POLICY2(-2, 2, "dyn");       // far dynamic, near static chunked
                f = fsave;
                nr = nsave;
            }
            POLICY2(-2, 2, "p2nt");
            POLICY2(-2, 4, "p4nt");
            POLICY2(2, 2, "p2bufnt");
            POLICY2(-1, 2, "p2plain");
            POLICY2(1, 2, "p2bufsc0");
            POLICY2(16, 2, "p2bufsc1");
            POLICY2(17, 2, "p2bufsc0sc1");
            POLICY2(18, 2, "p2bufsc1nt");
            POLICY2(19, 2, "p2bufsc0sc1nt");
            POLICY(-2, "nt");
            POLICY(-1, "plain");
            POLICY(0, "buf");
            POLICY(2, "bufnt");
            POLICY(16, "bufsc1");
            POLICY(18, "bufsc1nt");
            POLICY(17, "bufsc0sc1");
            POLICY(19, "bufsc0sc1nt");
            CK(hipFree(Ln.wb));
            CK(hipFree(Lf.wb));
            CK(hipFree(Ln.wd));
            CK(hipFree(Lf.wd));
            CK(hipFree(Ln.wd3));
            CK(hipFree(Lf.wd3));
            CK(hipFree(ja_n));
            CK(hipFree(ja_f));
            CK(hipFree(val_n));
            CK(hipFree(val_f));
            CK(hipFree(ia_n));
            CK(hipFree(ia_f));
            CK(hipFree(perm));
            if (xt) CK(hipFree(xt));
        }
    }
    qbh_csr_destroy(A);
    return 0;
}
