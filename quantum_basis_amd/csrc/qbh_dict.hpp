// qbh_dict.hpp -- device-side pieces of the lossless value dictionary, shared by the coder in
// qbh_kernels.hip (values already in HBM) and by generators that emit codes directly without ever
// materialising the 16 B/nnz value array (qbh_gen.hip).
//
// Collection is a two-level open-addressing hash keyed by a 64-bit fingerprint of the bit pattern
// of the complex128 value: one table in LDS per workgroup, flushed into a small global table.  The
// final dictionary is ordered by bit pattern, so codes do not depend on atomic races.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "qbh_internal.hpp"

namespace qbh {

constexpr int kDictLocal = 512;      // LDS hash slots per workgroup (power of two)
constexpr int kDictGlobal = 1024;    // global hash slots
// flags: [0] overflow (>256 distinct), [1] global entries claimed, [2] final count, [3] verify mismatch

__device__ __forceinline__ unsigned long long dict_fp(d2 v)
{
    unsigned long long a = (unsigned long long)__double_as_longlong(v.x);
    unsigned long long b = (unsigned long long)__double_as_longlong(v.y);
    unsigned long long x = a * 0x9E3779B97F4A7C15ULL ^ ((b << 31) | (b >> 33)) * 0xC2B2AE3D27D4EB4FULL;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 32;
    return x ? x : 1ULL;
}

// LDS state of one collecting workgroup
struct DictCollect {
    unsigned long long lf[kDictLocal];
    d2 lv[kDictLocal];
    int lcount;
};

__device__ __forceinline__ void dict_collect_init(DictCollect &D)
{
    for (int i = threadIdx.x; i < kDictLocal; i += blockDim.x) D.lf[i] = 0ULL;
    if (threadIdx.x == 0) D.lcount = 0;
    __syncthreads();
}

// true while the workgroup has seen at most 256 distinct values
__device__ __forceinline__ bool dict_collect_ok(DictCollect &D) { return ((volatile int *)&D.lcount)[0] <= 256; }

__device__ __forceinline__ void dict_collect_insert(DictCollect &D, d2 v, int *flags)
{
    const unsigned long long f = dict_fp(v);
    int s = (int)(f & (kDictLocal - 1));
    for (int probe = 0; probe < kDictLocal; ++probe) {
        const unsigned long long cur = ((volatile unsigned long long *)D.lf)[s];
        if (cur == f) return;
        if (cur == 0ULL) {
            const unsigned long long old = atomicCAS(&D.lf[s], 0ULL, f);
            if (old == 0ULL) {
                D.lv[s] = v;
                if (atomicAdd(&D.lcount, 1) >= 256) flags[0] = 1;
                return;
            }
            if (old == f) return;
        }
        s = (s + 1) & (kDictLocal - 1);
    }
}

// merge the workgroup's table into the global one (all threads; barrier inside)
__device__ __forceinline__ void dict_collect_flush(DictCollect &D, unsigned long long *gf, d2 *gv, int *flags)
{
    __syncthreads();
    for (int i = threadIdx.x; i < kDictLocal; i += blockDim.x) {
        const unsigned long long f = D.lf[i];
        if (f == 0ULL) continue;
        int s = (int)(f & (kDictGlobal - 1));
        for (int probe = 0; probe < kDictGlobal; ++probe) {
            const unsigned long long old = atomicCAS(&gf[s], 0ULL, f);
            if (old == 0ULL) {
                gv[s] = D.lv[i];
                if (atomicAdd(&flags[1], 1) >= 256) flags[0] = 1;
                break;
            }
            if (old == f) break;
            s = (s + 1) & (kDictGlobal - 1);
        }
    }
}

// LDS state of one encoding workgroup
struct DictEncode {
    unsigned long long lf[kDictLocal];
    int lc[kDictLocal];
    d2 ds[256];
};

__device__ __forceinline__ void dict_encode_init(DictEncode &E, const d2 *dict, int n)
{
    for (int i = threadIdx.x; i < kDictLocal; i += blockDim.x) E.lf[i] = 0ULL;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) E.ds[i] = dict[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int c = 0; c < n; ++c) {
            const unsigned long long f = dict_fp(E.ds[c]);
            int s = (int)(f & (kDictLocal - 1));
            while (E.lf[s] != 0ULL) s = (s + 1) & (kDictLocal - 1);
            E.lf[s] = f;
            E.lc[s] = c;
        }
    }
    __syncthreads();
}

// code of v; a value that is not in the dictionary (bitwise) raises flags[3] and codes as 0
__device__ __forceinline__ uint8_t dict_encode_one(const DictEncode &E, d2 v, int *flags)
{
    const unsigned long long f = dict_fp(v);
    int s = (int)(f & (kDictLocal - 1));
    int c = -1;
    for (int probe = 0; probe < kDictLocal; ++probe) {
        if (E.lf[s] == f) {
            c = E.lc[s];
            break;
        }
        if (E.lf[s] == 0ULL) break;
        s = (s + 1) & (kDictLocal - 1);
    }
    bool ok = c >= 0;
    if (ok) {
        const d2 w = E.ds[c];
        ok = __double_as_longlong(w.x) == __double_as_longlong(v.x) && __double_as_longlong(w.y) == __double_as_longlong(v.y);
    }
    if (!ok) {
        flags[3] = 1;
        c = 0;
    }
    return (uint8_t)c;
}

// host side (qbh_kernels.hip): scratch tables of one dictionary build
struct DictBuild {
    unsigned long long *gf = nullptr;
    d2 *gv = nullptr;
    int *flags = nullptr;
};
int dict_build_begin(DictBuild *b, hipStream_t s);
// order the collected values into d_dict[256]; *n_out = number of entries, 0 when there are more than 256
int dict_build_finalize(DictBuild *b, d2 *d_dict, int *n_out, hipStream_t s);
// 1 when an encode pass met a value outside the dictionary
int dict_build_mismatch(DictBuild *b, int *bad, hipStream_t s);
void dict_build_end(DictBuild *b);

}  // namespace qbh
