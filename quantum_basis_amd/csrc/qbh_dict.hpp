// qbh_dict.hpp -- device-side pieces of the lossless value dictionary, shared by the coder in
// qbh_kernels.hip (values already in HBM) and by generators that emit codes directly without ever
// materialising the 16 B/nnz value array (qbh_gen.hip).
//
// A matrix with at most 256 distinct complex128 values is stored with 1-byte codes, one with at most
// 65536 with 2-byte codes; products always use the exact original doubles.  Collection is an
// open-addressing hash in HBM keyed by a 64-bit fingerprint of the value's bit pattern, fronted by a
// direct-mapped cache in LDS so that the common case (few values, seen over and over) never leaves
// the CU.  The final dictionary is ordered by bit pattern on the host, so codes do not depend on
// atomic races; every emitted code is verified bitwise against the value it stands for.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "qbh_internal.hpp"

namespace qbh {

constexpr int kDictCache = 1024;         // LDS cache entries per workgroup (power of two)
constexpr int kDictSlots = 1 << 18;      // global hash slots (power of two, 4x the largest dictionary)
constexpr int kDictMax = 65536;          // 2-byte codes

// device view of the global table.  flags: [0] overflow (> cap distinct), [1] entries claimed,
// [2] unused, [3] a value met while encoding is not in the dictionary
struct DictTab {
    unsigned long long *fp;              // [kDictSlots] fingerprint, 0 = empty
    d2 *val;                             // [kDictSlots] the value that claimed the slot
    uint32_t *code;                      // [kDictSlots] its code (valid after dict_build_finalize)
    int *flags;
    int cap;
};

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

__device__ __forceinline__ int dict_slot0(unsigned long long f) { return (int)((f >> 20) & (unsigned long long)(kDictSlots - 1)); }

// ---- collection ----
struct DictCollect {
    unsigned long long seen[kDictCache];
};

__device__ __forceinline__ void dict_collect_init(DictCollect &D)
{
    for (int i = threadIdx.x; i < kDictCache; i += blockDim.x) D.seen[i] = 0ULL;
    __syncthreads();
}

// returns false once the table has overflowed (the caller may stop feeding it)
__device__ __forceinline__ bool dict_collect_insert(DictCollect &D, const DictTab &T, d2 v)
{
    const unsigned long long f = dict_fp(v);
    const int c = (int)(f & (unsigned long long)(kDictCache - 1));
    if (((volatile unsigned long long *)D.seen)[c] == f) return true;
    if (((volatile int *)T.flags)[0]) return false;
    int s = dict_slot0(f);
    for (int probe = 0; probe < kDictSlots; ++probe) {
        const unsigned long long old = atomicCAS(&T.fp[s], 0ULL, f);
        if (old == 0ULL) {
            T.val[s] = v;
            if (atomicAdd(&T.flags[1], 1) >= T.cap) T.flags[0] = 1;
            break;
        }
        if (old == f) break;
        s = (s + 1) & (kDictSlots - 1);
    }
    ((volatile unsigned long long *)D.seen)[c] = f;
    return true;
}

// ---- encoding ----
// cache word: fingerprint with its low 16 bits replaced by the code (one 8-byte LDS store, so a
// reader never pairs one value's fingerprint with another's code)
struct DictEncode {
    unsigned long long hit[kDictCache];
};

__device__ __forceinline__ void dict_encode_init(DictEncode &E)
{
    for (int i = threadIdx.x; i < kDictCache; i += blockDim.x) E.hit[i] = 0ULL;
    __syncthreads();
}

// code of v; a value that is not in the dictionary (bitwise) raises flags[3] and codes as 0
__device__ __forceinline__ uint32_t dict_encode_one(DictEncode &E, const DictTab &T, const d2 *dict, d2 v)
{
    const unsigned long long f = dict_fp(v);
    const int c = (int)((f >> 16) & (unsigned long long)(kDictCache - 1));
    const unsigned long long e = ((volatile unsigned long long *)E.hit)[c];
    int code = -1;
    if (e != 0ULL && ((e ^ f) >> 16) == 0ULL) {
        code = (int)(e & 0xFFFFULL);
    } else {
        int s = dict_slot0(f);
        for (int probe = 0; probe < kDictSlots; ++probe) {
            const unsigned long long g = T.fp[s];
            if (g == f) {
                code = (int)T.code[s];
                break;
            }
            if (g == 0ULL) break;
            s = (s + 1) & (kDictSlots - 1);
        }
        if (code >= 0) ((volatile unsigned long long *)E.hit)[c] = (f & ~0xFFFFULL) | (unsigned long long)code;
    }
    bool ok = code >= 0;
    if (ok) {
        const d2 w = dict[code];
        ok = __double_as_longlong(w.x) == __double_as_longlong(v.x) && __double_as_longlong(w.y) == __double_as_longlong(v.y);
    }
    if (!ok) {
        T.flags[3] = 1;
        code = 0;
    }
    return (uint32_t)code;
}

// ---- host side (qbh_kernels.hip) ----
struct DictBuild {
    DictTab tab{nullptr, nullptr, nullptr, nullptr, 0};
};
// cap: the largest dictionary the caller can use (256 or kDictMax)
int dict_build_begin(DictBuild *b, int cap, hipStream_t s);
// order the collected values; *d_dict_out = new device array of max(n, kDictLds) entries (zero padded), *n_out = n;
// n = 0 (and no array) when there are more than cap distinct values
int dict_build_finalize(DictBuild *b, d2 **d_dict_out, int *n_out, hipStream_t s);
// 1 when an encode pass met a value outside the dictionary
int dict_build_mismatch(DictBuild *b, int *bad, hipStream_t s);
void dict_build_end(DictBuild *b);
inline int dict_code_width(int n_dict) { return n_dict <= 256 ? 1 : 2; }

}  // namespace qbh
