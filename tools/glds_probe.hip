// glds_probe.hip -- can a producer wavefront that streams with LDS-DMA (global_load_lds, no destination VGPRs) keep the
// fabric busy while consumer wavefronts gather?  (measurement tool, not product)
// A workgroup is NCONS consumer wavefronts + NPROD producer wavefronts.  The producers copy "row blocks" of NPB
// (value 16 B, column 4 B) pairs into one of NBUF LDS buffers; the consumers read every staged pair once
// (ds_read_b128 + ds_read_b32), optionally gather x[col & mask] (16 B) for it, and accumulate.  One barrier per block.
// build: hipcc -O3 --offload-arch=gfx950 tools/glds_probe.hip -o tools/glds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NPB, int NPROD, int NBUF, int GATHER>
__global__ __launch_bounds__(256 + 64 * NPROD) void k_probe(const d2 *val, const int *col, const d2 *x, int mask, int64_t nblk, d2 *out)
{
    constexpr int BUF = NPB * 20;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * BUF];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t first = blockIdx.x, step = gridDim.x;
    if (wave >= 4) {
        const int pw = wave - 4;
        int64_t j = 0;
        // prologue: NBUF - 1 blocks ahead
        for (int64_t b = first; b < nblk; b += step, ++j) {
            unsigned char *buf = lds + (j % NBUF) * BUF;
            const d2 *vp = val + b * NPB;
            const int *cp = col + b * NPB;
#pragma unroll
            for (int k = pw; k < NPB / 64; k += NPROD)
                __builtin_amdgcn_global_load_lds((gvoid *)(vp + k * 64 + lane), (lvoid *)(buf + k * 1024), 16, 0, 2);
#pragma unroll
            for (int k = pw; k < NPB / 256; k += NPROD)
                __builtin_amdgcn_global_load_lds((gvoid *)(cp + k * 256 + lane * 4), (lvoid *)(buf + NPB * 16 + k * 1024), 16, 0, 2);
            if (NBUF == 2) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                // leave one block (this one) in flight across the barrier: wait for the previous block only
                constexpr int PER = (NPB / 64 + NPROD - 1) / NPROD + (NPB / 256 + NPROD - 1) / NPROD;
                static_assert(PER < 64, "vmcnt range");
                if (j > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                asm volatile("s_barrier" ::: "memory");
            }
        }
        if (NBUF > 2) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        d2 s = {0.0, 0.0};
        int64_t j = 0;
        if (NBUF > 2) asm volatile("s_barrier" ::: "memory");
        for (int64_t b = first; b < nblk; b += step, ++j) {
            asm volatile("s_barrier" ::: "memory");
            unsigned char *buf = lds + (j % NBUF) * BUF;
            const d2 *sv = (const d2 *)buf;
            const int *sc = (const int *)(buf + NPB * 16);
            int c[NPB / 256];
            d2 v[NPB / 256], xv[NPB / 256];
#pragma unroll
            for (int u = 0; u < NPB / 256; ++u) c[u] = sc[tid + u * 256];
            if (GATHER) {
#pragma unroll
                for (int u = 0; u < NPB / 256; ++u) xv[u] = x[c[u] & mask];
            }
#pragma unroll
            for (int u = 0; u < NPB / 256; ++u) v[u] = sv[tid + u * 256];
#pragma unroll
            for (int u = 0; u < NPB / 256; ++u) {
                if (GATHER) { s.x += v[u].x * xv[u].x - v[u].y * xv[u].y; s.y += v[u].x * xv[u].y + v[u].y * xv[u].x; }
                else        { s.x += v[u].x * (double)c[u]; s.y += v[u].y; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        out[(size_t)blockIdx.x * 256 + tid] = s;
    }
}

// NBUF == 2 protocol: producer stages block j, waits, barrier j; consumers barrier j, read block j; the NEXT barrier (j+1)
// is reached by the consumers after reading block j and by the producer after staging j+1 into the other buffer -- but
// then the producer would overwrite buffer (j+1)%2 = the one being ... no: block j+1 goes to buffer (j+1)%2, block j is
// in buffer j%2.  Consumers read j between barriers j and j+1; producer stages j+1 between barriers j and j+1: disjoint.

template <int NPB, int NPROD, int NBUF, int GATHER>
static void run(const char *name, const d2 *val, const int *col, const d2 *x, int mask, int64_t nblk, d2 *out, int wg_per_cu)
{
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_probe<NPB, NPROD, NBUF, GATHER>, 256 + 64 * NPROD, 0));
    const int use = wg_per_cu > 0 && wg_per_cu < occ ? wg_per_cu : occ;
    const int grid = 256 * use;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<NPB, NPROD, NBUF, GATHER>), dim3(grid), dim3(256 + 64 * NPROD), 0, 0, val, col, x, mask, nblk, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nblk * NPB * 20.0;
    printf("%-44s occ %d used %d  %.3f ms  %.2f TB/s stream  (%.2f Gnnz/s)\n", name, occ, use, ms, bytes / ms * 1e-9, nblk * (double)NPB / ms * 1e-6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int64_t nnz = (argc > 1 ? atoll(argv[1]) : 800) * 1000000LL / 4096 * 4096;   // 800 M pairs = 16 GB
    d2 *val, *x, *out;
    int *col;
    CK(hipMalloc(&val, nnz * sizeof(d2)));
    CK(hipMalloc(&col, nnz * sizeof(int)));
    CK(hipMalloc(&x, (size_t)1 << 30));
    CK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(d2)));
    CK(hipMemset(val, 0, nnz * sizeof(d2)));
    CK(hipMemset(x, 0, (size_t)1 << 30));
    // columns: a multiplicative hash, so that the masked gather is random inside the window
    {
        int *h = (int *)malloc(nnz * sizeof(int));
        uint32_t s = 12345;
        for (int64_t i = 0; i < nnz; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)(s >> 6); }
        CK(hipMemcpy(col, h, nnz * sizeof(int), hipMemcpyHostToDevice));
        free(h);
    }
    printf("stream: %.1f GB (%lld pairs)\n", nnz * 20e-9, (long long)nnz);
    const int m16k = (16 << 10) / 16 - 1, m2m = (2 << 20) / 16 - 1, m1g = (1 << 30) / 16 - 1;
    run<1024, 1, 2, 0>("1024 1prod 2buf stream-only", val, col, x, 0, nnz / 1024, out, 0);
    run<1024, 2, 2, 0>("1024 2prod 2buf stream-only", val, col, x, 0, nnz / 1024, out, 0);
    run<1024, 1, 3, 0>("1024 1prod 3buf stream-only", val, col, x, 0, nnz / 1024, out, 0);
    run<2048, 1, 2, 0>("2048 1prod 2buf stream-only", val, col, x, 0, nnz / 2048, out, 0);
    run<2048, 2, 2, 0>("2048 2prod 2buf stream-only", val, col, x, 0, nnz / 2048, out, 0);
    run<512, 1, 2, 0>("512 1prod 2buf stream-only", val, col, x, 0, nnz / 512, out, 0);
    run<512, 1, 3, 0>("512 1prod 3buf stream-only", val, col, x, 0, nnz / 512, out, 0);
    run<1024, 1, 2, 1>("1024 1prod 2buf gather 16KB window", val, col, x, m16k, nnz / 1024, out, 0);
    run<1024, 1, 2, 1>("1024 1prod 2buf gather 2MB window", val, col, x, m2m, nnz / 1024, out, 0);
    run<1024, 2, 2, 1>("1024 2prod 2buf gather 2MB window", val, col, x, m2m, nnz / 1024, out, 0);
    run<1024, 1, 3, 1>("1024 1prod 3buf gather 2MB window", val, col, x, m2m, nnz / 1024, out, 0);
    run<512, 1, 3, 1>("512 1prod 3buf gather 2MB window", val, col, x, m2m, nnz / 512, out, 0);
    run<2048, 1, 2, 1>("2048 1prod 2buf gather 2MB window", val, col, x, m2m, nnz / 2048, out, 0);
    run<1024, 1, 2, 1>("1024 1prod 2buf gather 1GB (random lines)", val, col, x, m1g, nnz / 1024, out, 0);
    return 0;
}
