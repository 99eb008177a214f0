// gather_probe.hip -- cost of scattered 16-byte (and 8-byte) gathers from a vector much larger than the caches, per
// cache-policy variant of the load instruction (measurement tool, not product).  The momentum-sector SpMV (C5) is
// bound by exactly this: one 128-byte line fetched per 16 useful bytes.  Question: does any policy make the L2 fetch
// less than a full line per miss?
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe.bin
// run:   tools/gather_probe.bin [vector_GB=4] [variant=-1 (all)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__device__ __forceinline__ d2 ld(const d2 *p)
{
    d2 r;
    if (V == 0) return *p;
    if (V == 1) return __builtin_nontemporal_load(p);
    if (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (V == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// every lane issues 8 independent gathers per iteration (the asm variants wait inside ld, so they are issued by 8
// different iterations of an unrolled loop only when the compiler keeps them apart -- to be fair to them the plain
// variants are also measured with UN = 1)
template <int V, int UN>
__global__ __launch_bounds__(256) void k_gather(const d2 *x, uint64_t mask, int iters, double *out)
{
    uint64_t h = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    d2 acc = {0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
        d2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            v[u] = ld<V>(x + ((h >> 24) & mask));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u];
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}

// same with 8-byte elements (the real fast path)
__global__ __launch_bounds__(256) void k_gather8(const double *x, uint64_t mask, int iters, double *out)
{
    uint64_t h = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            v[u] = x[(h >> 24) & mask];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int V, int UN>
static void run(const char *name, const d2 *x, uint64_t mask, double *out, hipEvent_t e0, hipEvent_t e1)
{
    const int grid = 256 * 8, iters = 512 / UN * 4;
    k_gather<V, UN><<<grid, 256>>>(x, mask, iters, out);
    CK(hipEventRecord(e0));
    k_gather<V, UN><<<grid, 256>>>(x, mask, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)grid * 256 * iters * UN;
    printf("%-28s UN=%d  %7.2f Ggather/s  (%6.1f GB/s useful, %7.1f GB/s if 128B lines, %7.1f if 64B)\n", name, UN, n / ms / 1e6,
           n * 16 / ms / 1e6, n * 128 / ms / 1e6, n * 64 / ms / 1e6);
}

int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? atoi(argv[1]) : 4;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const size_t bytes = gb << 30, n = bytes / 16;
    d2 *x; double *out;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(x, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t mask = n - 1;     // gb is a power of two
    printf("# random 16-byte gathers from a %zu GB vector\n", gb);
    if (only < 0 || only == 0) run<0, 8>("plain", x, mask, out, e0, e1);
    if (only < 0 || only == 1) run<1, 8>("nontemporal", x, mask, out, e0, e1);
    if (only < 0 || only == 10) run<0, 1>("plain", x, mask, out, e0, e1);
    if (only < 0 || only == 2) run<2, 1>("sc0", x, mask, out, e0, e1);
    if (only < 0 || only == 3) run<3, 1>("sc1", x, mask, out, e0, e1);
    if (only < 0 || only == 4) run<4, 1>("sc0 sc1", x, mask, out, e0, e1);
    if (only < 0 || only == 5) run<5, 1>("sc0 sc1 nt", x, mask, out, e0, e1);
    if (only < 0 || only == 8) {
        const int grid = 256 * 8, iters = 256;
        k_gather8<<<grid, 256>>>((const double *)x, 2 * n - 1, iters, out);
        CK(hipEventRecord(e0));
        k_gather8<<<grid, 256>>>((const double *)x, 2 * n - 1, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ng = (double)grid * 256 * iters * 8;
        printf("%-28s UN=8  %7.2f Ggather/s\n", "8-byte plain", ng / ms / 1e6);
    }
    return 0;
}
