// l2_probe.hip -- rate of 16-byte gathers that HIT the XCD's L2 (window far larger than the 32 KB L1, smaller than
// the 4 MB L2), lanes independent or in groups of 8 lanes reading one full 128-byte line (measurement tool).
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/l2_probe.hip -o tools/lab/l2_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int GROUP, int UN>
__global__ __launch_bounds__(256) void k_probe(const d2 *x, uint64_t mask, int iters, double *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t id = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / GROUP;
    uint64_t h = id * 0x9E3779B97F4A7C15ull + 777;
    d2 acc = {0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
        d2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t e = ((h >> 24) & mask);
            v[u] = x[GROUP == 1 ? e : ((e & ~(uint64_t)(GROUP - 1)) | (lane & (GROUP - 1)))];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u];
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}
template <int GROUP, int UN>
static void run(const char *name, const d2 *x, uint64_t mask, double *out, int wgs_per_cu)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu, iters = 2048 / UN;
    k_probe<GROUP, UN><<<grid, 256>>>(x, mask, iters, out);
    CK(hipEventRecord(e0));
    k_probe<GROUP, UN><<<grid, 256>>>(x, mask, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)grid * 256 * iters * UN;
    printf("  %-34s wg/cu %d: %8.2f G lanes/s  %8.2f G lines/s (%7.1f GB/s of lines)\n", name, wgs_per_cu, n / ms / 1e6, n / GROUP / ms / 1e6,
           n / GROUP * 128 / ms / 1e6);
}
int main(int argc, char **argv)
{
    d2 *x; double *out;
    const size_t bytes = 1ull << 30;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(x, 0, bytes));
    for (size_t win : {16384ul, 262144ul, 1048576ul, 2097152ul, 33554432ul, 1ul << 30}) {
        const uint64_t mask = win / 16 - 1;
        printf("window %zu KB (shared by all XCDs)\n", win >> 10);
        for (int w : {4, 8}) {
            run<1, 8>("independent lanes", x, mask, out, w);
            run<8, 8>("8 lanes per 128-byte line", x, mask, out, w);
            run<64, 8>("whole wave: 8 consecutive lines", x, mask, out, w);
        }
    }
    return 0;
}
