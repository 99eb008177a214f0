// mix_probe.hip -- how L2-resident gathers interfere with an HBM stream issued by the same wavefronts (measurement tool).
// Every wavefront loops: NS streaming 16-byte loads per lane (fresh HBM lines, non-temporal), then NG 16-byte loads per lane
// from a 1 MB window (L2 hits) in one of three lane patterns, then waits for everything.  Reports the stream rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/mix_probe.hip -o tools/lab/mix_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PAT: 0 independent lanes | 1 groups of 8 lanes on one 128-byte line | 2 whole wave on 1 KB contiguous
template <int NS, int NG, int PAT>
__global__ __launch_bounds__(256) void k_mix(const d2 *stream, size_t n_stream, const d2 *win, uint64_t mask, int iters, double *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    uint64_t h = (wave * 64 + (PAT == 0 ? lane : PAT == 1 ? (lane >> 3) : 0)) * 0x9E3779B97F4A7C15ull + 777;
    d2 acc = {0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
        const size_t base = (((size_t)it * nwaves + wave) * NS * 64) % (n_stream - NS * 64);
        d2 s[NS > 0 ? NS : 1], g[NG > 0 ? NG : 1];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t e = (h >> 24) & mask;
            if (PAT == 1) e = (e & ~7ull) | (lane & 7);
            if (PAT == 2) e = (e & ~63ull) | lane;
            g[u] = win[e];
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) s[u] = __builtin_nontemporal_load(stream + base + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < NG; ++u) acc += g[u];
#pragma unroll
        for (int u = 0; u < NS; ++u) acc += s[u];
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}
template <int NS, int NG, int PAT>
static void run(const d2 *stream, size_t n_stream, const d2 *win, double *out, int wgs)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs, iters = 400;
    const uint64_t mask = (1u << 20) / 16 - 1;
    k_mix<NS, NG, PAT><<<grid, 256>>>(stream, n_stream, win, mask, iters, out);
    CK(hipEventRecord(e0));
    k_mix<NS, NG, PAT><<<grid, 256>>>(stream, n_stream, win, mask, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double sb = (double)grid * 256 * iters * NS * 16, gl = (double)grid * 256 * iters * NG;
    const char *pn[] = {"independent", "8 per line", "contiguous"};
    printf("  NS %2d NG %2d %-12s wg/cu %d: %7.3f ms  stream %7.1f GB/s  gathers %7.1f G lanes/s\n", NS, NG, pn[PAT], wgs, ms, sb / ms / 1e6,
           gl / ms / 1e6);
}
int main()
{
    d2 *stream, *win; double *out;
    const size_t sbytes = 16ull << 30;
    CK(hipMalloc(&stream, sbytes)); CK(hipMalloc(&win, 1 << 20)); CK(hipMalloc(&out, 64));
    CK(hipMemset(stream, 0, sbytes)); CK(hipMemset(win, 0, 1 << 20));
    const size_t ns = sbytes / 16;
    for (int wgs : {1, 2, 3, 4, 6, 8}) {     // bytes in flight: wgs * 4 wavefronts * NS KB per CU
        run<4, 0, 0>(stream, ns, win, out, wgs);
        run<8, 0, 0>(stream, ns, win, out, wgs);
        run<16, 0, 0>(stream, ns, win, out, wgs);
    }
    for (int wgs : {3}) {
        run<8, 0, 0>(stream, ns, win, out, wgs);
        run<8, 2, 0>(stream, ns, win, out, wgs);
        run<8, 4, 0>(stream, ns, win, out, wgs);
        run<8, 8, 0>(stream, ns, win, out, wgs);
        run<8, 2, 1>(stream, ns, win, out, wgs);
        run<8, 4, 1>(stream, ns, win, out, wgs);
        run<8, 8, 1>(stream, ns, win, out, wgs);
        run<8, 4, 2>(stream, ns, win, out, wgs);
        run<8, 8, 2>(stream, ns, win, out, wgs);
        run<16, 0, 0>(stream, ns, win, out, wgs);
        run<16, 8, 0>(stream, ns, win, out, wgs);
        run<16, 8, 1>(stream, ns, win, out, wgs);
    }
    return 0;
}
