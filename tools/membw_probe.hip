// membw_probe.hip -- what the memory system of one MI355X gives a read-mostly kernel (measurement tool, not product).
//   1. read-only streaming bandwidth vs buffer size (32 MB .. 16 GB): HBM ceiling for reads and what a buffer that
//      fits the 256 MB Infinity Cache gets;
//   2. the same with non-temporal loads;
//   3. a "stream + hot window" mix: every workgroup alternates between a private streamed chunk (never reused) and
//      a shared window of W bytes that all workgroups re-read -- does the window stay in the Infinity Cache while
//      the stream flows, and what do those hits cost compared with HBM misses?
// build: hipcc -O3 --offload-arch=gfx950 tools/membw_probe.hip -o tools/membw_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const d2 *p, size_t n, double *out)
{
    d2 acc = {0.0, 0.0};
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
            else v[u] = d2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}

__global__ __launch_bounds__(256) void k_copy(const d2 *p, d2 *q, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) q[i] = p[i];
}

// stream + window: per iteration a workgroup reads 4 KB * SR of its private stream (nt) and 4 KB of the window at a
// pseudo-random 4 KB-aligned offset (the same sequence of offsets is visited by every workgroup, shifted, so each
// window line is re-read ~ (n_wg * iters * 4 KB / W) times).
template <int SR>
__global__ __launch_bounds__(256) void k_mix(const d2 *stream, size_t n_stream, const d2 *win, size_t n_win_pages, int iters,
                                              double *out)
{
    d2 acc = {0.0, 0.0};
    const size_t per_wg = (size_t)iters * SR * 256;
    const d2 *sp = stream + ((size_t)blockIdx.x * per_wg) % (n_stream - per_wg);
    uint64_t h = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
    for (int it = 0; it < iters; ++it) {
        d2 v[SR];
#pragma unroll
        for (int u = 0; u < SR; ++u) v[u] = __builtin_nontemporal_load(sp + ((size_t)it * SR + u) * 256 + threadIdx.x);
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const size_t page = (h >> 20) % n_win_pages;
        const d2 w = win[page * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < SR; ++u) acc += v[u];
        acc += w;
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv)
{
    size_t max_gb = argc > 1 ? atoi(argv[1]) : 16;
    const size_t max_bytes = max_gb << 30;
    d2 *buf; double *out;
    CK(hipMalloc(&buf, max_bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096, 8192, 16384};
    printf("# read-only sweep: size_MB  plain_GBps  nt_GBps\n");
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        if (bytes > max_bytes) break;
        const size_t n = bytes / 16;
        const int reps = (int)(((size_t)32 << 30) / bytes) < 4 ? 4 : (int)(((size_t)32 << 30) / bytes);
        double gbps[2];
        for (int nt = 0; nt < 2; ++nt) {
            for (int w = 0; w < 2; ++w) { if (nt) k_read<true><<<grid, 256>>>(buf, n, out); else k_read<false><<<grid, 256>>>(buf, n, out); }
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) { if (nt) k_read<true><<<grid, 256>>>(buf, n, out); else k_read<false><<<grid, 256>>>(buf, n, out); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            gbps[nt] = (double)bytes * reps / time_ms(e0, e1) / 1e6;
        }
        printf("read %6zu MB  %8.1f  %8.1f\n", mb, gbps[0], gbps[1]);
    }
    {
        const size_t bytes = (size_t)4 << 30, n = bytes / 16;
        k_copy<<<grid, 256>>>(buf, buf + n, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) k_copy<<<grid, 256>>>(buf, buf + n, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("copy 4096 MB: %.1f GB/s (read+write)\n", 2.0 * bytes * 8 / time_ms(e0, e1) / 1e6);
    }
    printf("# mix: SR streamed 4KB pages per 1 window page; window size sweep; GB/s over ALL bytes loaded\n");
    const size_t n_stream = ((size_t)8 << 30) / 16;
    const d2 *win = buf + n_stream;
    const size_t win_mb[] = {2, 16, 64, 128, 192, 256, 512, 2048, 6144};
    for (int sr : {1, 3}) {
        for (size_t wm : win_mb) {
            if (((size_t)8 << 30) + (wm << 20) > max_bytes) break;
            const size_t pages = (wm << 20) / 4096;
            const int iters = 2048 / (sr + 1);
            for (int w = 0; w < 2; ++w) {
                if (sr == 1) k_mix<1><<<grid, 256>>>(buf, n_stream, win, pages, iters, out);
                else         k_mix<3><<<grid, 256>>>(buf, n_stream, win, pages, iters, out);
            }
            CK(hipEventRecord(e0));
            const int reps = 5;
            for (int r = 0; r < reps; ++r) {
                if (sr == 1) k_mix<1><<<grid, 256>>>(buf, n_stream, win, pages, iters, out);
                else         k_mix<3><<<grid, 256>>>(buf, n_stream, win, pages, iters, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            const double bytes = (double)grid * iters * (sr + 1) * 4096.0 * reps;
            printf("mix SR=%d window %5zu MB: %8.1f GB/s total  (stream share %.0f%%)\n", sr, wm, bytes / time_ms(e0, e1) / 1e6,
                   100.0 * sr / (sr + 1));
        }
    }
    return 0;
}
