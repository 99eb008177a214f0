// region_probe.hip -- does HBM deliver the same rate when every XCD streams its OWN contiguous region (the ordered walk of the
// Kronecker-split passes) as when the whole chip streams one region in lock step (mix_probe's pattern)?  (measurement tool)
// Every wavefront loops: NS non-temporal 16-byte loads per lane (1 KB per instruction) [+ NS 4-byte loads per lane from a second
// array: the column stream] [+ one 16-byte store per 16 lanes: the row sums], waits, accumulates.
// REG 0: block index = it * nwaves + wave (one region) | 1: XCD k = blockIdx & 7 walks the k-th eighth | 2: as 1, but every XCD
// takes its blocks from a per-XCD atomic counter in chunks of 4 (the DynWalk of the kernels)
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/region_probe.hip -o tools/lab/region_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NS, int REG, int COL, int ST>
__global__ __launch_bounds__(256) void k_reg(const d2 *stream, const int *cols, d2 *outv, size_t n_blocks, unsigned int *ctr, double *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int xcd = blockIdx.x & 7;
    const size_t per = n_blocks / 8;
    const size_t wx = (((size_t)(blockIdx.x >> 3)) * 4) + ((threadIdx.x >> 6) & 3), nwx = nwaves / 8;     // wave index inside its XCD
    d2 acc = {0.0, 0.0};
    int cacc = 0;
    size_t it = 0;
    unsigned int chunk = 0;
    if (REG == 2) {
        if (lane == 0) chunk = atomicInc(ctr + xcd * 32, 0xFFFFFFFFu);
        chunk = __builtin_amdgcn_readfirstlane(chunk);
    }
    for (;; ++it) {
        size_t blk;
        if (REG == 0) {
            blk = it * nwaves + wave;
            if (blk >= n_blocks) break;
        } else if (REG == 1) {
            const size_t b = it * nwx + wx;
            if (b >= per) break;
            blk = xcd * per + b;
        } else if (REG == 3 || REG == 4) {            // static, but a wavefront takes CH consecutive blocks (the kernels' chunks): 3: CH 4, 4: CH 16
            constexpr size_t CH = REG == 3 ? 4 : 16;
            const size_t b = (it / CH) * nwx * CH + wx * CH + it % CH;
            if (b >= per) break;
            blk = xcd * per + b;
        } else {
            if ((it & 3) == 0 && it > 0) {
                if (lane == 0) chunk = atomicInc(ctr + xcd * 32, 0xFFFFFFFFu);
                chunk = __builtin_amdgcn_readfirstlane(chunk);
            }
            const size_t b = (size_t)chunk * 4 + (it & 3);
            if (b >= per) break;
            blk = xcd * per + b;
        }
        const size_t base = blk * NS * 64;
        d2 s[NS];
        int c[NS];
        d2 yold = {0.0, 0.0};
        if (ST == 5 && (lane & 1) == 0) yold = outv[blk * 32 + (lane >> 1)];
#pragma unroll
        for (int u = 0; u < NS; ++u) s[u] = __builtin_nontemporal_load(stream + base + u * 64 + lane);
        if (COL == 1) {
#pragma unroll
            for (int u = 0; u < NS; ++u) c[u] = __builtin_nontemporal_load(cols + base + u * 64 + lane);
        }
        if (COL == 2) {                      // the same column bytes as NS / 4 16-byte loads per lane (block-transposed column storage)
            typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int u = 0; u < NS / 4; ++u) {
                const i4 q = __builtin_nontemporal_load(reinterpret_cast<const i4 *>(cols + base + u * 256) + lane);
                c[4 * u] = q.x; c[4 * u + 1] = q.y; c[4 * u + 2] = q.z; c[4 * u + 3] = q.w;
            }
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) acc += s[u];
        if (COL) {
#pragma unroll
            for (int u = 0; u < NS; ++u) cacc += c[u];
        }
        if (ST == 1 && (lane & 1) == 0) outv[blk * 32 + (lane >> 1)] = acc;       // 32 row sums of 16 bytes per block
        if (ST == 2 && (lane & 1) == 0) __builtin_nontemporal_store(acc, outv + blk * 32 + (lane >> 1));
        if (ST == 3 && (it & 3) == 3) {                                            // the same bytes, every 4th block: 2 x 1 KB
            outv[(blk & ~(size_t)3) * 32 + lane] = acc;
            outv[(blk & ~(size_t)3) * 32 + 64 + lane] = acc;
        }
        if (ST == 4 && (lane & 1) == 0) outv[(wave * 64 + (it & 63)) * 32 + (lane >> 1)] = acc;      // a small, cache-resident target
        if (ST == 5) {                                                              // read-modify-write target read with the stream
            if ((lane & 1) == 0) outv[blk * 32 + (lane >> 1)] = acc + yold;
        }
    }
    if (acc.x == 12345.678 || cacc == 77) out[0] = acc.y;
}
template <int NS, int REG, int COL, int ST>
static void run(const d2 *stream, const int *cols, d2 *outv, size_t n_elems, unsigned int *ctr, double *out, int wgs)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs;
    const size_t n_blocks = (n_elems / (NS * 64)) / 8 * 8;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 8 * 32 * sizeof(unsigned int)));
        CK(hipEventRecord(e0));
        k_reg<NS, REG, COL, ST><<<grid, 256>>>(stream, cols, outv, n_blocks, ctr, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)n_blocks * NS * 64 * (16 + (COL ? 4 : 0)) + (ST ? (double)n_blocks * 512 * (ST == 5 ? 2 : 1) : 0);
    const char *rn[] = {"one region, lock step", "own eighth per XCD, static", "own eighth per XCD, counter", "own eighth, 4 blocks per wave", "own eighth, 16 blocks per wave"};
    printf("  NS %2d %-28s cols %d stores %d wg/cu %d: %8.3f ms  %7.1f GB/s\n", NS, rn[REG], COL, ST, wgs, best, bytes / best / 1e6);
}
int main()
{
    d2 *stream, *outv; int *cols; double *out; unsigned int *ctr;
    const size_t sbytes = 32ull << 30;
    const size_t n = sbytes / 16;
    CK(hipMalloc(&stream, sbytes)); CK(hipMalloc(&cols, n * 4)); CK(hipMalloc(&outv, n / 16 * 16 + 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&ctr, 8 * 32 * 4));
    CK(hipMemset(stream, 0, sbytes)); CK(hipMemset(cols, 0, n * 4));
    for (int wgs : {3}) {
        run<8, 0, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 2, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 0, 1, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 1, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 2, 1, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 0, 1, 1>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 1, 1>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 2, 1, 1>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 1>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 2>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 3>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 4>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 0, 5>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 1, 2>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 1, 3>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 3, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 4, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 3, 1, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 3, 1, 3>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 2, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 2, 3>(stream, cols, outv, n, ctr, out, wgs);
        run<8, 1, 2, 1>(stream, cols, outv, n, ctr, out, wgs);
        run<16, 1, 0, 0>(stream, cols, outv, n, ctr, out, wgs);
        run<16, 1, 0, 1>(stream, cols, outv, n, ctr, out, wgs);
    }
    return 0;
}
