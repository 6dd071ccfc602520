// bench_micro/pattern_bench.hip -- developer microbenchmark: what read bandwidth does HBM give for
// (a) a linear stream and (b) the strip-march access pattern of the fused pyrDown chain (no compute)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct alignas(16) V4 { unsigned x, y, z, w; };

__global__ __launch_bounds__(256) void k_linear(const V4 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        V4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

// one wave per (frame, strip, segment): rows of `piece16` 16-byte chunks, row pitch `pitch16`, PF rows in flight
template <int PF, int NL>
__global__ __launch_bounds__(64) void k_strips(const V4 *p, size_t frame16, int pitch16, int piece16, int strips, int segs, int rows_per_seg,
                                               int halo_rows, int H, int T, int xcd_map, unsigned *out)
{
    const int per_frame = strips * segs;
    int t, inner;
    if (xcd_map) { int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; t = (j / per_frame) * 8 + xcd; inner = j % per_frame; }
    else { t = blockIdx.x / per_frame; inner = blockIdx.x % per_frame; }
    if (t >= T) return;
    const int seg = inner / strips, strip = inner % strips;
    int r0 = seg * rows_per_seg - halo_rows, r1 = (seg + 1) * rows_per_seg + halo_rows;
    if (r0 < 0) r0 = 0;
    if (r1 > H) r1 = H;
    const V4 *base = p + (size_t)t * frame16 + (size_t)strip * (piece16 - 8) + threadIdx.x;
    V4 regs[PF][NL];
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int q = 0; q < NL; ++q) { int j = threadIdx.x + 64 * q; regs[i][q] = base[(size_t)min(r0 + i, r1 - 1) * pitch16 + (j < piece16 ? 64 * q : 0)]; }
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int q = 0; q < NL; ++q) acc += regs[0][q].x ^ regs[0][q].w;
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
            for (int q = 0; q < NL; ++q) regs[i][q] = regs[i + 1][q];
#pragma unroll
        for (int q = 0; q < NL; ++q) { int j = threadIdx.x + 64 * q; regs[PF - 1][q] = base[(size_t)min(r + PF, r1 - 1) * pitch16 + (j < piece16 ? 64 * q : 0)]; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

int main()
{
    const int T = 256, H = 1080, W = 1920;
    const size_t bytes = (size_t)T * H * W * 8, n16 = bytes / 16;
    V4 *p; unsigned *out;
    CK(hipMalloc(&p, bytes + (1 << 22))); CK(hipMalloc(&out, 64));
    CK(hipMemset(p, 1, bytes));
    for (int blocks : {2048, 8192, 32768}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, p, n16, out); });
        printf("linear read, %d blocks: %.3f ms  %.1f GB/s\n", blocks, ms, bytes / (ms * 1e-3) / 1e9);
    }
    const int pitch16 = W * 8 / 16, frame16 = H * pitch16;
    struct Cfg { int strips, segs; };
    for (Cfg c : {Cfg{6, 4}, Cfg{6, 1}, Cfg{3, 4}, Cfg{1, 8}}) {
        int piece16 = pitch16 / c.strips + 8;  // + halo columns
        int rows_per_seg = (H + c.segs - 1) / c.segs, halo = c.segs > 1 ? 30 : 0;
        unsigned grid = T * c.strips * c.segs;
        for (int xm = 0; xm < 2; ++xm) {
            float ms;
            if (piece16 <= 192) ms = timeit([&] { hipLaunchKernelGGL((k_strips<3, 3>), dim3(grid), dim3(64), 0, 0, p, (size_t)frame16, pitch16, piece16, c.strips, c.segs, rows_per_seg, halo, H, T, xm, out); });
            else if (piece16 <= 384) ms = timeit([&] { hipLaunchKernelGGL((k_strips<3, 6>), dim3(grid), dim3(64), 0, 0, p, (size_t)frame16, pitch16, piece16, c.strips, c.segs, rows_per_seg, halo, H, T, xm, out); });
            else ms = timeit([&] { hipLaunchKernelGGL((k_strips<2, 15>), dim3(grid), dim3(64), 0, 0, p, (size_t)frame16, pitch16, piece16, c.strips, c.segs, rows_per_seg, halo, H, T, xm, out); });
            printf("strips=%d segs=%d piece=%d B xcd_map=%d grid=%u: %.3f ms  %.1f GB/s (algorithmic)\n", c.strips, c.segs, piece16 * 16, xm, grid, ms,
                   bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
