// bench_micro/dc8_bench.hip -- developer microbenchmark of the narrow-dtype pyrDown chain (rm_down_chain_u8.h) alone.
// Variants via -D (RM_NARROW_WAVES, RM_NARROW_HOT, RM_NARROW_FENCE, RM_U8_PREFETCH, RM_F16_PREFETCH) or -DRM_DC8_HEADER="..." for
// another copy of the header; prints ms per launch and a checksum of the output (equal checksums = equal bits).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../respmon_amd/csrc/rm_kernels.h"
#ifdef RM_DC8_HEADER
#include "../respmon_amd/csrc/rm_down_chain.h"
#include RM_DC8_HEADER
#else
#include "../respmon_amd/csrc/rm_down_chain_u8.h"
#endif
using namespace rm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename Tin> __global__ void k_fill(Tin *p, size_t n, int W, int H)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), t = (int)(i / ((size_t)W * H));
        const int lvl = 128 + (int)(60.0f * __sinf(x * 0.01f) * __cosf(y * 0.013f)) + (int)((((unsigned)i * 2654435761u) >> 13) % 5) + (t & 3);
        if constexpr (sizeof(Tin) == 1) p[i] = (Tin)lvl; else p[i] = (Tin)(float)(lvl * (1.0 / 255));
    }
}
__global__ __launch_bounds__(256) void k_sum(const double *p, size_t n, double *out)   // deterministic: fixed partition, fixed order
{
    __shared__ double part[256];
    double s = 0;
    for (size_t i = threadIdx.x; i < n; i += 256) s += p[i] * (1 + (i % 7));
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 256; ++i) t += part[i]; *out = t; }
}

#ifdef RM_DC8_HEADER
#define RING_BYTES(Tin) 0
#define KERNEL_OF(S, Tin) k_down_chain_u8<S, Tin>
#else
template <int S, typename Tin> struct KernelOf { static constexpr auto fn = k_down_chain_u8<S, Tin>; };
template <int S> struct KernelOf<S, float> { static constexpr auto fn = k_down_chain_narrow<S, float>; };
#define KERNEL_OF(S, Tin) KernelOf<S, Tin>::fn
#define RING_BYTES(Tin) narrow_ring_bytes<Tin>()
#endif

// the chain's access pattern alone: every wave reads the rows of its (frame, strip, segment) -- 16 pixels per lane and row -- and
// does nothing with them; DEPTH rows in flight
template <int S, typename Tin, int DEPTH> __global__ __launch_bounds__(64) void k_pattern(const Tin *frames, size_t frame_stride, DownGeom g, unsigned *sink)
{
    constexpr int NLD = RegTraits<Tin>::NLD, VPER = 16 / NLD;
    const int per_frame = g.strips * g.segs;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = (j / per_frame) * 8 + xcd;
    if (t >= g.T) return;
    const int inner = j % per_frame;
    const int seg = inner / g.strips, strip = inner - seg * g.strips;
    const int W = g.w[0], lane = threadIdx.x;
    int y0 = seg * g.seg_h, y1 = min(y0 + g.seg_h, g.h[S]) - 1;
    int r0 = max(0, (y0 << S) - ((1 << (S + 1)) - 2)), r1 = min(g.h[0] - 1, (y1 << S) + ((1 << (S + 1)) - 2));
    const int c_first = strip * U8_STRIP_PX - 32 + 16 * lane;
    const Tin *src = frames + (size_t)t * frame_stride + min(max(c_first, 0), W - 16);
    unsigned acc = 0;
    for (int r = r0; r <= r1; r += DEPTH) {
        Raw16 v[DEPTH][NLD];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < NLD; ++q) v[d][q] = *reinterpret_cast<const Raw16 *>(src + (size_t)min(r + d, r1) * W + q * VPER);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < NLD; ++q) acc ^= v[d][q].x ^ v[d][q].y ^ v[d][q].z ^ v[d][q].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int S, typename Tin> void run_pattern(int T, int H, int W, const char *name)
{
    std::vector<int> h(S + 1), w(S + 1);
    h[0] = H; w[0] = W;
    for (int k = 1; k <= S; ++k) { h[k] = (h[k - 1] + 1) / 2; w[k] = (w[k - 1] + 1) / 2; }
    size_t n = (size_t)T * H * W;
    Tin *src; unsigned *sink;
    CK(hipMalloc(&src, n * sizeof(Tin))); CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill<Tin>, dim3(4096), dim3(256), 0, 0, src, n, W, H);
    DownGeom g;
    if (!make_down_geom_u8(S, h.data(), w.data(), T, g, false)) return;
    const unsigned grid = (unsigned)(((T + 7) / 8) * 8 * g.strips * g.segs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto kern, const char *label) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, src, (size_t)H * W, g, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, src, (size_t)H * W, g, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("pattern %-18s %s : %.3f ms  %.0f GB/s\n", name, label, ms, n * sizeof(Tin) / (ms * 1e-3) / 1e9);
    };
    time(k_pattern<S, Tin, 2>, "2 rows in flight");
    time(k_pattern<S, Tin, 4>, "4 rows in flight");
    time(k_pattern<S, Tin, 8>, "8 rows in flight");
    CK(hipFree(src)); CK(hipFree(sink));
}

template <int S, typename Tin> void run(int T, int H, int W, const char *name, int segs_override)
{
    std::vector<int> h(S + 1), w(S + 1);
    h[0] = H; w[0] = W;
    for (int k = 1; k <= S; ++k) { h[k] = (h[k - 1] + 1) / 2; w[k] = (w[k - 1] + 1) / 2; }
    size_t n = (size_t)T * H * W, no = (size_t)T * h[S] * w[S];
    Tin *src; double *dst, *d_sum;
    CK(hipMalloc(&src, n * sizeof(Tin)));
    CK(hipMalloc(&dst, no * sizeof(double)));
    CK(hipMalloc(&d_sum, 8));
    hipLaunchKernelGGL(k_fill<Tin>, dim3(4096), dim3(256), 0, 0, src, n, W, H);
    CK(hipMemset(dst, 0, no * sizeof(double)));
    DownGeom g;
    if (!make_down_geom_u8(S, h.data(), w.data(), T, g, false)) { printf("geometry refused\n"); return; }
    if (segs_override > 0) { g.seg_h = (h[S] + segs_override - 1) / segs_override; g.segs = (h[S] + g.seg_h - 1) / g.seg_h; }
    const unsigned grid = (unsigned)(((T + 7) / 8) * 8 * g.strips * g.segs);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((KERNEL_OF(S, Tin)), dim3(grid), dim3(64), RING_BYTES(Tin), 0, src, (size_t)H * W, g, dst);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    float best = 1e9f, tot = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((KERNEL_OF(S, Tin)), dim3(grid), dim3(64), RING_BYTES(Tin), 0, src, (size_t)H * W, g, dst);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    CK(hipMemset(d_sum, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, dst, no, d_sum);
    double sum; CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
    printf("%-22s S=%d strips=%d segs=%d grid=%u : avg %.3f ms  best %.3f ms  %.0f GB/s  checksum %.17g\n", name, S, g.strips, g.segs, grid, tot / iters, best,
           n * sizeof(Tin) / (tot / iters * 1e-3) / 1e9, sum);
    CK(hipFree(src)); CK(hipFree(dst)); CK(hipFree(d_sum));
}

int main(int argc, char **argv)
{
    const int segs = argc > 1 ? atoi(argv[1]) : 0;
    run<4, uint8_t>(256, 1080, 1920, "P u8 256x1080p", segs);
    run<4, float>(256, 1080, 1920, "P f32 256x1080p", segs);
    run<2, __half>(512, 2160, 3840, "R f16 512x4K", segs);
    run<2, float>(128, 720, 1280, "Q f32 128x720p", segs);
#ifndef RM_DC8_HEADER
    if (argc > 2) {
        run_pattern<4, uint8_t>(256, 1080, 1920, "P u8");
        run_pattern<4, float>(256, 1080, 1920, "P f32");
        run_pattern<2, __half>(512, 2160, 3840, "R f16");
    }
#endif
    return 0;
}
