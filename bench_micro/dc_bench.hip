// bench_micro/dc_bench.hip -- developer microbenchmark of the fused pyrDown chain kernel alone
// (variants selected with -DRM_DC_SW4=.. -DRM_DC_PREFETCH=..).  Not part of the product or the tests.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../respmon_amd/csrc/rm_down_chain.h"
using namespace rm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename Tin> void run(int T, int H, int W, const char *name)
{
    constexpr int S = RM_BENCH_S;
    std::vector<int> h(S + 1), w(S + 1);
    h[0] = H; w[0] = W;
    for (int k = 1; k <= S; ++k) { h[k] = (h[k - 1] + 1) / 2; w[k] = (w[k - 1] + 1) / 2; }
    size_t n = (size_t)T * H * W;
    Tin *src; double *dst;
    CK(hipMalloc(&src, n * sizeof(Tin)));
    CK(hipMalloc(&dst, (size_t)T * h[S] * w[S] * sizeof(double)));
    {
        std::vector<Tin> hbuf((size_t)H * W);
        // camera-like data (uint8 levels / 255, smooth): switching activity -- and with it the clocks the kernel
        // runs at -- depends on the operand bits; random mantissas measured ~10 % slower than real frames
#ifdef RM_BENCH_RANDOM
        for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = (Tin)((i * 2654435761u % 1000) / 1000.0);
#else
        for (size_t i = 0; i < hbuf.size(); ++i) {
            const int x = (int)(i % W), y = (int)(i / W);
            const int lvl = 128 + (int)(60.0 * sin(x * 0.01) * cos(y * 0.013)) + (int)((i * 2654435761u >> 13) % 5);
            if (sizeof(Tin) == 1) hbuf[i] = (Tin)lvl; else hbuf[i] = (Tin)(lvl * (1.0 / 255));
        }
#endif
        for (int t = 0; t < T; ++t) CK(hipMemcpy(src + (size_t)t * H * W, hbuf.data(), hbuf.size() * sizeof(Tin), hipMemcpyHostToDevice));
    }
    DownGeom g;
    int y0 = 0, y1 = h[S];
    make_down_geom(S, h.data(), w.data(), T, 1, y0, y1, g);
    unsigned grid = down_chain_grid(g), block = down_chain_block(g);
    size_t shmem = sizeof(double) * down_chain_lds_doubles<Tin, S>() * g.wpg;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_down_chain<Tin, S, false>), dim3(grid), dim3(block), shmem, 0, src, (size_t)H * W, g, dst);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_down_chain<Tin, S, false>), dim3(grid), dim3(block), shmem, 0, src, (size_t)H * W, g, dst);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%s rows[%d,%d) SW=%d PF=%d strips=%d segs=%d wpg=%d grid=%u lds=%zuB : %.3f ms  %.1f GB/s\n", name, y0, y1, StripWidth<S>::SW, DC_PREFETCH, g.strips, g.segs, g.wpg,
           grid, shmem, ms, n * sizeof(Tin) / (ms * 1e-3) / 1e9);
    CK(hipFree(src)); CK(hipFree(dst));
}

int main(int argc, char **argv)
{
    int T = 256, H = 1080, W = 1920;
    run<double>(T, H, W, "f64");
    if (argc > 1) { run<float>(T, H, W, "f32"); run<uint8_t>(T, H, W, "u8"); }
    return 0;
}
