// respmon_amd/csrc/rm_down_bgr.hip -- the all-register chain for [T,H,W,3] uint8 BGR frame buffers (rm_down_chain_u8.h bgr8_t): base.py:230-231's
// cvtColor(BGR2GRAY) + uint8_to_float fused into the first row pass of the Gaussian chain
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

int launch_down_chain_bgr(rm_ctx *ctx, const void *frames, int T, const std::vector<int> &h, const std::vector<int> &w, int S, double *out,
                          hipStream_t s, bool tiny)
{
    DownGeom g8;
    // three times the bytes of a gray row per lane: one wave per SIMD with four rows (192 bytes per lane) in flight beats two waves with two,
    // and half the segments re-read half the halo rows (1080p x 256: two segments per frame, 11 % of the rows twice instead of 22 %)
    // (depth 2 keeps one row in flight -- RegChain::PF_T -- and wants the waves: 720p x 128 0.121 ms with 2048, 0.188 with 1024)
    if (!make_down_geom_u8(S, h.data(), w.data(), T, g8, tiny, ctx->dbg.dc_segs, S == 2 ? 2048 : 1024)) return 1;
    g8.prio = ctx->dbg.dc_prio;
    const size_t fs = (size_t)h[0] * w[0];   // in pixels: the kernel's pointer arithmetic is in bgr8_t
    const unsigned grid = (unsigned)(((T + 7) / 8) * 8 * g8.strips * g8.segs);
    const bgr8_t *f = (const bgr8_t *)frames;
    switch (S) {
    case 1: hipLaunchKernelGGL((k_down_chain_u8<1, bgr8_t>), dim3(grid), dim3(64), 0, s, f, fs, g8, out); break;
    case 2: hipLaunchKernelGGL((k_down_chain_u8<2, bgr8_t>), dim3(grid), dim3(64), 0, s, f, fs, g8, out); break;
    case 3: hipLaunchKernelGGL((k_down_chain_u8<3, bgr8_t>), dim3(grid), dim3(64), 0, s, f, fs, g8, out); break;
    default: hipLaunchKernelGGL((k_down_chain_u8<4, bgr8_t>), dim3(grid), dim3(64), 0, s, f, fs, g8, out); break;
    }
    LAUNCH_CHECK();
    return RM_OK;
}
