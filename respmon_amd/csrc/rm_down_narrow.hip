// respmon_amd/csrc/rm_down_narrow.hip -- the all-register chains for uint8 / half / float frame buffers (rm_down_chain_u8.h)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

int launch_down_chain_narrow(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S, double *out,
                             hipStream_t s, bool tiny)
{
    DownGeom g8;
    // (float32: four rows = 256 bytes per lane in flight.  Deep chains, whose segments re-read 2 (2^(S+1) - 2) rows each, take one wave
    //  per SIMD -- 1080p x 256 skip 4: two segments per frame 0.378 ms, three 0.416; at skip 2 the halo is 12 rows and more waves win --
    //  720p x 128: six segments 0.139 ms, four 0.151, three 0.180)
    if (make_down_geom_u8(S, h.data(), w.data(), T, g8, tiny, ctx->dbg.dc_segs, dtype == RM_F32 ? (S >= 3 ? 1024 : 1536) : 2048)) {
        g8.prio = ctx->dbg.dc_prio;
        const size_t fs = (size_t)h[0] * w[0];
        const unsigned grid = (unsigned)(((T + 7) / 8) * 8 * g8.strips * g8.segs);
#define RM_REG_CASE(KK, SS, TT, ptr) case SS: hipLaunchKernelGGL((KK<SS, TT>), dim3(grid), dim3(64), narrow_ring_bytes<TT>(), s, ptr, fs, g8, out); break;
#define RM_REG_SWITCH(KK, TT)                                                                               \
        {                                                                                               \
            const TT *f = (const TT *)frames;                                                           \
            switch (S) { RM_REG_CASE(KK, 1, TT, f) RM_REG_CASE(KK, 2, TT, f) RM_REG_CASE(KK, 3, TT, f) default: hipLaunchKernelGGL((KK<4, TT>), dim3(grid), dim3(64), narrow_ring_bytes<TT>(), s, f, fs, g8, out); break; } \
        }
        if (dtype == RM_U8) RM_REG_SWITCH(k_down_chain_u8, uint8_t)
        else if (dtype == RM_F16) RM_REG_SWITCH(k_down_chain_u8, __half)
        else RM_REG_SWITCH(k_down_chain_narrow, float)
#undef RM_REG_SWITCH
#undef RM_REG_CASE
        LAUNCH_CHECK();
        return RM_OK;
    }
    return 1;
}
