// respmon_amd/csrc/rm_down_launch.h -- launcher templates of k_down_chain (rm_down_chain.h), instantiated per frame dtype by
// rm_down_f64.hip (the reference's buffer dtype: the roofline kernel) and rm_down_generic.hip (narrow dtypes through its LDS front end)
#pragma once
#include "rm_internal.h"

// ------------------------------------------------------------------------------------------
// fused Gaussian chain (rm_down_chain.h)
// ------------------------------------------------------------------------------------------
template <typename Tin, bool VB>
static int launch_down_chain_g(const Tin *f, int T, const DownGeom &g, double *out, hipStream_t s)
{
    const size_t fs = (size_t)g.h[0] * g.w[0];
    (void)T;
    const unsigned grid = down_chain_grid(g), block = down_chain_block(g);
#define RM_DC_CASE(SS)                                                                                         \
    case SS:                                                                                                   \
        hipLaunchKernelGGL((k_down_chain<Tin, SS, VB>), dim3(grid), dim3(block),                               \
                           (sizeof(double) * down_chain_lds_doubles<Tin, SS>() * g.wpg), s, f, fs, g, out);    \
        break;
    switch (g.S) {
        RM_DC_CASE(1) RM_DC_CASE(2) RM_DC_CASE(3) RM_DC_CASE(4) RM_DC_CASE(5)
    default: return fail(RM_E_UNSUPPORTED, "fused pyrDown chain supports 1..5 levels, got %d", g.S);
    }
#undef RM_DC_CASE
    LAUNCH_CHECK();
    return RM_OK;
}

// one launch over all level-S rows: the hot instantiation whenever every level has >= 3 rows
template <typename Tin>
static int launch_down_chain_t(rm_ctx *ctx, const void *frames, int T, const std::vector<int> &h, const std::vector<int> &w, int S, int vec_ok,
                               double *out, hipStream_t s, bool tiny)
{
    const Tin *f = (const Tin *)frames;
    DownGeom g;
    if (!make_down_geom(S, h.data(), w.data(), T, vec_ok, 0, h[S], g, tiny, ctx->dbg.dc_segs, ctx->dbg.dc_wpg, ctx->dbg.dc_split)) return fail(RM_E_UNSUPPORTED, "down chain geometry");
    g.prio = ctx->dbg.dc_prio;
    if (ctx->dbg.dc_prio_shift >= 1 && ctx->dbg.dc_prio_shift <= 12) g.prio_shift = ctx->dbg.dc_prio_shift;
    if (down_chain_hot_ok(S, h.data())) return launch_down_chain_g<Tin, false>(f, T, g, out, s);
    return launch_down_chain_g<Tin, true>(f, T, g, out, s);
}


