// respmon_amd/csrc/rm_calibrate.hip -- rm_calibrate, the frame-sharded stages and the materialising eulerian_magnification_bandpass
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// nothing is filtered (skip >= levels - 1): the band-passed pyramid, raw and the heatmap are all zeros.  The
// heatmap extrema (0, 0) go into the state like after a real calibration, so rm_locate normalises 0/0 -> NaN
// -> uint8 0 -> no contour, as the reference does (base.py:563-570).
int zero_result(rm_ctx *ctx, size_t npix, double *heat, double *minmax_host, hipStream_t s)
{
    ctx->state_fresh = false;
    HIP_TRY(hipMemsetAsync(heat, 0, sizeof(double) * npix, s));
    hipLaunchKernelGGL(k_heat_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_heat_minmax<>, dim3(1), dim3(256), 0, s, (const double *)heat, (size_t)1, ctx->d_state);
    LAUNCH_CHECK();
    if (minmax_host) { minmax_host[0] = 0.0; minmax_host[1] = 0.0; HIP_TRY(stream_wait(s)); }
    return RM_OK;
}

int calibrate_impl(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin,
                          double fmax, double amp, int levels, int skip, double thr, unsigned flags, double *heat,
                          double *minmax_host, void *stream, CollapsePlan *plan_out)
{
    if (!ctx || !frames || !heat || T < 1 || H < 1 || W < 1 || levels < 1 || skip < 0 || !(fps > 0) || !valid_buffer_dtype(dtype))
        return fail(RM_E_BADARG, "rm_calibrate: bad argument");
    if (T > MAX_T) return fail(RM_E_UNSUPPORTED, "rm_calibrate: T=%d > %d", T, MAX_T);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const size_t npix = (size_t)H * W;
    ctx->nkept_H = ctx->nkept_W = 0;   // set again by collapse_sum when the tile bookkeeping of this call exists
    SmallLevels sl;
    RM_TRY(front_half(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, flags, sl, s));
    if (ctx->prof_on) ctx->prof_calls++;
    if (sl.all_zero) return zero_result(ctx, npix, heat, minmax_host, s);
    CollapseState *st = ctx->d_state;
    PhaseTimer *pt_collapse = new PhaseTimer(ctx, 2, s);
    struct Guard { PhaseTimer *&p; ~Guard() { delete p; p = nullptr; } } guard{pt_collapse};
    CollapsePlan cp;
    RM_TRY(collapse_eval(ctx, sl, T, 0, T, thr, flags, cp, s));
    RM_TRY(collapse_sum(ctx, cp, thr, heat, s, T, plan_out != nullptr));   // time average and heatmap extrema ride the sum kernel
    if (plan_out) *plan_out = cp;
    delete pt_collapse; pt_collapse = nullptr;
    if (minmax_host) {
        HIP_TRY(hipMemcpyAsync(ctx->h_state, st, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
        HIP_TRY(stream_wait(s));
        minmax_host[0] = ctx->h_state->min_val;
        minmax_host[1] = ctx->h_state->max_val;
    }
    return RM_OK;
}

extern "C" int rm_calibrate(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin,
                            double fmax, double amp, int levels, int skip, double thr, unsigned flags, double *heat,
                            double *minmax_host, void *stream)
{
    return calibrate_impl(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, thr, flags, heat, minmax_host, stream, nullptr);
}

// ------------------------------------------------------------------------------------------
// frame-sharded calibration (SURVEY 8e "Mode A"): ONE [T,H,W] buffer split by frame index over the ranks.
// The library does the per-rank stages; the caller (respmon_amd/dist.py) runs the three collectives between
// them with torch.distributed (RCCL): all-gather of the small pyramid, all-reduce(MAX) of {-min, max},
// all-reduce(SUM) of the [H,W] heat sum.
// ------------------------------------------------------------------------------------------
extern "C" int rm_shard_layout_flags(int H, int W, int levels, int skip, unsigned flags, size_t *np_out)
{
    if (!np_out || H < 1 || W < 1 || levels < 1 || skip < 0) return fail(RM_E_BADARG, "rm_shard_layout: bad argument");
    PyrGeom pg;
    pyr_geom(H, W, levels, skip, flags, pg);
    *np_out = pg.all_zero ? 0 : pg.NP;
    return RM_OK;
}

extern "C" int rm_shard_layout(int H, int W, int levels, int skip, size_t *np_out) { return rm_shard_layout_flags(H, W, levels, skip, 0, np_out); }

extern "C" int rm_shard_pyramid(rm_ctx *ctx, const void *frames, int dtype, int Tl, int H, int W, int levels, int skip,
                                unsigned flags, double *lap_local, void *stream)
{
    if (!ctx || !frames || !lap_local || Tl < 1 || H < 1 || W < 1 || levels < 1 || skip < 1 || !valid_buffer_dtype(dtype))
        return fail(RM_E_BADARG, "rm_shard_pyramid: bad argument (frame-sharded calibration needs skip_levels_at_top >= 1)");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    PyrGeom pg;
    pyr_geom(H, W, levels, skip, flags, pg);
    if (pg.all_zero) return RM_OK;  // nothing is filtered: rm_shard_layout reported NP = 0
    return front_pyramid(ctx, frames, dtype, Tl, H, W, pg, flags, lap_local, s);
}

extern "C" int rm_shard_collapse(rm_ctx *ctx, const double *lap_all, int T, int t0, int t1, int H, int W, double fps, double fmin,
                                 double fmax, double amp, int levels, int skip, double thr, unsigned flags, double *negmin_max_dev,
                                 void *stream)
{
    if (!ctx || !negmin_max_dev || T < 1 || t0 < 0 || t1 < t0 || t1 > T || H < 1 || W < 1 || levels < 1 || skip < 1 || !(fps > 0))
        return fail(RM_E_BADARG, "rm_shard_collapse: bad argument");
    if (T > MAX_T) return fail(RM_E_UNSUPPORTED, "rm_shard_collapse: T=%d > %d", T, MAX_T);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    PyrGeom pg;
    pyr_geom(H, W, levels, skip, flags, pg);
    CollapsePlan &cp = ctx->shard_plan;
    cp.valid = false;
    ctx->nkept_H = ctx->nkept_W = 0;
    if (ctx->prof_on) ctx->prof_calls++;
    if (pg.all_zero) {  // band-passed pyramid is all zeros: min = max = 0, heat sum = 0
        cp.T = T; cp.t0 = t0; cp.t1 = t1; cp.H = H; cp.W = W; cp.S = -1; cp.valid = true;
        HIP_TRY(hipMemsetAsync(negmin_max_dev, 0, 2 * sizeof(double), s));
        return RM_OK;
    }
    if (!lap_all) return fail(RM_E_BADARG, "rm_shard_collapse: lap_all is NULL");
    SmallLevels sl;
    RM_TRY(front_filter(ctx, lap_all, T, pg, fps, fmin, fmax, amp, sl, s));
    PhaseTimer pt(ctx, 2, s);
    RM_TRY(collapse_eval(ctx, sl, T, t0, t1, thr, flags, cp, s));
    hipLaunchKernelGGL(k_export_minmax<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state, negmin_max_dev);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_shard_heat(rm_ctx *ctx, const double *negmin_max_dev, double thr, double *heat_sum, void *stream)
{
    if (!ctx || !negmin_max_dev || !heat_sum) return fail(RM_E_BADARG, "rm_shard_heat: bad argument");
    const CollapsePlan &cp = ctx->shard_plan;
    if (!cp.valid) return fail(RM_E_BADARG, "rm_shard_heat: no rm_shard_collapse result on this context");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    if (cp.S < 0) { HIP_TRY(hipMemsetAsync(heat_sum, 0, sizeof(double) * (size_t)cp.H * cp.W, s)); return RM_OK; }
    PhaseTimer pt(ctx, 2, s);
    hipLaunchKernelGGL(k_import_minmax<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state, negmin_max_dev);
    LAUNCH_CHECK();
    return collapse_sum(ctx, cp, thr, heat_sum, s);
}

extern "C" int rm_shard_finish(rm_ctx *ctx, const double *heat_sum, int T, int H, int W, int threshold, double *heatmap,
                               int32_t *xywh, void *stream)
{
    if (!ctx || !heat_sum || !heatmap || T < 1 || H < 1 || W < 1) return fail(RM_E_BADARG, "rm_shard_finish: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const size_t npix = (size_t)H * W;
    hipLaunchKernelGGL(k_heat_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
    ctx->state_fresh = false;
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_heat_avg_minmax<>, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, heat_sum, npix, T, heatmap, ctx->d_state);
    LAUNCH_CHECK();
    if (!xywh) return RM_OK;
    return heatmap_to_roi_impl(ctx, heatmap, H, W, threshold, xywh, nullptr, nullptr, stream, true);
}

extern "C" int rm_eulerian_magnification_bandpass(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps,
                                                  double fmin, double fmax, double amp, int levels, int skip, double thr,
                                                  double *masked, double *raw, double *minmax_host, void *stream)
{
    if (!ctx || !frames || T < 1 || H < 1 || W < 1 || levels < 1 || skip < 0 || !(fps > 0) || !valid_buffer_dtype(dtype))
        return fail(RM_E_BADARG, "rm_eulerian_magnification_bandpass: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const size_t n = (size_t)T * H * W;
    SmallLevels sl;
    RM_TRY(front_half(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, 0, sl, s));
    if (sl.all_zero) {
        if (masked) HIP_TRY(hipMemsetAsync(masked, 0, sizeof(double) * n, s));
        if (raw) HIP_TRY(hipMemsetAsync(raw, 0, sizeof(double) * n, s));
        if (minmax_host) { minmax_host[0] = minmax_host[1] = 0.0; }
        HIP_TRY(stream_wait(s));
        return RM_OK;
    }
    double *raw_buf = raw;
    if (!raw_buf) RM_TRY(ws(ctx, "raw_full", n, &raw_buf));
    // materialised collapse of the all-zero levels below `skip` (pyramid.py:55 with zero levels), for the unique frames; the
    // frames past T / 2 are their mirror images (rm_kernels.h sym_frame)
    const int Th = sym_frames(T);
    const double *cur = sl.cS;
    for (int l = sl.S - 1; l >= 0; --l) {
        double *dst = nullptr;
        if (l == 0) dst = raw_buf;
        else RM_TRY(ws(ctx, (l & 1) ? "collapse_a" : "collapse_b", (size_t)Th * sl.h[l] * sl.w[l], &dst));
        RM_TRY(launch_pyr_up(cur, Th, sl.h[l + 1], sl.w[l + 1], dst, sl.h[l], sl.w[l], 0, nullptr, s));
        cur = dst;
    }
    if (sl.S == 0) HIP_TRY(hipMemcpyAsync(raw_buf, sl.cS, sizeof(double) * (size_t)Th * H * W, hipMemcpyDeviceToDevice, s));
    if (T > Th) {
        hipLaunchKernelGGL(k_mirror_frames<>, dim3(nblk((size_t)H * W, 256, 1024), (unsigned)(T - Th)), dim3(256), 0, s, raw_buf, T, (size_t)H * W);
        LAUNCH_CHECK();
    }
    CollapseState *st = ctx->d_state;
    hipLaunchKernelGGL(k_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_plain<>, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, raw_buf, n, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_minmax<>, dim3(1), dim3(NSTRIPE), 0, s, st, thr);
    LAUNCH_CHECK();
    if (masked) {
        hipLaunchKernelGGL(k_mask_plain<>, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, raw_buf, n, st, masked);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(ctx->h_state, st, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    if (minmax_host) { minmax_host[0] = ctx->h_state->min_val; minmax_host[1] = ctx->h_state->max_val; }
    return RM_OK;
}

