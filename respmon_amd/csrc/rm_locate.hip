// respmon_amd/csrc/rm_locate.hip -- rm_locate and its two-call form (base.py:547-601)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// kept pairs of a call at skip >= 3 beyond which the next call refines its bounds first (k_bounds_up1, ~13 us at 1080p x 256), and below
// which a refined call turns that off again.  On = what the value store starts with (16 384 slots): a selection that overflows it goes to the
// store-less sum, whose time is the number of kept pairs.  The streams the sparse path serves lose by the refinement (measured, same
// process, on / off: sixteen blobs -- 5 100 pairs -- 0.966 / 0.960 ms, four blobs + noise 0.919 / 0.908; headline, 2 600 pairs: never on).
constexpr int REFINE_ON_PAIRS = 16384, REFINE_OFF_PAIRS = 8192;

extern "C" int rm_locate(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax,
                         double amp, int levels, int skip, double temporal_thr, int threshold, unsigned flags, int32_t *xywh,
                         void *stream)
{
    if (!ctx || !xywh) return fail(RM_E_BADARG, "rm_locate: bad argument");
    ctx->host_enter = std::chrono::steady_clock::now();
    ctx->host_marks[0] = ctx->host_have_return ? std::chrono::duration<double, std::micro>(ctx->host_enter - ctx->host_last_return).count() : 0.0;
    struct ReturnStamp { rm_ctx *c; ~ReturnStamp() { c->host_last_return = std::chrono::steady_clock::now(); c->host_have_return = true; } } stamp{ctx};
    ctx->cur_slot = 0;
    RoiSlot &rs = ctx->slots[0];
    if (rs.h_unserved) *rs.h_unserved = 0;   // (a failed earlier call must not leave its "dense sum wanted" behind)
    double *heat = nullptr;
    RM_TRY(ws(ctx, "heat", (size_t)H * W, &heat));
    CollapsePlan cp;
    RM_TRY(calibrate_impl(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, temporal_thr, flags, heat, nullptr, stream, &cp));
    const bool clip_once = (flags & RM_FLAG_CONTOUR_CLIP_FRAME) != 0;
    ctx->clip_frame_once = clip_once;
    ctx->tiles_const_once = cp.valid && cp.S >= 1;
    int rc = heatmap_to_roi_impl(ctx, heat, H, W, threshold, xywh, nullptr, nullptr, stream, true);
    const int unserved_word = rs.h_unserved ? *rs.h_unserved : 0;   // 1: the sparse kernel stood down and nothing took the sum; 2: the stand-in did
    const bool unserved = unserved_word == 1;
    // how many pairs did this call's selection keep (skip >= 3 with a capped value store: the sparse sum kernel leaves the count beside the
    // word)?  Many: the next call of the context refines its bounds one level down before it selects (k_bounds_up1: ~10 us that the
    // headline stream -- 2 600 pairs -- must not pay); on until a REFINED selection keeps few
    if (rs.h_unserved && cp.valid && cp.S >= 3) ctx->refine_hint = rs.h_unserved[1] > (ctx->refine_hint ? REFINE_OFF_PAIRS : REFINE_ON_PAIRS) ? 1 : 0;
    if (rs.h_unserved) *rs.h_unserved = 0;
    if (ctx->dense_hint && unserved_word != 2) ctx->dense_hint = 0;   // (the stand-in enqueued on the hint was not needed: back to the plain path)
    if (rc >= 0 && cp.valid && unserved) {
        // the selection kept more pairs than the value store holds and the sparse sum kernel stood down (the ROI stage above ran on
        // a heatmap nobody wrote -- the price of not putting a host synchronisation in front of the ROI stage of EVERY call, which
        // is what looking at the flag first would take): take the sum with the dense kernel, now that the stream is idle, and
        // extract the ROI again
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(k_heat_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
        LAUNCH_CHECK();
        // how many pairs did the selection keep?  A store that holds them (up to STORE_MAX_SLOTS) is allocated -- for this call and the
        // later ones of the context -- and the evaluation + sum run again through it; beyond that the store-less sum takes over
        HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
        HIP_TRY(stream_wait(s));
        const long long kept = (long long)ctx->h_state->n_slots;
        CollapsePlan again = cp;
        const bool dense_sel = kept * 4 > (long long)cp.sp.npairs_mine;   // a dense selection: the store would be most of the materialised video
        if (dense_sel) ctx->dense_hint = 1;
        if (!dense_sel && kept <= STORE_MAX_SLOTS && !(flags & RM_FLAG_TINY_STORE) && ctx->dbg.store_slots <= 0 && cp.sp.mode == 0) {
            const long long want = std::min(STORE_MAX_SLOTS, kept + kept / 8 + 64);
            ctx->store_hint_slots = std::max(ctx->store_hint_slots, want);
            RM_TRY(ws(ctx, "value_store", (size_t)want * CT_H * CT_W, &again.store));
            again.sp.cap_slots = (unsigned)std::min<long long>(want, (long long)again.sp.npairs_mine);
            ctx->dbg_cap = again.sp.cap_slots;
            RM_TRY(launch_eval_pairs(ctx, again, s));
            RM_TRY(collapse_sum(ctx, again, temporal_thr, heat, s, T));
        } else {
            again.sp.mode = 1;
            RM_TRY(collapse_sum(ctx, again, temporal_thr, heat, s, T));
        }
        ctx->clip_frame_once = clip_once;
        ctx->tiles_const_once = true;
        rc = heatmap_to_roi_impl(ctx, heat, H, W, threshold, xywh, nullptr, nullptr, stream, true);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// rm_locate in two calls: rm_locate_submit enqueues everything up to the packed thresholded image and returns; rm_locate_result
// waits for it and runs the host contour stage.  Between the two the caller may submit the NEXT buffer (two tickets per context),
// so its frame-buffer kernel runs while the host follows the borders of this one: the 30-60 us the GPU idles per synchronous step
// (stream_wait + contour stage + the next call's launch latency) disappear from a back-to-back sequence of calibration buffers
// (base.py:547-601 called once per buffer: BASELINE config 4's streams, the state machine's recalibrations).
// All submissions of a context go on ONE stream (stream order is what keeps the second submission's kernels off the workspace of
// the first); the results are pinned per ticket.  A selection that overflows the value store is taken again by the synchronous
// rm_locate inside rm_locate_result (frames_dev must stay valid until then).
// ------------------------------------------------------------------------------------------
extern "C" int rm_locate_submit(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax,
                                double amp, int levels, int skip, double temporal_thr, int threshold, unsigned flags, void *stream,
                                int *ticket_out)
{
    if (!ctx || !ticket_out) return fail(RM_E_BADARG, "rm_locate_submit: bad argument");
    int ti = -1;
    for (int i = 0; i < ROI_SLOTS - 1; ++i)
        if (!ctx->tickets[i].active) { ti = i; break; }
    if (ti < 0) return fail(RM_E_BUSY, "rm_locate_submit: %d submissions are waiting for rm_locate_result", ROI_SLOTS - 1);
    for (int i = 0; i < ROI_SLOTS - 1; ++i)
        if (ctx->tickets[i].active && ctx->tickets[i].stream != (hipStream_t)stream)
            return fail(RM_E_BADARG, "rm_locate_submit: the submissions of a context share one stream");
    LocateTicket &t = ctx->tickets[ti];
    HIP_TRY(hipSetDevice(ctx->device));
    if (!t.done) HIP_TRY(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    struct SlotGuard { rm_ctx *c; ~SlotGuard() { c->cur_slot = 0; } } guard{ctx};
    ctx->cur_slot = ti + 1;
    RoiSlot &rs = ctx->slots[ctx->cur_slot];
    if (rs.h_unserved) *rs.h_unserved = 0;
    double *heat = nullptr;
    RM_TRY(ws(ctx, "heat", (size_t)H * W, &heat));
    CollapsePlan cp;
    RM_TRY(calibrate_impl(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, temporal_thr, flags, heat, nullptr, stream, &cp));
    ctx->clip_frame_once = (flags & RM_FLAG_CONTOUR_CLIP_FRAME) != 0;
    ctx->tiles_const_once = cp.valid && cp.S >= 1;
    RM_TRY(roi_launch(ctx, heat, H, W, threshold, nullptr, nullptr, stream, true, t.roi, true));
    HIP_TRY(hipEventRecord(t.done, (hipStream_t)stream));
    t.stream = (hipStream_t)stream; t.frames = frames; t.dtype = dtype; t.T = T; t.H = H; t.W = W; t.fps = fps; t.fmin = fmin; t.fmax = fmax;
    t.amp = amp; t.levels = levels; t.skip = skip; t.temporal_thr = temporal_thr; t.threshold = threshold; t.flags = flags;
    t.plan_valid = cp.valid;
    t.active = true;
    *ticket_out = ti;
    return RM_OK;
}

extern "C" int rm_locate_result(rm_ctx *ctx, int ticket, int32_t *xywh)
{
    if (!ctx || !xywh || ticket < 0 || ticket >= ROI_SLOTS - 1 || !ctx->tickets[ticket].active)
        return fail(RM_E_BADARG, "rm_locate_result: bad argument (no such submission)");
    LocateTicket &t = ctx->tickets[ticket];
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(event_wait(t.done));   // (a failed wait leaves the ticket active: its kernels may still write the pinned slot)
    t.active = false;
    int rc = roi_finish(ctx, t.roi, xywh);
    RoiSlot &rs = ctx->slots[ticket + 1];
    const int unserved_word = rs.h_unserved ? *rs.h_unserved : 0;
    if (rs.h_unserved) *rs.h_unserved = 0;
    if (ctx->dense_hint && unserved_word != 2) ctx->dense_hint = 0;
    if (rc >= 0 && t.plan_valid && unserved_word == 1)   // nobody took the sum (value store overflow): the synchronous call sorts it out
        rc = rm_locate(ctx, t.frames, t.dtype, t.T, t.H, t.W, t.fps, t.fmin, t.fmax, t.amp, t.levels, t.skip, t.temporal_thr, t.threshold, t.flags,
                       xywh, (void *)t.stream);
    return rc;
}

