// respmon_amd/csrc/rm_ctx.hip -- contexts, developer switches, profiling hooks, dtype helpers
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

static thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int ws_get(rm_ctx *ctx, const std::string &name, size_t bytes, void **out)
{
    DevBuf &b = ctx->bufs[name];
    if (b.cap < bytes) {
        if (b.p) {
            // kernels of a submission nobody has fetched yet may still hold the old pointer: wait for them, then free
            for (LocateTicket &t : ctx->tickets)
                if (t.active && t.done) HIP_TRY(hipEventSynchronize(t.done));
            HIP_TRY(hipFree(b.p));
        }
        b.p = nullptr; b.cap = 0;
        size_t cap = (bytes + 255) / 256 * 256;
        HIP_TRY(hipMalloc(&b.p, cap));
        b.cap = cap;
    }
    *out = b.p;
    return RM_OK;
}

int ctx_stream_ok(rm_ctx *ctx, void *stream, const char *who)
{
    for (const LocateTicket &t : ctx->tickets)
        if (t.active && t.stream != (hipStream_t)stream)
            return fail(RM_E_BUSY, "%s: a rm_locate_submit of this context is in flight on another stream (fetch it with rm_locate_result first, or "
                                   "use its stream)", who);
    return RM_OK;
}

extern "C" int rm_debug_roi_path(rm_ctx *ctx, int *path_out)
{
    if (!ctx || !path_out) return fail(RM_E_BADARG, "rm_debug_roi_path: bad argument");
    *path_out = ctx->roi_path;
    return RM_OK;
}

extern "C" int rm_debug_host_timeline(rm_ctx *ctx, double *out)
{
    if (!ctx || !out) return fail(RM_E_BADARG, "rm_debug_host_timeline: bad argument");
    for (int i = 0; i < RM_HOST_MARKS; ++i) out[i] = ctx->host_marks[i];
    return RM_OK;
}

extern "C" int rm_abi_version(void) { return 1; }
extern "C" const char *rm_last_error_string(void) { return g_err.c_str(); }

extern "C" int rm_ctx_create(int device, rm_ctx **out)
{
    if (!out) return fail(RM_E_BADARG, "rm_ctx_create: out is NULL");
    HIP_TRY(hipSetDevice(device));
    rm_ctx *c = new rm_ctx();
    c->device = device;
    HIP_TRY(hipMalloc((void **)&c->d_state, sizeof(CollapseState)));
    HIP_TRY(hipHostMalloc((void **)&c->h_state, sizeof(CollapseState), hipHostMallocDefault));
    *out = c;
    return RM_OK;
}

extern "C" int rm_ctx_destroy(rm_ctx *ctx)
{
    if (!ctx) return RM_OK;
    (void)hipSetDevice(ctx->device);
    for (LocateTicket &t : ctx->tickets) {
        if (t.active && t.done) (void)hipEventSynchronize(t.done);   // (a submission nobody fetched still writes into the pinned slots)
        if (t.done) (void)hipEventDestroy(t.done);
    }
    for (auto &kv : ctx->bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->h_state) (void)hipHostFree(ctx->h_state);
    for (RoiSlot &rs : ctx->slots) {
        if (rs.h_bin) (void)hipHostFree(rs.h_bin);
        if (rs.h_comps) (void)hipHostFree(rs.h_comps);
        if (rs.h_unserved) (void)hipHostFree(rs.h_unserved);
    }
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    (void)rm_comm_destroy(ctx);
    for (int p = 0; p < RM_PROFILE_PHASES; ++p)
        for (hipEvent_t e : ctx->prof_ev[p]) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
    return RM_OK;
}

extern "C" int rm_debug_set(rm_ctx *ctx, const char *key, long long value)
{
    if (!ctx || !key) return fail(RM_E_BADARG, "rm_debug_set: bad argument");
    DebugKnobs &d = ctx->dbg;
    const std::string k(key);
    if (k == "temporal_valu") d.temporal_valu = (int)value;
    else if (k == "temporal_wide") d.temporal_wide = (int)value;
    else if (k == "dc_lds_front_end") d.dc_lds_front_end = (int)value;
    else if (k == "bgr_unfused") d.bgr_unfused = (int)value;
    else if (k == "no_fused_bounds") d.no_fused_bounds = (int)value;
    else if (k == "bounds_table_bytes") d.bounds_table_bytes = value;
    else if (k == "bounds_scalar") d.bounds_scalar = (int)value;
    else if (k == "bounds_l1") d.bounds_l1 = (int)value;
    else if (k == "bounds_up1") d.bounds_up1 = (int)value;
    else if (k == "dense_wf_list") d.dense_wf_list = (int)value;
    else if (k == "xs") d.xs = (int)value;
    else if (k == "xs_waves") d.xs_waves = (int)value;
    else if (k == "xs_budget_words") d.xs_budget_words = value;
    else if (k == "bounds_l1_rows") d.bounds_l1_rows = (int)value;
    else if (k == "dense_rows") d.dense_rows = (int)value;
    else if (k == "dense_general") d.dense_general = (int)value;
    else if (k == "dense_wave") d.dense_wave = (int)value;
    else if (k == "dense_frames") d.dense_frames = (int)value;
    else if (k == "dense_split") d.dense_split = (int)value;
    else if (k == "dc_segs") d.dc_segs = (int)value;
    else if (k == "dc_wpg") d.dc_wpg = (int)value;
    else if (k == "dc_split") d.dc_split = (int)value;
    else if (k == "dc_prio") d.dc_prio = (int)value;
    else if (k == "dc_prio_shift") d.dc_prio_shift = (int)value;
    else if (k == "store_slots") d.store_slots = value;
    else if (k == "store_default_slots") d.store_default_slots = value;
    else if (k == "collapse_fused") d.collapse_fused = (int)value;
    else if (k == "tile_sum_half") d.tile_sum_half = (int)value;
    else if (k == "dense_tiles") d.dense_tiles = (int)value;
    else if (k == "eval_fast") d.eval_fast = (int)value;
    else if (k == "exchange_dense") d.exchange_dense = (int)value;
    else if (k == "host_simple_shape") d.host_simple_shape = (int)value;
    else if (k == "host_area_bound") d.host_area_bound = (int)value;
    else if (k == "label_lazy") d.label_lazy = (int)value;
    else if (k == "heat_rows") d.heat_rows = (int)value;
    else if (k == "ff_parts") d.ff_parts = (int)value;
    else if (k == "heat_const_tiles") d.heat_const_tiles = (int)value;
    else if (k == "dense_t_low") d.dense_t_low = (int)value;
    else if (k == "dense_exact_top") d.dense_exact_top = (int)value;
    else if (k == "ccl_table") d.ccl_table = (int)value;
    else if (k == "ccl_tiles") d.ccl_tiles = (int)value;
    else if (k == "ccl_tile_waves") d.ccl_tile_waves = (int)value;
    else if (k == "label_host_steps") d.label_host_steps = value;
    else if (k == "sum_sym") d.sum_sym = (int)value;
    else if (k == "sum_rows") d.sum_rows = (int)value;
    else return fail(RM_E_BADARG, "rm_debug_set: unknown key '%s'", key);
    return RM_OK;
}

extern "C" int rm_profile_enable(rm_ctx *ctx, int on)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_profile_enable: ctx is NULL");
    ctx->prof_mode = on < 0 ? 0 : on > 2 ? 2 : on;
    ctx->prof_on = on != 0;
    return RM_OK;
}

extern "C" int rm_profile_read(rm_ctx *ctx, double *ms, int *n)
{
    if (!ctx || !ms) return fail(RM_E_BADARG, "rm_profile_read: bad argument");
    for (int p = 0; p < RM_PROFILE_PHASES; ++p) {
        double total = ctx->prof_host_ms[p];
        ctx->prof_host_ms[p] = 0;
        std::vector<hipEvent_t> &v = ctx->prof_ev[p];
        if (p == 0) ctx->prof_sampled = (int)(v.size() / 2);
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            HIP_TRY(hipEventSynchronize(v[i + 1]));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, v[i], v[i + 1]));
            total += t;
        }
        for (hipEvent_t e : v) ctx->prof_pool.push_back(e);
        v.clear();
        ms[p] = total;
    }
    if (n) *n = ctx->prof_sampled;      // calls whose phase 0 was bracketed (every call in mode 2, every 8th in mode 1)
    ctx->prof_calls = 0; ctx->prof_sampled = 0;
    return RM_OK;
}

extern "C" int rm_debug_counters(rm_ctx *ctx, long long *out, void *stream)
{
    if (!ctx || !out) return fail(RM_E_BADARG, "rm_debug_counters: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    const CollapseState &h = *ctx->h_state;
    const bool dense = ctx->dbg_mode == 1 || (long long)h.n_slots > ctx->dbg_cap ||
                       (ctx->dbg_mode == 0 && ctx->dbg_auto_dense && (unsigned long long)h.n_slots * DENSE_ONE_IN > (unsigned long long)ctx->dbg_mine);
    out[0] = ctx->dbg_pairs; out[1] = (long long)h.n_list_a + (dense ? 0 : (long long)h.n_list_b); out[2] = h.n_slots;
    out[3] = dense ? 0 : ctx->dbg_cap;
    if (ctx->dbg_fused) { out[1] = (long long)h.n_list_a; out[3] = -1; }   // store-less path: C pairs evaluated for the extrema; the kept pairs where they are summed
    return RM_OK;
}

#ifndef RM_FRAME_KERNEL_SRC_SHA
#include "build/rm_stamp.h"   // written by the Makefile: first 16 hex digits of the sha256 over the frame-buffer kernel sources
#endif
extern "C" const char *rm_debug_kernel_source_stamp(void) { return RM_FRAME_KERNEL_SRC_SHA; }

extern "C" int rm_debug_workspace(rm_ctx *ctx, const char *name, void *out_host, size_t bytes, void *stream)
{
    if (!ctx || !name || !out_host) return fail(RM_E_BADARG, "rm_debug_workspace: bad argument");
    auto it = ctx->bufs.find(name);
    if (it == ctx->bufs.end() || !it->second.p) return fail(RM_E_BADARG, "rm_debug_workspace: no workspace buffer '%s'", name);
    if (bytes > it->second.cap) return fail(RM_E_BADARG, "rm_debug_workspace: '%s' holds %zu bytes, %zu asked for", name, it->second.cap, bytes);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out_host, it->second.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    return RM_OK;
}

extern "C" size_t rm_ctx_workspace_bytes(const rm_ctx *ctx)
{
    size_t n = 0;
    if (ctx)
        for (auto &kv : ctx->bufs) n += kv.second.cap;
    return n;
}

// ------------------------------------------------------------------------------------------
// dtype helpers
// ------------------------------------------------------------------------------------------
extern "C" int rm_uint8_to_float(rm_ctx *ctx, const uint8_t *src, double *dst, size_t n, void *stream)
{
    if (!ctx || !src || !dst) return fail(RM_E_BADARG, "rm_uint8_to_float: NULL argument");
    if (n == 0) return RM_OK;
    hipLaunchKernelGGL(k_u8_to_f64<>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_float_to_uint8(rm_ctx *ctx, const double *src, uint8_t *dst, size_t n, void *stream)
{
    if (!ctx || !src || !dst) return fail(RM_E_BADARG, "rm_float_to_uint8: NULL argument");
    if (n == 0) return RM_OK;
    hipLaunchKernelGGL(k_f64_to_u8<>, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    LAUNCH_CHECK();
    return RM_OK;
}

int launch_bgr_to_gray(const uint8_t *bgr, size_t npix, uint8_t *gray, hipStream_t s)
{
    if (npix == 0) return RM_OK;
    size_t done = 0;
    if ((((uintptr_t)bgr | (uintptr_t)gray) & 3) == 0 && npix >= 4) {
        const size_t nquads = npix / 4;
        hipLaunchKernelGGL(k_bgr_to_gray_quads<>, dim3(nblk(nquads, 256, 1 << 16)), dim3(256), 0, s, (const unsigned *)bgr, nquads, (unsigned *)gray);
        LAUNCH_CHECK();
        done = nquads * 4;
    }
    if (done < npix) {
        hipLaunchKernelGGL(k_bgr_to_gray<>, dim3(nblk(npix - done, 256, 1 << 16)), dim3(256), 0, s, bgr + 3 * done, npix - done, gray + done);
        LAUNCH_CHECK();
    }
    return RM_OK;
}

extern "C" int rm_bgr_to_gray(rm_ctx *ctx, const uint8_t *bgr, size_t npix, uint8_t *gray, void *stream)
{
    if (!ctx || !bgr || !gray) return fail(RM_E_BADARG, "rm_bgr_to_gray: NULL argument");
    return launch_bgr_to_gray(bgr, npix, gray, (hipStream_t)stream);
}

// RM_BGR8 frame buffers on the paths that have no fused conversion (rm_down_chain_u8.h bgr8_t): cvtColor of the whole buffer into a
// gray uint8 workspace of the context, which then stands for the frame buffer
int bgr_buffer_to_gray(rm_ctx *ctx, const void *frames, size_t npix, const void **gray_out, hipStream_t s)
{
    uint8_t *gray = nullptr;
    RM_TRY(ws(ctx, "gray_of_bgr", (npix + 7) / 8, (double **)&gray));
    RM_TRY(launch_bgr_to_gray((const uint8_t *)frames, npix, gray, s));
    *gray_out = gray;
    return RM_OK;
}

#ifdef RM_TRACE
static TraceRec *g_trace_dev = nullptr;
extern "C" int rm_trace_start(void)
{

    const size_t bytes = sizeof(TraceRec) * (size_t)TRACE_KERNELS * TRACE_BLOCKS;
    if (!g_trace_dev) HIP_TRY(hipMalloc((void **)&g_trace_dev, bytes));
    HIP_TRY(hipMemset(g_trace_dev, 0, bytes));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &g_trace_dev, sizeof(g_trace_dev)));
    HIP_TRY(hipDeviceSynchronize());
    return RM_OK;
}
extern "C" int rm_trace_read(void *host, size_t bytes)
{
    const size_t all = sizeof(TraceRec) * (size_t)TRACE_KERNELS * TRACE_BLOCKS;
    if (!g_trace_dev || !host || bytes < all) return fail(RM_E_BADARG, "rm_trace_read: need %zu bytes", all);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, g_trace_dev, all, hipMemcpyDeviceToHost));
    return RM_OK;
}
#endif

