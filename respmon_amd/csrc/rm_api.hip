// respmon_amd/csrc/rm_api.hip -- C-ABI host side of librespmon_hip.so (see include/respmon_hip.h).
// Owns the context (device workspace, cached temporal operator, pinned staging) and sequences
// the kernels of rm_kernels.h on the caller's HIP stream.
#include "../../include/respmon_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rm_contour.h"
#include "rm_kernels.h"
#include "rm_down_chain.h"
#include "rm_down_chain_u8.h"
#include "rm_dense_sum.h"
#include "rm_tile_eval.h"
#include "rm_ccl.h"
#include "rm_flow.h"

using namespace rm;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(RM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define RM_TRY(expr)                \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ < 0) return rc_;    \
    } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

struct DevBuf { void *p = nullptr; size_t cap = 0; };
constexpr long long STORE_MAX_SLOTS = 524288;
struct ExchangeState { int cap = RM_SPARSE_CAP_TILES; int dense_left = 0; };   // heatmap exchange policy of a communicator (rm_locate_streams / _sharded)   // value store: at most 4 GiB (8 KB per kept (tile, frame) pair)

// what the collapse passes share (see collapse_eval / collapse_sum below)
struct CollapsePlan {
    ChainGeom g;
    int ntiles = 0, npairs = 0;            // npairs: unique (tile, frame) pairs = ntiles * sym_frames(T)
    double *lo = nullptr, *hi = nullptr, *store = nullptr;
    unsigned int *list_a = nullptr, *list_b = nullptr, *heavy = nullptr;
    int *slot_of = nullptr, *sel_cnt = nullptr;
    size_t shmem = 0;
    const double *cS = nullptr;
    int T = 0, t0 = 0, t1 = 0, H = 0, W = 0, S = 0;
    bool valid = false;
    bool no_prune = false;
    bool fused = false;                    // k_eval_c + k_tile_sum (rm_tile_eval.h): no value store, no separate evaluation of the kept pairs
    SumPlan sp{0, 0, 0, 0};                // sparse or dense sum: decided on the device (rm_kernels.h sum_is_dense)
};


// developer / test switches (rm_debug_set): they select between implementations that produce identical results, or shrink a
// tuning constant so that a test reaches a rare code path.  The library never reads the process environment.
struct DebugKnobs {
    int temporal_valu = 0;        // 1: the two-stage VALU temporal kernels instead of k_temporal_sym
    int temporal_wide = -1;       // 0 / 1: k_temporal_sym / k_temporal_sym_px whatever the level size (-1: by size)
    int dc_lds_front_end = 0;     // 1: narrow frame buffers through the LDS front end of k_down_chain instead of rm_down_chain_u8.h
    int no_fused_bounds = 0;      // 1: k_small_collapse + k_frame_bounds instead of k_small_collapse_bounds
    long long bounds_table_bytes = 0;   // > 0: LDS budget of k_frame_bounds' row-extrema table (forces small bands)
    int bounds_scalar = 0;        // 1: k_frame_bounds (a thread per row and tile column) also for wide levels instead of k_frame_bounds_rows
    int dense_rows = 0;           // 16 / 32 / 64: super-tile rows of the dense sum kernel
    int dense_general = 0;        // 1: k_dense_sum instead of the table-driven k_dense_sum_s2 at skip <= 2
    int dense_frames = 0;         // 1 / 2: frames per trip of k_dense_sum_w (0: by the number of tiles)
    int dense_split = 0;          // 1 / 2 / 4: waves per tile (k_dense_sum_wf for 2 and 4; 0: by the number of tiles)
    int dense_wave = 1;           // 0: the workgroup kernels (k_dense_sum_s2 / k_dense_sum) instead of the wave-private k_dense_sum_w at skip <= 2
    int dc_segs = 0, dc_wpg = 0;  // > 0: segments per frame / waves per workgroup of k_down_chain
    int dc_split = 0;             // > 0: share (per mille) of the level-S rows the upper of exactly two segments takes (default 513)
    int collapse_fused = 0;       // 1: collapse passes without a value store wherever TileEval applies (rm_tile_eval.h k_eval_c + k_tile_sum); 0: only as the stand-in for an overflowing store at skip >= 3
    int sum_rows = 0;             // 1: k_masked_sum_rows (one wave per tile row, LDS-DMA staging) instead of k_masked_sum_tiles for whole-buffer sums (measured slower: 35 us against 21)
    int sum_sym = 0;              // 1: k_masked_sum_sym instead of k_masked_sum_tiles for whole-buffer sums (measured slower: 31 us against 21 at 1080p x 256)
    int label_host_us = 250;      // host border following slower than this (+ the labelled stage's own host time) -> device labelling next time
    int ccl_table = -1;           // k_ccl_bbox: 1 with / 0 without the per-tile LDS table of boxes, -1 by the last component count
    int heat_const_tiles = 1;     // 0: k_heat_to_u8 reads every pixel of rm_locate's heatmap (no use of the sum kernel's constant-tile flags)
    int ff_parts = 0;             // > 0: workgroups per frame of k_small_filter_first (default: 2 when one per frame would leave CUs idle)
    int host_simple_shape = 1;    // 0: the host contour stage always follows the borders (no one-blob shortcut on the packed rows)
    int exchange_dense = 0;       // 1: rm_locate_streams / rm_locate_sharded exchange the heatmaps by the dense all-reduce only
    int eval_fast = 1;            // 0: the generic k_eval_pairs instead of k_eval_pairs_fast (rm_tile_eval.h) where the latter applies
    int dense_t_low = -1;         // k_dense_sum_t (TileEval) at skip <= 2 instead of k_dense_sum_w / wf: 1 always, 0 never, -1 on large frames
    int dense_tiles = 1;          // 0: k_tile_sum (rounds of sixteen waves per tile) instead of k_dense_sum_t (one wave per tile) where a store-less sum at skip >= 3 is due
    int tile_sum_half = -1;       // 0 / 1: k_tile_sum works on whole tiles / half tiles whatever the number of heavy tiles (-1: by that number)
    long long store_default_slots = 0;   // > 0: slots the value store starts with before any selection has made it grow (default 16 384)
    long long store_slots = 0;    // > 0: capacity of the value store in (tile, frame) slots (forces the overflow path)
};

// pinned result areas of ONE ROI extraction in flight
struct RoiSlot {
    uint8_t *h_bin = nullptr; size_t h_bin_cap = 0;       // bit-packed thresholded image + H row flags (k_heat_to_u8)
    uint8_t *h_rows_dirty = nullptr;                       // the row-flag part of h_bin that is known to be all zero
    CclComp *h_comps = nullptr; size_t h_comps_cap = 0;    // [0] = {count, -, -, -}, then one record per component
    int *h_unserved = nullptr;      // set by k_masked_sum_tiles when it left the sum to a dense kernel nobody enqueued (rm_locate)
};
constexpr int ROI_SLOTS = 3;
// what the host half of the ROI stage has to know about the launches it finishes
struct RoiPending { int H = 0, W = 0, slot = 0; size_t nwords = 0, comps_cap = 0; bool label = false, clip = false; };
// one rm_locate_submit whose rm_locate_result has not been called yet (the arguments: a selection that overflows the value store is
// taken again through the synchronous rm_locate)
struct LocateTicket {
    bool active = false;
    RoiPending roi;
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;
    const void *frames = nullptr;
    int dtype = 0, T = 0, H = 0, W = 0, levels = 0, skip = 0, threshold = 0;
    double fps = 0, fmin = 0, fmax = 0, amp = 0, temporal_thr = 0;
    unsigned flags = 0;
    bool plan_valid = false;
};

struct rm_ctx {
    int device = 0;
    DebugKnobs dbg;
    std::map<std::string, DevBuf> bufs;
    CollapseState *d_state = nullptr;
    CollapseState *h_state = nullptr;  // pinned
    RoiSlot slots[ROI_SLOTS];                              // pinned result areas; slot 0 serves the synchronous entries
    int cur_slot = 0;                                      // the slot the launches being enqueued write to
    LocateTicket tickets[ROI_SLOTS - 1];                   // rm_locate_submit / rm_locate_result (ticket i uses slot i + 1)
    bool tiles_const_once = false;                         // the next ROI stage reads the heatmap rm_locate's own sum kernel has just written (tile_nkept is valid for it)
    bool clip_frame = false, clip_frame_once = false;      // cv2.findContours of OpenCV <= 3.1 (rm_set_contour_clip_frame / RM_FLAG_CONTOUR_CLIP_FRAME)
    // device labelling of the thresholded image (rm_ccl.h): taken when the previous ROI extraction of this geometry met
    // more than LABEL_MIN_CONTOURS components (label_mode -1 = that rule, 0 = never, 1 = always: rm_set_contour_labelling)
    int label_mode = -1, label_H = 0, label_W = 0, label_last_n = 0, label_used = 0;
    // ... or when following every border on the host took long last time (few components with long borders: a frame of noise blobs):
    // host time of the last unlabelled stage of this geometry (< 0: none) with its contour count, host time of the last labelled
    // stage, labelled stages in a row (every LABEL_REPROBE-th one is run unlabelled to refresh the first figure)
    double label_unl_us = -1.0, label_lab_host_us = 0.0;
    int label_unl_n = 0, label_streak = 0;
    // cached temporal operator
    int op_T = 0, op_nk = 0; double op_fps = 0, op_fmin = 0, op_fmax = 0;
    bool state_fresh = false;   // d_state was reset by the last kernel of front_pyramid and nothing has reduced into it since
    int op_mfma = 0;        // > 0: the cached operator also exists in the fragment-major form of k_temporal_mfma, with this many 16-row tiles
    FlowWorkspace flow;
    CollapsePlan shard_plan;   // rm_shard_collapse -> rm_shard_heat
    size_t eval_shmem = ~(size_t)0; int eval_per_cu = 0, eval_cus = 0;   // k_eval_pairs: resident workgroups per CU at this LDS footprint
    int nkept_H = 0, nkept_W = 0;   // geometry the "tile_nkept" workspace buffer (last rm_calibrate) belongs to; 0 = none
    int *h_flag = nullptr;          // pinned: {overflow flag, largest per-rank tile count} of the sparse heatmap merge
    void *comm = nullptr; int comm_rank = 0, comm_world = 1;   // RCCL communicator (rm_comm_init); none: one rank
    ExchangeState xp_streams, xp_sharded;
    int dense_hint = 0;             // the last rm_locate of this context met a dense selection (more than a quarter of the pairs kept)
    long long store_hint_slots = 0; // slots a selection of this context needed when it overflowed the value store (rm_locate grows the store to it)
    // measurement hook (rm_profile_*)
    long long dbg_pairs = 0, dbg_cap = 0, dbg_mine = 0; int dbg_mode = 0, dbg_auto_dense = 0, dbg_fused = 0;   // the SumPlan of the last collapse (host copy)
    int prof_mode = 0;                     // 0 off, 1 frame-buffer kernel only, 2 all phases
    bool prof_on = false;
    int prof_calls = 0, prof_sampled = 0;
    std::vector<hipEvent_t> prof_ev[RM_PROFILE_PHASES];  // start/stop pairs per phase
    std::vector<hipEvent_t> prof_pool;
    double prof_host_ms[RM_PROFILE_PHASES] = {0, 0, 0, 0};
};

// RAII bracket: records a start event now and a stop event at scope exit (no-op when profiling is off)
struct PhaseTimer {
    rm_ctx *c; int phase; hipStream_t s; bool on;
    static hipEvent_t get(rm_ctx *c)
    {
        hipEvent_t e = nullptr;
        if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    }
    // mode 1 brackets the frame-buffer kernel of every 8th call only: the two event records in front of that launch sit on the
    // host's critical path between two steps (~4 us each time, rocprofv3 --hip-runtime-trace)
    PhaseTimer(rm_ctx *c_, int phase_, hipStream_t s_) : c(c_), phase(phase_), s(s_), on(c_->prof_mode == 2 || (c_->prof_mode == 1 && phase_ == 0 && (c_->prof_calls & 7) == 0))
    {
        if (!on) return;
        hipEvent_t e = get(c);
        (void)hipEventRecord(e, s);
        c->prof_ev[phase].push_back(e);
    }
    ~PhaseTimer()
    {
        if (!on) return;
        hipEvent_t e = get(c);
        (void)hipEventRecord(e, s);
        c->prof_ev[phase].push_back(e);
    }
};

static int ws_get(rm_ctx *ctx, const std::string &name, size_t bytes, void **out)
{
    DevBuf &b = ctx->bufs[name];
    if (b.cap < bytes) {
        if (b.p) HIP_TRY(hipFree(b.p));
        b.p = nullptr; b.cap = 0;
        size_t cap = (bytes + 255) / 256 * 256;
        HIP_TRY(hipMalloc(&b.p, cap));
        b.cap = cap;
    }
    *out = b.p;
    return RM_OK;
}
template <typename T> static int ws(rm_ctx *ctx, const std::string &name, size_t count, T **out)
{
    void *p = nullptr;
    RM_TRY(ws_get(ctx, name, count * sizeof(T), &p));
    *out = (T *)p;
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
    else if (k == "no_fused_bounds") d.no_fused_bounds = (int)value;
    else if (k == "bounds_table_bytes") d.bounds_table_bytes = value;
    else if (k == "bounds_scalar") d.bounds_scalar = (int)value;
    else if (k == "dense_rows") d.dense_rows = (int)value;
    else if (k == "dense_general") d.dense_general = (int)value;
    else if (k == "dense_wave") d.dense_wave = (int)value;
    else if (k == "dense_frames") d.dense_frames = (int)value;
    else if (k == "dense_split") d.dense_split = (int)value;
    else if (k == "dc_segs") d.dc_segs = (int)value;
    else if (k == "dc_wpg") d.dc_wpg = (int)value;
    else if (k == "dc_split") d.dc_split = (int)value;
    else if (k == "store_slots") d.store_slots = value;
    else if (k == "store_default_slots") d.store_default_slots = value;
    else if (k == "collapse_fused") d.collapse_fused = (int)value;
    else if (k == "tile_sum_half") d.tile_sum_half = (int)value;
    else if (k == "dense_tiles") d.dense_tiles = (int)value;
    else if (k == "eval_fast") d.eval_fast = (int)value;
    else if (k == "exchange_dense") d.exchange_dense = (int)value;
    else if (k == "host_simple_shape") d.host_simple_shape = (int)value;
    else if (k == "ff_parts") d.ff_parts = (int)value;
    else if (k == "heat_const_tiles") d.heat_const_tiles = (int)value;
    else if (k == "dense_t_low") d.dense_t_low = (int)value;
    else if (k == "ccl_table") d.ccl_table = (int)value;
    else if (k == "label_host_us") d.label_host_us = (int)value;
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

extern "C" size_t rm_ctx_workspace_bytes(const rm_ctx *ctx)
{
    size_t n = 0;
    if (ctx)
        for (auto &kv : ctx->bufs) n += kv.second.cap;
    return n;
}

static inline unsigned nblk(size_t n, unsigned per, unsigned cap = 8192)
{
    size_t b = (n + per - 1) / per;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// ------------------------------------------------------------------------------------------
// dtype helpers
// ------------------------------------------------------------------------------------------
extern "C" int rm_uint8_to_float(rm_ctx *ctx, const uint8_t *src, double *dst, size_t n, void *stream)
{
    if (!ctx || !src || !dst) return fail(RM_E_BADARG, "rm_uint8_to_float: NULL argument");
    if (n == 0) return RM_OK;
    hipLaunchKernelGGL(k_u8_to_f64, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_float_to_uint8(rm_ctx *ctx, const double *src, uint8_t *dst, size_t n, void *stream)
{
    if (!ctx || !src || !dst) return fail(RM_E_BADARG, "rm_float_to_uint8: NULL argument");
    if (n == 0) return RM_OK;
    hipLaunchKernelGGL(k_f64_to_u8, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_bgr_to_gray(rm_ctx *ctx, const uint8_t *bgr, size_t npix, uint8_t *gray, void *stream)
{
    if (!ctx || !bgr || !gray) return fail(RM_E_BADARG, "rm_bgr_to_gray: NULL argument");
    if (npix == 0) return RM_OK;
    hipLaunchKernelGGL(k_bgr_to_gray, dim3(nblk(npix, 256)), dim3(256), 0, (hipStream_t)stream, bgr, npix, gray);
    LAUNCH_CHECK();
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// pyramid building blocks
// ------------------------------------------------------------------------------------------
static int launch_pyr_down(const void *src, int dtype, int T, int h, int w, double *dst, hipStream_t s)
{
    int dh = (h + 1) / 2, dw = (w + 1) / 2;
    dim3 grid((dw + PD_TX - 1) / PD_TX, (dh + PD_TY - 1) / PD_TY, T), block(256);
    size_t fs = (size_t)h * w;
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_pyr_down<uint8_t>), grid, block, 0, s, (const uint8_t *)src, h, w, fs, dst, dh, dw); break;
    case RM_F16: hipLaunchKernelGGL((k_pyr_down<__half>), grid, block, 0, s, (const __half *)src, h, w, fs, dst, dh, dw); break;
    case RM_F32: hipLaunchKernelGGL((k_pyr_down<float>), grid, block, 0, s, (const float *)src, h, w, fs, dst, dh, dw); break;
    case RM_F64: hipLaunchKernelGGL((k_pyr_down<double>), grid, block, 0, s, (const double *)src, h, w, fs, dst, dh, dw); break;
    default: return fail(RM_E_BADARG, "unknown dtype %d", dtype);
    }
    LAUNCH_CHECK();
    return RM_OK;
}

static int launch_pyr_up(const double *src, int T, int sh, int sw, double *dst, int dh, int dw, int mode,
                         const double *other, hipStream_t s, size_t src_fs = 0, size_t dst_fs = 0, size_t other_fs = 0)
{
    if (!src_fs) src_fs = (size_t)sh * sw;
    if (!dst_fs) dst_fs = (size_t)dh * dw;
    if (!other_fs) other_fs = (size_t)dh * dw;
    if (!((dw == 2 * sw || dw == 2 * sw - 1) && (dh == 2 * sh || dh == 2 * sh - 1)))
        return fail(RM_E_BADARG, "pyrUp: dstsize (%d,%d) incompatible with source (%d,%d)", dw, dh, sw, sh);
    if (dw >= 128 && dh >= 8 && dst != src) {   // large levels: 2 x 2 outputs per thread
        dim3 grid((dw + 127) / 128, (dh + 7) / 8, T), block(256);
        hipLaunchKernelGGL(k_pyr_up_2x2, grid, block, 0, s, src, sh, sw, src_fs, dst, dh, dw, dst_fs, mode, other, other_fs);
        LAUNCH_CHECK();
        return RM_OK;
    }
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, T), block(256);
    hipLaunchKernelGGL(k_pyr_up, grid, block, 0, s, src, sh, sw, src_fs, dst, dh, dw, dst_fs, mode, other, other_fs);
    LAUNCH_CHECK();
    return RM_OK;
}

static bool valid_dtype(int d) { return d == RM_U8 || d == RM_F16 || d == RM_F32 || d == RM_F64; }

extern "C" int rm_pyr_down(rm_ctx *ctx, const void *src, int dtype, int T, int h, int w, double *dst, void *stream)
{
    if (!ctx || !src || !dst || T < 0 || h < 1 || w < 1 || !valid_dtype(dtype))
        return fail(RM_E_BADARG, "rm_pyr_down: bad argument");
    if (T == 0) return RM_OK;
    return launch_pyr_down(src, dtype, T, h, w, dst, (hipStream_t)stream);
}

extern "C" int rm_pyr_up(rm_ctx *ctx, const double *src, int T, int sh, int sw, double *dst, int dh, int dw, int mode,
                         const double *other, void *stream)
{
    if (!ctx || !src || !dst || T < 0 || sh < 1 || sw < 1 || mode < 0 || mode > 2 || (mode != 0 && !other))
        return fail(RM_E_BADARG, "rm_pyr_up: bad argument");
    if (T == 0) return RM_OK;
    return launch_pyr_up(src, T, sh, sw, dst, dh, dw, mode, other, (hipStream_t)stream);
}

template <typename Tin>
__global__ __launch_bounds__(256) void k_to_f64(const Tin *src, double *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = load_px(src, i);
}

static int launch_to_f64(const void *src, int dtype, size_t n, double *dst, hipStream_t s)
{
    dim3 grid(nblk(n, 256)), block(256);
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_to_f64<uint8_t>), grid, block, 0, s, (const uint8_t *)src, dst, n); break;
    case RM_F16: hipLaunchKernelGGL((k_to_f64<__half>), grid, block, 0, s, (const __half *)src, dst, n); break;
    case RM_F32: hipLaunchKernelGGL((k_to_f64<float>), grid, block, 0, s, (const float *)src, dst, n); break;
    case RM_F64: hipLaunchKernelGGL((k_to_f64<double>), grid, block, 0, s, (const double *)src, dst, n); break;
    default: return fail(RM_E_BADARG, "unknown dtype %d", dtype);
    }
    LAUNCH_CHECK();
    return RM_OK;
}

static void level_sizes(int H, int W, int levels, std::vector<int> &h, std::vector<int> &w)
{
    h.assign(levels, 0); w.assign(levels, 0);
    h[0] = H; w[0] = W;
    for (int l = 1; l < levels; ++l) { h[l] = (h[l - 1] + 1) / 2; w[l] = (w[l - 1] + 1) / 2; }
}

extern "C" int rm_create_laplacian_video_pyramid(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W,
                                                 int levels, double *const *lv, void *stream)
{
    if (!ctx || !frames || !lv || T < 1 || H < 1 || W < 1 || levels < 1 || !valid_dtype(dtype))
        return fail(RM_E_BADARG, "rm_create_laplacian_video_pyramid: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> h, w;
    level_sizes(H, W, levels, h, w);
    // Gaussian chain into the level arrays themselves (pyramid.py:9-17), then in place
    // L_i = G_i - pyrUp(G_{i+1}) from fine to coarse (pyramid.py:23-27)
    RM_TRY(launch_to_f64(frames, dtype, (size_t)T * H * W, lv[0], s));
    for (int l = 1; l < levels; ++l) RM_TRY(launch_pyr_down(lv[l - 1], RM_F64, T, h[l - 1], w[l - 1], lv[l], s));
    for (int l = 0; l + 1 < levels; ++l)
        RM_TRY(launch_pyr_up(lv[l + 1], T, h[l + 1], w[l + 1], lv[l], h[l], w[l], 1, lv[l], s));
    return RM_OK;
}

extern "C" int rm_collapse_laplacian_video_pyramid(rm_ctx *ctx, const double *const *lv, int T, int H, int W, int levels,
                                                   double *out, void *stream)
{
    if (!ctx || !lv || !out || T < 1 || H < 1 || W < 1 || levels < 1)
        return fail(RM_E_BADARG, "rm_collapse_laplacian_video_pyramid: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> h, w;
    level_sizes(H, W, levels, h, w);
    if (levels == 1) {
        if (out != lv[0]) HIP_TRY(hipMemcpyAsync(out, lv[0], sizeof(double) * (size_t)T * H * W, hipMemcpyDeviceToDevice, s));
        return RM_OK;
    }
    const double *cur = lv[levels - 1];
    for (int l = levels - 2; l >= 0; --l) {
        double *dst = nullptr;
        if (l == 0) dst = out;
        else RM_TRY(ws(ctx, (l & 1) ? "collapse_a" : "collapse_b", (size_t)T * h[l] * w[l], &dst));
        RM_TRY(launch_pyr_up(cur, T, h[l + 1], w[l + 1], dst, h[l], w[l], 2, lv[l], s));
        cur = dst;
    }
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// temporal operator  (transforms.py:82-102)
// ------------------------------------------------------------------------------------------
// scipy.fftpack.fftfreq(n, d) == numpy.fft.fftfreq: val = 1.0/(n*d); f[i] = i*val (i < (n-1)/2+1),
// f[i] = (i - n)*val otherwise (i.e. -(n/2) .. -1)
static void band_bounds(int n, double fps, double fmin, double fmax, int *lo, int *hi)
{
    double d = 1.0 / fps;
    double val = 1.0 / (n * d);
    int N = (n - 1) / 2 + 1;
    double best_lo = 0, best_hi = 0;
    *lo = 0; *hi = 0;
    for (int i = 0; i < n; ++i) {
        double f = (double)(i < N ? i : i - n) * val;
        double a = std::fabs(f - fmin), b = std::fabs(f - fmax);
        if (i == 0 || a < best_lo) { best_lo = a; *lo = i; }  // argmin keeps the first minimum
        if (i == 0 || b < best_hi) { best_hi = b; *hi = i; }
    }
}

extern "C" int rm_temporal_operator(int T, double fps, double fmin, double fmax, double *M, int *blo, int *bhi)
{
    if (T < 1 || !(fps > 0) || !M) return fail(RM_E_BADARG, "rm_temporal_operator: bad argument");
    int lo, hi;
    band_bounds(T, fps, fmin, fmax, &lo, &hi);
    if (blo) *blo = lo;
    if (bhi) *bhi = hi;
    const int n = T;
    // keep[k] for the PACKED rfft array: fft[hi:-hi] = 0; if lo != 0: fft[:lo] = 0, fft[-lo:] = 0
    std::vector<char> keep(n, 1);
    {
        // python slice [hi : n-hi] (when hi == 0 the stop is -0 == 0 -> empty slice)
        int start = hi, stop = (hi == 0) ? 0 : n - hi;
        for (int k = start; k < stop; ++k) keep[k] = 0;
        if (lo != 0) {
            for (int k = 0; k < lo && k < n; ++k) keep[k] = 0;
            for (int k = (n - lo > 0 ? n - lo : 0); k < n; ++k) keep[k] = 0;
        }
    }
    // packed real FFT rows: R[0,t] = 1; R[2j-1,t] = cos(2 pi j t / n); R[2j,t] = -sin(2 pi j t / n);
    // (n even) R[n-1,t] = (-1)^t.  Inverse as the reference applies it: Re(ifft(packed))[s] =
    // (1/n) sum_k packed[k] cos(2 pi k s / n).   M[s,t] = (1/n) sum_{k kept} cos(2 pi k s/n) R[k,t]
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<double> R((size_t)n * n, 0.0);
    for (int k = 0; k < n; ++k) {
        if (!keep[k]) continue;
        for (int t = 0; t < n; ++t) {
            double v;
            if (k == 0) v = 1.0;
            else if ((n % 2 == 0) && k == n - 1) v = (t % 2 == 0) ? 1.0 : -1.0;
            else {
                int j = (k + 1) / 2;
                long long jt = ((long long)j * t) % n;  // exact argument reduction
                double ang = two_pi * (double)jt / (double)n;
                v = (k % 2 == 1) ? std::cos(ang) : -std::sin(ang);
            }
            R[(size_t)k * n + t] = v;
        }
    }
    for (int s = 0; s < n; ++s)
        for (int t = 0; t < n; ++t) {
            long double acc = 0.0L;
            for (int k = 0; k < n; ++k) {
                if (!keep[k]) continue;
                long long ks = ((long long)k * s) % n;
                acc += (long double)std::cos(two_pi * (double)ks / (double)n) * (long double)R[(size_t)k * n + t];
            }
            M[(size_t)s * n + t] = (double)(acc / (long double)n);
        }
    return RM_OK;
}

// packed-rfft indices that survive the reference's mask (transforms.py:91-94), in increasing order
static void kept_packed_indices(int n, double fps, double fmin, double fmax, std::vector<int> &kept)
{
    int lo, hi;
    band_bounds(n, fps, fmin, fmax, &lo, &hi);
    std::vector<char> keep(n, 1);
    int start = hi, stop = (hi == 0) ? 0 : n - hi;  // python slice [hi:-hi]; -0 == 0 gives an empty slice
    for (int k = start; k < stop; ++k) keep[k] = 0;
    if (lo != 0) {
        for (int k = 0; k < lo && k < n; ++k) keep[k] = 0;
        for (int k = (n - lo > 0 ? n - lo : 0); k < n; ++k) keep[k] = 0;
    }
    kept.clear();
    for (int k = 0; k < n; ++k)
        if (keep[k]) kept.push_back(k);
}

// Two-stage form of the operator, MERGED and for the UNIQUE output frames (rm_kernels.h sym_frames):
//   packed index k contributes  Re(ifft)[s] += cos(2 pi k s / n) / n * y[k]  (transforms.py:98 on the packed array), and
//   cos(2 pi (n - k) s / n) == cos(2 pi k s / n): the kept indices k and n - k share their inverse column, so their forward rows
//   are added here once and for all.  m = min(k, n - k) names the merged row;
//     Rz[i][t] = R[m_i][t] (if kept) + R[n - m_i][t] (if kept, and a different index)      i < nm, t < n
//     Cz[s][i] = cos(2 pi m_i s / n) / n                                                   s < n / 2 + 1
//   The rows s > n / 2 of the inverse are the mirror images of these (out[n - s] == out[s], as in the reference: scipy's ifft of
//   a real array is exactly Hermitian), so they are never computed.
static double packed_row(int n, int k, int t)
{
    const double two_pi = 6.283185307179586476925286766559;
    if (k == 0) return 1.0;
    if ((n % 2 == 0) && k == n - 1) return (t % 2 == 0) ? 1.0 : -1.0;
    const int j = (k + 1) / 2;
    const long long jt = ((long long)j * t) % n;  // exact argument reduction
    const double ang = two_pi * (double)jt / (double)n;
    return (k % 2 == 1) ? std::cos(ang) : -std::sin(ang);
}

static void merged_operator(int n, const std::vector<int> &kept, std::vector<int> &ms, std::vector<double> &Rz, std::vector<double> &Cz)
{
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<char> is_kept(n, 0);
    for (int k : kept) is_kept[k] = 1;
    ms.clear();
    for (int m = 0; m <= n / 2; ++m) {
        const int k2 = n - m;
        if (is_kept[m] || (m != 0 && k2 < n && is_kept[k2])) ms.push_back(m);
    }
    const int nm = (int)ms.size(), Th = n / 2 + 1;
    Rz.assign((size_t)nm * n, 0.0);
    Cz.assign((size_t)Th * nm, 0.0);
    for (int i = 0; i < nm; ++i) {
        const int m = ms[i], k2 = n - m;
        const bool second = m != 0 && k2 != m && k2 < n && is_kept[k2];
        for (int t = 0; t < n; ++t) {
            double v = is_kept[m] ? packed_row(n, m, t) : 0.0;
            if (second) v = is_kept[m] ? v + packed_row(n, k2, t) : packed_row(n, k2, t);
            Rz[(size_t)i * n + t] = v;
        }
        for (int sidx = 0; sidx < Th; ++sidx) {
            const long long ks = ((long long)m * sidx) % n;
            Cz[(size_t)sidx * nm + i] = std::cos(two_pi * (double)ks / (double)n) / (double)n;
        }
    }
}

// symmetry class of merged row m when n is even: rows 0, odd m (cosine rows, and (-1)^t for k = n - 1) are even in t, even m > 0
// (sine rows) odd in t; the partner n - m has the parity of m, so a merged row never mixes the classes
static bool merged_row_is_even(int m) { return m == 0 || (m & 1); }

struct TemporalOp { const double *R = nullptr, *C = nullptr, *Rf = nullptr, *Cf = nullptr; int nk = 0, tiles = 0; };  // nk: merged rows; Rf / Cf: fragment-major copies for k_temporal_sym<tiles>

static int get_operator(rm_ctx *ctx, int T, double fps, double fmin, double fmax, TemporalOp *op, hipStream_t s)
{
    const int Th = T / 2 + 1, nks = (Th + 3) / 4, mt = (Th + 15) / 16;
    if (!(ctx->op_T == T && ctx->op_fps == fps && ctx->op_fmin == fmin && ctx->op_fmax == fmax)) {
        std::vector<int> kept, ms;
        kept_packed_indices(T, fps, fmin, fmax, kept);
        std::vector<double> R, C;
        merged_operator(T, kept, ms, R, C);
        const int nm = (int)ms.size();
        ctx->op_nk = nm;
        double *dR = nullptr, *dC = nullptr;
        RM_TRY(ws(ctx, "temporal_R", R.size() + 1, &dR));
        RM_TRY(ws(ctx, "temporal_C", C.size() + 1, &dC));
        // fragment-major copies for the matrix-core kernel (k_temporal_sym): class-pure tiles of 16 merged rows, NH "even" tiles
        // then NH "odd" ones, zero padded; frames folded to t <= n / 2 (needs an even n)
        std::vector<int> rows_e, rows_o;
        for (int i = 0; i < nm; ++i) (merged_row_is_even(ms[i]) ? rows_e : rows_o).push_back(i);
        const int NH = std::max(1, (int)std::max((rows_e.size() + 15) / 16, (rows_o.size() + 15) / 16));
        const bool mf = nm >= 1 && T % 2 == 0 && T >= 8 && NH <= TM_MAX_HALF;
        const int NT = 2 * NH;
        std::vector<double> Rf(mf ? (size_t)nks * NT * 64 : 1, 0.0), Cf(mf ? (size_t)mt * 4 * NT * 64 : 1, 0.0);
        if (mf) {
            auto row_of = [&](int q, int i) -> int {   // merged row held by row i of tile q, or -1
                const std::vector<int> &v = q < NH ? rows_e : rows_o;
                const size_t j = (size_t)(q < NH ? q : q - NH) * 16 + i;
                return j < v.size() ? v[j] : -1;
            };
            for (int ks = 0; ks < nks; ++ks)
                for (int q = 0; q < NT; ++q)
                    for (int l = 0; l < 64; ++l) {
                        const int r = row_of(q, l & 15), t = 4 * ks + (l >> 4);
                        Rf[((size_t)ks * NT + q) * 64 + l] = (r >= 0 && t < Th) ? R[(size_t)r * T + t] : 0.0;
                    }
            for (int m = 0; m < mt; ++m)
                for (int q = 0; q < NT; ++q)
                    for (int rr = 0; rr < 4; ++rr)
                        for (int l = 0; l < 64; ++l) {
                            const int sI = 16 * m + (l & 15), r = row_of(q, 4 * rr + (l >> 4));
                            Cf[(((size_t)m * NT + q) * 4 + rr) * 64 + l] = (r >= 0 && sI < Th) ? C[(size_t)sI * nm + r] : 0.0;
                        }
        }
        double *dRf = nullptr, *dCf = nullptr;
        RM_TRY(ws(ctx, "temporal_Rf", Rf.size(), &dRf));
        RM_TRY(ws(ctx, "temporal_Cf", Cf.size(), &dCf));
        ctx->op_mfma = mf ? NH : 0;
        if (!R.empty()) {
            HIP_TRY(hipMemcpyAsync(dR, R.data(), sizeof(double) * R.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dC, C.data(), sizeof(double) * C.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dRf, Rf.data(), sizeof(double) * Rf.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(dCf, Cf.data(), sizeof(double) * Cf.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(stream_wait(s));  // the sources are stack-lifetime vectors
        }
        ctx->op_T = T; ctx->op_fps = fps; ctx->op_fmin = fmin; ctx->op_fmax = fmax;
    }
    double *dR = nullptr, *dC = nullptr;
    RM_TRY(ws(ctx, "temporal_R", (size_t)ctx->op_nk * T + 1, &dR));
    RM_TRY(ws(ctx, "temporal_C", (size_t)ctx->op_nk * Th + 1, &dC));
    op->R = dR; op->C = dC; op->nk = ctx->op_nk;
    if (ctx->op_mfma) {
        double *dRf = nullptr, *dCf = nullptr;
        RM_TRY(ws(ctx, "temporal_Rf", (size_t)nks * 2 * ctx->op_mfma * 64, &dRf));
        RM_TRY(ws(ctx, "temporal_Cf", (size_t)mt * 8 * ctx->op_mfma * 64, &dCf));
        op->Rf = dRf; op->Cf = dCf; op->tiles = ctx->op_mfma;
    }
    return RM_OK;
}

// out[Th, NP] = amp * Cz (Rz x), x[T, NP]: the Th = T / 2 + 1 unique frames of the band-passed signal (transforms.py:86-99);
// full = true: out is [T, NP] and the mirrored frames are stored as well
static int launch_temporal(rm_ctx *ctx, const double *x, int T, size_t NP, const TemporalOp &op, double amp, double *out, hipStream_t s,
                           CollapseState *st_init = nullptr, bool full = false)
{
    const int Th = sym_frames(T);
    if (op.nk == 0) {  // nothing survives the mask
        HIP_TRY(hipMemsetAsync(out, 0, sizeof(double) * (size_t)(full ? T : Th) * NP, s));
        if (st_init) { hipLaunchKernelGGL(k_state_init, dim3(1), dim3(NSTRIPE), 0, s, st_init); LAUNCH_CHECK(); }
        return RM_OK;
    }
    const int mirror_n = full ? T : 0;
#ifndef RM_HIPEMU
    // large levels: one wave per 16 pixel columns, no K-split (k_temporal_sym_px); the choice depends on (T, NP) only
    int cus_t = 256;
    (void)hipDeviceGetAttribute(&cus_t, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const bool wide = ctx->dbg.temporal_wide >= 0 ? ctx->dbg.temporal_wide != 0 : NP >= (size_t)64 * 4 * cus_t;
    if (op.Rf && !ctx->dbg.temporal_valu && wide) {
        const dim3 grid((unsigned)((NP + 63) / 64)), block(256);
        if (op.tiles == 1) hipLaunchKernelGGL((k_temporal_sym_px<1>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else if (op.tiles == 2) hipLaunchKernelGGL((k_temporal_sym_px<2>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else hipLaunchKernelGGL((k_temporal_sym_px<3>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        LAUNCH_CHECK();
        return RM_OK;
    }
    if (op.Rf && !ctx->dbg.temporal_valu) {
        const dim3 grid((unsigned)((NP + 15) / 16)), block(64 * TM_W);
        if (op.tiles == 1) hipLaunchKernelGGL((k_temporal_sym<1>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else if (op.tiles == 2) hipLaunchKernelGGL((k_temporal_sym<2>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        else hipLaunchKernelGGL((k_temporal_sym<3>), grid, block, 0, s, x, T, NP, op.Rf, op.Cf, amp, out, mirror_n, st_init);
        LAUNCH_CHECK();
        return RM_OK;
    }
#endif
    const size_t sh1 = sizeof(double) * (size_t)T * TF_KC, sh2 = sizeof(double) * (size_t)op.nk * TF_SC;
    if (sh1 > 64 * 1024 || sh2 > 64 * 1024) return fail(RM_E_UNSUPPORTED, "temporal filter: T=%d exceeds the LDS-staged operator (T <= 2048)", T);
    double *y = nullptr;
    RM_TRY(ws(ctx, "temporal_y", (size_t)op.nk * NP, &y));
    dim3 g1((unsigned)((NP + 63) / 64), (op.nk + TF_KC - 1) / TF_KC), g2((unsigned)((NP + 63) / 64), (Th + TF_SC - 1) / TF_SC);
    hipLaunchKernelGGL(k_temporal_fwd, g1, dim3(64), sh1, s, x, T, NP, op.R, op.nk, y, st_init);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_temporal_inv, g2, dim3(64), sh2, s, y, op.nk, NP, op.C, Th, amp, out, mirror_n);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_temporal_bandpass_filter_fft(rm_ctx *ctx, const double *data, int T, size_t npix, double fps, double fmin,
                                               double fmax, double amp, double *out, void *stream)
{
    if (!ctx || !data || !out || T < 1 || !(fps > 0)) return fail(RM_E_BADARG, "rm_temporal_bandpass_filter_fft: bad argument");
    if (npix == 0) return RM_OK;
    if (data == out) return fail(RM_E_BADARG, "rm_temporal_bandpass_filter_fft: in-place filtering is not supported");
    hipStream_t s = (hipStream_t)stream;
    TemporalOp op;
    RM_TRY(get_operator(ctx, T, fps, fmin, fmax, &op, s));
    return launch_temporal(ctx, data, T, npix, op, amp, out, s, nullptr, true);
}

extern "C" int rm_time_average(rm_ctx *ctx, const void *data, int dtype, int T, size_t npix, double *out, void *stream)
{
    if (!ctx || !data || !out || T < 1 || !valid_dtype(dtype)) return fail(RM_E_BADARG, "rm_time_average: bad argument");
    if (npix == 0) return RM_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_time_average<uint8_t>), grid, block, 0, s, (const uint8_t *)data, T, npix, out); break;
    case RM_F16: hipLaunchKernelGGL((k_time_average<__half>), grid, block, 0, s, (const __half *)data, T, npix, out); break;
    case RM_F32: hipLaunchKernelGGL((k_time_average<float>), grid, block, 0, s, (const float *)data, T, npix, out); break;
    default: hipLaunchKernelGGL((k_time_average<double>), grid, block, 0, s, (const double *)data, T, npix, out); break;
    }
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_lfilter(rm_ctx *ctx, const double *data, int T, size_t npix, const double *b_host, const double *a_host, int ncoef,
                          double scale, double *out, void *stream)
{
    if (!ctx || !data || !out || !b_host || !a_host || T < 1 || ncoef < 1) return fail(RM_E_BADARG, "rm_lfilter: bad argument");
    if (ncoef > IIR_MAX) return fail(RM_E_UNSUPPORTED, "rm_lfilter: %d coefficients > %d", ncoef, IIR_MAX);
    if (a_host[0] == 0.0) return fail(RM_E_BADARG, "rm_lfilter: a[0] must not be zero");
    if (npix == 0) return RM_OK;
    if (data == out) return fail(RM_E_BADARG, "rm_lfilter: in-place filtering is not supported");
    IirCoef c;
    c.n = ncoef;
    for (int i = 0; i < IIR_MAX; ++i) {  // scipy normalises both polynomials by a[0] first
        c.b[i] = i < ncoef ? b_host[i] / a_host[0] : 0.0;
        c.a[i] = i < ncoef ? a_host[i] / a_host[0] : 0.0;
    }
    hipLaunchKernelGGL(k_lfilter, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, (hipStream_t)stream, data, T, npix, c, scale, out);
    LAUNCH_CHECK();
    return RM_OK;
}

// transforms.py:184-192 on a materialised [n] array: min, max, top = max - (max - min) * threshold,
// masked = raw with every value >= top replaced by min
extern "C" int rm_threshold_mask(rm_ctx *ctx, const double *raw, size_t n, double threshold, double *masked, double *minmax_host,
                                 void *stream)
{
    if (!ctx || !raw || n == 0) return fail(RM_E_BADARG, "rm_threshold_mask: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    CollapseState *st = ctx->d_state;
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(NSTRIPE), 0, s, st);
    ctx->state_fresh = false;   // this call reduces into the state
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_plain, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, raw, n, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_minmax, dim3(1), dim3(NSTRIPE), 0, s, st, threshold);
    LAUNCH_CHECK();
    if (masked) {
        hipLaunchKernelGGL(k_mask_plain, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, raw, n, st, masked);
        LAUNCH_CHECK();
    }
    if (minmax_host) {
        HIP_TRY(hipMemcpyAsync(ctx->h_state, st, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
        HIP_TRY(stream_wait(s));
        minmax_host[0] = ctx->h_state->min_val;
        minmax_host[1] = ctx->h_state->max_val;
    }
    return RM_OK;
}

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
    if (down_chain_hot_ok(S, h.data())) return launch_down_chain_g<Tin, false>(f, T, g, out, s);
    return launch_down_chain_g<Tin, true>(f, T, g, out, s);
}

static int dtype_vec(int dtype) { return dtype == RM_F64 ? 2 : dtype == RM_F32 ? 4 : dtype == RM_F16 ? 8 : 16; }
static size_t dtype_size(int dtype) { return dtype == RM_F64 ? 8 : dtype == RM_F32 ? 4 : dtype == RM_F16 ? 2 : 1; }

// frames[T,H,W] -> G_S[T,h_S,w_S] in one launch
static int launch_down_chain(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S,
                             double *out, hipStream_t s, bool tiny)
{
    const int V = dtype_vec(dtype);
    const size_t esz = dtype_size(dtype);
    const bool vec_ok = (w[0] % V == 0) && (((size_t)h[0] * w[0] * esz) % 16 == 0) && (((uintptr_t)frames) % 16 == 0);
    if (S < 1 || S > 5) return fail(RM_E_UNSUPPORTED, "fused pyrDown chain supports 1..5 levels, got %d", S);
    const int vo = vec_ok ? 1 : 0;
    if ((dtype == RM_U8 || dtype == RM_F16 || dtype == RM_F32) && vec_ok && !ctx->dbg.dc_lds_front_end) {
        // all-register variant for narrow frame buffers (rm_down_chain_u8.h): a lane owns 16 adjacent pixels
        DownGeom g8;
        if (make_down_geom_u8(S, h.data(), w.data(), T, g8, tiny, ctx->dbg.dc_segs, dtype == RM_F32 ? 1536 : 2048)) {
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
    }
    switch (dtype) {
    case RM_U8: return launch_down_chain_t<uint8_t>(ctx, frames, T, h, w, S, vo, out, s, tiny);
    case RM_F16: return launch_down_chain_t<__half>(ctx, frames, T, h, w, S, vo, out, s, tiny);
    case RM_F32: return launch_down_chain_t<float>(ctx, frames, T, h, w, S, vo, out, s, tiny);
    case RM_F64: return launch_down_chain_t<double>(ctx, frames, T, h, w, S, vo, out, s, tiny);
    }
    return fail(RM_E_BADARG, "unknown dtype %d", dtype);
}

// ------------------------------------------------------------------------------------------
// front half of calibration in two steps:
//   front_pyramid: frames[T,H,W] -> Laplacian levels S..L-2 side by side, lap[T,NP]     (per frame)
//   front_filter : lap[T,NP]     -> collapsed band-passed level S, C_S[T,hS,wS]         (needs every frame)
// rm_calibrate runs them back to back; the frame-sharded path (rm_shard_*) all-gathers lap in between.
// ------------------------------------------------------------------------------------------
struct SmallLevels {
    std::vector<int> h, w;
    int S = 0;             // level the collapse stopped at (== skip when any level is filtered)
    const double *cS = nullptr;
    bool all_zero = false;  // no level is filtered: the band-passed pyramid is all zeros
    bool state_ready = false;  // the collapse kernel that produced cS has already reset ctx->d_state (k_state_init's job)
    bool bounds_ready = false; // ... and left the tile bounds in the workspace buffers tile_lo / tile_hi and their extrema in the state
};

struct PyrGeom {
    std::vector<int> h, w;
    std::vector<size_t> off;   // offset of level l inside a [NP] frame of the small pyramid (levels S..L-2)
    size_t NP = 0;             // filtered pixels per frame
    size_t lds_levels = 0;     // doubles needed to hold G_S..G_{L-1} of one frame
    int S = 0, L = 0;
    bool all_zero = false;
    bool chain = false;        // the fused pyrDown chain builds G_S
    bool fuse_small = false;   // per-frame LDS kernels build / collapse the small pyramid
    bool filter_first = false; // ... in the filter-first form (k_small_filter_first): the [T, NP] array between the stages is G_S itself
    bool ff_levels = false;    // filter-first with one launch per level (the small pyramid does not fit LDS): same arithmetic
    SmallGeom sg;
};

static void pyr_geom(int H, int W, int levels, int skip, unsigned flags, PyrGeom &pg)
{
    level_sizes(H, W, levels, pg.h, pg.w);
    const int L = levels, S = skip;
    pg.L = L; pg.S = S;
    pg.all_zero = skip >= L - 1;
    if (pg.all_zero) { pg.S = 0; return; }
    pg.chain = S >= 1 && S <= 5 && !(flags & RM_FLAG_UNFUSED_DOWN);
    pg.off.assign(L, 0);
    pg.NP = 0;
    for (int l = S; l <= L - 2; ++l) { pg.off[l] = pg.NP; pg.NP += (size_t)pg.h[l] * pg.w[l]; }
    pg.lds_levels = 0;
    for (int l = S; l < L; ++l) pg.lds_levels += (size_t)pg.h[l] * pg.w[l];
    const size_t LDS_LIMIT = 150 * 1024;
    pg.fuse_small = pg.chain && !(flags & RM_FLAG_UNFUSED_SMALL) && L <= SMALL_MAX_LEVELS &&
                    pg.lds_levels * sizeof(double) <= LDS_LIMIT && pg.NP * sizeof(double) <= LDS_LIMIT;
    if (pg.fuse_small) {
        SmallGeom &sg = pg.sg;
        sg.S = S; sg.L = L; sg.NP = (int)pg.NP;
        int o = 0;
        for (int l = 0; l < L; ++l) {
            sg.h[l] = pg.h[l]; sg.w[l] = pg.w[l];
            sg.g_off[l] = 0; sg.np_off[l] = (int)pg.off[l];
            if (l >= S) { sg.g_off[l] = o; o += pg.h[l] * pg.w[l]; }
        }
        // filter-first form (rm_kernels.h k_small_filter_first): the Gaussian levels and the row-extrema table of the tile
        // bounds must fit LDS together
        if (!(flags & RM_FLAG_FILTER_LAPLACIANS) && S >= 1 && S < MAX_CHAIN) {
            const size_t nS = (size_t)pg.h[S] * pg.w[S];
            const size_t tiles_x = (size_t)(W + CT_W - 1) / CT_W;
            const size_t need = sizeof(double) * (pg.lds_levels + 2 * (size_t)pg.h[S] * tiles_x);
            if (need <= LDS_LIMIT && !(flags & RM_FLAG_FF_PER_LEVEL)) { pg.filter_first = true; pg.NP = nS; }   // (flag: test hook)
        }
    }
    // the same form with one launch per pyramid level when the levels are too large for LDS (4K, skip 2): the temporal filter
    // runs over G_S only and the Laplacian levels are never materialised
    if (pg.chain && !pg.filter_first && (!pg.fuse_small || (flags & RM_FLAG_FF_PER_LEVEL)) &&
        !(flags & (RM_FLAG_FILTER_LAPLACIANS | RM_FLAG_UNFUSED_SMALL)) && S >= 1 && S < MAX_CHAIN) {
        pg.ff_levels = true; pg.fuse_small = false; pg.NP = (size_t)pg.h[S] * pg.w[S];
    }
}

static int front_pyramid(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, const PyrGeom &pg, unsigned flags,
                         double *lap, hipStream_t s)
{
    const std::vector<int> &h = pg.h, &w = pg.w;
    const int L = pg.L, S = pg.S;
    const size_t NP = pg.NP;
    // Gaussian chain (pyramid.py:9-17).  Levels < S are stepping stones (ping-pong scratch);
    // levels S..L-1 are kept for the Laplacians.
    std::vector<double *> g(L, nullptr);
    const void *cur = frames; int cur_dtype = dtype;
    int first = 1;
    if (pg.filter_first || pg.ff_levels) {
        // filter-first form: the array the stages exchange is G_S itself; the small pyramid is built after the temporal filter
        PhaseTimer pt(ctx, 0, s);
        RM_TRY(launch_down_chain(ctx, frames, dtype, T, h, w, S, lap, s, (flags & RM_FLAG_TINY_STRIPS) != 0));
        ctx->state_fresh = false;
        return RM_OK;
    }
    if (pg.chain) {
        // one launch reads the frame buffer once and writes only G_S
        double *dst = nullptr;
        RM_TRY(ws(ctx, "g" + std::to_string(S), (size_t)T * h[S] * w[S], &dst));
        {
            PhaseTimer pt(ctx, 0, s);
            RM_TRY(launch_down_chain(ctx, frames, dtype, T, h, w, S, dst, s, (flags & RM_FLAG_TINY_STRIPS) != 0));
        }
        g[S] = dst; cur = dst; cur_dtype = RM_F64;
        first = S + 1;
    }
    if (!pg.fuse_small) {
        for (int l = first; l < L; ++l) {
            double *dst = nullptr;
            if (l < S) RM_TRY(ws(ctx, (l & 1) ? "g_ping" : "g_pong", (size_t)T * h[l] * w[l], &dst));
            else RM_TRY(ws(ctx, "g" + std::to_string(l), (size_t)T * h[l] * w[l], &dst));
            {
                PhaseTimer pt(ctx, (l == 1) ? 0 : 1, s);
                RM_TRY(launch_pyr_down(cur, cur_dtype, T, h[l - 1], w[l - 1], dst, s));
            }
            g[l] = dst; cur = dst; cur_dtype = RM_F64;
        }
    }
    if (S == 0) {
        double *g0 = nullptr;
        RM_TRY(ws(ctx, "g0", (size_t)T * H * W, &g0));
        RM_TRY(launch_to_f64(frames, dtype, (size_t)T * H * W, g0, s));
        g[0] = g0;
    }
    PhaseTimer pt_small(ctx, 1, s);
    // The filtered levels S .. L-2 live side by side in [T, NP] buffers (level S first), so the temporal
    // filter is two launches for the whole small pyramid.
    if (pg.fuse_small) {
        // Gaussian levels S+1..L-1 and all Laplacians in one launch, one workgroup per frame, in LDS
        const size_t shmem = pg.lds_levels * sizeof(double);
        if (shmem > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_pyramid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k_small_pyramid, dim3(T), dim3(SMALL_NT), shmem, s, (const double *)g[S], pg.sg, lap, ctx->d_state);
        LAUNCH_CHECK();
        ctx->state_fresh = true;
    } else {
        // Laplacian levels (pyramid.py:23-26): L_l = G_l - pyrUp(G_{l+1})
        for (int l = L - 2; l >= S; --l)
            RM_TRY(launch_pyr_up(g[l + 1], T, h[l + 1], w[l + 1], lap + pg.off[l], h[l], w[l], 1, g[l], s, 0, NP, 0));
    }
    return RM_OK;
}

static int make_geom(const SmallLevels &sl, ChainGeom &g);

static int front_filter(rm_ctx *ctx, const double *lap, int T, const PyrGeom &pg, double fps, double fmin, double fmax, double amp,
                        SmallLevels &out, hipStream_t s)
{
    const std::vector<int> &h = pg.h, &w = pg.w;
    const int L = pg.L, S = pg.S;
    const size_t NP = pg.NP;
    const int Th = sym_frames(T);   // the band-passed signal is even in time: everything below handles the unique frames only
    out.h = pg.h; out.w = pg.w; out.S = S; out.all_zero = false;
    // consumed here, on every path: whatever follows reduces into d_state, so the reset by front_pyramid's last kernel
    // vouches for this call only (a later rm_shard_collapse with a foreign lap buffer must reset the state itself)
    const bool state_fresh = ctx->state_fresh;
    ctx->state_fresh = false;
    TemporalOp op;
    RM_TRY(get_operator(ctx, T, fps, fmin, fmax, &op, s));
    PhaseTimer pt_small(ctx, 1, s);
    double *bp = nullptr;
    RM_TRY(ws(ctx, "bp_all", (size_t)Th * NP, &bp));
    if (pg.filter_first) {
        // X = B(G_S) for the unique frames (its workgroup 0 resets the reduction state), then ONE per-frame kernel: Gaussian levels
        // of X, Laplacians, collapse to C_S, tile bounds and lattice samples
        RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s, ctx->d_state));
        ChainGeom cg;
        SmallLevels probe; probe.h = pg.h; probe.w = pg.w; probe.S = S;
        RM_TRY(make_geom(probe, cg));
        const long long npairs = (long long)cg.tiles_x * cg.tiles_y * Th;
        if (npairs >= (1ll << 31)) return fail(RM_E_UNSUPPORTED, "calibration: %lld (tile, frame) pairs exceed 2^31", npairs);
        double *dst = nullptr, *lo = nullptr, *hi = nullptr;
        int *sel_cnt = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * NP, &dst));
        RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &lo));
        RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &hi));
        RM_TRY(ws(ctx, "sel_cnt", (size_t)cg.tiles_x * cg.tiles_y, &sel_cnt));
        const size_t sh = sizeof(double) * (pg.lds_levels + 2 * (size_t)h[S] * cg.tiles_x);
        if (sh > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_filter_first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        // two workgroups per frame when one per frame leaves CUs idle and the frame has tile rows to share
        int cus_ff = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus_ff, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        int parts = (2 * Th <= cus_ff && cg.tiles_y >= 8) ? 2 : 1;   // (120 KB of LDS: one workgroup per CU, so 2 Th must fit the chip in ONE round -- 258 workgroups at T = 256 took 45 us instead of 27)
        if (ctx->dbg.ff_parts > 0) parts = std::min(ctx->dbg.ff_parts, std::max(1, cg.tiles_y / 2));
        hipLaunchKernelGGL(k_small_filter_first, dim3(Th * parts), dim3(SMALL_NT), sh, s, (const double *)bp, pg.sg, (int)pg.lds_levels, dst, ctx->d_state, cg,
                           cg.tiles_x * cg.tiles_y, lo, hi, sel_cnt, parts);
        LAUNCH_CHECK();
        out.state_ready = true; out.bounds_ready = true;
        out.cS = dst;
        return RM_OK;
    }
    if (pg.ff_levels) {
        // X_S = B(G_S); X_{L-1} = pyrDown^(L-1-S)(X_S); U_{L-1} = X_{L-1}, U_l = pyrUp(U_{l+1}); C_S = X_S - pyrUp(U_{S+1})
        // (rm_kernels.h k_small_filter_first: the telescoped collapse, here with one launch per step): only the COARSEST level of
        // the filtered pyramid is needed, so the way down is the fused pyrDown chain (rm_down_chain.h) on the float64 level X_S
        RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s, ctx->d_state));   // (its workgroup 0 resets the reduction state: no k_state_init launch)
        out.state_ready = true;
        std::vector<double *> x(L, nullptr);
        x[S] = bp;
        const int depth = L - 1 - S;
        const bool fused_down = depth >= 1 && depth <= 5;
        auto level_buf = [&](int l) { return x[l] ? RM_OK : ws(ctx, "g" + std::to_string(l), (size_t)Th * h[l] * w[l], &x[l]); };
        RM_TRY(level_buf(L - 1));
        if (fused_down) {
            std::vector<int> hh(h.begin() + S, h.end()), ww(w.begin() + S, w.end());
            RM_TRY(launch_down_chain(ctx, bp, RM_F64, Th, hh, ww, depth, x[L - 1], s, false));
        } else {
            for (int l = S + 1; l < L; ++l) { RM_TRY(level_buf(l)); RM_TRY(launch_pyr_down(x[l - 1], RM_F64, Th, h[l - 1], w[l - 1], x[l], s)); }
        }
        double *dst = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * NP, &dst));
        // the way back up and the subtraction in one launch (k_ff_collapse): levels S .. L-1 as a pyrUp chain of `depth` steps
        SmallLevels up; up.S = depth;
        up.h.assign(h.begin() + S, h.end()); up.w.assign(w.begin() + S, w.end());
        ChainGeom ug;
        if (depth >= 1 && depth < MAX_CHAIN && make_geom(up, ug) == RM_OK && (long long)ug.tiles_x * ug.tiles_y * Th < (1ll << 31)) {
            const int utiles = ug.tiles_x * ug.tiles_y, nitems = utiles * Th;
            const size_t sh = sizeof(double) * (size_t)ug.lds_total;
            if (sh > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_ff_collapse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL(k_ff_collapse, dim3((unsigned)std::min(nitems, 256 * 64)), dim3(64), sh, s, (const double *)bp, (const double *)x[L - 1],
                               ug, utiles, nitems, dst);
            LAUNCH_CHECK();
        } else {
            for (int l = L - 2; l > S; --l) {
                RM_TRY(level_buf(l));
                RM_TRY(launch_pyr_up(x[l + 1], Th, h[l + 1], w[l + 1], x[l], h[l], w[l], 0, nullptr, s));
            }
            RM_TRY(launch_pyr_up(x[S + 1], Th, h[S + 1], w[S + 1], dst, h[S], w[S], 1, bp, s));
        }
        out.cS = dst;
        return RM_OK;
    }
    // temporal band-pass of every level at once (transforms.py:162,169)
    RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s));
    // collapse of the band-passed levels L-2 .. S (pyramid.py:51-57; the coarsest level is zeros: 0 + x == x);
    // the result is a contiguous [Th,h_S,w_S] array for the full-resolution passes
    const double *c = bp + pg.off[L - 2];
    if (L - 2 == S) {
        // single filtered level: NP == h_S*w_S, bp_all is already C_S
    } else if (pg.fuse_small) {
        double *dst = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * h[S] * w[S], &dst));
        const size_t shmem = NP * sizeof(double);
        if (shmem > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_collapse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        // when the row-extrema table of a frame fits beside its small pyramid, the tile bounds of the collapse passes
        // are taken here, from the LDS copy of C_S (k_small_collapse_bounds)
        ChainGeom cg;
        SmallLevels probe; probe.h = pg.h; probe.w = pg.w; probe.S = S;
        const bool geom_ok = S >= 1 && S < MAX_CHAIN && make_geom(probe, cg) == RM_OK;
        const size_t tbl = geom_ok ? 2 * sizeof(double) * (size_t)h[S] * cg.tiles_x : 0;
        const long long npairs = geom_ok ? (long long)cg.tiles_x * cg.tiles_y * Th : 0;
        if (geom_ok && shmem + tbl <= 150 * 1024 && npairs < (1ll << 31) && !ctx->dbg.no_fused_bounds) {
            double *lo = nullptr, *hi = nullptr;
            int *sel_cnt = nullptr;
            RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &lo));
            RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &hi));
            RM_TRY(ws(ctx, "sel_cnt", (size_t)cg.tiles_x * cg.tiles_y, &sel_cnt));
            if (!state_fresh) {   // the lap buffer did not come from front_pyramid on this context just now
                hipLaunchKernelGGL(k_state_init, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
                LAUNCH_CHECK();
            }
            const size_t sh2 = shmem + tbl;
            if (sh2 > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)k_small_collapse_bounds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
            hipLaunchKernelGGL(k_small_collapse_bounds, dim3(Th), dim3(SMALL_NT), sh2, s, (const double *)bp, pg.sg, dst, ctx->d_state, cg,
                               cg.tiles_x * cg.tiles_y, lo, hi, sel_cnt);
            out.state_ready = true; out.bounds_ready = true;
        } else {
            hipLaunchKernelGGL(k_small_collapse, dim3(Th), dim3(SMALL_NT), shmem, s, (const double *)bp, pg.sg, dst, ctx->d_state);
            out.state_ready = true;
        }
        LAUNCH_CHECK();
        c = dst;
    } else {
        size_t c_fs = NP;
        for (int l = L - 3; l >= S; --l) {
            double *dst = bp + pg.off[l];
            size_t dst_fs = NP;
            if (l == S) {
                RM_TRY(ws(ctx, "cS", (size_t)Th * h[S] * w[S], &dst));
                dst_fs = (size_t)h[S] * w[S];
            }
            RM_TRY(launch_pyr_up(c, Th, h[l + 1], w[l + 1], dst, h[l], w[l], 2, bp + pg.off[l], s, c_fs, dst_fs, NP));
            c = dst; c_fs = dst_fs;
        }
    }
    out.cS = c;
    return RM_OK;
}

static int front_half(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax,
                      double amp, int levels, int skip, unsigned flags, SmallLevels &out, hipStream_t s)
{
    PyrGeom pg;
    pyr_geom(H, W, levels, skip, flags, pg);
    out.h = pg.h; out.w = pg.w;
    if (pg.all_zero) { out.all_zero = true; out.S = 0; return RM_OK; }
    double *lap = nullptr;
    RM_TRY(ws(ctx, "lap_all", (size_t)T * pg.NP, &lap));
    RM_TRY(front_pyramid(ctx, frames, dtype, T, H, W, pg, flags, lap, s));
    return front_filter(ctx, lap, T, pg, fps, fmin, fmax, amp, out, s);
}

static int make_geom(const SmallLevels &sl, ChainGeom &g)
{
    const int S = sl.S;
    if (S < 1 || S >= MAX_CHAIN) return fail(RM_E_UNSUPPORTED, "fused collapse supports 1 <= skip_levels_at_top <= %d", MAX_CHAIN - 1);
    g.S = S;
    for (int k = 0; k <= S; ++k) { g.h[k] = sl.h[k]; g.w[k] = sl.w[k]; }
    // LDS layout of one evaluation workgroup (doubles).  Step k -> k-1 of the chain needs level k, level k-1 and the scratch of
    // its horizontal pass (every source row of level k at the destination columns of level k-1); level k is dead afterwards:
    //   [ level 1 ][ level 2 ][ B ]   B = levels 3 .. S and the scratch of steps S .. 3 behind them, reused as the (largest)
    //                                     scratch of step 2 -> 1 once those levels are dead
    // 889 doubles at S = 4 instead of 1034 side by side: 22 single-wave workgroups per CU instead of 18.
    auto lvl = [](int k) { return (chain_extent(CT_H, k) + 1) * (chain_extent(CT_W, k) + 1); };
    auto scratch = [](int k) { return (chain_extent(CT_H, k) + 1) * (chain_extent(CT_W, k - 1) + 1); };
    for (int k = 0; k < MAX_CHAIN; ++k) { g.lds_off[k] = 0; g.lds_hb[k] = 0; }
    int off = lvl(1);
    if (S >= 2) { g.lds_off[2] = off; off += lvl(2); }
    const int B = off;
    int small = 0, hb_small = 0;
    for (int k = 3; k <= S; ++k) { g.lds_off[k] = B + small; small += lvl(k); hb_small = std::max(hb_small, scratch(k)); }
    for (int k = 3; k <= S; ++k) g.lds_hb[k] = B + small;
    if (S >= 2) g.lds_hb[2] = B;
    g.lds_total = B + (S >= 2 ? std::max(scratch(2), S >= 3 ? small + hb_small : 0) : 0);
    g.tiles_x = (sl.w[0] + CT_W - 1) / CT_W;
    g.tiles_y = (sl.h[0] + CT_H - 1) / CT_H;
    // weights of the lattice samples (rm_kernels.h lattice_sample): a unit impulse pushed through S interior 1-D pyrUp
    // steps (even: (s[j-1] + 6 s[j] + s[j+1]) / 8, odd: (s[j] + s[j+1]) / 2), read at position 1 << S of a 3-pixel line
    {
        double w[3];
        for (int k = 0; k < 3; ++k) {
            std::vector<double> v(5, 0.0);
            v[1 + k] = 1.0;                      // pixels y-1, y, y+1 sit at 1, 2, 3; 0 and 4 are never reached from the lattice point
            int centre = 2;
            for (int i = 0; i < S; ++i) {
                std::vector<double> u(2 * v.size(), 0.0);
                for (size_t j = 0; j < v.size(); ++j) {
                    const double a = j > 0 ? v[j - 1] : 0.0, c = j + 1 < v.size() ? v[j + 1] : 0.0;
                    u[2 * j] = (a + 6 * v[j] + c) / 8;
                    u[2 * j + 1] = (v[j] + c) / 2;
                }
                v.swap(u);
                centre *= 2;
            }
            w[k] = v[centre];
        }
        g.lat_a = w[0]; g.lat_b = w[1];   // w[2] == w[0]
    }
    return RM_OK;
}

static int heatmap_to_roi_impl(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                               uint8_t *binary, void *stream, bool have_minmax);
__global__ __launch_bounds__(NSTRIPE) void k_heat_state_init(CollapseState *st);

// nothing is filtered (skip >= levels - 1): the band-passed pyramid, raw and the heatmap are all zeros.  The
// heatmap extrema (0, 0) go into the state like after a real calibration, so rm_locate normalises 0/0 -> NaN
// -> uint8 0 -> no contour, as the reference does (base.py:563-570).
static int zero_result(rm_ctx *ctx, size_t npix, double *heat, double *minmax_host, hipStream_t s)
{
    ctx->state_fresh = false;
    HIP_TRY(hipMemsetAsync(heat, 0, sizeof(double) * npix, s));
    hipLaunchKernelGGL(k_heat_state_init, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_heat_minmax, dim3(1), dim3(256), 0, s, (const double *)heat, (size_t)1, ctx->d_state);
    LAUNCH_CHECK();
    if (minmax_host) { minmax_host[0] = 0.0; minmax_host[1] = 0.0; HIP_TRY(stream_wait(s)); }
    return RM_OK;
}

// the flat evaluation pass over the listed pairs (exact extrema; values of the kept pairs into the value store)
static int launch_eval_pairs(rm_ctx *ctx, const CollapsePlan &cp, hipStream_t s)
{
    CollapseState *st = ctx->d_state;
    const ChainGeom &g = cp.g;
    const int ntiles = cp.ntiles, npairs = cp.npairs, Th = sym_frames(cp.T);
    const SumPlan &sp = cp.sp;
    struct { const double *cS; int S; } sl{cp.cS, cp.S};
    // one resident round of single-wave workgroups that loop over the lists: their lengths live on the device, and
    // dispatching thousands of workgroups that find nothing to do costs more than the loop.  "Resident" is what the
    // kernel's registers and this geometry's LDS footprint allow per CU (asked of the runtime once per footprint).
    unsigned egrid = 64;   // (host emulation: a fiber per lane -- few, looping workgroups compute the same thing)
#ifndef RM_HIPEMU
    {
        size_t &cached_shmem = ctx->eval_shmem;
        int &cached_per_cu = ctx->eval_per_cu, &cached_cus = ctx->eval_cus;
        if (cached_shmem != cp.shmem) {
            int per_cu = 0, cus = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_eval_pairs, 64, cp.shmem));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
            per_cu -= 1;   // (measured: the runtime's figure ignores the LDS allocation granule -- its last workgroup queues)
#ifdef RM_EVAL_PER_CU
            per_cu = RM_EVAL_PER_CU;
#endif
            cached_per_cu = per_cu < 1 ? 1 : per_cu; cached_cus = cus < 1 ? 1 : cus; cached_shmem = cp.shmem;
        }
        const long long capw = (long long)cached_per_cu * cached_cus;
        egrid = (unsigned)(npairs < capw ? npairs : capw);
    }
#else
    if ((long long)egrid > npairs) egrid = (unsigned)npairs;
#endif
    if (tile_eval_ok(g) && ctx->dbg.eval_fast) {
        // the wave-private evaluator (rm_tile_eval.h): ~70 VGPRs and < 5 KB of LDS per single-wave workgroup -- 24 per CU stay resident
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        const unsigned fgrid = (unsigned)std::min<long long>(npairs, 12ll * cus);   // (launching more single-wave workgroups than pairs costs ~0.5 us of ramp per thousand)
#else
        const unsigned fgrid = egrid;
#endif
#define RM_EVAL_FAST(SS)                                                                                                              \
        do {                                                                                                                          \
            using FootE = TileFoot<SS, false>;                                                                                        \
            hipLaunchKernelGGL((k_eval_pairs_fast<SS>), dim3(fgrid), dim3(64), sizeof(double) * FootE::TOTAL, s, sl.cS, g, ntiles, cp.list_a, cp.list_b, \
                               cp.slot_of, st, cp.store, sp, Th);                                                                     \
        } while (0)
        switch (sl.S) { case 1: RM_EVAL_FAST(1); break; case 2: RM_EVAL_FAST(2); break; case 3: RM_EVAL_FAST(3); break; default: RM_EVAL_FAST(4); break; }
#undef RM_EVAL_FAST
        (void)cus;
    } else {
        hipLaunchKernelGGL(k_eval_pairs, dim3(egrid), dim3(64), cp.shmem, s, sl.cS, g, ntiles, cp.list_a, cp.list_b, cp.slot_of, st, cp.store, sp, Th);
    }
    LAUNCH_CHECK();
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// back half: C_S -> exact raw.min()/raw.max() -> masked time sum, for the frames [t0, t1) of the buffer.
// The bounds and the pruning decisions always cover all T frames (they are cheap and every rank of a
// frame-sharded run must agree on them); full-resolution evaluation and the sum touch only [t0, t1).
// ------------------------------------------------------------------------------------------
static int collapse_eval(rm_ctx *ctx, const SmallLevels &sl, int T, int t0, int t1, double thr, unsigned flags, CollapsePlan &cp,
                         hipStream_t s)
{
    CollapseState *st = ctx->d_state;
    cp.valid = false;
    cp.cS = sl.cS; cp.T = T; cp.t0 = t0; cp.t1 = t1; cp.H = sl.h[0]; cp.W = sl.w[0]; cp.S = sl.S;
    const size_t npix = (size_t)cp.H * cp.W;
    const int Th = sym_frames(T);   // C_S, the bounds and the pairs exist for the unique frames only (rm_kernels.h sym_frame)
    if (!sl.state_ready) {
        hipLaunchKernelGGL(k_state_init, dim3(1), dim3(NSTRIPE), 0, s, st);
        LAUNCH_CHECK();
    }
    const int no_prune = (flags & RM_FLAG_NO_PRUNE) ? 1 : 0;
    if (sl.S == 0) {
        if (t0 != 0 || t1 != T) return fail(RM_E_UNSUPPORTED, "frame-sharded calibration needs skip_levels_at_top >= 1");
        size_t n = (size_t)Th * npix;
        hipLaunchKernelGGL(k_minmax_plain, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, sl.cS, n, st);
        LAUNCH_CHECK();
        cp.valid = true;
        return RM_OK;
    }
    ChainGeom &g = cp.g;
    RM_TRY(make_geom(sl, g));
    const int ntiles = g.tiles_x * g.tiles_y;
    const long long npairs_ll = (long long)ntiles * Th;
    if (npairs_ll >= (1ll << 31)) return fail(RM_E_UNSUPPORTED, "calibration: %lld (tile, frame) pairs exceed 2^31", npairs_ll);
    const int npairs = (int)npairs_ll;
    cp.ntiles = ntiles; cp.npairs = npairs;
    int mine_frames = 0;   // unique frames this rank's frame range [t0, t1) holds
    for (int u = 0; u < Th; ++u) mine_frames += sym_in_range(u, T, t0, t1) ? 1 : 0;
    const long long npairs_mine = (long long)ntiles * mine_frames;
    RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &cp.lo));
    RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &cp.hi));
    RM_TRY(ws(ctx, "pair_list_a", (size_t)npairs, &cp.list_a));
    RM_TRY(ws(ctx, "pair_list_b", (size_t)npairs, &cp.list_b));
    RM_TRY(ws(ctx, "pair_slot", (size_t)npairs, &cp.slot_of));
    // The value store: 8 KB slots for the pairs the selection keeps, handed out tile by tile (rm_kernels.h k_select_pairs).
    // Capped at STORE_BUDGET_SLOTS (1 GiB): a selection that keeps more takes the dense sum kernel, which needs no store --
    // decided on the device, in this same call (sum_is_dense).  The exhaustive-evaluation baseline (RM_FLAG_NO_PRUNE) and a
    // forced sparse path park every pair they are told to, so they get a slot per pair.
    // (round 4) The store starts at STORE_DEFAULT_SLOTS (128 MiB) and GROWS when a selection of this context has overflowed it
    // (rm_locate reads the kept count after its host synchronisation and runs the evaluation and the sum again with a store that
    // holds it: memory is committed for the streams that need it, up to STORE_MAX_SLOTS = 4 GiB; ctx->store_hint_slots).
    const long long STORE_DEFAULT_SLOTS = ctx->dbg.store_default_slots > 0 ? ctx->dbg.store_default_slots : 16384;   // (knob: test hook)
    const long long STORE_BUDGET_SLOTS = std::min(STORE_MAX_SLOTS, std::max(STORE_DEFAULT_SLOTS, ctx->store_hint_slots));
    cp.no_prune = no_prune != 0;
    SumPlan &sp = cp.sp;
    sp.mode = (flags & RM_FLAG_DENSE_SUM) ? 1 : ((flags & RM_FLAG_SPARSE_SUM) || no_prune) ? 2 : 0;
    sp.auto_dense_ok = sl.S <= 2 ? 1 : 0;
    sp.npairs_mine = (unsigned)npairs_mine;
    long long cap = sp.mode == 2 ? npairs_mine : std::min(npairs_mine, STORE_BUDGET_SLOTS);
    if (flags & RM_FLAG_TINY_STORE) cap = std::min(cap, (long long)8);   // test hook: nearly every selection overflows
    if (ctx->dbg.store_slots > 0) cap = std::min(npairs_mine, ctx->dbg.store_slots);
    if (sp.mode == 1) cap = 0;
    sp.cap_slots = (unsigned)cap;
    cp.store = nullptr;
    // skip 3 / 4 (locate()'s default): the kept pairs are evaluated where they are summed, tile by tile, and nothing is stored
    // (rm_tile_eval.h); the flags that name a sum kernel of the store-based path keep that path (tests compare the two bit for bit)
    cp.fused = tile_eval_ok(g) && !(flags & (RM_FLAG_DENSE_SUM | RM_FLAG_SPARSE_SUM | RM_FLAG_TINY_STORE)) && ctx->dbg.store_slots <= 0 &&
               ctx->dbg.collapse_fused > 0;
    if (cp.fused) { cap = 0; sp.cap_slots = 0; sp.mode = 0; }
    if (cap > 0) RM_TRY(ws(ctx, "value_store", (size_t)cap * CT_H * CT_W, &cp.store));
    ctx->dbg_pairs = npairs; ctx->dbg_cap = cap; ctx->dbg_mine = npairs_mine; ctx->dbg_mode = sp.mode; ctx->dbg_auto_dense = sp.auto_dense_ok;
    RM_TRY(ws(ctx, "sel_cnt", (size_t)ntiles, &cp.sel_cnt));
    RM_TRY(ws(ctx, "heavy_tiles", (size_t)ntiles, &cp.heavy));
    if (!sl.bounds_ready) {
        // per-frame separable form, in bands of tile rows whose row-extrema table fits 64 KB of LDS; the per-pair kernel
        // remains for geometries where even one tile row does not fit
        const size_t row_bytes = 2 * sizeof(double) * (size_t)g.tiles_x;
        int band = g.tiles_y;
        auto tbl_rows_of = [&](int b) {   // most level-S rows any band of b tile rows touches (exact: the device's own footprint rule)
            int most = 0;
            for (int ty0 = 0; ty0 < g.tiles_y; ty0 += b) {
                const int ty1 = std::min(ty0 + b, g.tiles_y) - 1;
                most = std::max(most, tile_region(g, ty1 * g.tiles_x, g.S).y1 - tile_region(g, ty0 * g.tiles_x, g.S).y0 + 1);
            }
            return most;
        };
        size_t tbl_max = 64 * 1024;
        // wide levels (>= 256 columns, at most 64 tile columns): a wave per FB_TR tile rows, streaming (k_frame_bounds_rows)
        const int wS_b = g.w[g.S];
        const size_t rowbufs = 4 * sizeof(double) * (size_t)fb_row_pitch(wS_b);
        // (small images -- 720p: 45 tile rows x 65 frames -- have too few waves of 8 tile rows to fill the chip: the table form stays,
        //  measured 31.7 us against 45 us with 2 tile rows per wave)
        const bool by_rows = (ctx->dbg.bounds_scalar == 2 || (!ctx->dbg.bounds_scalar && (long long)Th * ((g.tiles_y + 7) / 8) >= 2048)) && wS_b >= 256 &&
                             wS_b <= 64 * FB_MAXNL && g.tiles_x <= 64 && ntiles < (1 << 24);
        if (ctx->dbg.bounds_table_bytes > 0) tbl_max = (size_t)ctx->dbg.bounds_table_bytes;   // test hook: force small bands
        while (band > 1 && (size_t)tbl_rows_of(band) * row_bytes > tbl_max) band = (band + 1) / 2;
        // ... and enough workgroups to fill the chip: one workgroup per frame leaves half of it idle at T = 128
        while (band > 4 && (long long)Th * ((g.tiles_y + band - 1) / band) < 1024) band = (band + 1) / 2;
        const int tbl_rows = tbl_rows_of(band);
        const size_t tbl = (size_t)tbl_rows * row_bytes;
        if (by_rows) {
            hipLaunchKernelGGL(k_frame_bounds_rows<8>, dim3(Th, (unsigned)((g.tiles_y + 31) / 32)), dim3(256), rowbufs, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, cp.sel_cnt);
        } else if (tbl <= std::max(tbl_max, (size_t)64 * 1024) && ntiles < (1 << 24)) {
            const unsigned nbands = (unsigned)((g.tiles_y + band - 1) / band);
            hipLaunchKernelGGL(k_frame_bounds, dim3(Th, nbands), dim3(256), tbl, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, band, tbl_rows, cp.sel_cnt);
        } else {
            hipLaunchKernelGGL(k_tile_bounds, dim3((npairs + 255) / 256), dim3(256), 0, s, sl.cS, g, Th, ntiles, cp.lo, cp.hi, st, cp.sel_cnt);
        }
        LAUNCH_CHECK();
    }
    const int prune_ok = (!no_prune && thr >= 0.0 && thr <= 1.0) ? 1 : 0;
    hipLaunchKernelGGL(k_select_pairs, dim3((ntiles + SEL_TILES - 1) / SEL_TILES, (Th + SEL_PH * SEL_U - 1) / (SEL_PH * SEL_U)), dim3(256), 0, s,
                       cp.lo, cp.hi, ntiles, Th, T, t0, t1, st, cp.list_a, cp.list_b, cp.slot_of, prune_ok ? 0 : 1, thr, cp.sel_cnt, cp.heavy);
    LAUNCH_CHECK();
    if (cp.fused) {
        // exact extrema from the C pairs: one wave per pair, a grid that covers the few pairs of a pruned selection at once and loops
        // over an exhaustive one
        const unsigned cgrid = (unsigned)std::min<long long>(npairs, 8192);
        ctx->dbg_fused = 1;
#define RM_EVAL_C(SS)                                                                                            \
        do {                                                                                                     \
            using FootC = TileFoot<SS, false>;                                                                   \
            hipLaunchKernelGGL((k_eval_c<SS>), dim3(cgrid), dim3(64), sizeof(double) * FootC::TOTAL, s, sl.cS, g, ntiles, cp.list_a, st); \
        } while (0)
        switch (sl.S) { case 1: RM_EVAL_C(1); break; case 2: RM_EVAL_C(2); break; case 3: RM_EVAL_C(3); break; default: RM_EVAL_C(4); break; }
#undef RM_EVAL_C
        LAUNCH_CHECK();
        cp.valid = true;
        return RM_OK;
    }
    ctx->dbg_fused = 0;
    cp.shmem = sizeof(double) * (size_t)g.lds_total;
    RM_TRY(launch_eval_pairs(ctx, cp, s));
    cp.valid = true;
    return RM_OK;
}

// heat_sum[H*W] = sum over t in [t0, t1) of (raw >= top ? min : raw), with min/max as they stand in the state
// avg_T > 0: the sum covers the whole buffer, write heat = sum / avg_T and leave the heatmap's min / max in the state
// host_rescue: the caller synchronises the stream soon and looks at the slot's h_unserved (rm_locate): a dense kernel that could only
// be chosen because the value store overflowed is then not enqueued here
static int collapse_sum(rm_ctx *ctx, const CollapsePlan &cp, double thr, double *heat_sum, hipStream_t s, int avg_T = 0, bool host_rescue = false)
{
    CollapseState *st = ctx->d_state;
    const size_t npix = (size_t)cp.H * cp.W;
    if (cp.S == 0) {
        hipLaunchKernelGGL(k_finish_minmax, dim3(1), dim3(NSTRIPE), 0, s, st, thr);
        LAUNCH_CHECK();
        double *sum = heat_sum;
        if (avg_T > 0) RM_TRY(ws(ctx, "heat_sum", npix, &sum));
        hipLaunchKernelGGL(k_masked_sum_plain, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, cp.cS, cp.T, npix, st, sum);
        LAUNCH_CHECK();
        if (avg_T > 0) {
            hipLaunchKernelGGL(k_heat_avg_minmax, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, sum, npix, avg_T, heat_sum, st);
            LAUNCH_CHECK();
        }
        return RM_OK;
    }
    int *tile_nkept = nullptr;
    RM_TRY(ws(ctx, "tile_nkept", (size_t)cp.ntiles, &tile_nkept));
    int *unserved_dev = nullptr;
    auto launch_tile_sum = [&](int only_if_dense) -> int {
        // one workgroup of TS_NW waves per CU (the exchange takes most of a CU's LDS): the heavy tiles' items first, the workgroups
        // left without one fill the constant tiles
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#else
        cus = 6;   // (host emulation: few, looping workgroups compute the same thing)
#endif
        const int nworkers = std::max(1, std::min(2 * cp.ntiles, cus));
#define RM_TILE_SUM(SS)                                                                                                                  \
        do {                                                                                                                             \
            constexpr int exd = tile_sum_exchange_doubles<SS, false>();                                                                  \
            const size_t shb = sizeof(double) * (size_t)exd + sizeof(int) * (size_t)cp.T;                                                \
            if (shb > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_tile_sum<SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb)); \
            hipLaunchKernelGGL((k_tile_sum<SS>), dim3(nworkers), dim3(64 * TS_NW), shb, s, cp.cS, cp.g, cp.t0, cp.t1, cp.T, cp.ntiles, cp.slot_of, st, thr,  \
                               heat_sum, avg_T, tile_nkept, cp.sel_cnt, cp.heavy, nworkers, ctx->dbg.tile_sum_half, cp.sp, only_if_dense);          \
        } while (0)
        switch (cp.S) { case 1: RM_TILE_SUM(1); break; case 2: RM_TILE_SUM(2); break; case 3: RM_TILE_SUM(3); break; default: RM_TILE_SUM(4); break; }
#undef RM_TILE_SUM
        LAUNCH_CHECK();
        return RM_OK;
    };
    auto launch_dense_t = [&](int only_if_dense) -> int {
        // one wave per tile, frame after frame (rm_tile_eval.h k_dense_sum_t)
#define RM_DENSE_T(SS)                                                                                                                   \
        do {                                                                                                                             \
            using FootD = TileFoot<SS, false>;                                                                                           \
            hipLaunchKernelGGL((k_dense_sum_t<SS>), dim3(dense_tile_grid(cp.ntiles)), dim3(64), sizeof(double) * (FootD::TOTAL + DST_MAXW), s, cp.cS, cp.g, cp.t0, \
                               cp.t1, cp.T, cp.ntiles, cp.slot_of, st, thr, heat_sum, avg_T, tile_nkept, cp.sp, only_if_dense, unserved_dev); \
        } while (0)
        switch (cp.S) { case 1: RM_DENSE_T(1); break; case 2: RM_DENSE_T(2); break; case 3: RM_DENSE_T(3); break; default: RM_DENSE_T(4); break; }
#undef RM_DENSE_T
        LAUNCH_CHECK();
        return RM_OK;
    };
    if (cp.fused) {
        if (ctx->dbg.dense_tiles) RM_TRY(launch_dense_t(0)); else
        RM_TRY(launch_tile_sum(0));
        ctx->nkept_H = cp.H; ctx->nkept_W = cp.W;
        return RM_OK;
    }
    const SumPlan &sp = cp.sp;
    // Both sum kernels are enqueued and the one whose turn it is not returns at once (sum_is_dense, decided from this call's own
    // selection); a launch that can never be chosen is left out: the sparse one when the dense kernel is forced, the dense one when
    // the store has a slot for every pair and the automatic rule cannot pick it.
    const bool may_sparse = sp.mode != 1;
    const bool auto_dense = sp.mode == 0 && sp.auto_dense_ok;
    const bool overflow_only = !auto_dense && sp.mode != 1 && sp.cap_slots < sp.npairs_mine;
    // (a context whose last selection was dense enqueues the stand-in behind the sparse kernel instead of waiting for the host to
    //  find the store overflowed: ctx->dense_hint, set and cleared by rm_locate)
    const bool may_dense = sp.mode == 1 || auto_dense || (overflow_only && (!host_rescue || ctx->dense_hint));
    if (overflow_only && host_rescue) {
        RoiSlot &rs = ctx->slots[ctx->cur_slot];
        if (!rs.h_unserved) HIP_TRY(hipHostMalloc((void **)&rs.h_unserved, sizeof(int), hipHostMallocDefault));
        *rs.h_unserved = 0;
        HIP_TRY(hipHostGetDevicePointer((void **)&unserved_dev, rs.h_unserved, 0));
    }
    if (may_sparse) {
        // worker items for the tiles with kept pairs (MS_Q each); the workgroups left without an item fill the other tiles
#ifdef RM_HIPEMU
        const int nworkers = std::min(cp.ntiles * MS_Q, 24);    // (host emulation: fewer, looping workgroups compute the same thing)
#else
        const int nworkers = std::min(cp.ntiles * MS_Q, MS_B > 16 ? 512 : 768);   // 2-3 workgroups per CU (registers): one resident round
#endif
        if (cp.t0 == 0 && cp.t1 == cp.T && avg_T == cp.T && ctx->dbg.sum_rows) {
            // the whole buffer: one wave per (heavy tile, row), the kept unique frames' values staged by LDS-DMA (rm_tile_eval.h)
            const size_t shr = sizeof(double) * MSR_CHUNK * 64 + 2 * sizeof(int) * (size_t)sym_frames(cp.T);
#ifdef RM_HIPEMU
            const int nw3 = std::min(cp.ntiles * CT_H, 40);
#else
            // one resident round: what the LDS footprint lets a CU hold (a queued wave starts its chain of round trips late)
            int cus3 = 256;
            HIP_TRY(hipDeviceGetAttribute(&cus3, hipDeviceAttributeMultiprocessorCount, ctx->device));
            const int per_cu3 = (int)std::max<size_t>(1, std::min<size_t>(32, ((size_t)160 * 1024) / (shr + 512)));
            const int nw3 = std::min(cp.ntiles * CT_H, per_cu3 * cus3);
#endif
            hipLaunchKernelGGL(k_masked_sum_rows, dim3(nw3), dim3(64), shr, s, cp.T, cp.ntiles, cp.W, cp.H, cp.slot_of, cp.store, st, thr, heat_sum,
                               tile_nkept, cp.sel_cnt, cp.heavy, nw3, sp, unserved_dev);
        } else if (cp.t0 == 0 && cp.t1 == cp.T && avg_T == cp.T && ctx->dbg.sum_sym) {
            // the whole buffer: every unique frame loaded once and added on the way up and on the way down (rm_tile_eval.h)
            const int nw2 = std::min(nworkers, 512);   // 220 VGPRs: two workgroups per CU stay resident
            hipLaunchKernelGGL(k_masked_sum_sym, dim3(nw2), dim3(64 * MS_RQ), 2 * sizeof(int) * (size_t)sym_frames(cp.T), s, cp.T, cp.ntiles, cp.W, cp.H,
                               cp.slot_of, cp.store, st, thr, heat_sum, tile_nkept, cp.sel_cnt, cp.heavy, nw2, sp, unserved_dev);
        } else {
            hipLaunchKernelGGL(k_masked_sum_tiles, dim3(nworkers), dim3(64 * MS_RQ), 2 * sizeof(int) * (size_t)cp.T, s, cp.t0, cp.t1, cp.T, cp.ntiles,
                               cp.W, cp.H, cp.slot_of, cp.store, st, thr, heat_sum, avg_T, tile_nkept, cp.sel_cnt, cp.heavy, nworkers, sp, unserved_dev);
        }
        LAUNCH_CHECK();
    }
    // skip <= 2 on large frames (four waves' worth of tiles per SIMD): the TileEval kernel of the deeper chains is the faster one-wave-per-
    // tile form there too (4K x 512 skip 2: 2.26 -> 2.18 ms); smaller frames keep the several-waves-per-tile forms below
    const bool t_low = ctx->dbg.dense_t_low >= 0 ? ctx->dbg.dense_t_low != 0 : (cp.S == 2 && cp.ntiles >= 4096 && tile_eval_ok(cp.g));
    if (may_dense && cp.S <= 2 && ctx->dbg.dense_wave && !t_low && !ctx->dbg.dense_rows && !ctx->dbg.dense_general && dense_wave_ok(cp.g)) {
        // one wave per 64 x 16 tile, no barriers (rm_dense_sum.h k_dense_sum_w)
        const ChainGeom &g = cp.g;
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        // fewer than two waves per SIMD: two frames per trip, interleaved (the lone wave's dependency chain is what takes the time)
        int fr = cp.ntiles < 8 * cus ? 2 : 1;
        if (ctx->dbg.dense_frames == 1 || ctx->dbg.dense_frames == 2) fr = ctx->dbg.dense_frames;
        // fewer tiles than SIMDs: NW waves per tile, each evaluating every NW-th frame (k_dense_sum_wf)
        int split = cp.ntiles < 4 * cus ? 4 : 1;
        if (ctx->dbg.dense_split == 1 || ctx->dbg.dense_split == 2 || ctx->dbg.dense_split == 4) split = ctx->dbg.dense_split;
        if (split > 1) {
#define RM_DENSE_WF(SS, NN)                                                                                                               \
            do {                                                                                                                          \
                const size_t shf = sizeof(double) * (size_t)NN * 16 * 64;                                                                 \
                hipLaunchKernelGGL((k_dense_sum_wf<SS, NN>), dim3(dense_tile_grid(cp.ntiles)), dim3(64 * NN), shf, s, cp.cS, g, cp.t0, cp.t1, cp.T, st, thr,  \
                                   heat_sum, avg_T, tile_nkept, sp);                                                                      \
            } while (0)
            if (cp.S == 2) { if (split == 4) RM_DENSE_WF(2, 4); else RM_DENSE_WF(2, 2); }
            else { if (split == 4) RM_DENSE_WF(1, 4); else RM_DENSE_WF(1, 2); }
#undef RM_DENSE_WF
            LAUNCH_CHECK();
        } else {
#define RM_DENSE_W(SS, FF)                                                                                                                \
        hipLaunchKernelGGL((k_dense_sum_w<SS, FF>), dim3(dense_tile_grid(cp.ntiles)), dim3(64), sizeof(double) * DenseW<SS>::TOTAL * FF, s, cp.cS, g, cp.t0, \
                           cp.t1, cp.T, st, thr, heat_sum, avg_T, tile_nkept, sp)
        if (cp.S == 2) { if (fr == 2) RM_DENSE_W(2, 2); else RM_DENSE_W(2, 1); }
        else { if (fr == 2) RM_DENSE_W(1, 2); else RM_DENSE_W(1, 1); }
#undef RM_DENSE_W
        LAUNCH_CHECK();
        }
    } else if (may_dense && (cp.S >= 3 || t_low) && tile_eval_ok(cp.g) && !ctx->dbg.dense_rows && !ctx->dbg.dense_general) {
        // deeper chains: every kept pair evaluated where it is summed, tile by tile (rm_tile_eval.h k_tile_sum); it looks at the
        // selection itself when the sparse kernel was enqueued in front of it
        if (ctx->dbg.dense_tiles) RM_TRY(launch_dense_t(sp.mode == 1 ? 0 : 1));
        else RM_TRY(launch_tile_sum(sp.mode == 1 ? 0 : 1));
    } else if (may_dense) {
        // super-tiles of 64 x 64 pixels (four waves, 16 rows each) when that still gives every CU two workgroups, 64 x 32 (two
        // waves) next; with fewer tiles than that, one 64 x 16 tile per workgroup and four rows per wave: the per-frame latency counts
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        const ChainGeom &g = cp.g;
        int rows = 64;
        while (rows > 16 && (long long)g.tiles_x * ((cp.H + rows - 1) / rows) < 2ll * cus) rows >>= 1;
        { const int v = ctx->dbg.dense_rows; if (v == 16 || v == 32 || v == 64) rows = v; }   // test hook
        DenseGeom dg;
        dg.rows = rows; dg.nsx = g.tiles_x; dg.nsy = (cp.H + rows - 1) / rows;
        const int S = cp.S;
        auto lvl = [&](int k) { return (chain_extent(rows, k) + 1) * (chain_extent(CT_W, k) + 1); };
        auto scratch = [&](int k) { return (chain_extent(rows, k) + 1) * (chain_extent(CT_W, k - 1) + 1); };
        for (int k = 0; k < MAX_CHAIN; ++k) { dg.lds_off[k] = 0; dg.lds_hb[k] = 0; }
        int off = lvl(1);                                   // [ level 1 ][ level 2 ][ B ], as make_geom lays out k_eval_pairs
        if (S >= 2) { dg.lds_off[2] = off; off += lvl(2); }
        const int B = off;
        int small = 0, hb_small = 0;
        for (int k = 3; k <= S; ++k) { dg.lds_off[k] = B + small; small += lvl(k); hb_small = std::max(hb_small, scratch(k)); }
        for (int k = 3; k <= S; ++k) dg.lds_hb[k] = B + small;
        if (S >= 2) dg.lds_hb[2] = B;
        dg.lds_total = B + (S >= 2 ? std::max(scratch(2), S >= 3 ? small + hb_small : 0) : 0);
        const size_t sh = sizeof(double) * (size_t)dg.lds_total;
        const unsigned grid = (unsigned)(dg.nsx * dg.nsy);
#define RM_DENSE_LAUNCH(KERNEL, NW, RPW)                                                                                                 \
        do {                                                                                                                             \
            if (sh > 64 * 1024)                                                                                                          \
                HIP_TRY(hipFuncSetAttribute((const void *)KERNEL<NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));       \
            hipLaunchKernelGGL((KERNEL<NW, RPW>), dim3(grid), dim3(64 * NW), sh, s, cp.cS, g, dg, cp.t0, cp.t1, cp.T, st, thr, heat_sum, \
                               avg_T, tile_nkept, sp);                                                                                   \
        } while (0)
        if (S <= 2 && !ctx->dbg.dense_general) {   // table-driven form (knob: test hook for the general kernel)
            if (rows == 64) RM_DENSE_LAUNCH(k_dense_sum_s2, 4, 16); else if (rows == 32) RM_DENSE_LAUNCH(k_dense_sum_s2, 2, 16); else RM_DENSE_LAUNCH(k_dense_sum_s2, 4, 4);
        } else {
            if (rows == 64) RM_DENSE_LAUNCH(k_dense_sum, 4, 16); else if (rows == 32) RM_DENSE_LAUNCH(k_dense_sum, 2, 16); else RM_DENSE_LAUNCH(k_dense_sum, 4, 4);
        }
#undef RM_DENSE_LAUNCH
        LAUNCH_CHECK();
    }
    ctx->nkept_H = cp.H; ctx->nkept_W = cp.W;   // the constant tiles of this heatmap (or partial heat sum of a frame shard) are known
    return RM_OK;
}

static int calibrate_impl(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin,
                          double fmax, double amp, int levels, int skip, double thr, unsigned flags, double *heat,
                          double *minmax_host, void *stream, CollapsePlan *plan_out)
{
    if (!ctx || !frames || !heat || T < 1 || H < 1 || W < 1 || levels < 1 || skip < 0 || !(fps > 0) || !valid_dtype(dtype))
        return fail(RM_E_BADARG, "rm_calibrate: bad argument");
    if (T > MAX_T) return fail(RM_E_UNSUPPORTED, "rm_calibrate: T=%d > %d", T, MAX_T);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
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
    if (!ctx || !frames || !lap_local || Tl < 1 || H < 1 || W < 1 || levels < 1 || skip < 1 || !valid_dtype(dtype))
        return fail(RM_E_BADARG, "rm_shard_pyramid: bad argument (frame-sharded calibration needs skip_levels_at_top >= 1)");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
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
    hipLaunchKernelGGL(k_export_minmax, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state, negmin_max_dev);
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
    if (cp.S < 0) { HIP_TRY(hipMemsetAsync(heat_sum, 0, sizeof(double) * (size_t)cp.H * cp.W, s)); return RM_OK; }
    PhaseTimer pt(ctx, 2, s);
    hipLaunchKernelGGL(k_import_minmax, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state, negmin_max_dev);
    LAUNCH_CHECK();
    return collapse_sum(ctx, cp, thr, heat_sum, s);
}

extern "C" int rm_shard_finish(rm_ctx *ctx, const double *heat_sum, int T, int H, int W, int threshold, double *heatmap,
                               int32_t *xywh, void *stream)
{
    if (!ctx || !heat_sum || !heatmap || T < 1 || H < 1 || W < 1) return fail(RM_E_BADARG, "rm_shard_finish: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    hipLaunchKernelGGL(k_heat_state_init, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
    ctx->state_fresh = false;
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_heat_avg_minmax, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, heat_sum, npix, T, heatmap, ctx->d_state);
    LAUNCH_CHECK();
    if (!xywh) return RM_OK;
    return heatmap_to_roi_impl(ctx, heatmap, H, W, threshold, xywh, nullptr, nullptr, stream, true);
}

extern "C" int rm_eulerian_magnification_bandpass(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps,
                                                  double fmin, double fmax, double amp, int levels, int skip, double thr,
                                                  double *masked, double *raw, double *minmax_host, void *stream)
{
    if (!ctx || !frames || T < 1 || H < 1 || W < 1 || levels < 1 || skip < 0 || !(fps > 0) || !valid_dtype(dtype))
        return fail(RM_E_BADARG, "rm_eulerian_magnification_bandpass: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
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
        hipLaunchKernelGGL(k_mirror_frames, dim3(nblk((size_t)H * W, 256, 1024), (unsigned)(T - Th)), dim3(256), 0, s, raw_buf, T, (size_t)H * W);
        LAUNCH_CHECK();
    }
    CollapseState *st = ctx->d_state;
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(NSTRIPE), 0, s, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax_plain, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, raw_buf, n, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_minmax, dim3(1), dim3(NSTRIPE), 0, s, st, thr);
    LAUNCH_CHECK();
    if (masked) {
        hipLaunchKernelGGL(k_mask_plain, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, raw_buf, n, st, masked);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(ctx->h_state, st, sizeof(CollapseState), hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    if (minmax_host) { minmax_host[0] = ctx->h_state->min_val; minmax_host[1] = ctx->h_state->max_val; }
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// heatmap -> ROI  (base.py:563-575)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NSTRIPE) void k_heat_state_init(CollapseState *st)
{
    st->heat_min_keys[threadIdx.x] = ~0ull; st->heat_max_keys[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) { st->heat_min_key = ~0ull; st->heat_max_key = 0ull; }
}

constexpr int CCL_TABLE_MAX_COMPONENTS = 2048;   // more components than this last time: k_ccl_bbox without its LDS table
constexpr int LABEL_REPROBE = 64;         // labelled stages in a row before the host-only stage is timed again
constexpr int LABEL_MIN_CONTOURS = 512;   // ~0.13 us per followed border on the host against ~40 us of labelling kernels
static_assert(sizeof(CclComp) == sizeof(LabelComp), "record layout shared by rm_ccl.h and rm_contour.h");

// The ROI stage in two halves: roi_launch enqueues the device work (threshold -> packed image in the pinned memory of slot
// ctx->cur_slot, component labelling when the rule asks for it), roi_finish -- once the stream (or the event recorded behind the
// launches) has been waited for -- runs the host contour stage on that slot.  heatmap_to_roi_impl is the two with stream_wait
// between them; rm_locate_submit / rm_locate_result put the next call's frame-buffer kernel there instead.
static int roi_launch(rm_ctx *ctx, const double *heat, int H, int W, int threshold, uint8_t *avg_u8, uint8_t *binary, void *stream,
                      bool have_minmax, RoiPending &pd, bool xywh_given)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_heatmap_to_roi: bad argument");
    // the one-call clip request of rm_locate (RM_FLAG_CONTOUR_CLIP_FRAME) is consumed here, whatever happens below
    const bool clip_once = ctx->clip_frame_once;
    ctx->clip_frame_once = false;
    if (!heat || !xywh_given || H < 1 || W < 1) return fail(RM_E_BADARG, "rm_heatmap_to_roi: bad argument");
    RoiSlot &rs = ctx->slots[ctx->cur_slot];
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    CollapseState *st = ctx->d_state;
    ctx->state_fresh = false;   // the heatmap extrema are (or have been) reduced into the state
    // the thresholded image goes to the host bit-packed (npix / 8 bytes) plus one "row holds foreground" byte per row: the
    // kernel stores both straight into pinned, device-mapped host memory (no copy-engine hop)
    const size_t nwords = (npix + 63) / 64;
    const size_t need = nwords * 8 + (size_t)H;
    if (rs.h_bin_cap < need) {
        if (rs.h_bin) { HIP_TRY(stream_wait(s)); HIP_TRY(hipHostFree(rs.h_bin)); }
        rs.h_bin = nullptr; rs.h_bin_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&rs.h_bin, need, hipHostMallocDefault));
        rs.h_bin_cap = need;
        rs.h_rows_dirty = nullptr;
    }
    uint8_t *h_rows = rs.h_bin + nwords * 8;
    // invariant between calls: image words and row flags are all zero (the rows a call read are zeroed again below)
    if (rs.h_rows_dirty != h_rows) { std::memset(rs.h_bin, 0, need); rs.h_rows_dirty = h_rows; }
    uint8_t *dev_bin = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dev_bin, rs.h_bin, 0));
    PhaseTimer *pt_roi = new PhaseTimer(ctx, 3, s);
    struct Guard { PhaseTimer *&p; ~Guard() { delete p; p = nullptr; } } guard{pt_roi};
    if (!have_minmax) {  // rm_calibrate has just left the heatmap's min / max in the state
        hipLaunchKernelGGL(k_heat_state_init, dim3(1), dim3(NSTRIPE), 0, s, st);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_heat_minmax, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, heat, npix, st);
        LAUNCH_CHECK();
    }
    // noisy images: label the components on the device so that the host follows only borders that can win (rm_ccl.h)
    const bool clip = ctx->clip_frame || clip_once;
    const bool same_geom = ctx->label_H == H && ctx->label_W == W;
    if (!same_geom) { ctx->label_unl_us = -1.0; ctx->label_lab_host_us = 0.0; ctx->label_streak = 0; }
    const bool many = ctx->label_last_n > LABEL_MIN_CONTOURS;
    bool slow_host = ctx->label_unl_us >= 0.0 && ctx->label_unl_us > (double)ctx->dbg.label_host_us + ctx->label_lab_host_us;
    if (slow_host && !many && ctx->label_mode < 0 && ctx->label_streak >= LABEL_REPROBE) { slow_host = false; ctx->label_streak = 0; }
    const bool label = !clip && npix < (size_t)0x7fffffff &&
                       (ctx->label_mode == 1 || (ctx->label_mode < 0 && same_geom && (many || slow_host)));
    // rm_locate's own heatmap: the sum kernel that wrote it knows which 64 x 16 tiles are one constant (tile_nkept == 0)
    const int *tile_const = nullptr;
    if (ctx->tiles_const_once && ctx->nkept_H == H && ctx->nkept_W == W && W % CT_W == 0 && ctx->dbg.heat_const_tiles) {
        int *tk = nullptr;
        RM_TRY(ws(ctx, "tile_nkept", (size_t)((W + CT_W - 1) / CT_W) * ((H + CT_H - 1) / CT_H), &tk));
        tile_const = tk;
    }
    ctx->tiles_const_once = false;
    unsigned long long *d_bits = nullptr;
    const size_t comps_cap = std::min<size_t>(npix / 4 + 2, (size_t)1 << 18);
    if (label) {
        int *d_label = nullptr; CclBox *d_box = nullptr; unsigned int *d_cnt = nullptr; int *d_list = nullptr;
        RM_TRY(ws(ctx, "ccl_list", comps_cap, &d_list));
        RM_TRY(ws(ctx, "ccl_bits", nwords, &d_bits));
        RM_TRY(ws(ctx, "ccl_label", npix, &d_label));
        RM_TRY(ws(ctx, "ccl_box", npix, &d_box));
        RM_TRY(ws(ctx, "ccl_counters", (size_t)2, &d_cnt));
        if (rs.h_comps_cap < comps_cap + 1) {
            if (rs.h_comps) { HIP_TRY(stream_wait(s)); HIP_TRY(hipHostFree(rs.h_comps)); }
            rs.h_comps = nullptr; rs.h_comps_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&rs.h_comps, (comps_cap + 1) * sizeof(CclComp), hipHostMallocDefault));
            rs.h_comps_cap = comps_cap + 1;
        }
        CclComp *dev_comps = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dev_comps, rs.h_comps, 0));
        hipLaunchKernelGGL(k_heat_to_u8, dim3(nblk(npix, 256, 2048)), dim3(256), 0, s, heat, npix, W, st, threshold, avg_u8, binary,
                           (unsigned long long *)dev_bin, dev_bin + nwords * 8, d_bits, d_label, d_box, d_cnt, tile_const);
        LAUNCH_CHECK();
        // (a thread per 64-bit word walking its set bits through the same rule was measured: 97 / 95 us instead of 21 / 19 at 720p --
        //  ten dependent find / atomic round trips per thread cost more than launching 84 % idle threads)
        const dim3 grid((unsigned)((npix + 255) / 256));
        hipLaunchKernelGGL(k_ccl_union, grid, dim3(256), 0, s, d_bits, npix, H, W, d_label);
        LAUNCH_CHECK();
        // per-tile LDS table of boxes unless the last extraction of this geometry met thousands of components (specks: see k_ccl_bbox)
        const bool table = ctx->dbg.ccl_table >= 0 ? ctx->dbg.ccl_table != 0 : !(same_geom && ctx->label_last_n > CCL_TABLE_MAX_COMPONENTS);
        const dim3 bgrid((unsigned)((W + 63) / 64), (unsigned)((H + CCL_BOX_ROWS - 1) / CCL_BOX_ROWS));
        if (table) hipLaunchKernelGGL(k_ccl_bbox<true>, bgrid, dim3(64 * CCL_BOX_ROWS), 0, s, d_bits, npix, H, W, d_label, d_box, d_cnt, d_list, (unsigned int)comps_cap);
        else hipLaunchKernelGGL(k_ccl_bbox<false>, bgrid, dim3(64 * CCL_BOX_ROWS), 0, s, d_bits, npix, H, W, d_label, d_box, d_cnt, d_list, (unsigned int)comps_cap);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_ccl_publish, dim3(64), dim3(256), 0, s, d_list, d_box, W, d_cnt, (unsigned int)comps_cap, dev_comps);
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_heat_to_u8, dim3(nblk(npix, 256, 2048)), dim3(256), 0, s, heat, npix, W, st, threshold, avg_u8, binary,
                           (unsigned long long *)dev_bin, dev_bin + nwords * 8, (unsigned long long *)nullptr, (int *)nullptr,
                           (CclBox *)nullptr, (unsigned int *)nullptr, tile_const);
        LAUNCH_CHECK();
    }
    delete pt_roi; pt_roi = nullptr;
    pd.H = H; pd.W = W; pd.slot = ctx->cur_slot; pd.nwords = nwords; pd.comps_cap = comps_cap; pd.label = label; pd.clip = clip;
    return RM_OK;
}

static int roi_finish(rm_ctx *ctx, const RoiPending &pd, int32_t *xywh)
{
    RoiSlot &rs = ctx->slots[pd.slot];
    const int H = pd.H, W = pd.W;
    const size_t nwords = pd.nwords, comps_cap = pd.comps_cap;
    const bool label = pd.label, clip = pd.clip;
    uint8_t *h_rows = rs.h_bin + nwords * 8;
    RoiResult r;
    {
        const auto t0 = std::chrono::steady_clock::now();
        int y0 = H, y1 = -1;   // rows that hold foreground
        for (int y = 0; y < H; ++y)
            if (h_rows[y]) { if (y < y0) y0 = y; y1 = y; h_rows[y] = 0; }
        if (clip && y1 >= y0) {
            // OpenCV <= 3.1: the 1-pixel image frame is zeroed before tracing (the host copy is ours to change)
            uint64_t *hb = (uint64_t *)rs.h_bin;
            auto clear_bit = [&](size_t p) { hb[p >> 6] &= ~(1ull << (p & 63)); };
            for (int y = y0; y <= y1; ++y) {
                const size_t r0 = (size_t)y * W;
                if (y == 0 || y == H - 1) { for (int x = 0; x < W; ++x) clear_bit(r0 + x); }
                else { clear_bit(r0); clear_bit(r0 + W - 1); }
            }
        }
        const size_t ncomp = label ? (size_t)(unsigned int)rs.h_comps[0].root : 0;
        ctx->label_used = label && ncomp <= comps_cap;
        if (ctx->label_used)   // (an overflowing record list falls through to the full scan: the image is here either way)
            largest_external_contour_labelled((const uint64_t *)rs.h_bin, H, W, (const LabelComp *)(rs.h_comps + 1), ncomp, &r);
        else if (!(ctx->dbg.host_simple_shape && y1 >= y0 && simple_shape_bits_rows((const uint64_t *)rs.h_bin, H, W, y0, y1, &r)))
            largest_external_contour_bits_rows((const uint64_t *)rs.h_bin, H, W, y0, y1, &r);
        ctx->label_H = H; ctx->label_W = W; ctx->label_last_n = r.n_contours;
        if (y1 >= y0) {   // restore the all-zero image: the words that cover rows y0 .. y1
            const size_t w0 = ((size_t)y0 * W) >> 6, w1 = (((size_t)(y1 + 1) * W) - 1) >> 6;
            std::memset(rs.h_bin + w0 * 8, 0, (w1 - w0 + 1) * 8);
        }
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->label_used) {
            ctx->label_lab_host_us = host_us;
            ++ctx->label_streak;
            if ((long long)ncomp * 2 < (long long)ctx->label_unl_n) ctx->label_unl_us = -1.0;   // a different kind of image: time the host stage afresh
        } else if (!clip) {
            ctx->label_unl_us = host_us; ctx->label_unl_n = r.n_contours; ctx->label_streak = 0;
        }
        if (ctx->prof_on) ctx->prof_host_ms[3] += host_us * 1e-3;
    }
    if (!r.found) { xywh[0] = xywh[1] = xywh[2] = xywh[3] = 0; return RM_NO_CONTOUR; }
    xywh[0] = r.x; xywh[1] = r.y; xywh[2] = r.w; xywh[3] = r.h;
    return RM_OK;
}

static int heatmap_to_roi_impl(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                               uint8_t *binary, void *stream, bool have_minmax)
{
    RoiPending pd;
    RM_TRY(roi_launch(ctx, heat, H, W, threshold, avg_u8, binary, stream, have_minmax, pd, xywh != nullptr));
    HIP_TRY(stream_wait((hipStream_t)stream));
    return roi_finish(ctx, pd, xywh);
}

// ------------------------------------------------------------------------------------------
// sparse heatmap exchange (kernels: k_sparse_*)
// ------------------------------------------------------------------------------------------
extern "C" size_t rm_heat_sparse_packet_doubles(int cap_tiles)
{
    return cap_tiles < 1 ? 0 : (size_t)SP_HDR + (size_t)cap_tiles + (size_t)cap_tiles * CT_H * CT_W;
}

extern "C" int rm_heat_sparse_pack(rm_ctx *ctx, const double *heat, int H, int W, int cap_tiles, double *packet, void *stream)
{
    if (!ctx || !heat || !packet || H < 1 || W < 1 || cap_tiles < 1) return fail(RM_E_BADARG, "rm_heat_sparse_pack: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, ntiles = tiles_x * tiles_y;
    if (ctx->nkept_H != H || ctx->nkept_W != W) {
        // no pruning bookkeeping for this heatmap (skip 0, zero result, foreign heatmap): report overflow -> dense exchange
        HIP_TRY(hipMemsetAsync(packet, 0, sizeof(double) * SP_HDR, s));
        HIP_TRY(hipMemsetAsync(packet, 0xff, sizeof(unsigned int), s));   // SP_DENSE_ONLY
        return RM_OK;
    }
    int *tile_nkept = nullptr;
    RM_TRY(ws(ctx, "tile_nkept", (size_t)ntiles, &tile_nkept));
    hipLaunchKernelGGL(k_sparse_background, dim3(1), dim3(64), 0, s, heat, W, tiles_x, ntiles, tile_nkept, cap_tiles, packet);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sparse_pack, dim3(ntiles), dim3(256), 0, s, heat, H, W, tiles_x, tile_nkept, cap_tiles, packet);
    LAUNCH_CHECK();
    return RM_OK;
}

static int heatmap_to_roi_impl(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                               uint8_t *binary, void *stream, bool have_minmax);

extern "C" int rm_heat_sparse_merge_roi(rm_ctx *ctx, const double *packets, int world, int H, int W, int cap_tiles, int threshold,
                                        int avg_T, double *fused, int32_t *xywh, void *stream)
{
    if (!ctx || !packets || !fused || !xywh || world < 1 || H < 1 || W < 1 || cap_tiles < 1)
        return fail(RM_E_BADARG, "rm_heat_sparse_merge_roi: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, ntiles = tiles_x * tiles_y;
    const size_t pd = rm_heat_sparse_packet_doubles(cap_tiles);
    int *map = nullptr, *any = nullptr, *flag_dev = nullptr;
    RM_TRY(ws(ctx, "sparse_map", (size_t)world * ntiles, &map));
    RM_TRY(ws(ctx, "sparse_any", (size_t)ntiles, &any));
    if (!ctx->h_flag) {
        HIP_TRY(hipHostMalloc((void **)&ctx->h_flag, 2 * sizeof(int), hipHostMallocDefault));
        ctx->h_flag[0] = ctx->h_flag[1] = 0;
    }
    HIP_TRY(hipHostGetDevicePointer((void **)&flag_dev, ctx->h_flag, 0));
    hipLaunchKernelGGL(k_sparse_index, dim3(1), dim3(256), 0, s, packets, pd, world, cap_tiles, ntiles, map, any, flag_dev,
                       ctx->d_state, avg_T);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sparse_merge, dim3(ntiles), dim3(256), 0, s, packets, pd, world, cap_tiles, H, W, tiles_x, ntiles, map, any,
                       fused, ctx->d_state, avg_T);
    LAUNCH_CHECK();
    // the ROI stage synchronises the stream; the overflow flag is in pinned memory by then
    const int rc = heatmap_to_roi_impl(ctx, fused, H, W, threshold, xywh, nullptr, nullptr, stream, true);
    if (rc < 0) return rc;
    if (ctx->h_flag[0]) return RM_SPARSE_FALLBACK;
    return rc;
}

extern "C" int rm_heat_sparse_tiles_needed(rm_ctx *ctx, int *tiles)
{
    if (!ctx || !tiles) return fail(RM_E_BADARG, "rm_heat_sparse_tiles_needed: bad argument");
    *tiles = ctx->h_flag ? ctx->h_flag[1] : 0;
    return RM_OK;
}

extern "C" int rm_set_contour_labelling(rm_ctx *ctx, int mode)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_set_contour_labelling: ctx is NULL");
    ctx->label_mode = mode < 0 ? -1 : mode > 0 ? 1 : 0;
    return RM_OK;
}

extern "C" int rm_get_contour_labelling(rm_ctx *ctx, int *mode)
{
    if (!ctx || !mode) return fail(RM_E_BADARG, "rm_get_contour_labelling: bad argument");
    *mode = ctx->label_mode;
    return RM_OK;
}

extern "C" int rm_get_contour_clip_frame(rm_ctx *ctx, int *on)
{
    if (!ctx || !on) return fail(RM_E_BADARG, "rm_get_contour_clip_frame: bad argument");
    *on = ctx->clip_frame ? 1 : 0;
    return RM_OK;
}

extern "C" int rm_contour_stats(rm_ctx *ctx, int *n_components, int *labelled)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_contour_stats: ctx is NULL");
    if (n_components) *n_components = ctx->label_last_n;
    if (labelled) *labelled = ctx->label_used;
    return RM_OK;
}

extern "C" int rm_set_contour_clip_frame(rm_ctx *ctx, int on)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_set_contour_clip_frame: ctx is NULL");
    ctx->clip_frame = on != 0;
    return RM_OK;
}

extern "C" int rm_heatmap_to_roi(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                                 uint8_t *binary, void *stream)
{
    return heatmap_to_roi_impl(ctx, heat, H, W, threshold, xywh, avg_u8, binary, stream, false);
}

extern "C" int rm_locate(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax,
                         double amp, int levels, int skip, double temporal_thr, int threshold, unsigned flags, int32_t *xywh,
                         void *stream)
{
    if (!ctx || !xywh) return fail(RM_E_BADARG, "rm_locate: bad argument");
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
    if (rs.h_unserved) *rs.h_unserved = 0;
    if (ctx->dense_hint && unserved_word != 2) ctx->dense_hint = 0;   // (the stand-in enqueued on the hint was not needed: back to the plain path)
    if (rc >= 0 && cp.valid && unserved) {
        // the selection kept more pairs than the value store holds and the sparse sum kernel stood down (the ROI stage above ran on
        // a heatmap nobody wrote -- the price of not putting a host synchronisation in front of the ROI stage of EVERY call, which
        // is what looking at the flag first would take): take the sum with the dense kernel, now that the stream is idle, and
        // extract the ROI again
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(k_heat_state_init, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
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
    t.active = false;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(event_wait(t.done));
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

// ------------------------------------------------------------------------------------------
// multi-GPU steps behind the C-ABI: RCCL on the caller's stream (the call site this replaces is base.py:444 run once per
// GPU).  librccl is opened at run time (dlopen: a process that already holds RCCL -- PyTorch-ROCm's copy carries the same SONAME --
// shares it), so a single-GPU user of the library never needs it.
//   Mode B (rm_locate_streams): calibrate -> sparse packet -> ncclAllGather -> merge + ROI   (dense: ncclAllReduce(sum) of the heatmap)
//   Mode A (rm_locate_sharded): pyramid of the local frames -> ncclAllGather -> collapse -> ncclAllReduce(max) of {-min, max} ->
//                               masked sum of the local frames -> sparse packets / dense all-reduce -> ROI
// one host synchronisation per step on the common (sparse) path.
// ------------------------------------------------------------------------------------------
#ifndef RM_HIPEMU
#include <dlfcn.h>
#include <rccl/rccl.h>
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static RcclApi &rccl_api()
{
    static RcclApi a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
    if (!a.lib) { a.err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return a; }
    auto sym = [&](const char *s) { void *p = dlsym(a.lib, s); if (!p && a.err.empty()) a.err = std::string("librccl has no symbol ") + s; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    return a;
}
#define RCCL_TRY(expr)                                                                                                          \
    do {                                                                                                                        \
        ncclResult_t r_ = (expr);                                                                                               \
        if (r_ != ncclSuccess) return fail(RM_E_COMM, "%s: %s", #expr, rccl_api().GetErrorString ? rccl_api().GetErrorString(r_) : "RCCL error"); \
    } while (0)
#endif

extern "C" int rm_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(RM_E_BADARG, "rm_comm_unique_id: id_out is NULL");
#ifndef RM_HIPEMU
    RcclApi &a = rccl_api();
    if (!a.err.empty()) return fail(RM_E_COMM, "rm_comm_unique_id: %s", a.err.c_str());
    static_assert(sizeof(ncclUniqueId) == RM_COMM_ID_BYTES, "RM_COMM_ID_BYTES is the size of ncclUniqueId");
    ncclUniqueId id;
    RCCL_TRY(a.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return RM_OK;
#else
    std::memset(id_out, 0, RM_COMM_ID_BYTES);
    return RM_OK;
#endif
}

extern "C" int rm_comm_destroy(rm_ctx *ctx)
{
    if (!ctx) return RM_OK;
#ifndef RM_HIPEMU
    if (ctx->comm) { (void)hipSetDevice(ctx->device); (void)rccl_api().CommDestroy((ncclComm_t)ctx->comm); }
#endif
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 1;
    return RM_OK;
}

extern "C" int rm_comm_init(rm_ctx *ctx, int rank, int world, const void *unique_id)
{
    if (!ctx || world < 1 || rank < 0 || rank >= world) return fail(RM_E_BADARG, "rm_comm_init: bad argument");
    RM_TRY(rm_comm_destroy(ctx));
    ctx->xp_streams = ExchangeState(); ctx->xp_sharded = ExchangeState();   // every rank starts a communicator with the same policy state
    if (!unique_id) {
        if (world != 1) return fail(RM_E_BADARG, "rm_comm_init: %d ranks need the unique id rank 0 made (rm_comm_unique_id)", world);
        return RM_OK;   // one rank, no library: the collectives are the identity
    }
#ifndef RM_HIPEMU
    RcclApi &a = rccl_api();
    if (!a.err.empty()) return fail(RM_E_COMM, "rm_comm_init: %s", a.err.c_str());
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    RCCL_TRY(a.CommInitRank(&c, world, id, rank));
    ctx->comm = c; ctx->comm_rank = rank; ctx->comm_world = world;
    return RM_OK;
#else
    return fail(RM_E_UNSUPPORTED, "rm_comm_init: the host emulation has no RCCL");
#endif
}

extern "C" int rm_comm_info(rm_ctx *ctx, int *rank, int *world, int *rccl_ranks)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_comm_info: ctx is NULL");
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    if (rccl_ranks) {
        *rccl_ranks = 0;   // 0: no RCCL communicator behind this context
#ifndef RM_HIPEMU
        if (ctx->comm) { int n = 0; RCCL_TRY(rccl_api().CommCount((ncclComm_t)ctx->comm, &n)); *rccl_ranks = n; }
#endif
    }
    return RM_OK;
}

extern "C" int rm_shard_frames(int T, int rank, int world, int *t0, int *t1)
{
    if (T < 0 || world < 1 || rank < 0 || rank >= world || !t0 || !t1) return fail(RM_E_BADARG, "rm_shard_frames: bad argument");
    const int base = T / world, rem = T % world;
    *t0 = rank * base + std::min(rank, rem);
    *t1 = *t0 + base + (rank < rem ? 1 : 0);
    return RM_OK;
}

// collectives on float64 device buffers; a context without a communicator is one rank (copy / nothing)
static int comm_all_gather(rm_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t s)
{
#ifndef RM_HIPEMU
    if (ctx->comm) { RCCL_TRY(rccl_api().AllGather(send, recv, count, ncclFloat64, (ncclComm_t)ctx->comm, s)); return RM_OK; }
#endif
    if (recv != send) HIP_TRY(hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    return RM_OK;
}
static int comm_all_reduce(rm_ctx *ctx, double *buf, size_t count, bool max_not_sum, hipStream_t s)
{
#ifndef RM_HIPEMU
    if (ctx->comm) { RCCL_TRY(rccl_api().AllReduce(buf, buf, count, ncclFloat64, max_not_sum ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, s)); return RM_OK; }
#endif
    (void)buf; (void)count; (void)max_not_sum; (void)s;
    return RM_OK;
}

// what a heatmap exchange remembers from one step to the next (respmon_amd/dist.py ExchangePolicy, now per communicator): a packet
// that overflowed tells every rank how many tiles the fullest rank needed -- all of them read all the headers, so all switch together
static bool exchange_use_sparse(ExchangeState &x)
{
    if (x.dense_left > 0) { --x.dense_left; return false; }
    return true;
}
static void exchange_overflowed(ExchangeState &x, int needed)
{
    const int want = ((int)(needed * 1.25) + 63) / 64 * 64;
    if (needed > 0 && want <= RM_SPARSE_MAX_TILES) x.cap = std::max(x.cap, want);
    else x.dense_left = RM_DENSE_HOLD;
}

// heat (this rank's heatmap, or its partial heat SUM when avg_T > 0) -> the ranks' sum (/ avg_T) -> ROI; fused_out (nullable): the sum
static int exchange_heat_roi(rm_ctx *ctx, ExchangeState &xp, double *heat, int H, int W, int threshold, int avg_T, double *fused_out, int32_t *xywh,
                             int *exchange_out, hipStream_t s)
{
    const size_t npix = (size_t)H * W;
    const int world = ctx->comm_world;
    const bool clip_once = ctx->clip_frame_once;   // (consumed by the first ROI stage: a second one after an overflow asks again)
    if (!(ctx->dbg.exchange_dense) && exchange_use_sparse(xp)) {
        const int cap = xp.cap;
        const size_t pd = rm_heat_sparse_packet_doubles(cap);
        double *packet = nullptr, *all = nullptr, *fused = fused_out;
        RM_TRY(ws(ctx, "xp_packet", pd, &packet));
        RM_TRY(ws(ctx, "xp_packets", pd * (size_t)world, &all));
        if (!fused) RM_TRY(ws(ctx, "xp_fused", npix, &fused));
        RM_TRY(rm_heat_sparse_pack(ctx, heat, H, W, cap, packet, (void *)s));
        RM_TRY(comm_all_gather(ctx, packet, all, pd, s));
        const int rc = rm_heat_sparse_merge_roi(ctx, all, world, H, W, cap, threshold, avg_T, fused, xywh, (void *)s);
        if (rc < 0) return rc;
        if (rc != RM_SPARSE_FALLBACK) { if (exchange_out) *exchange_out = RM_EXCHANGE_SPARSE; return rc; }
        exchange_overflowed(xp, ctx->h_flag ? ctx->h_flag[1] : 0);
    }
    if (exchange_out) *exchange_out = RM_EXCHANGE_DENSE;
    ctx->clip_frame_once = clip_once;
    RM_TRY(comm_all_reduce(ctx, heat, npix, false, s));
    if (avg_T > 0) {
        double *fused = fused_out;
        if (!fused) RM_TRY(ws(ctx, "xp_fused", npix, &fused));
        return rm_shard_finish(ctx, heat, avg_T, H, W, threshold, fused, xywh, (void *)s);
    }
    if (fused_out && fused_out != heat) HIP_TRY(hipMemcpyAsync(fused_out, heat, sizeof(double) * npix, hipMemcpyDeviceToDevice, s));
    return heatmap_to_roi_impl(ctx, heat, H, W, threshold, xywh, nullptr, nullptr, (void *)s, false);
}

extern "C" int rm_locate_streams(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp,
                                 int levels, int skip, double temporal_thr, int threshold, unsigned flags, double *fused_heat, int32_t *xywh,
                                 int *exchange_out, void *stream)
{
    if (!ctx || !xywh) return fail(RM_E_BADARG, "rm_locate_streams: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *heat = nullptr;
    RM_TRY(ws(ctx, "heat", (size_t)H * W, &heat));
    RM_TRY(calibrate_impl(ctx, frames, dtype, T, H, W, fps, fmin, fmax, amp, levels, skip, temporal_thr, flags, heat, nullptr, stream, nullptr));
    ctx->clip_frame_once = (flags & RM_FLAG_CONTOUR_CLIP_FRAME) != 0;
    return exchange_heat_roi(ctx, ctx->xp_streams, heat, H, W, threshold, 0, fused_heat, xywh, exchange_out, s);
}

extern "C" int rm_locate_sharded(rm_ctx *ctx, const void *frames_local, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp,
                                 int levels, int skip, double temporal_thr, int threshold, unsigned flags, double *heatmap, int32_t *xywh,
                                 int *exchange_out, void *stream)
{
    if (!ctx || !xywh || !frames_local || T < 1 || H < 1 || W < 1) return fail(RM_E_BADARG, "rm_locate_sharded: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rank = ctx->comm_rank, world = ctx->comm_world;
    int t0 = 0, t1 = 0;
    RM_TRY(rm_shard_frames(T, rank, world, &t0, &t1));
    if (t1 - t0 < 1) return fail(RM_E_BADARG, "rm_locate_sharded: every rank needs at least one frame (T=%d, world=%d)", T, world);
    size_t NP = 0;
    RM_TRY(rm_shard_layout_flags(H, W, levels, skip, flags, &NP));
    const int cmax = (T + world - 1) / world;   // frames of the longest shard: what every rank sends (shorter shards are padded)
    double *lap_pad = nullptr, *lap_gath = nullptr, *lap_all = nullptr, *mm = nullptr, *heat_sum = nullptr;
    if (NP) {
        RM_TRY(ws(ctx, "sh_lap_pad", (size_t)cmax * NP, &lap_pad));
        RM_TRY(rm_shard_pyramid(ctx, frames_local, dtype, t1 - t0, H, W, levels, skip, flags, lap_pad, stream));
        const bool even = T % world == 0;
        RM_TRY(ws(ctx, "sh_lap_all", (size_t)T * NP, &lap_all));
        if (even) {
            RM_TRY(comm_all_gather(ctx, lap_pad, lap_all, (size_t)cmax * NP, s));
        } else {
            RM_TRY(ws(ctx, "sh_lap_gath", (size_t)world * cmax * NP, &lap_gath));
            if (t1 - t0 < cmax) HIP_TRY(hipMemsetAsync(lap_pad + (size_t)(t1 - t0) * NP, 0, sizeof(double) * (size_t)(cmax - (t1 - t0)) * NP, s));
            RM_TRY(comm_all_gather(ctx, lap_pad, lap_gath, (size_t)cmax * NP, s));
            for (int r = 0; r < world; ++r) {   // compact: rank r's frames in frame order
                int a = 0, b = 0;
                RM_TRY(rm_shard_frames(T, r, world, &a, &b));
                HIP_TRY(hipMemcpyAsync(lap_all + (size_t)a * NP, lap_gath + (size_t)r * cmax * NP, sizeof(double) * (size_t)(b - a) * NP, hipMemcpyDeviceToDevice, s));
            }
        }
    }
    RM_TRY(ws(ctx, "sh_minmax", (size_t)2, &mm));
    RM_TRY(rm_shard_collapse(ctx, lap_all, T, t0, t1, H, W, fps, fmin, fmax, amp, levels, skip, temporal_thr, flags, mm, stream));
    RM_TRY(comm_all_reduce(ctx, mm, 2, true, s));
    RM_TRY(ws(ctx, "sh_heat_sum", (size_t)H * W, &heat_sum));
    RM_TRY(rm_shard_heat(ctx, mm, temporal_thr, heat_sum, stream));
    ctx->clip_frame_once = (flags & RM_FLAG_CONTOUR_CLIP_FRAME) != 0;
    return exchange_heat_roi(ctx, ctx->xp_sharded, heat_sum, H, W, threshold, T, heatmap, xywh, exchange_out, s);
}

// ------------------------------------------------------------------------------------------
// ROI reductions (base.py:355-358, 364)
// ------------------------------------------------------------------------------------------
static bool roi_ok(int H, int W, int x, int y, int w, int h) { return x >= 0 && y >= 0 && w >= 1 && h >= 1 && x + w <= W && y + h <= H; }

extern "C" int rm_roi_mean(rm_ctx *ctx, const void *frame, int dtype, int H, int W, int x, int y, int w, int h, double *out,
                           void *stream)
{
    if (!ctx || !frame || !out || !valid_dtype(dtype) || !roi_ok(H, W, x, y, w, h)) return fail(RM_E_BADARG, "rm_roi_mean: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *d = nullptr;
    RM_TRY(ws(ctx, "roi_mean", 1, &d));
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_roi_mean<uint8_t>), dim3(1), dim3(256), 0, s, (const uint8_t *)frame, W, x, y, w, h, d); break;
    case RM_F16: hipLaunchKernelGGL((k_roi_mean<__half>), dim3(1), dim3(256), 0, s, (const __half *)frame, W, x, y, w, h, d); break;
    case RM_F32: hipLaunchKernelGGL((k_roi_mean<float>), dim3(1), dim3(256), 0, s, (const float *)frame, W, x, y, w, h, d); break;
    default: hipLaunchKernelGGL((k_roi_mean<double>), dim3(1), dim3(256), 0, s, (const double *)frame, W, x, y, w, h, d); break;
    }
    LAUNCH_CHECK();
    HIP_TRY(hipMemcpyAsync(out, d, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    return RM_OK;
}

extern "C" int rm_roi_to_uint8(rm_ctx *ctx, const void *frame, int dtype, int H, int W, int x, int y, int w, int h, uint8_t *dst,
                               void *stream)
{
    if (!ctx || !frame || !dst || !valid_dtype(dtype) || !roi_ok(H, W, x, y, w, h)) return fail(RM_E_BADARG, "rm_roi_to_uint8: bad argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk((size_t)w * h, 256, 1024)), block(256);
    switch (dtype) {
    case RM_U8: hipLaunchKernelGGL((k_roi_to_u8<uint8_t>), grid, block, 0, s, (const uint8_t *)frame, W, x, y, w, h, dst); break;
    case RM_F16: hipLaunchKernelGGL((k_roi_to_u8<__half>), grid, block, 0, s, (const __half *)frame, W, x, y, w, h, dst); break;
    case RM_F32: hipLaunchKernelGGL((k_roi_to_u8<float>), grid, block, 0, s, (const float *)frame, W, x, y, w, h, dst); break;
    default: hipLaunchKernelGGL((k_roi_to_u8<double>), grid, block, 0, s, (const double *)frame, W, x, y, w, h, dst); break;
    }
    LAUNCH_CHECK();
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// optical-flow path (rm_flow.h)
// ------------------------------------------------------------------------------------------
extern "C" int rm_good_features_to_track(rm_ctx *ctx, const uint8_t *img, int h, int w, int max_corners, double quality,
                                         double min_distance, int block_size, float *pts, int *n, void *stream)
{
    if (!ctx || !img || !pts || !n || h < 3 || w < 3 || block_size < 1 || (block_size & 1) == 0)
        return fail(RM_E_BADARG, "rm_good_features_to_track: bad argument");
    std::string err;
    int rc = flow_good_features(ctx->flow, img, h, w, max_corners, quality, min_distance, block_size, pts, n, (hipStream_t)stream, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return rc;
}

extern "C" int rm_calc_optical_flow_pyr_lk(rm_ctx *ctx, const uint8_t *prev, const uint8_t *next, int h, int w, const float *pts_in,
                                           int npts, int win_w, int win_h, int max_level, int max_count, double epsilon,
                                           float *pts_out, uint8_t *status, void *stream)
{
    if (!ctx || !prev || !next || !pts_in || !pts_out || !status || h < 1 || w < 1 || npts < 0 || win_w < 3 || win_h < 3 || max_level < 0)
        return fail(RM_E_BADARG, "rm_calc_optical_flow_pyr_lk: bad argument");
    std::string err;
    int rc = flow_pyr_lk(ctx->flow, prev, next, h, w, pts_in, npts, win_w, win_h, max_level, max_count, epsilon, pts_out, status,
                         (hipStream_t)stream, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return rc;
}

extern "C" int rm_mean_flow(rm_ctx *ctx, const float *old_pts, const float *new_pts, const uint8_t *status, int npts, float *mean_xy,
                            int *n_good, void *stream)
{
    if (!ctx || !old_pts || !new_pts || !status || !mean_xy || !n_good || npts < 0) return fail(RM_E_BADARG, "rm_mean_flow: bad argument");
    std::string err;
    int rc = flow_mean(ctx->flow, old_pts, new_pts, status, npts, mean_xy, n_good, (hipStream_t)stream, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return rc;
}

extern "C" int rm_pca_reduce(rm_ctx *ctx, const float *motion, int n, double *out, void *stream)
{
    if (!ctx || !motion || !out || n < 0) return fail(RM_E_BADARG, "rm_pca_reduce: bad argument");
    std::string err;
    int rc = flow_pca(ctx->flow, motion, n, out, (hipStream_t)stream, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return rc;
}

// ---- one C-ABI call per frame of extract_motion('flow') (base.py:363-388); crops and points stay on the device ----------
static int flow_crop(rm_ctx *ctx, const void *frame, int dtype, int H, int W, int x, int y, int w, int h, uint8_t *dst, hipStream_t s)
{
    return rm_roi_to_uint8(ctx, frame, dtype, H, W, x, y, w, h, dst, (void *)s);
}

struct rm_flow_state {
    int device = 0;
    FlowState fs;
};

extern "C" int rm_flow_state_create(rm_ctx *ctx, rm_flow_state **out)
{
    if (!ctx || !out) return fail(RM_E_BADARG, "rm_flow_state_create: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    rm_flow_state *st = new rm_flow_state();
    st->device = ctx->device;
    *out = st;
    return RM_OK;
}

extern "C" int rm_flow_state_destroy(rm_flow_state *st)
{
    if (!st) return RM_OK;
    (void)hipSetDevice(st->device);
    delete st;   // (FlowState / FlowWorkspace release their device and pinned memory)
    return RM_OK;
}

static int flow_state_pts(FlowState &fs, float **pts_a, float **pts_b)
{
    std::string err;
    int rc;
    if ((rc = fs.ws.get("pts_a", sizeof(float) * 2 * (size_t)fs.cap, (void **)pts_a, err)) < 0) return fail(rc, "%s", err.c_str());
    if ((rc = fs.ws.get("pts_b", sizeof(float) * 2 * (size_t)fs.cap, (void **)pts_b, err)) < 0) return fail(rc, "%s", err.c_str());
    return RM_OK;
}

static int flow_state_crop(FlowState &fs, int side, uint8_t **crop)
{
    std::string err;
    const int rc = flow_side_buf(fs, side, "pyr", 0, (size_t)fs.w * fs.h, (void **)crop, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return RM_OK;
}

extern "C" int rm_flow_begin(rm_ctx *ctx, rm_flow_state *state, const void *frame, int dtype, int H, int W, int x, int y, int w, int h, int max_corners,
                             double quality, double min_distance, int block_size, float *pts_host, int *n_host, void *stream)
{
    if (!ctx || !state || !frame || !pts_host || !n_host || !valid_dtype(dtype) || !roi_ok(H, W, x, y, w, h) || h < 3 || w < 3 || block_size < 1 ||
        (block_size & 1) == 0)
        return fail(RM_E_BADARG, "rm_flow_begin: bad argument");
    if (state->device != ctx->device) return fail(RM_E_BADARG, "rm_flow_begin: the flow state belongs to device %d, the context to %d", state->device, ctx->device);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    FlowState &fs = state->fs;
    fs.w = w; fs.h = h; fs.cap = std::max(max_corners, 1); fs.flip = 0; fs.npts = 0; fs.begun = false;
    fs.pyr_levels[0] = fs.pyr_levels[1] = -1; fs.deriv_levels[0] = fs.deriv_levels[1] = -1;
    if (!fs.res) HIP_TRY(hipHostMalloc((void **)&fs.res, 4 * sizeof(float), hipHostMallocDefault));
    uint8_t *crop = nullptr; float *pa = nullptr, *pb = nullptr;
    RM_TRY(flow_state_crop(fs, 0, &crop));
    RM_TRY(flow_state_pts(fs, &pa, &pb));
    RM_TRY(flow_crop(ctx, frame, dtype, H, W, x, y, w, h, crop, s));
    fs.pyr_levels[0] = 0;
    std::string err;
    int rc = flow_good_features(fs.ws, crop, h, w, max_corners, quality, min_distance, block_size, pts_host, n_host, s, err);
    if (rc < 0) return fail(rc, "%s", err.c_str());
    fs.npts = *n_host;
    if (*n_host > 0) {
        HIP_TRY(hipMemcpyAsync(pa, pts_host, sizeof(float) * 2 * (size_t)*n_host, hipMemcpyHostToDevice, s));
        HIP_TRY(stream_wait(s));   // pts_host is the caller's again
    }
    fs.begun = true;
    return RM_OK;
}

extern "C" int rm_flow_step(rm_ctx *ctx, rm_flow_state *state, const void *frame, int dtype, int H, int W, int x, int y, int w, int h, int win_w,
                            int win_h, int max_level, int max_count, double epsilon, float *mean_xy_host, int *n_good_host, void *stream)
{
    if (!ctx || !state || !frame || !mean_xy_host || !n_good_host || !valid_dtype(dtype) || !roi_ok(H, W, x, y, w, h) || win_w < 3 || win_h < 3 ||
        max_level < 0)
        return fail(RM_E_BADARG, "rm_flow_step: bad argument");
    FlowState &fs = state->fs;
    if (!fs.begun || w != fs.w || h != fs.h) return fail(RM_E_BADARG, "rm_flow_step: rm_flow_begin has not been called on this state for this ROI size");
    if (state->device != ctx->device) return fail(RM_E_BADARG, "rm_flow_step: the flow state belongs to another device");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    const int prev_side = fs.flip, cur_side = fs.flip ^ 1;
    uint8_t *cur = nullptr; float *pa = nullptr, *pb = nullptr;
    RM_TRY(flow_state_crop(fs, cur_side, &cur));
    RM_TRY(flow_state_pts(fs, &pa, &pb));
    float *pts = fs.flip ? pb : pa, *pts_next = fs.flip ? pa : pb;
    RM_TRY(flow_crop(ctx, frame, dtype, H, W, x, y, w, h, cur, s));
    fs.pyr_levels[cur_side] = 0; fs.deriv_levels[cur_side] = -1;   // a new image on this side: its old pyramid and derivatives are void
    const int npts = fs.npts;
    mean_xy_host[0] = mean_xy_host[1] = 0.f; *n_good_host = 0;
    if (npts > 0) {
        std::string err;
        float *d_out = nullptr; uint8_t *d_st = nullptr; float *dev_res = nullptr;
        int rc;
        if ((rc = fs.ws.get("lk_pts_out", sizeof(float) * 2 * (size_t)npts, (void **)&d_out, err)) < 0) return fail(rc, "%s", err.c_str());
        if ((rc = fs.ws.get("lk_status", (size_t)npts, (void **)&d_st, err)) < 0) return fail(rc, "%s", err.c_str());
        rc = flow_track_resident(fs, prev_side, cur_side, pts, npts, win_w, win_h, max_level, max_count, epsilon, d_out, d_st, s, err);
        if (rc < 0) return fail(rc, "%s", err.c_str());
        HIP_TRY(hipHostGetDevicePointer((void **)&dev_res, fs.res, 0));
        if (npts <= FLOW_FINISH_MAX) hipLaunchKernelGGL(k_flow_finish, dim3(1), dim3(64), 2 * sizeof(float) * (size_t)flow_finish_pitch(npts), s, pts, d_out, d_st, npts, dev_res, pts_next);
        else hipLaunchKernelGGL(k_flow_finish_seq, dim3(1), dim3(1), 0, s, pts, d_out, d_st, npts, dev_res, pts_next);
        LAUNCH_CHECK();
        HIP_TRY(stream_wait(s));
        mean_xy_host[0] = fs.res[0]; mean_xy_host[1] = fs.res[1]; *n_good_host = (int)fs.res[2];
        fs.npts = *n_good_host;
    }
    fs.flip ^= 1;   // the crop just made is the next call's previous image, the packed points its input (base.py:381-382)
    return RM_OK;
}

extern "C" int rm_flow_points(rm_ctx *ctx, rm_flow_state *state, float *pts_host, int cap, int *n_host, void *stream)
{
    if (!ctx || !state || !n_host || cap < 0 || (cap > 0 && !pts_host)) return fail(RM_E_BADARG, "rm_flow_points: bad argument");
    FlowState &fs = state->fs;
    *n_host = fs.npts;
    if (fs.npts == 0 || cap == 0) return RM_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    float *pa = nullptr, *pb = nullptr;
    RM_TRY(flow_state_pts(fs, &pa, &pb));
    const int n = std::min(cap, fs.npts);
    HIP_TRY(hipMemcpyAsync(pts_host, fs.flip ? pb : pa, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(stream_wait(s));
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// developer build only (-DRM_TRACE, librespmon_hip_trace.so; tools/trace_tail.py): workgroup timelines
// ------------------------------------------------------------------------------------------
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
