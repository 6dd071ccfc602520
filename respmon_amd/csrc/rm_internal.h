// respmon_amd/csrc/rm_internal.h -- what the translation units of librespmon_hip.so share on the HOST side: the context, the
// error convention, the workspace, and the stage functions one unit calls in another.  Nothing here is exported.
//
//   rm_ctx.hip            contexts, developer switches, profiling hooks, dtype helpers
//   rm_pyramid.hip        pyramid building blocks + materialising pyramid API          (pyramid.py:9-69)
//   rm_temporal.hip       temporal filters, materialised min/max mask                   (transforms.py:38-102, 184-192)
//   rm_down.hip           launchers of the frame-buffer kernels                          (rm_down_chain.h, rm_down_chain_u8.h)
//   rm_front.hip          frames -> collapsed band-passed level C_S                      (transforms.py:144-182)
//   rm_collapse_eval.hip  tile bounds, pruning, exact extrema                            (transforms.py:184-186)
//   rm_collapse_sum.hip   masked time sum -> heatmap                                     (transforms.py:187-192, base.py:562)
//   rm_calibrate.hip      rm_calibrate, frame-sharded stages, materialising eulerian_magnification_bandpass
//   rm_roi.hip            heatmap -> ROI, sparse heatmap packets                         (base.py:563-575)
//   rm_locate.hip         rm_locate, rm_locate_submit / rm_locate_result                 (base.py:547-601)
//   rm_comm.hip           RCCL behind the C-ABI                                          
//   rm_motion.hip         ROI mean / crop, corners, LK, PCA                              (base.py:354-407)
//   rm_unity.hip          all of the above as ONE unit: the tracing build and the host emulation of the tests
// Every kernel header is included by every unit; non-template kernels are `static`, so a unit generates code only for the kernels
// it launches.
#pragma once
#include "../../include/respmon_hip.h"
#include "../../include/respmon_hip_debug.h"

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
#include <type_traits>
#include <vector>

#include "rm_contour.h"
#include "rm_kernels.h"
#include "rm_down_chain.h"
#include "rm_down_chain_u8.h"
#include "rm_dense_sum.h"
#include "rm_tile_eval.h"
#include "rm_bounds_l1.h"
#include "rm_xstore.h"
#include "rm_ccl.h"
#include "rm_flow.h"

// sets the thread's error string (rm_last_error_string) and returns `code`
int fail(int code, const char *fmt, ...);

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

using namespace rm;

struct DevBuf { void *p = nullptr; size_t cap = 0; };
constexpr long long STORE_MAX_SLOTS = 524288;   // value store: at most 4 GiB (8 KB per kept (tile, frame) pair)
struct ExchangeState { int cap = RM_SPARSE_CAP_TILES; int dense_left = 0; };   // heatmap exchange policy of a communicator (rm_locate_streams / _sharded)

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
    bool l1_bounds = false;                // lo / hi are the extrema of the LEVEL-1 footprints (rm_bounds_l1.h): a pair they keep cannot stop at level 1
    unsigned long long *xs_tab = nullptr;  // exception store table [tile][unique frame] (rm_xstore.h XsEntry), written XS_NONE by k_select_pairs
};


// developer / test switches (rm_debug_set): they select between implementations that produce identical results, or shrink a
// tuning constant so that a test reaches a rare code path.  The library never reads the process environment.
struct DebugKnobs {
    int temporal_valu = 0;        // 1: the two-stage VALU temporal kernels instead of k_temporal_sym
    int temporal_wide = -1;       // 0 / 1: k_temporal_sym / k_temporal_sym_px whatever the level size (-1: by size)
    int bgr_unfused = 0;          // 1: RM_BGR8 buffers are converted to gray as a whole before the chain (the path of shapes the fused kernel does not take)
    int dc_lds_front_end = 0;     // 1: narrow frame buffers through the LDS front end of k_down_chain instead of rm_down_chain_u8.h
    int no_fused_bounds = 0;      // 1: k_small_collapse + k_frame_bounds instead of k_small_collapse_bounds
    long long bounds_table_bytes = 0;   // > 0: LDS budget of k_frame_bounds' row-extrema table (forces small bands)
    int bounds_l1 = 2;            // skip 2: tile bounds from the level-1 footprint (rm_bounds_l1.h) -- 2: formed in packed float32 and widened (default), 1: in float64 (the exact extrema); 0: from the level-2 footprint (k_frame_bounds / k_frame_bounds_rows)
    int bounds_l1_rows = 0;       // > 0: tile rows per wave of k_frame_bounds_l1 (default: 16, fewer on small frames)
    int xs = 0;                   // 1: dense selections go through the exception store (rm_xstore.h) instead of the store-less sum kernels (measured slower: DESIGN 7)
    int xs_waves = 0;             // 1 / 2 / 4: waves per tile of k_xs_sum (0: by the number of tiles)
    long long xs_budget_words = 0;   // > 0: capacity of the exception store in 8-byte words (default: the worst case of the geometry, at most 1 GiB)
    int dense_wf_list = 1;        // 0: k_dense_sum_wf evaluates every frame of a tile (no kept list from slot_of / lo)
    int bounds_up1 = -1;          // skip 3 / 4: bounds refined from the level-(S - 1) footprint (k_bounds_up1): 1 always, 0 never, -1 behind a call that kept many pairs (ctx->refine_hint)
    int bounds_scalar = 0;        // 1: k_frame_bounds (a thread per row and tile column) also for wide levels instead of k_frame_bounds_rows
    int dense_rows = 0;           // 16 / 32 / 64: super-tile rows of the dense sum kernel
    int dense_general = 0;        // 1: k_dense_sum instead of the table-driven k_dense_sum_s2 at skip <= 2
    int dense_frames = 0;         // 1 / 2: frames per trip of k_dense_sum_w (0: by the number of tiles)
    int dense_split = 0;          // 1 / 2 / 4: waves per tile (k_dense_sum_wf for 2 and 4; 0: by the number of tiles)
    int dense_wave = 1;           // 0: the workgroup kernels (k_dense_sum_s2 / k_dense_sum) instead of the wave-private k_dense_sum_w at skip <= 2
    int dc_segs = 0, dc_wpg = 0;  // > 0: segments per frame / waves per workgroup of k_down_chain
    int dc_prio_shift = 0;        // 1 .. 12: log2 of the rotation period of dc_prio 2 in input rows (default 4)
    int dc_prio = 2;              // k_down_chain: issue priority of its waves (2 rotating -- the default --, 0 none, 1 younger workgroups higher, 3 older higher)
    int dc_split = 0;             // > 0: share (per mille) of the level-S rows the upper of exactly two segments takes (default 513)
    int collapse_fused = 0;       // 1: collapse passes without a value store wherever TileEval applies (rm_tile_eval.h k_eval_c + k_tile_sum); 0: only as the stand-in for an overflowing store at skip >= 3
    int sum_rows = 0;             // 1: k_masked_sum_rows (one wave per tile row, LDS-DMA staging) instead of k_masked_sum_tiles for whole-buffer sums (measured slower: 35 us against 21)
    int sum_sym = 0;              // 1: k_masked_sum_sym instead of k_masked_sum_tiles for whole-buffer sums (measured slower: 31 us against 21 at 1080p x 256)
    long long label_host_steps = 0;   // > 0: border steps of an unlabelled host stage beyond which the next extraction is labelled on the device (default LABEL_MIN_STEPS)
    int ccl_tiles = 1;            // 0: rows of whole words labelled through global memory as the others are (k_ccl_union / k_ccl_bbox), not tile by tile in LDS
    int ccl_tile_waves = 0;       // waves of a k_ccl_tile workgroup (4 / 8 / 16: eight / four / two rows of the tile per wave; 0: by the number of tiles)
    int ccl_table = -1;           // k_ccl_bbox: 1 with / 0 without the per-tile LDS table of boxes, -1 by the last component count
    int heat_const_tiles = 1;     // 0: k_heat_to_u8 reads every pixel of rm_locate's heatmap (no use of the sum kernel's constant-tile flags)
    int ff_parts = 0;             // > 0: workgroups per frame of k_small_filter_first (default: 2 when one per frame would leave CUs idle)
    int heat_rows = 1;            // 0: k_heat_to_u8 (flat over the pixels, row flags) also where rows are whole words, instead of k_heat_rows_u8 (row records)
    int label_lazy = 1;           // 0: the labelled path always sends the packed image and the full component list to the host
    int host_area_bound = 1;      // 0: the labelled path always follows the top component's border (no shortcut on its area's lower bound, rm_ccl.h)
    int host_simple_shape = 1;    // 0: the host contour stage always follows the borders (no one-blob shortcut on the packed rows)
    int exchange_dense = 0;       // 1: rm_locate_streams / rm_locate_sharded exchange the heatmaps by the dense all-reduce only
    int eval_fast = 1;            // 0: the generic k_eval_pairs instead of k_eval_pairs_fast (rm_tile_eval.h) where the latter applies
    int dense_exact_top = 1;      // 0: the store-less sum kernels evaluate every pair the selection kept (no second look with the exact `top`)
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
    // record path (k_heat_rows_u8): what the last extraction on this slot left non-zero -- image words [dirty_w0, dirty_w1], records
    // [dirty_r0, dirty_r1] --, zeroed by the next roi_launch on the slot; dirty_geom: the buffer layout they belong to
    size_t dirty_w0 = 1, dirty_w1 = 0, dirty_geom = 0; int dirty_r0 = 0, dirty_r1 = -1;
    int *h_unserved = nullptr;      // set by k_masked_sum_tiles when it left the sum to a dense kernel nobody enqueued (rm_locate)
};
constexpr int ROI_SLOTS = 3;
// what the host half of the ROI stage has to know about the launches it finishes
struct RoiPending {
    int H = 0, W = 0, slot = 0; size_t nwords = 0, comps_cap = 0, rec_off = 0; bool label = false, clip = false, rows = false;
    // lazy labelled stage: the packed image and the full component list stayed on the device (d_bits / d_list, buffers of this slot); only
    // the summary records travelled.  roi_finish copies the rest over `stream` when the summaries do not settle the winner.
    bool lazy = false; hipStream_t stream = nullptr; const void *d_bits = nullptr, *d_list = nullptr;
};
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
    bool label_lazy = false;      // the last labelled extraction of this geometry was settled by the summary records alone: the next one keeps image and list on the device
    int roi_path = 0;             // RM_ROI_PATH_* of the last host contour stage (rm_debug_roi_path)
    // ... or when following every border on the host was long last time (few components with long borders: a frame of noise blobs):
    // border steps the last unlabelled stage of this geometry walked (< 0: none) with its contour count, labelled stages in a row
    // (every LABEL_REPROBE-th one is run unlabelled to refresh the first figure).  Counts, not clock readings (round 6): the same
    // stream takes the same path in every run.
    long long label_unl_steps = -1;
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
    int refine_hint = 0;            // the last rm_locate of this context kept many pairs at skip >= 3: the next one refines its bounds one level down (rm_bounds_l1.h k_bounds_up1)
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
    // host timeline of the last rm_locate (rm_debug_host_timeline): microseconds after its entry; [0] = what the caller spent since the previous return
    double host_marks[RM_HOST_MARKS] = {0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point host_enter{}, host_last_return{};
    bool host_have_return = false;
};

static inline void host_mark(rm_ctx *c, int i)
{
    c->host_marks[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c->host_enter).count();
}

// A context with rm_locate_submit tickets in flight takes further work only on the tickets' stream (stream order is what keeps a
// later call's kernels off the workspaces an in-flight submission still reads: heat, state, tile_nkept, ccl_*, value_store);
// RM_E_BUSY otherwise.  Called by every entry point that enqueues on the context's workspaces.
int ctx_stream_ok(rm_ctx *ctx, void *stream, const char *who);

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


int ws_get(rm_ctx *ctx, const std::string &name, size_t bytes, void **out);
template <typename T> static int ws(rm_ctx *ctx, const std::string &name, size_t count, T **out)
{
    void *p = nullptr;
    RM_TRY(ws_get(ctx, name, count * sizeof(T), &p));
    *out = (T *)p;
    return RM_OK;
}

static inline unsigned nblk(size_t n, unsigned per, unsigned cap = 8192)
{
    size_t b = (n + per - 1) / per;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}
static inline bool valid_dtype(int d) { return d == RM_U8 || d == RM_F16 || d == RM_F32 || d == RM_F64; }
// frame BUFFERS of the calibration entry points may also be RM_BGR8 ([T,H,W,3] uint8: base.py:230's cvtColor happens on the device)
static inline bool valid_buffer_dtype(int d) { return valid_dtype(d) || d == RM_BGR8; }
static inline int dtype_vec(int dtype) { return dtype == RM_F64 ? 2 : dtype == RM_F32 ? 4 : dtype == RM_F16 ? 8 : 16; }   // pixels per 16-byte aligned lane-load unit
static inline size_t dtype_size(int dtype) { return dtype == RM_F64 ? 8 : dtype == RM_F32 ? 4 : dtype == RM_F16 ? 2 : dtype == RM_BGR8 ? 3 : 1; }

// ---- rm_ctx.hip
int launch_bgr_to_gray(const uint8_t *bgr, size_t npix, uint8_t *gray, hipStream_t s);
int bgr_buffer_to_gray(rm_ctx *ctx, const void *frames, size_t npix, const void **gray_out, hipStream_t s);

// ---- rm_pyramid.hip
int launch_pyr_down(const void *src, int dtype, int T, int h, int w, double *dst, hipStream_t s);
int launch_pyr_up(const double *src, int T, int sh, int sw, double *dst, int dh, int dw, int mode, const double *other, hipStream_t s,
                  size_t src_fs = 0, size_t dst_fs = 0, size_t other_fs = 0);
int launch_to_f64(const void *src, int dtype, size_t n, double *dst, hipStream_t s);
void level_sizes(int H, int W, int levels, std::vector<int> &h, std::vector<int> &w);

// ---- rm_temporal.hip
struct TemporalOp { const double *R = nullptr, *C = nullptr, *Rf = nullptr, *Cf = nullptr; int nk = 0, tiles = 0; };  // nk: merged rows; Rf / Cf: fragment-major copies for k_temporal_sym<tiles>
int get_operator(rm_ctx *ctx, int T, double fps, double fmin, double fmax, TemporalOp *op, hipStream_t s);
int launch_temporal(rm_ctx *ctx, const double *x, int T, size_t NP, const TemporalOp &op, double amp, double *out, hipStream_t s,
                    rm::CollapseState *st_init = nullptr, bool full = false);

// ---- rm_down.hip: frames[T,H,W] -> G_S[T,h_S,w_S] in one launch
int launch_down_chain(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S,
                      double *out, hipStream_t s, bool tiny);

// ---- rm_front.hip
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

void pyr_geom(int H, int W, int levels, int skip, unsigned flags, PyrGeom &pg);
int front_pyramid(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, const PyrGeom &pg, unsigned flags, double *lap, hipStream_t s);
int front_filter(rm_ctx *ctx, const double *lap, int T, const PyrGeom &pg, double fps, double fmin, double fmax, double amp, SmallLevels &out,
                 hipStream_t s);
int front_half(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp, int levels,
               int skip, unsigned flags, SmallLevels &out, hipStream_t s);
int make_geom(const SmallLevels &sl, rm::ChainGeom &g);

// ---- rm_collapse_eval.hip / rm_collapse_sum.hip
int launch_eval_pairs(rm_ctx *ctx, const CollapsePlan &cp, hipStream_t s);
int collapse_eval(rm_ctx *ctx, const SmallLevels &sl, int T, int t0, int t1, double thr, unsigned flags, CollapsePlan &cp, hipStream_t s);
int collapse_sum(rm_ctx *ctx, const CollapsePlan &cp, double thr, double *heat_sum, hipStream_t s, int avg_T = 0, bool host_rescue = false);

// ---- rm_calibrate.hip
int zero_result(rm_ctx *ctx, size_t npix, double *heat, double *minmax_host, hipStream_t s);
int calibrate_impl(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp, int levels,
                   int skip, double thr, unsigned flags, double *heat, double *minmax_host, void *stream, CollapsePlan *plan_out);

// ---- rm_roi.hip: the ROI stage in two halves (device launches, then -- once they have been waited for -- the host contour stage)
int roi_launch(rm_ctx *ctx, const double *heat, int H, int W, int threshold, uint8_t *avg_u8, uint8_t *binary, void *stream, bool have_minmax,
               RoiPending &pd, bool xywh_given);
int roi_finish(rm_ctx *ctx, const RoiPending &pd, int32_t *xywh);
int heatmap_to_roi_impl(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8, uint8_t *binary, void *stream,
                        bool have_minmax);
