// respmon_amd/csrc/rm_comm.hip -- multi-GPU steps: RCCL behind the C-ABI
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// ------------------------------------------------------------------------------------------
// multi-GPU steps behind the C-ABI: RCCL on the caller's stream (the call site this replaces is base.py:444 run once per
// GPU).  librccl is opened at run time (dlopen: a process that already holds RCCL -- PyTorch-ROCm's copy carries the same SONAME --
// shares it), so a single-GPU user of the library never needs it.
//   Mode B (rm_locate_streams): calibrate -> sparse packet -> ncclAllGather -> merge + ROI   (dense: ncclAllReduce(sum) of the heatmap)
//   Mode A (rm_locate_sharded): pyramid of the local frames -> ncclAllGather -> collapse -> ncclAllReduce(max) of {-min, max} ->
//                               masked sum of the local frames -> sparse packets / dense all-reduce -> ROI
// one host synchronisation per step on the common (sparse) path.
// ------------------------------------------------------------------------------------------
#if !defined(RM_HIPEMU) && defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#define RM_HAVE_RCCL 1
#endif
#endif
#ifdef RM_HAVE_RCCL
#include <dlfcn.h>
#include <mutex>
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
static void rccl_open(RcclApi &a);
static RcclApi &rccl_api()
{
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] { rccl_open(a); });
    return a;
}
static void rccl_open(RcclApi &a)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
    if (!a.lib) { a.err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return; }
    auto sym = [&](const char *s) { void *p = dlsym(a.lib, s); if (!p && a.err.empty()) a.err = std::string("librccl has no symbol ") + s; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
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
#ifdef RM_HAVE_RCCL
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
#ifdef RM_HAVE_RCCL
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
#ifdef RM_HAVE_RCCL
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
    return fail(RM_E_COMM, "rm_comm_init: this build of the library has no RCCL (host emulation, or built without <rccl/rccl.h>)");
#endif
}

extern "C" int rm_comm_info(rm_ctx *ctx, int *rank, int *world, int *rccl_ranks)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_comm_info: ctx is NULL");
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    if (rccl_ranks) {
        *rccl_ranks = 0;   // 0: no RCCL communicator behind this context
#ifdef RM_HAVE_RCCL
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
#ifdef RM_HAVE_RCCL
    if (ctx->comm) { RCCL_TRY(rccl_api().AllGather(send, recv, count, ncclFloat64, (ncclComm_t)ctx->comm, s)); return RM_OK; }
#endif
    if (recv != send) HIP_TRY(hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    return RM_OK;
}
static int comm_all_reduce(rm_ctx *ctx, double *buf, size_t count, bool max_not_sum, hipStream_t s)
{
#ifdef RM_HAVE_RCCL
    if (ctx->comm) { RCCL_TRY(rccl_api().AllReduce(buf, buf, count, ncclFloat64, max_not_sum ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, s)); return RM_OK; }
#endif
    (void)buf; (void)count; (void)max_not_sum; (void)s;
    return RM_OK;
}

// what a heatmap exchange remembers from one step to the next (respmon_amd/dist.py ExchangePolicy, now per communicator): a packet
// that overflowed tells every rank how many tiles the fullest rank needed -- all of them read all the headers, so all switch together
// (only PEEKS at the dense hold: the counter moves in exchange_heat_roi once the step's collective has been enqueued -- a rank whose
//  step fails before that must not run ahead of its peers, or it would issue ncclAllGather while they issue ncclAllReduce)
static bool exchange_use_sparse(const ExchangeState &x) { return x.dense_left <= 0; }
static void exchange_overflowed(ExchangeState &x, int needed)
{
    const int want = ((int)(needed * 1.25) + 63) / 64 * 64;
    if (needed > 0 && want <= RM_SPARSE_MAX_TILES) x.cap = std::max(x.cap, want);
    else x.dense_left = RM_DENSE_HOLD;
}

// heat (this rank's heatmap, or its partial heat SUM when avg_T > 0) -> the ranks' sum (/ avg_T) -> ROI; fused_out (nullable): the sum
int exchange_heat_roi(rm_ctx *ctx, ExchangeState &xp, double *heat, int H, int W, int threshold, int avg_T, double *fused_out, int32_t *xywh,
                             int *exchange_out, hipStream_t s)
{
    const size_t npix = (size_t)H * W;
    const int world = ctx->comm_world;
    const bool clip_once = ctx->clip_frame_once;   // (consumed by the first ROI stage: a second one after an overflow asks again)
    // ("exchange_dense" is a per-rank developer switch that changes the collective: set it identically on every rank)
    const bool held_dense = !ctx->dbg.exchange_dense && !exchange_use_sparse(xp);
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
    if (held_dense) --xp.dense_left;   // one step of the dense hold is spent: its all-reduce is on the stream
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
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
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

