// respmon_amd/csrc/rm_motion.hip -- ROI reductions and motion extraction (base.py:354-407)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

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
        if (npts <= FLOW_FINISH_MAX) hipLaunchKernelGGL(k_flow_finish<>, dim3(1), dim3(64), 2 * sizeof(float) * (size_t)flow_finish_pitch(npts), s, pts, d_out, d_st, npts, dev_res, pts_next);
        else hipLaunchKernelGGL(k_flow_finish_seq<>, dim3(1), dim3(1), 0, s, pts, d_out, d_st, npts, dev_res, pts_next);
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

