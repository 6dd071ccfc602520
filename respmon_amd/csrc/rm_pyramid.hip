// respmon_amd/csrc/rm_pyramid.hip -- pyramid building blocks and the materialising pyramid API (pyramid.py:9-69)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// ------------------------------------------------------------------------------------------
// pyramid building blocks
// ------------------------------------------------------------------------------------------
int launch_pyr_down(const void *src, int dtype, int T, int h, int w, double *dst, hipStream_t s)
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

int launch_pyr_up(const double *src, int T, int sh, int sw, double *dst, int dh, int dw, int mode,
                         const double *other, hipStream_t s, size_t src_fs, size_t dst_fs, size_t other_fs)
{
    if (!src_fs) src_fs = (size_t)sh * sw;
    if (!dst_fs) dst_fs = (size_t)dh * dw;
    if (!other_fs) other_fs = (size_t)dh * dw;
    if (!((dw == 2 * sw || dw == 2 * sw - 1) && (dh == 2 * sh || dh == 2 * sh - 1)))
        return fail(RM_E_BADARG, "pyrUp: dstsize (%d,%d) incompatible with source (%d,%d)", dw, dh, sw, sh);
    if (dw >= 128 && dh >= 8 && dst != src) {   // large levels: 2 x 2 outputs per thread
        dim3 grid((dw + 127) / 128, (dh + 7) / 8, T), block(256);
        hipLaunchKernelGGL(k_pyr_up_2x2<>, grid, block, 0, s, src, sh, sw, src_fs, dst, dh, dw, dst_fs, mode, other, other_fs);
        LAUNCH_CHECK();
        return RM_OK;
    }
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, T), block(256);
    hipLaunchKernelGGL(k_pyr_up<>, grid, block, 0, s, src, sh, sw, src_fs, dst, dh, dw, dst_fs, mode, other, other_fs);
    LAUNCH_CHECK();
    return RM_OK;
}

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

int launch_to_f64(const void *src, int dtype, size_t n, double *dst, hipStream_t s)
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

void level_sizes(int H, int W, int levels, std::vector<int> &h, std::vector<int> &w)
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

