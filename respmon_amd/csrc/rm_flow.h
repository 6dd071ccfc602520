// respmon_amd/csrc/rm_flow.h -- optical-flow motion extraction (reference base.py:360-407):
//   cv2.goodFeaturesToTrack (base.py:365-366)   -> Shi-Tomasi min-eigenvalue map on the GPU, greedy
//                                                   min-distance selection (inherently sequential) on the host
//   cv2.calcOpticalFlowPyrLK (base.py:371-372)  -> uint8 pyramids + Scharr derivatives + one wavefront per point
//   np.mean(good_old - good_new, axis=0)        (base.py:388)
//   np.cov / np.linalg.eig / row-unpack / dot   (base.py:396-405)
// Arithmetic follows OpenCV 3.4's generic code paths in the oracle's operation order:
// float32 where OpenCV uses float32, exact integers for the fixed-point bilinear taps, and the float
// accumulations of the LK tracker summed in raster order (every lane runs the same chain over the
// per-tap terms the 64 lanes computed in parallel), so results agree with the CPU oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/respmon_hip.h"
#include "rm_kernels.h"

namespace rm {

struct FlowWorkspace {
    struct Buf { void *p = nullptr; size_t cap = 0; };
    std::map<std::string, Buf> bufs;
    int get(const std::string &name, size_t bytes, void **out, std::string &err)
    {
        Buf &b = bufs[name];
        if (b.cap < bytes) {
            if (b.p) (void)hipFree(b.p);
            b.p = nullptr; b.cap = 0;
            size_t cap = (bytes + 255) / 256 * 256;
            if (hipMalloc(&b.p, cap) != hipSuccess) { err = "hipMalloc failed in flow workspace"; return RM_E_NOMEM; }
            b.cap = cap;
        }
        *out = b.p;
        return RM_OK;
    }
    ~FlowWorkspace()
    {
        for (auto &kv : bufs)
            if (kv.second.p) (void)hipFree(kv.second.p);
    }
};

#define FLOW_HIP(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RM_E_HIP; } \
    } while (0)
#define FLOW_TRY(expr)           \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ < 0) return rc_; \
    } while (0)

// ----------------------------------------------------------------------------------------
// goodFeaturesToTrack: cornerMinEigenVal(blockSize, ksize = 3) on uint8  
// ----------------------------------------------------------------------------------------
// Sobel 3x3 products: cov[.,0] = Dx*Dx, [.,1] = Dx*Dy, [.,2] = Dy*Dy   (float32)
RM_KERNEL __launch_bounds__(256) void k_gftt_cov(const uint8_t *img, int h, int w, float k1, float k2, float *cov)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    int y = i / w, x = i - y * w;
    int y0 = reflect101(y - 1, h), y2 = reflect101(y + 1, h);
    int x0 = reflect101(x - 1, w), x2 = reflect101(x + 1, w);
    const int rows[3] = {y0, y, y2};
    float rdx[3], rdy[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint8_t *s = img + (size_t)rows[r] * w;
        float a = (float)s[x0], b = (float)s[x], c = (float)s[x2];
        float t = -1.0f * a; t += 0.0f * b; t += 1.0f * c;   // row filter [-1 0 1]
        rdx[r] = t;
        float u = k1 * a; u += k2 * b; u += k1 * c;          // row filter scale*[1 2 1]
        rdy[r] = u;
    }
    float dx = (rdx[0] + rdx[2]) * k1 + rdx[1] * k2;         // column filter scale*[1 2 1]
    float dy = rdy[2] - rdy[0];                              // column filter [-1 0 1]
    cov[3 * (size_t)i] = dx * dx; cov[3 * (size_t)i + 1] = dx * dy; cov[3 * (size_t)i + 2] = dy * dy;
}

// un-normalised block x block box sums (row sums then column sums, in double like OpenCV's float path),
// then the smaller eigenvalue of the 2x2 structure tensor in float32
RM_KERNEL __launch_bounds__(256) void k_gftt_eig(const float *cov, int h, int w, int block, float *eig, unsigned int *max_key)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    float e = 0.0f;
    if (i < h * w) {
        int y = i / w, x = i - y * w;
        const int r = block / 2;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int ky = 0; ky < block; ++ky) {
                int yy = reflect101(y - r + ky, h);
                double rs = 0;
                for (int kx = 0; kx < block; ++kx) rs += (double)cov[3 * ((size_t)yy * w + reflect101(x - r + kx, w)) + c];
                s += rs;
            }
            v[c] = (float)s;
        }
        float a = v[0] * 0.5f, b = v[1], c2 = v[2] * 0.5f;
        e = (float)((a + c2) - sqrtf((a - c2) * (a - c2) + b * b));
        eig[i] = e;
    }
    // max over the image (minMaxLoc); eigenvalues are >= -tiny, order floats as sign-fixed ints
    unsigned key = (unsigned)__float_as_int(e);
    key = (key & 0x80000000u) ? ~key : (key | 0x80000000u);
    for (int m = 32; m >= 1; m >>= 1) { unsigned o = __shfl_xor(key, m); key = o > key ? o : key; }
    if ((threadIdx.x & 63) == 0 && i - (int)(threadIdx.x & 63) < h * w) atomicMax(max_key, key);
}

// THRESH_TOZERO, 3x3 dilate, local-maximum test (excluding the 1-pixel frame); candidates appended unordered
RM_KERNEL __launch_bounds__(256) void k_gftt_candidates(const float *eig, int h, int w, float thr, float *cand_val, int *cand_idx,
                                                         int *n_cand, int cap)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    int y = i / w, x = i - y * w;
    if (y < 1 || y >= h - 1 || x < 1 || x >= w - 1) return;
    float v = eig[i];
    v = (v > thr) ? v : 0.0f;
    if (v == 0.0f) return;
    float m = v;
    for (int yy = y - 1; yy <= y + 1; ++yy)
        for (int xx = x - 1; xx <= x + 1; ++xx) {
            float o = eig[(size_t)yy * w + xx];
            o = (o > thr) ? o : 0.0f;
            m = o > m ? o : m;
        }
    if (v == m) {
        int k = atomicAdd(n_cand, 1);
        if (k < cap) { cand_val[k] = v; cand_idx[k] = i; }
    }
}

inline int flow_good_features(FlowWorkspace &ws, const uint8_t *img, int h, int w, int max_corners, double quality, double min_distance,
                              int block_size, float *pts, int *n_out, hipStream_t s, std::string &err)
{
    const size_t n = (size_t)h * w;
    float *cov = nullptr, *eig = nullptr, *cval = nullptr;
    int *cidx = nullptr;
    unsigned int *scal = nullptr;  // [0] max key, [1] candidate count
    FLOW_TRY(ws.get("gftt_cov", n * 3 * sizeof(float), (void **)&cov, err));
    FLOW_TRY(ws.get("gftt_eig", n * sizeof(float), (void **)&eig, err));
    FLOW_TRY(ws.get("gftt_cval", n * sizeof(float), (void **)&cval, err));
    FLOW_TRY(ws.get("gftt_cidx", n * sizeof(int), (void **)&cidx, err));
    FLOW_TRY(ws.get("gftt_scal", 2 * sizeof(unsigned), (void **)&scal, err));
    double scale = (double)(1 << 2) * block_size * 255.0;
    scale = 1.0 / scale;
    const float k1 = (float)(1.0 * scale), k2 = (float)(2.0 * scale);
    const unsigned grid = (unsigned)((n + 255) / 256);
    FLOW_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_gftt_cov<>, dim3(grid), dim3(256), 0, s, img, h, w, k1, k2, cov);
    hipLaunchKernelGGL(k_gftt_eig<>, dim3(grid), dim3(256), 0, s, cov, h, w, block_size, eig, scal);
    FLOW_HIP(hipGetLastError());
    unsigned host_scal[2];
    FLOW_HIP(hipMemcpyAsync(host_scal, scal, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    FLOW_HIP(stream_wait(s));
    unsigned key = host_scal[0];
    key = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    float max_val_f;
    std::memcpy(&max_val_f, &key, 4);
    const double max_val = (double)max_val_f;
    const float thr = (float)(max_val * quality);
    hipLaunchKernelGGL(k_gftt_candidates<>, dim3(grid), dim3(256), 0, s, eig, h, w, thr, cval, cidx, (int *)(scal + 1), (int)n);
    FLOW_HIP(hipGetLastError());
    FLOW_HIP(hipMemcpyAsync(host_scal, scal, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    FLOW_HIP(stream_wait(s));
    int nc = (int)host_scal[1];
    if (nc > (int)n) nc = (int)n;
    std::vector<float> hv(nc);
    std::vector<int> hi(nc);
    if (nc) {
        FLOW_HIP(hipMemcpyAsync(hv.data(), cval, sizeof(float) * nc, hipMemcpyDeviceToHost, s));
        FLOW_HIP(hipMemcpyAsync(hi.data(), cidx, sizeof(int) * nc, hipMemcpyDeviceToHost, s));
        FLOW_HIP(stream_wait(s));
    }
    // sort by value descending, ties by raster index descending (OpenCV >= 3.4 greaterThanPtr), then the
    // greedy minimum-distance acceptance -- sequential by definition
    std::vector<int> order(nc);
    for (int i = 0; i < nc; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (hv[a] != hv[b]) return hv[a] > hv[b];
        return hi[a] > hi[b];
    });
    int ncorners = 0;
    const int cap = max_corners > 0 ? max_corners : nc;
    if (min_distance >= 1) {
        const float md2 = (float)(min_distance * min_distance);
        for (int k = 0; k < nc && ncorners < cap; ++k) {
            int idx = hi[order[k]];
            int y = idx / w, x = idx % w;
            bool good = true;
            for (int j = 0; j < ncorners; ++j) {
                float ddx = (float)x - pts[2 * j], ddy = (float)y - pts[2 * j + 1];
                if (ddx * ddx + ddy * ddy < md2) { good = false; break; }
            }
            if (good) { pts[2 * ncorners] = (float)x; pts[2 * ncorners + 1] = (float)y; ++ncorners; }
        }
    } else {
        for (int k = 0; k < nc && ncorners < cap; ++k) {
            int idx = hi[order[k]];
            pts[2 * ncorners] = (float)(idx % w); pts[2 * ncorners + 1] = (float)(idx / w); ++ncorners;
        }
    }
    *n_out = ncorners;
    return RM_OK;
}

// ----------------------------------------------------------------------------------------
// calcOpticalFlowPyrLK  
// ----------------------------------------------------------------------------------------
// uint8 pyrDown: integer 5-tap, (sum + 128) >> 8, BORDER_REFLECT_101
RM_KERNEL __launch_bounds__(256) void k_pyr_down_u8(const uint8_t *src, int h, int w, uint8_t *dst, int dh, int dw)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dh * dw) return;
    int y = i / dw, x = i - y * dw;
    int rows[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint8_t *s = src + (size_t)reflect101(2 * y - 2 + k, h) * w;
        rows[k] = s[reflect101(2 * x, w)] * 6 + (s[reflect101(2 * x - 1, w)] + s[reflect101(2 * x + 1, w)]) * 4 +
                  s[reflect101(2 * x - 2, w)] + s[reflect101(2 * x + 2, w)];
    }
    int v = rows[2] * 6 + (rows[1] + rows[3]) * 4 + rows[0] + rows[4];
    dst[i] = (uint8_t)((v + 128) >> 8);
}

// calcSharrDeriv: int16 (Ix, Iy) interleaved; reflect-101 inside the image
RM_KERNEL __launch_bounds__(256) void k_scharr(const uint8_t *src, int h, int w, short *d)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    int y = i / w, x = i - y * w;
    const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * w;
    const uint8_t *r1 = src + (size_t)y * w;
    const uint8_t *r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * w;
    int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    auto t0 = [&](int c) { return (int)(short)((r0[c] + r2[c]) * 3 + r1[c] * 10); };
    auto t1 = [&](int c) { return (int)(short)(r2[c] - r0[c]); };
    d[2 * (size_t)i] = (short)(t0(xp) - t0(xm));
    d[2 * (size_t)i + 1] = (short)((t1(xp) + t1(xm)) * 3 + t1(x) * 10);
}

constexpr int LK_MAX_LEVELS = 8;
constexpr int LK_MAX_WIN = 1024;  // taps per window (winSize up to 32x32)

struct LKLevels {
    int n;  // number of levels (maxLevel + 1)
    int h[LK_MAX_LEVELS], w[LK_MAX_LEVELS];
    const uint8_t *prev[LK_MAX_LEVELS], *next[LK_MAX_LEVELS];
    const short *deriv[LK_MAX_LEVELS];
};

__device__ __forceinline__ int lk_px(const uint8_t *img, int h, int w, int y, int x)
{
    return img[(size_t)reflect101(y, h) * w + reflect101(x, w)];  // image + REFLECT_101 pad of winSize
}
__device__ __forceinline__ int lk_dv(const short *d, int h, int w, int y, int x, int c)
{
    if (y < 0 || y >= h || x < 0 || x >= w) return 0;             // derivative + zero pad
    return d[2 * ((size_t)y * w + x) + c];
}
#define LK_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

// The raster-order float sums of the tracker (OpenCV's generic path adds the window's terms one after the other): every lane runs
// the same chain.  The terms come from LDS sixteen bytes at a time (broadcast reads, requested well ahead of the additions that
// use them): a loop of scalar reads paid one LDS round trip (~64 cycles) per term, 450 terms per iteration of the tracker.
struct alignas(16) LkF4 { float x, y, z, w; };
__device__ __forceinline__ void lk_seq_sum2(const float *a, const float *b, int n, float &sa, float &sb)
{
    const int n4 = n & ~3;
#pragma unroll 4
    for (int k = 0; k < n4; k += 4) {
        const LkF4 va = *reinterpret_cast<const LkF4 *>(a + k), vb = *reinterpret_cast<const LkF4 *>(b + k);
        sa += va.x; sb += vb.x; sa += va.y; sb += vb.y; sa += va.z; sb += vb.z; sa += va.w; sb += vb.w;
    }
    for (int k = n4; k < n; ++k) { sa += a[k]; sb += b[k]; }
}
__device__ __forceinline__ void lk_seq_sum3(const float *a, const float *b, const float *c, int n, float &sa, float &sb, float &sc)
{
    const int n4 = n & ~3;
#pragma unroll 4
    for (int k = 0; k < n4; k += 4) {
        const LkF4 va = *reinterpret_cast<const LkF4 *>(a + k), vb = *reinterpret_cast<const LkF4 *>(b + k), vc = *reinterpret_cast<const LkF4 *>(c + k);
        sa += va.x; sb += vb.x; sc += vc.x; sa += va.y; sb += vb.y; sc += vc.y; sa += va.z; sb += vb.z; sc += vc.z; sa += va.w; sb += vb.w; sc += vc.w;
    }
    for (int k = n4; k < n; ++k) { sa += a[k]; sb += b[k]; sc += c[k]; }
}

// One wavefront per point.  The 64 lanes compute the window's fixed-point terms in parallel; the float
// accumulations run as one raster-order chain (replicated in every lane) to match OpenCV's generic path.
// ROUNDS: trips of the tap loops (lane k, k + 64, ...), unrolled so that the gathers of all of a window's taps are in flight
// together: 4 covers winSize up to 16 x 16 (the reference's 15 x 15, base.py:96), 16 the 32 x 32 maximum.
template <int ROUNDS>
__global__ __launch_bounds__(64) void k_lk_track(LKLevels L, const float *pts_in, int npts, int win_w, int win_h, int max_count,
                                                 double epsilon, float *pts_out, uint8_t *status)
{
    __shared__ short s_I[LK_MAX_WIN];
    __shared__ short s_dI[2 * LK_MAX_WIN];
    __shared__ __attribute__((aligned(16))) float s_t0[LK_MAX_WIN], s_t1[LK_MAX_WIN], s_t2[LK_MAX_WIN];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= npts) return;
    const int ntap = win_w * win_h;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int W_BITS = 14;
    const float min_eig_threshold = (float)1e-4;
    const float px_in = pts_in[2 * p], py_in = pts_in[2 * p + 1];
    int tap_y[ROUNDS], tap_x[ROUNDS];     // window position of this lane's taps (one integer division each, once)
#pragma unroll
    for (int rnd = 0; rnd < ROUNDS; ++rnd) { const int k = lane + 64 * rnd; tap_y[rnd] = k / win_w; tap_x[rnd] = k - tap_y[rnd] * win_w; }
    float out_x = 0.f, out_y = 0.f;
    int st = 1;
    for (int level = L.n - 1; level >= 0; --level) {
        const int h = L.h[level], w = L.w[level];
        const uint8_t *I = L.prev[level], *J = L.next[level];
        const short *dI = L.deriv[level];
        const float sc = (float)(1. / (1 << level));
        float prev_x = px_in * sc, prev_y = py_in * sc;
        float next_x, next_y;
        if (level == L.n - 1) { next_x = prev_x; next_y = prev_y; }
        else { next_x = out_x * 2.f; next_y = out_y * 2.f; }
        out_x = next_x; out_y = next_y;
        prev_x -= half_x; prev_y -= half_y;
        const int ipx = (int)floorf(prev_x), ipy = (int)floorf(prev_y);
        if (ipx < -win_w || ipx >= w || ipy < -win_h || ipy >= h) {
            if (level == 0) st = 0;
            continue;
        }
        float a = prev_x - ipx, b = prev_y - ipy;
        int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        __syncthreads();
#pragma unroll
        for (int rnd = 0; rnd < ROUNDS; ++rnd) {
            const int k = lane + 64 * rnd;
            if (k >= ntap) continue;
            const int yy = ipy + tap_y[rnd], xx = ipx + tap_x[rnd];
            int ival = LK_DESCALE(lk_px(I, h, w, yy, xx) * iw00 + lk_px(I, h, w, yy, xx + 1) * iw01 + lk_px(I, h, w, yy + 1, xx) * iw10 +
                                  lk_px(I, h, w, yy + 1, xx + 1) * iw11, W_BITS - 5);
            int ixval = LK_DESCALE(lk_dv(dI, h, w, yy, xx, 0) * iw00 + lk_dv(dI, h, w, yy, xx + 1, 0) * iw01 +
                                   lk_dv(dI, h, w, yy + 1, xx, 0) * iw10 + lk_dv(dI, h, w, yy + 1, xx + 1, 0) * iw11, W_BITS);
            int iyval = LK_DESCALE(lk_dv(dI, h, w, yy, xx, 1) * iw00 + lk_dv(dI, h, w, yy, xx + 1, 1) * iw01 +
                                   lk_dv(dI, h, w, yy + 1, xx, 1) * iw10 + lk_dv(dI, h, w, yy + 1, xx + 1, 1) * iw11, W_BITS);
            s_I[k] = (short)ival; s_dI[2 * k] = (short)ixval; s_dI[2 * k + 1] = (short)iyval;
            s_t0[k] = (float)(ixval * ixval); s_t1[k] = (float)(ixval * iyval); s_t2[k] = (float)(iyval * iyval);
        }
        __syncthreads();
        float iA11 = 0, iA12 = 0, iA22 = 0;
        lk_seq_sum3(s_t0, s_t1, s_t2, ntap, iA11, iA12, iA22);
        const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win_w * win_h);
        if (min_eig < min_eig_threshold || D < 1.1920928955078125e-07f) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        next_x -= half_x; next_y -= half_y;
        float pdx = 0, pdy = 0;
        for (int j = 0; j < max_count; ++j) {
            const int inx = (int)floorf(next_x), iny = (int)floorf(next_y);
            if (inx < -win_w || inx >= w || iny < -win_h || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            a = next_x - inx; b = next_y - iny;
            iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            __syncthreads();
#pragma unroll
            for (int rnd = 0; rnd < ROUNDS; ++rnd) {
                const int k = lane + 64 * rnd;
                if (k >= ntap) continue;
                const int yy = iny + tap_y[rnd], xx = inx + tap_x[rnd];
                int diff = LK_DESCALE(lk_px(J, h, w, yy, xx) * iw00 + lk_px(J, h, w, yy, xx + 1) * iw01 + lk_px(J, h, w, yy + 1, xx) * iw10 +
                                      lk_px(J, h, w, yy + 1, xx + 1) * iw11, W_BITS - 5) - s_I[k];
                s_t0[k] = (float)(diff * s_dI[2 * k]); s_t1[k] = (float)(diff * s_dI[2 * k + 1]);
            }
            __syncthreads();
            float ib1 = 0, ib2 = 0;
            lk_seq_sum2(s_t0, s_t1, ntap, ib1, ib2);
            const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
            const float dx = (float)((A12 * b2 - A22 * b1) * D);
            const float dy = (float)((A12 * b1 - A11 * b2) * D);
            next_x += dx; next_y += dy;
            out_x = next_x + half_x; out_y = next_y + half_y;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                out_x -= dx * 0.5f; out_y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
    }
    if (lane == 0) { pts_out[2 * p] = out_x; pts_out[2 * p + 1] = out_y; status[p] = (uint8_t)st; }
}

// buildOpticalFlowPyramid keeps levels while the next one stays larger than the window
inline int lk_max_level(int h, int w, int win_w, int win_h, int max_level)
{
    int sh = h, sw = w;
    for (int level = 0; level <= max_level; ++level) {
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= win_w || sh <= win_h) return level;
    }
    return max_level;
}

// device half of flow_pyr_lk: pyramids, derivatives and the tracking kernel; points in / out and status stay on the device,
// nothing is copied and the stream is not waited for.  Returns the number of pyramid levels above level 0 that were used.
inline int flow_pyr_lk_dev(FlowWorkspace &ws, const uint8_t *prev, const uint8_t *next, int h, int w, const float *d_in, int npts,
                           int win_w, int win_h, int max_level, int max_count, double epsilon, float *d_out, uint8_t *d_st,
                           hipStream_t s, std::string &err)
{
    if (win_w * win_h > LK_MAX_WIN) { err = "winSize too large"; return RM_E_UNSUPPORTED; }
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0) epsilon = 0;
    if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    max_level = lk_max_level(h, w, win_w, win_h, max_level);
    if (max_level + 1 > LK_MAX_LEVELS) { err = "too many pyramid levels"; return RM_E_UNSUPPORTED; }
    LKLevels L;
    L.n = max_level + 1;
    int sh = h, sw = w;
    for (int l = 0; l <= max_level; ++l) {
        L.h[l] = sh; L.w[l] = sw;
        uint8_t *pp = nullptr, *nn = nullptr;
        short *dd = nullptr;
        const size_t n = (size_t)sh * sw;
        if (l == 0) { pp = const_cast<uint8_t *>(prev); nn = const_cast<uint8_t *>(next); }
        else {
            FLOW_TRY(ws.get("lk_prev" + std::to_string(l), n, (void **)&pp, err));
            FLOW_TRY(ws.get("lk_next" + std::to_string(l), n, (void **)&nn, err));
            const unsigned grid = (unsigned)((n + 255) / 256);
            hipLaunchKernelGGL(k_pyr_down_u8<>, dim3(grid), dim3(256), 0, s, L.prev[l - 1], L.h[l - 1], L.w[l - 1], pp, sh, sw);
            hipLaunchKernelGGL(k_pyr_down_u8<>, dim3(grid), dim3(256), 0, s, L.next[l - 1], L.h[l - 1], L.w[l - 1], nn, sh, sw);
        }
        FLOW_TRY(ws.get("lk_deriv" + std::to_string(l), n * 2 * sizeof(short), (void **)&dd, err));
        hipLaunchKernelGGL(k_scharr<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pp, sh, sw, dd);
        L.prev[l] = pp; L.next[l] = nn; L.deriv[l] = dd;
        sh = (sh + 1) / 2; sw = (sw + 1) / 2;
    }
    if (win_w * win_h <= 256) hipLaunchKernelGGL(k_lk_track<4>, dim3(npts), dim3(64), 0, s, L, d_in, npts, win_w, win_h, max_count, epsilon, d_out, d_st);
    else hipLaunchKernelGGL(k_lk_track<16>, dim3(npts), dim3(64), 0, s, L, d_in, npts, win_w, win_h, max_count, epsilon, d_out, d_st);
    FLOW_HIP(hipGetLastError());
    return max_level;
}

inline int flow_pyr_lk(FlowWorkspace &ws, const uint8_t *prev, const uint8_t *next, int h, int w, const float *pts_in, int npts, int win_w,
                       int win_h, int max_level, int max_count, double epsilon, float *pts_out, uint8_t *status, hipStream_t s,
                       std::string &err)
{
    if (npts == 0) return RM_OK;
    float *d_in = nullptr, *d_out = nullptr;
    uint8_t *d_st = nullptr;
    FLOW_TRY(ws.get("lk_pts_in", sizeof(float) * 2 * npts, (void **)&d_in, err));
    FLOW_TRY(ws.get("lk_pts_out", sizeof(float) * 2 * npts, (void **)&d_out, err));
    FLOW_TRY(ws.get("lk_status", npts, (void **)&d_st, err));
    FLOW_HIP(hipMemcpyAsync(d_in, pts_in, sizeof(float) * 2 * npts, hipMemcpyHostToDevice, s));
    const int used = flow_pyr_lk_dev(ws, prev, next, h, w, d_in, npts, win_w, win_h, max_level, max_count, epsilon, d_out, d_st, s, err);
    if (used < 0) return used;
    FLOW_HIP(hipMemcpyAsync(pts_out, d_out, sizeof(float) * 2 * npts, hipMemcpyDeviceToHost, s));
    FLOW_HIP(hipMemcpyAsync(status, d_st, npts, hipMemcpyDeviceToHost, s));
    FLOW_HIP(stream_wait(s));
    return used;
}

// One frame of extract_motion('flow') without a host round trip in the middle (base.py:377-388): the mean of old - new over
// the points with status == 1 (float32, in point order: k_mean_flow's arithmetic) AND p1[st == 1] packed in order into the
// buffer the next frame tracks from.  res (pinned host memory): {mean_x, mean_y, (float) n_good}.
// One wave: the lanes pack the surviving points and their differences (ballot + prefix popcount, 64 points per trip), then
// lane 0 adds the differences in point order from LDS.  (One THREAD walking global memory took 150 us for 1 000 points.)
constexpr int FLOW_FINISH_MAX = 6000;   // points whose differences fit the LDS staging (2 floats each)
__host__ __device__ __forceinline__ int flow_finish_pitch(int n) { return (n + 3) & ~3; }   // floats per staged component
RM_KERNEL __launch_bounds__(64) void k_flow_finish(const float *o, const float *nw, const uint8_t *st, int n, float *res, float *next_pts)
{
    HIP_DYNAMIC_SHARED(float, s_d)     // [2 * flow_finish_pitch(n)]: dx of the survivors, then dy (both 16-byte aligned: lk_seq_sum2 reads float4)
    const int lane = threadIdx.x;
    float *s_dx = s_d, *s_dy = s_d + flow_finish_pitch(n);
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool good = i < n && st[i] == 1;
        float ox = 0.f, oy = 0.f, nx = 0.f, ny = 0.f;
        if (good) { ox = o[2 * i]; oy = o[2 * i + 1]; nx = nw[2 * i]; ny = nw[2 * i + 1]; }
        const unsigned long long m = __ballot(good);
        if (good) {
            const int slot = base + (int)__popcll(m & ((1ull << lane) - 1ull));
            next_pts[2 * slot] = nx; next_pts[2 * slot + 1] = ny;
            s_dx[slot] = ox - nx; s_dy[slot] = oy - ny;
        }
        base += (int)__popcll(m);
    }
    __syncthreads();
    if (lane == 0) {
        float sx = 0.f, sy = 0.f;
        lk_seq_sum2(s_dx, s_dy, base, sx, sy);
        res[0] = base ? sx / (float)base : 0.f;
        res[1] = base ? sy / (float)base : 0.f;
        res[2] = (float)base;
    }
}
// (more points than the staging holds: one thread, global memory)
RM_KERNEL void k_flow_finish_seq(const float *o, const float *nw, const uint8_t *st, int n, float *res, float *next_pts)
{
    float sx = 0.f, sy = 0.f;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (st[i] == 1) {
            sx += o[2 * i] - nw[2 * i]; sy += o[2 * i + 1] - nw[2 * i + 1];
            next_pts[2 * m] = nw[2 * i]; next_pts[2 * m + 1] = nw[2 * i + 1];
            ++m;
        }
    res[0] = m ? sx / (float)m : 0.f;
    res[1] = m ? sy / (float)m : 0.f;
    res[2] = (float)m;
}

// Device-resident state of one extract_motion('flow') session (rm_flow_state): the two ROI crops a step works on -- the
// previous one and its successor, alternating -- each with its LK pyramid (uint8 levels) and Scharr derivatives, the tracked
// points, the pinned result words.  Every RespiratoryMonitor owns one, so two monitors on one GPU never see each other's
// crops or points.  What a step builds for the NEW crop (pyramid) or for the crop it tracks FROM (derivatives) is kept: the crop
// that was "next" in step i is "previous" in step i + 1 and arrives with its pyramid, so a step builds one pyramid and one set
// of derivatives instead of two and two (base.py:381: `previous_cropped_image = cropped_image`).
struct FlowState {
    FlowWorkspace ws;
    int w = 0, h = 0, npts = 0, cap = 0, flip = 0;
    float *res = nullptr;            // pinned {mean_x, mean_y, n_good}
    int pyr_levels[2] = {-1, -1};    // highest pyramid level that exists for side 0 / 1 (0: the crop only; -1: no crop)
    int deriv_levels[2] = {-1, -1};  // highest level whose Scharr derivatives exist
    bool begun = false;
    ~FlowState() { if (res) (void)hipHostFree(res); }
};

inline int flow_side_buf(FlowState &fs, int side, const char *what, int level, size_t bytes, void **out, std::string &err)
{
    return fs.ws.get(std::string(what) + (side ? "_b" : "_a") + std::to_string(level), bytes, out, err);
}

// pyramids / derivatives of the two sides as far as this step needs them, then the tracking kernel (points stay on the device)
inline int flow_track_resident(FlowState &fs, int prev_side, int cur_side, const float *d_in, int npts, int win_w, int win_h, int max_level,
                               int max_count, double epsilon, float *d_out, uint8_t *d_st, hipStream_t s, std::string &err)
{
    const int h = fs.h, w = fs.w;
    if (win_w * win_h > LK_MAX_WIN) { err = "winSize too large"; return RM_E_UNSUPPORTED; }
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0) epsilon = 0;
    if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    max_level = lk_max_level(h, w, win_w, win_h, max_level);
    if (max_level + 1 > LK_MAX_LEVELS) { err = "too many pyramid levels"; return RM_E_UNSUPPORTED; }
    LKLevels L;
    L.n = max_level + 1;
    int sh = h, sw = w;
    for (int l = 0; l <= max_level; ++l) {
        L.h[l] = sh; L.w[l] = sw;
        const size_t n = (size_t)sh * sw;
        uint8_t *pp = nullptr, *nn = nullptr;
        short *dd = nullptr;
        FLOW_TRY(flow_side_buf(fs, prev_side, "pyr", l, n, (void **)&pp, err));
        FLOW_TRY(flow_side_buf(fs, cur_side, "pyr", l, n, (void **)&nn, err));
        FLOW_TRY(flow_side_buf(fs, prev_side, "deriv", l, n * 2 * sizeof(short), (void **)&dd, err));
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (l > 0) {
            if (fs.pyr_levels[prev_side] < l) {
                hipLaunchKernelGGL(k_pyr_down_u8<>, dim3(grid), dim3(256), 0, s, L.prev[l - 1], L.h[l - 1], L.w[l - 1], pp, sh, sw);
                fs.pyr_levels[prev_side] = l;
            }
            if (fs.pyr_levels[cur_side] < l) {
                hipLaunchKernelGGL(k_pyr_down_u8<>, dim3(grid), dim3(256), 0, s, L.next[l - 1], L.h[l - 1], L.w[l - 1], nn, sh, sw);
                fs.pyr_levels[cur_side] = l;
            }
        }
        if (fs.deriv_levels[prev_side] < l) {
            hipLaunchKernelGGL(k_scharr<>, dim3(grid), dim3(256), 0, s, pp, sh, sw, dd);
            fs.deriv_levels[prev_side] = l;
        }
        L.prev[l] = pp; L.next[l] = nn; L.deriv[l] = dd;
        sh = (sh + 1) / 2; sw = (sw + 1) / 2;
    }
    if (win_w * win_h <= 256) hipLaunchKernelGGL(k_lk_track<4>, dim3(npts), dim3(64), 0, s, L, d_in, npts, win_w, win_h, max_count, epsilon, d_out, d_st);
    else hipLaunchKernelGGL(k_lk_track<16>, dim3(npts), dim3(64), 0, s, L, d_in, npts, win_w, win_h, max_count, epsilon, d_out, d_st);
    FLOW_HIP(hipGetLastError());
    return max_level;
}

// ----------------------------------------------------------------------------------------
// np.mean(good_old - good_new, axis=0): float32, sequential over the points with status == 1  (base.py:377-388)
// ----------------------------------------------------------------------------------------
RM_KERNEL void k_mean_flow(const float *o, const float *nw, const uint8_t *st, int n, float *mean_xy, int *n_good)
{
    float sx = 0.f, sy = 0.f;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (st[i] == 1) { sx += o[2 * i] - nw[2 * i]; sy += o[2 * i + 1] - nw[2 * i + 1]; ++m; }
    mean_xy[0] = m ? sx / (float)m : 0.f;
    mean_xy[1] = m ? sy / (float)m : 0.f;
    *n_good = m;
}

inline int flow_mean(FlowWorkspace &ws, const float *old_pts, const float *new_pts, const uint8_t *status, int npts, float *mean_xy,
                     int *n_good, hipStream_t s, std::string &err)
{
    if (npts == 0) { mean_xy[0] = mean_xy[1] = 0.f; *n_good = 0; return RM_OK; }
    float *d_o = nullptr, *d_n = nullptr, *d_m = nullptr;
    uint8_t *d_s = nullptr;
    int *d_c = nullptr;
    FLOW_TRY(ws.get("mf_old", sizeof(float) * 2 * npts, (void **)&d_o, err));
    FLOW_TRY(ws.get("mf_new", sizeof(float) * 2 * npts, (void **)&d_n, err));
    FLOW_TRY(ws.get("mf_st", npts, (void **)&d_s, err));
    FLOW_TRY(ws.get("mf_mean", sizeof(float) * 2, (void **)&d_m, err));
    FLOW_TRY(ws.get("mf_cnt", sizeof(int), (void **)&d_c, err));
    FLOW_HIP(hipMemcpyAsync(d_o, old_pts, sizeof(float) * 2 * npts, hipMemcpyHostToDevice, s));
    FLOW_HIP(hipMemcpyAsync(d_n, new_pts, sizeof(float) * 2 * npts, hipMemcpyHostToDevice, s));
    FLOW_HIP(hipMemcpyAsync(d_s, status, npts, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_mean_flow<>, dim3(1), dim3(1), 0, s, d_o, d_n, d_s, npts, d_m, d_c);
    FLOW_HIP(hipGetLastError());
    FLOW_HIP(hipMemcpyAsync(mean_xy, d_m, sizeof(float) * 2, hipMemcpyDeviceToHost, s));
    FLOW_HIP(hipMemcpyAsync(n_good, d_c, sizeof(int), hipMemcpyDeviceToHost, s));
    FLOW_HIP(stream_wait(s));
    return RM_OK;
}

// ----------------------------------------------------------------------------------------
// PCA reduction (base.py:396-405): np.cov (ddof = 1, float64) -> np.linalg.eig -> argsort descending ->
// evec1, evec2 = eig_vecs[:, idx] (ROW unpack: evec1 = x-components of the major and minor vectors) ->
// dot(motion_data, evec1)[-1].  One wavefront; sums by wave shuffles; the 2x2 eigen-decomposition restates
// LAPACK dgeev's path for a 2x2 matrix (dlanv2 standardisation, dtrevc back-substitution, unit 2-norm), which
// fixes the eigenvector SIGNS the reference's result depends on.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ inline double d_sign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

// LAPACK dgeev on [[a, b], [c, d]] with real eigenvalues -> eigenvalues w[2], eigenvectors v (columns), unit norm
__device__ inline void eig2x2_dgeev(double a, double b, double c, double d, double *w, double (*v)[2])
{
    const double eps = 2.220446049250313e-16 * 0.5;  // dlamch('P') = eps*base/2... LAPACK: precision = eps*radix
    const double multpl = 4.0;
    double cs, sn;
    // dlanv2
    if (c == 0.0) { cs = 1.0; sn = 0.0; }
    else if (b == 0.0) { cs = 0.0; sn = 1.0; double temp = d; d = a; a = temp; b = -c; c = 0.0; }
    else if ((a - d) == 0.0 && d_sign(1.0, b) != d_sign(1.0, c)) { cs = 1.0; sn = 0.0; }
    else {
        double temp = a - d;
        double p = 0.5 * temp;
        double bcmax = fmax(fabs(b), fabs(c));
        double bcmis = fmin(fabs(b), fabs(c)) * d_sign(1.0, b) * d_sign(1.0, c);
        double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * (2.0 * eps)) {
            // real eigenvalues
            z = p + d_sign(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            double tau = hypot(c, z);
            cs = z / tau; sn = c / tau;
            b = b - c; c = 0.0;
        } else {
            // complex or nearly equal eigenvalues: cannot occur for a symmetric PSD covariance with distinct values;
            // fall back to the symmetric standardisation
            double sigma = b + c;
            double tau = hypot(sigma, temp);
            cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * d_sign(1.0, sigma);
            double aa = a * cs + b * sn, bb = -a * sn + b * cs, cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn; c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d); a = temp; d = temp;
            if (c != 0.0 && b != 0.0 && d_sign(1.0, b) == d_sign(1.0, c)) {
                double sab = sqrt(fabs(b)), sac = sqrt(fabs(c));
                p = d_sign(sab * sac, c);
                tau = 1.0 / sqrt(fabs(b + c));
                a = temp + p; d = temp - p;
                b = b - c; c = 0.0;
                double cs1 = sab * tau, sn1 = sac * tau;
                temp = cs * cs1 - sn * sn1; sn = cs * sn1 + sn * cs1; cs = temp;
            }
        }
    }
    // Schur form T = [[a, b], [0, d]], Schur vectors Z = [[cs, -sn], [sn, cs]]
    w[0] = a; w[1] = d;
    // dtrevc (back-transformed right eigenvectors): x1 = e1, x2 = [-b/(a-d), 1] (scaled so that max |.| = 1 after Z)
    double x2_0 = (a - d != 0.0) ? -b / (a - d) : 0.0, x2_1 = 1.0;
    double v1[2] = {cs, sn};
    double v2[2] = {cs * x2_0 - sn * x2_1, sn * x2_0 + cs * x2_1};
    double e1 = fmax(fabs(v1[0]), fabs(v1[1])), e2 = fmax(fabs(v2[0]), fabs(v2[1]));
    v1[0] /= e1; v1[1] /= e1; v2[0] /= e2; v2[1] /= e2;
    double n1 = 1.0 / hypot(v1[0], v1[1]), n2 = 1.0 / hypot(v2[0], v2[1]);
    v[0][0] = v1[0] * n1; v[1][0] = v1[1] * n1; v[0][1] = v2[0] * n2; v[1][1] = v2[1] * n2;
}

RM_KERNEL __launch_bounds__(64) void k_pca_reduce(const float *motion, int n, double *out)
{
    const int lane = threadIdx.x;
    double sx = 0, sy = 0;
    for (int i = lane; i < n; i += 64) { sx += (double)motion[2 * i]; sy += (double)motion[2 * i + 1]; }
    sx = wave_sum(sx); sy = wave_sum(sy);
    const double mx = sx / n, my = sy / n;
    double cxx = 0, cxy = 0, cyy = 0;
    for (int i = lane; i < n; i += 64) {
        double dx = (double)motion[2 * i] - mx, dy = (double)motion[2 * i + 1] - my;
        cxx += dx * dx; cxy += dx * dy; cyy += dy * dy;
    }
    cxx = wave_sum(cxx); cxy = wave_sum(cxy); cyy = wave_sum(cyy);
    if (lane == 0) {
        const double f = 1.0 / (double)(n - 1);
        cxx *= f; cxy *= f; cyy *= f;
        double w[2], v[2][2];
        eig2x2_dgeev(cxx, cxy, cxy, cyy, w, v);
        // sort_indices = argsort(eig_vals)[::-1]; evec1 = FIRST ROW of eig_vecs[:, idx]
        const int i0 = (w[0] > w[1]) ? 0 : (w[0] < w[1] ? 1 : 1), i1 = 1 - i0;
        const double e0 = v[0][i0], e1 = v[0][i1];
        out[0] = (double)motion[2 * (n - 1)] * e0 + (double)motion[2 * (n - 1) + 1] * e1;
    }
}

inline int flow_pca(FlowWorkspace &ws, const float *motion, int n, double *out, hipStream_t s, std::string &err)
{
    if (n < 2) { *out = 0.0; return RM_OK; }  // base.py:406-407
    float *d_m = nullptr;
    double *d_o = nullptr;
    FLOW_TRY(ws.get("pca_in", sizeof(float) * 2 * n, (void **)&d_m, err));
    FLOW_TRY(ws.get("pca_out", sizeof(double), (void **)&d_o, err));
    FLOW_HIP(hipMemcpyAsync(d_m, motion, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pca_reduce<>, dim3(1), dim3(64), 0, s, d_m, n, d_o);
    FLOW_HIP(hipGetLastError());
    FLOW_HIP(hipMemcpyAsync(out, d_o, sizeof(double), hipMemcpyDeviceToHost, s));
    FLOW_HIP(stream_wait(s));
    return RM_OK;
}

}  // namespace rm
