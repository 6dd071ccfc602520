// respmon_amd/csrc/rm_kernels.h -- gfx950 HIP kernels of the calibration path.
//
// All arithmetic that decides the ROI is float64 with the operation order of the reference's
// numpy / OpenCV-scalar path (no FMA contraction: the library is built with -ffp-contract=off),
// so results are reproducible op-for-op against the CPU oracle.
//
// Reference functions restated (paths relative to the reference root):
//   pyramid.py:9-28   Gaussian / Laplacian image pyramid (cv2.pyrDown / cv2.pyrUp)
//   pyramid.py:51-69  collapse
//   transforms.py:82-102  temporal FFT band-pass (as the explicit linear operator M)
//   transforms.py:184-192 global min/max mask
//   base.py:562-566   time average, normalise, float_to_uint8, threshold
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <chrono>

// Non-template kernels are function templates with one defaulted parameter (launched as k_name<>): every translation unit of the
// library includes every kernel header, and a unit generates device code only for the kernels it launches.
#define RM_KERNEL template <int RM_UNIT_ = 0> __global__
// N-element vector type (clang spells it ext_vector_type, the g++ of the tests' host emulation vector_size): MFMA accumulators, 16-byte loads
#ifdef RM_HIPEMU
#define RM_VEC(T, N) T __attribute__((vector_size(sizeof(T) * (N))))
#else
#define RM_VEC(T, N) T __attribute__((ext_vector_type(N)))
#endif

namespace rm {

// Host wait for a result on `s`.  hipStreamSynchronize sleeps on an interrupt (tens of microseconds to wake up);
// the calls that return host results are latency-critical (one per locate(), one per measured frame), so poll
// first and only fall back to the blocking wait when the stream is still busy after a few milliseconds.
inline hipError_t stream_wait(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) return hipStreamSynchronize(s);
    }
}

// ... for the work in front of an event (rm_locate_result: the stream already carries the next submission)
inline hipError_t event_wait(hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) return hipEventSynchronize(ev);
    }
}


// ----------------------------------------------------------------------------------------
// Workgroup timeline tracing -- DEVELOPER BUILD ONLY (make librespmon_hip_trace.so, tools/trace_tail.py).  In the product
// library RM_TRACE is undefined and both macros expand to nothing.  A traced workgroup records the 100 MHz wall clock at
// entry and exit plus the CU it ran on, so launch ramps, imbalance and per-workgroup latency can be read off per kernel.
// ----------------------------------------------------------------------------------------
#ifdef RM_TRACE
struct TraceRec { unsigned long long t0, t1, c0, c1; unsigned int hwid, xcc; };   // t: 100 MHz wall clock, c: shader clock (s_memtime)
constexpr int TRACE_KERNELS = 16, TRACE_BLOCKS = 20480;
__device__ TraceRec *g_trace_buf = nullptr;
__device__ __forceinline__ void trace_end(int kid, unsigned long long t0, unsigned long long c0)
{
    if (threadIdx.x != 0 || !g_trace_buf) return;
    const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
    if (b >= (unsigned)TRACE_BLOCKS) return;
    TraceRec &r = g_trace_buf[(size_t)kid * TRACE_BLOCKS + b];
    r.t0 = t0; r.t1 = wall_clock64(); r.c0 = c0; r.c1 = clock64();
    r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
}
struct TraceScope {
    int kid; unsigned long long t0, c0;
    __device__ __forceinline__ TraceScope(int k) : kid(k), t0(wall_clock64()), c0(clock64()) {}
    __device__ __forceinline__ ~TraceScope() { trace_end(kid, t0, c0); }
};
#define RM_TRACE_SCOPE(KID) TraceScope rm_trace_scope_(KID)
// phase marks inside ONE workgroup (block 5) of a kernel: mark i = wall clock when thread 0 passes the call; they live in
// the records of the last pseudo-kernel (TRACE_KERNELS - 1), field t0 of record kid * TRACE_MARKS + i
constexpr int TRACE_MARKS = 16;
__device__ __forceinline__ void trace_mark(int kid, int i)
{
    if (threadIdx.x == 0 && blockIdx.x == 5 && blockIdx.y == 0 && g_trace_buf)
        g_trace_buf[(size_t)(TRACE_KERNELS - 1) * TRACE_BLOCKS + kid * TRACE_MARKS + i].t0 = wall_clock64();
}
#define RM_TRACE_MARK(KID, I) trace_mark(KID, I)
#else
#define RM_TRACE_SCOPE(KID)
#define RM_TRACE_MARK(KID, I)
#endif

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

// The band-passed video is EVEN in time.  transforms.py:98 takes Re(ifft(.)) of a REAL (packed-rfft) array, so the filter output
// satisfies out[s] == out[n - s] for 0 < s < n -- bit for bit in the reference (scipy's ifft of a real sequence is exactly
// Hermitian) and here (the stage-2 operator is evaluated for s <= n / 2 only, rm_temporal.hip get_operator).  Every per-frame stage
// after the filter is a function of the frame alone, so C_S, the tile bounds and raw inherit the symmetry: the library computes and
// stores the T / 2 + 1 UNIQUE frames and the time-ordered consumers (the masked sum) read frame t through sym_frame().
__host__ __device__ __forceinline__ int sym_frames(int T) { return T / 2 + 1; }
__host__ __device__ __forceinline__ int sym_frame(int t, int T) { return 2 * t <= T ? t : T - t; }
// does the frame range [t0, t1) of a T-frame buffer hold a frame whose unique frame is u?
__host__ __device__ __forceinline__ bool sym_in_range(int u, int T, int t0, int t1)
{
    return (u >= t0 && u < t1) || (u > 0 && T - u >= t0 && T - u < t1);
}

// slot_of[] (which pairs the masked time sum keeps, k_select_pairs) is TILE-major -- [tile][unique frame] -- so that the sum kernels,
// which walk the frames of one tile, find them in a few cache lines (frame-major: one line per frame, 129 scattered lines per tile
// and reader at 1080p x 256); the pair index of the lists and bounds stays u * ntiles + tile
__host__ __device__ __forceinline__ size_t slot_index(int u, int tile, int Th) { return (size_t)tile * Th + u; }

// widening loads; uint8 applies uint8_to_float's  k * (1./255)  (transforms.py:20-23)
__device__ __forceinline__ double load_px(const uint8_t *p, size_t i) { return (double)p[i] * (1.0 / 255); }
__device__ __forceinline__ double load_px(const __half *p, size_t i) { return (double)__half2float(p[i]); }
__device__ __forceinline__ double load_px(const float *p, size_t i) { return (double)p[i]; }
__device__ __forceinline__ double load_px(const double *p, size_t i) { return p[i]; }

// order-preserving map double -> uint64 so that atomicMin/atomicMax on the key orders doubles
__device__ __forceinline__ unsigned long long f64_key(double d)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unkey(unsigned long long k)
{
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// float_to_uint8 (transforms.py:26-29): v*255 then C cast (truncate toward zero, low byte;
// NaN / |v| >= 2^31 -> 0, the x86 cvttsd2si result numpy produces)
__device__ __forceinline__ uint8_t f64_to_u8_trunc(double v255)
{
    if (!(v255 == v255) || v255 >= 2147483648.0 || v255 <= -2147483649.0) return 0;
    return (uint8_t)((int)v255 & 255);
}

// Wave (64-lane) reductions; the result is valid in every lane.
// On the GPU they run on DPP row shifts / broadcasts (VALU speed): a __shfl_xor butterfly compiles to ds_bpermute, i.e. six
// dependent LDS round trips per 32-bit word -- the latency-bound tail kernels spent 1-3 us each in their folds and extrema.
//   row_shr:1,2,4,8 leave the reduction of each 16-lane row in its last lane (a lane without a source keeps its own value:
//   harmless for min / max), row_bcast:15 folds rows 0->1 and 2->3, row_bcast:31 folds the lower half into lane 63.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v)
{
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)(unsigned)v), hi = dpp_i32<CTRL, ROW_MASK>((int)(unsigned)(v >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long bits_of(double d) { return (unsigned long long)__double_as_longlong(d); }
__device__ __forceinline__ unsigned long long bits_of(unsigned long long d) { return d; }
__device__ __forceinline__ void from_bits(unsigned long long b, double &d) { d = __longlong_as_double((long long)b); }
__device__ __forceinline__ void from_bits(unsigned long long b, unsigned long long &d) { d = b; }
template <int CTRL, int ROW_MASK, typename T> __device__ __forceinline__ T dpp_get(T v)
{
    T o;
    from_bits(dpp_u64<CTRL, ROW_MASK>(bits_of(v)), o);
    return o;
}
template <typename T, typename Op> __device__ __forceinline__ T wave_reduce(T v, Op op)   // T: double or unsigned long long
{
    v = op(dpp_get<0x111, 0xF>(v), v);   // row_shr:1
    v = op(dpp_get<0x112, 0xF>(v), v);   // row_shr:2
    v = op(dpp_get<0x114, 0xF>(v), v);   // row_shr:4
    v = op(dpp_get<0x118, 0xF>(v), v);   // row_shr:8
    v = op(dpp_get<0x142, 0xA>(v), v);   // row_bcast:15 into rows 1 and 3
    v = op(dpp_get<0x143, 0xC>(v), v);   // row_bcast:31 into rows 2 and 3
    const unsigned long long b = bits_of(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    T r;
    from_bits(((unsigned long long)hi << 32) | lo, r);
    return r;
}
// arg-reduction: the value and the position that holds it travel together
template <typename Less> __device__ __forceinline__ void wave_arg_reduce(double &v, int &idx, Less less)
{
#define RM_ARG_STEP(CTRL, MASK)                                                  \
    {                                                                            \
        const double ov = dpp_get<CTRL, MASK>(v);                                \
        const int oi = dpp_i32<CTRL, MASK>(idx);                                 \
        if (less(ov, v)) { v = ov; idx = oi; }                                   \
    }
    RM_ARG_STEP(0x111, 0xF) RM_ARG_STEP(0x112, 0xF) RM_ARG_STEP(0x114, 0xF) RM_ARG_STEP(0x118, 0xF)
    RM_ARG_STEP(0x142, 0xA) RM_ARG_STEP(0x143, 0xC)
#undef RM_ARG_STEP
    const unsigned long long b = bits_of(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    from_bits(((unsigned long long)hi << 32) | lo, v);
    idx = __builtin_amdgcn_readlane(idx, 63);
}
// min / max of two doubles as ONE instruction.  __builtin_fmin / fmax put a canonicalising v_max_f64 x, x, x in front of every operand
// the compiler cannot prove quiet (loads, DPP moves, loop-carried values): three instructions instead of one in kernels that do
// little else (rm_bounds_l1.h).  The operands here are never signalling NaNs (arithmetic results and loaded image data); a quiet NaN
// operand yields the other operand, as fmin / fmax do.
__device__ __forceinline__ double f64_min(double a, double b)
{
#ifndef RM_HIPEMU
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fmin(a, b);
#endif
}
__device__ __forceinline__ double f64_max(double a, double b)
{
#ifndef RM_HIPEMU
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fmax(a, b);
#endif
}

__device__ __forceinline__ double wave_min(double v) { return wave_reduce(v, [](double o, double w) { return (o < w) ? o : w; }); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce(v, [](double o, double w) { return (o > w) ? o : w; }); }

// ----------------------------------------------------------------------------------------
// K1  cv2.pyrDown (pyramid.py:14) on every frame: [T,h,w] Tin -> [T,dh,dw] f64
//     LDS-staged tile; horizontal 5-tap first, vertical second (OpenCV order).
// ----------------------------------------------------------------------------------------
constexpr int PD_TY = 8, PD_TX = 64;
constexpr int PD_SY = 2 * PD_TY + 3, PD_SX = 2 * PD_TX + 3;

template <typename Tin>
__global__ __launch_bounds__(256) void k_pyr_down(const Tin *src, int h, int w, size_t frame_stride,
                                                  double *dst, int dh, int dw)
{
    __shared__ double s_src[PD_SY][PD_SX + 1];
    __shared__ double s_row[PD_SY][PD_TX];
    const int tid = threadIdx.x;
    const int t = blockIdx.z, ty0 = blockIdx.y * PD_TY, tx0 = blockIdx.x * PD_TX;
    const Tin *sp = src + (size_t)t * frame_stride;
    for (int i = tid; i < PD_SY * PD_SX; i += 256) {
        int r = i / PD_SX, c = i - r * PD_SX;
        int sy = reflect101(2 * ty0 - 2 + r, h), sx = reflect101(2 * tx0 - 2 + c, w);
        s_src[r][c] = load_px(sp, (size_t)sy * w + sx);
    }
    __syncthreads();
    for (int i = tid; i < PD_SY * PD_TX; i += 256) {
        int r = i / PD_TX, x = i - r * PD_TX;
        const double *s = &s_src[r][2 * x];
        s_row[r][x] = s[2] * 6 + (s[1] + s[3]) * 4 + s[0] + s[4];
    }
    __syncthreads();
    for (int i = tid; i < PD_TY * PD_TX; i += 256) {
        int y = i / PD_TX, x = i - y * PD_TX;
        int oy = ty0 + y, ox = tx0 + x;
        if (oy < dh && ox < dw) {
            int r = 2 * y;
            double v = (s_row[r + 2][x] * 6 + (s_row[r + 1][x] + s_row[r + 3][x]) * 4 + s_row[r][x] + s_row[r + 4][x]) *
                       (1.0 / 256);
            dst[((size_t)t * dh + oy) * dw + ox] = v;
        }
    }
}

// ----------------------------------------------------------------------------------------
// K2  cv2.pyrUp with explicit dstsize (pyramid.py:25-26, 55), SURVEY B2.
//     `up_at` evaluates one output pixel from a source image addressed through a functor so the
//     same code serves global memory (materialising kernels) and LDS tiles (fused collapse).
// ----------------------------------------------------------------------------------------
// horizontal value of source row r at destination column x (unnormalised, x8 kernel)
template <typename Src>
__device__ __forceinline__ double up_h(const Src &s, int r, int x, int sw)
{
    if (sw == 1) return s(r, 0) * 8;
    int j = x >> 1;
    if (x & 1) {
        if (j == sw - 1) return s(r, j) * 8;
        return (s(r, j) + s(r, j + 1)) * 4;
    }
    if (j == 0) return s(r, 0) * 6 + s(r, 1) * 2;
    if (j == sw - 1) return s(r, j - 1) + s(r, j) * 7;
    return s(r, j - 1) + s(r, j) * 6 + s(r, j + 1);
}

template <typename Src>
__device__ __forceinline__ double up_at(const Src &s, int y, int x, int sh, int sw)
{
    int i = y >> 1;
    if (y & 1) {
        int r2 = (i == sh - 1) ? i : i + 1;
        return ((up_h(s, i, x, sw) + up_h(s, r2, x, sw)) * 4) * (1.0 / 64);
    }
    int r0 = (i == 0) ? (sh > 1 ? 1 : 0) : i - 1;
    int r2 = (i == sh - 1) ? i : i + 1;
    return (up_h(s, r0, x, sw) + up_h(s, i, x, sw) * 6 + up_h(s, r2, x, sw)) * (1.0 / 64);
}

struct GlobalImg {
    const double *p; int w;
    __device__ __forceinline__ double operator()(int r, int c) const { return p[(size_t)r * w + c]; }
};

// mode 0: dst = up(src); 1: dst = other - up(src); 2: dst = up(src) + other
// src_fs / dst_fs / other_fs: frame strides in doubles (frames of several levels may share one [T, NP] buffer)
RM_KERNEL __launch_bounds__(256) void k_pyr_up(const double *src, int sh, int sw, size_t src_fs, double *dst, int dh, int dw,
                                                size_t dst_fs, int mode, const double *other, size_t other_fs)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int t = blockIdx.z;
    if (x >= dw || y >= dh) return;
    GlobalImg s{src + (size_t)t * src_fs, sw};
    double u = up_at(s, y, x, sh, sw);
    size_t o = (size_t)y * dw + x;
    if (mode == 1) u = other[(size_t)t * other_fs + o] - u;
    else if (mode == 2) u = u + other[(size_t)t * other_fs + o];
    dst[(size_t)t * dst_fs + o] = u;
}

// The same for large levels: a thread produces the 2 x 2 outputs that hang under source pixel (i, j).  Their taps all lie
// in its 3 x 3 neighbourhood, so the four up_at() calls share 9 loads (instead of 9 + 6 + 6 + 4 for four threads), the
// address arithmetic is paid once, and a lane stores 16 contiguous bytes per row.  Same expressions per output, same bits.
RM_KERNEL __launch_bounds__(256) void k_pyr_up_2x2(const double *__restrict__ src, int sh, int sw, size_t src_fs,
                                                    double *dst, int dh, int dw, size_t dst_fs, int mode,
                                                    const double *other, size_t other_fs)
{
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = 2 * (blockIdx.y * 4 + (threadIdx.x >> 6));
    const int t = blockIdx.z;
    if (x >= dw || y >= dh) return;
    const bool x1 = x + 1 < dw, y1 = y + 1 < dh;
    GlobalImg s{src + (size_t)t * src_fs, sw};
    double u00 = up_at(s, y, x, sh, sw);
    double u01 = x1 ? up_at(s, y, x + 1, sh, sw) : 0.0;
    double u10 = y1 ? up_at(s, y + 1, x, sh, sw) : 0.0;
    double u11 = (x1 && y1) ? up_at(s, y + 1, x + 1, sh, sw) : 0.0;
    const size_t o0 = (size_t)y * dw + x, o1 = o0 + dw;
    if (mode != 0) {
        const double *op = other + (size_t)t * other_fs;
        const double a00 = op[o0], a01 = x1 ? op[o0 + 1] : 0.0, a10 = y1 ? op[o1] : 0.0, a11 = (x1 && y1) ? op[o1 + 1] : 0.0;
        if (mode == 1) { u00 = a00 - u00; u01 = a01 - u01; u10 = a10 - u10; u11 = a11 - u11; }
        else { u00 = u00 + a00; u01 = u01 + a01; u10 = u10 + a10; u11 = u11 + a11; }
    }
    double *dp = dst + (size_t)t * dst_fs;
    dp[o0] = u00;
    if (x1) dp[o0 + 1] = u01;
    if (y1) { dp[o1] = u10; if (x1) dp[o1 + 1] = u11; }
}

// ----------------------------------------------------------------------------------------
// K5-K8  temporal band-pass (transforms.py:82-102): packed rfft -> index mask -> Re(ifft) -> *amp,
//        a fixed real linear operator along T, applied in its two-stage form.
// ----------------------------------------------------------------------------------------
// Two-stage form (the reference's own rfft -> mask -> ifft order; ~T/(2*nk) times cheaper than the dense
// T x T product M = C R that rm_temporal_operator() exports for inspection):
//   stage 1 (packed real FFT rows that survive the mask):  y[k,p]   = sum_t R[k,t] x[t,p]          k < nk
//   stage 2 (Re(ifft) of the packed array, then *amp):     out[s,p] = amp * sum_k C[s,k] y[k,p]    s < T
// One single-wave workgroup = 64 pixels x KC (resp. SC) outputs; coefficient chunks are staged in LDS and
// read as broadcasts; the pixel loads are issued U deep.  All levels of the small pyramid sit side by side
// in one [T, NP] buffer, so one launch per stage serves every filtered level.
constexpr int TF_KC = 4, TF_SC = 8, TF_U = 16;

// Measured and rejected (1080p x 256, both stages 0.064 ms as written): splitting T over 4 waves per workgroup with an
// LDS reduction (stage 1 48 us vs 42 us); splitting T over 2 / 4 workgroups with partial y buffers (+11 / +56 us);
// 8 / 12 / 16 rows of R per workgroup instead of 4, i.e. fewer re-reads of x through L2 but fewer waves (+10 / +25 /
// +34 us); 16 output rows per workgroup in stage 2 (no change);
// one fused kernel per 64 pixel columns that reads x once, keeps all y[k] in registers and takes the
// coefficients through the scalar cache (128 us vs 64 us for both stages: one workgroup per CU exposes every
// scalar-load and global-load latency, the two-stage form has 8-20 waves per CU to hide them).
__device__ void state_init_lane(struct CollapseState *st, int i);   // defined with the state, below

// st_init (nullable): workgroup (0, 0) also resets the reduction state of the collapse passes that follow on the stream
RM_KERNEL __launch_bounds__(64) void k_temporal_fwd(const double *x, int T, size_t NP, const double *R, int nk, double *y, struct CollapseState *st_init)
{
    HIP_DYNAMIC_SHARED(double, s_r)  // [T][TF_KC]
    if (st_init && blockIdx.x == 0 && blockIdx.y == 0) state_init_lane(st_init, (int)threadIdx.x);
    const int k0 = blockIdx.y * TF_KC;
    for (int i = threadIdx.x; i < T * TF_KC; i += 64) {
        int t = i / TF_KC, k = i - t * TF_KC;
        s_r[i] = (k0 + k < nk) ? R[(size_t)(k0 + k) * T + t] : 0.0;
    }
    __syncthreads();
    size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= NP) return;
    double acc[TF_KC];
#pragma unroll
    for (int k = 0; k < TF_KC; ++k) acc[k] = 0.0;
    for (int t0 = 0; t0 < T; t0 += TF_U) {
        double v[TF_U];
#pragma unroll
        for (int u = 0; u < TF_U; ++u) v[u] = (t0 + u < T) ? x[(size_t)(t0 + u) * NP + p] : 0.0;
#pragma unroll
        for (int u = 0; u < TF_U; ++u) {
            if (t0 + u < T) {
                const double *r = &s_r[(t0 + u) * TF_KC];
#pragma unroll
                for (int k = 0; k < TF_KC; ++k) acc[k] = acc[k] + r[k] * v[u];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < TF_KC; ++k)
        if (k0 + k < nk) y[(size_t)(k0 + k) * NP + p] = acc[k];
}

// T = rows of C / frames written (the unique frames: sym_frames(n)); mirror_n > 0: also store row s as row mirror_n - s
// (0 < s, 2 s < mirror_n) -- the full [n, NP] array of the module-level filter call
RM_KERNEL __launch_bounds__(64) void k_temporal_inv(const double *y, int nk, size_t NP, const double *C, int T, double amp,
                                                     double *out, int mirror_n)
{
    HIP_DYNAMIC_SHARED(double, s_c)  // [nk][TF_SC]
    const int s0 = blockIdx.y * TF_SC;
    for (int i = threadIdx.x; i < nk * TF_SC; i += 64) {
        int k = i / TF_SC, j = i - k * TF_SC;
        s_c[i] = (s0 + j < T) ? C[(size_t)(s0 + j) * nk + k] : 0.0;
    }
    __syncthreads();
    size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= NP) return;
    double acc[TF_SC];
#pragma unroll
    for (int j = 0; j < TF_SC; ++j) acc[j] = 0.0;
    for (int k0 = 0; k0 < nk; k0 += TF_U) {
        double v[TF_U];
#pragma unroll
        for (int u = 0; u < TF_U; ++u) v[u] = (k0 + u < nk) ? y[(size_t)(k0 + u) * NP + p] : 0.0;
#pragma unroll
        for (int u = 0; u < TF_U; ++u) {
            if (k0 + u < nk) {
                const double *c = &s_c[(k0 + u) * TF_SC];
#pragma unroll
                for (int j = 0; j < TF_SC; ++j) acc[j] = acc[j] + c[j] * v[u];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TF_SC; ++j)
        if (s0 + j < T) {
            const int sr = s0 + j;
            const double v = acc[j] * amp;
            out[(size_t)sr * NP + p] = v;
            if (mirror_n > 0 && sr > 0 && 2 * sr < mirror_n) out[(size_t)(mirror_n - sr) * NP + p] = v;
        }
}

// Matrix-core form of the two stages (even n, at most 48 merged rows of either symmetry class).
// The band-pass IS a dense contraction along T -- z = Rz x, out = amp * Cz z -- and the only place on this path where MFMA fits.
// v_mfma_f64_16x16x4_f64 runs at the fp64 vector rate on gfx950, so what counts is the number of products and operand reuse:
//   * merged rows (host, get_operator): packed indices k and n - k multiply the same inverse column cos(2 pi k s / n), so their
//     forward rows are added once on the host: about half the rows of R and the columns of C;
//   * folded frames: a merged row is a cosine row (even in t) or a sine row (odd in t), never a mix (n even), so
//         z_even = sum_{t <= n/2} Rz[., t] e[t],   e[t] = x[t] + x[n - t]   (x[t] alone for t = 0 and t = n / 2)
//         z_odd  = sum_{t <  n/2} Rz[., t] o[t],   o[t] = x[t] - x[n - t]   (0 there)
//     -- half the K-steps; the tiles of 16 rows are class-pure (NH "even" tiles, then NH "odd" tiles, zero padded);
//   * unique output frames: only s <= n / 2 is produced (sym_frames): half the products of stage 2.
// 6 x fewer products than the plain two-stage form at n = 256 / 512.  A workgroup owns 16 pixel columns and reads its x[T, 16] tile
// ONCE, keeps z in 2 NH accumulator tiles and feeds them straight back as the B operands of the second product -- the D layout of
// the first product (row = (lane >> 4) + 4 * reg, col = lane & 15) is exactly the B layout the second one needs for K-step
// (tile, reg) -- so z never leaves registers.  The W wavefronts of a workgroup share the 16 columns: wave w contracts every W-th
// K-step of stage 1 (the partial z tiles meet in LDS, summed in wave order) and produces every W-th tile of 16 output frames.
// The operators arrive "fragment major" (built on the host), so that every A operand is one coalesced 512-byte load:
//   Rf[(ks * 2 NH + q) * 64 + lane] = Rz[row(q, lane & 15)][4 ks + (lane >> 4)]
//   Cf[(m * 8 NH + 4 q + r) * 64 + lane] = Cz[16 m + (lane & 15)][row(q, 4 r + (lane >> 4))]
// fused == materialised == per-level stays bit for bit: every path through the library uses this same kernel for a given n.
constexpr int TM_W = 4;            // waves per workgroup
constexpr int TM_MAX_HALF = 3;     // up to 48 merged rows per symmetry class (n = 1024 at 10 fps has 47 + 47)

typedef RM_VEC(double, 4) v4f64;

// mirror_n > 0: also store output frame s as frame mirror_n - s (the full [n, NP] array of the module-level filter call)
template <int NH>
__global__ __launch_bounds__(64 * TM_W, NH == 1 ? 3 : 2) void k_temporal_sym(const double *__restrict__ x, int T, size_t NP, const double *__restrict__ Rf,
                                                             const double *__restrict__ Cf, double amp, double *__restrict__ out, int mirror_n,
                                                             struct CollapseState *st_init)
{
    RM_TRACE_SCOPE(2);
    if (st_init && blockIdx.x == 0 && threadIdx.x < 64) state_init_lane(st_init, (int)threadIdx.x);
    constexpr int NT = 2 * NH;
    __shared__ double s_y[TM_W][4 * NT][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lo = lane & 15, hi = lane >> 4;
    const size_t p = (size_t)blockIdx.x * 16 + lo;
    const size_t pc = p < NP ? p : NP - 1;     // columns past the end repeat the last one (never stored)
    const int Th = sym_frames(T), nks = (Th + 3) >> 2;
    v4f64 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (v4f64){0.0, 0.0, 0.0, 0.0};
    RM_TRACE_MARK(2, 0);
    // x comes from HBM (one round trip ~2 us under load), the operator fragments from L2: a chunk of up to TM_XPF K-steps has ALL
    // its x operands requested up front (two doubles per K-step and lane), then the K-steps run in batches of TM_U whose operator
    // fragments are requested together.  A loop that loads one step's operands, waits and multiplies is a chain of round trips, and
    // at 2-3 waves per SIMD nothing hides them (4K x 512: 2.75 -> 1.16 ms with the symmetric operator, -> this).
    // The order of the products into each accumulator is fixed: K-steps wave, wave + W, ... in increasing order.
    constexpr int TM_U = NH == 1 ? 8 : 4;
    constexpr int TM_XPF = NH == 3 ? 8 : 16;
    for (int kc = wave; kc < nks; kc += TM_W * TM_XPF) {
        double xa[TM_XPF], xb[TM_XPF];
#pragma unroll
        for (int i = 0; i < TM_XPF; ++i) {
            const int ks = kc + i * TM_W;
            if (ks < nks) {   // (wave-uniform)
                const int t = 4 * ks + hi, tc = t < Th ? t : Th - 1, tp = tc == 0 ? 0 : T - tc;
                xa[i] = x[(size_t)tc * NP + pc];
                xb[i] = x[(size_t)tp * NP + pc];
            }
        }
#pragma unroll
        for (int i0 = 0; i0 < TM_XPF; i0 += TM_U) {
            if (kc + i0 * TM_W >= nks) break;   // (wave-uniform)
            double rr[TM_U][NT];
#pragma unroll
            for (int u = 0; u < TM_U; ++u) {
                const int ks = kc + (i0 + u) * TM_W;
                if (ks < nks) {
                    const double *rf = Rf + (size_t)ks * NT * 64 + lane;
#pragma unroll
                    for (int q = 0; q < NT; ++q) rr[u][q] = rf[q * 64];
                }
            }
#pragma unroll
            for (int u = 0; u < TM_U; ++u) {
                const int ks = kc + (i0 + u) * TM_W;
                if (ks < nks) {
                    const int t = 4 * ks + hi;
                    const bool self = t == 0 || 2 * t == T, valid = t < Th;
                    double e = self ? xa[i0 + u] : xa[i0 + u] + xb[i0 + u], o = self ? 0.0 : xa[i0 + u] - xb[i0 + u];
                    if (!valid) { e = 0.0; o = 0.0; }
#pragma unroll
                    for (int q = 0; q < NH; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[u][q], e, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = NH; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[u][q], o, acc[q], 0, 0, 0);
                }
            }
        }
        RM_TRACE_MARK(2, 8 + (kc - wave) / (TM_W * TM_XPF));
    }
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_y[wave][4 * q + r][lane] = acc[q][r];
    RM_TRACE_MARK(2, 1);
    const int mt = (Th + 15) >> 4;             // output tiles of 16 frames, dealt round-robin to the waves
    // the A operands of this wave's first output tile travel while the partial z tiles meet in LDS
    double cfv[4 * NT];
    if (wave < mt) {
        const double *cf = Cf + (size_t)wave * 4 * NT * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) cfv[q] = cf[q * 64];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = s_y[0][4 * q + r][lane];
#pragma unroll
            for (int w = 1; w < TM_W; ++w) v = v + s_y[w][4 * q + r][lane];
            acc[q][r] = v;
        }
    RM_TRACE_MARK(2, 2);
    for (int m = wave; m < mt; m += TM_W) {
        const int s0 = 16 * m;
        v4f64 o = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f64_16x16x4f64(cfv[4 * q + r], acc[q][r], o, 0, 0, 0);
        }
        if (m + TM_W < mt) {   // the next tile's operands travel while this one is stored
            const double *cf = Cf + (size_t)(m + TM_W) * 4 * NT * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4 * NT; ++q) cfv[q] = cf[q * 64];
        }
        if (p < NP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sr = s0 + hi + 4 * r;
                if (sr < Th) {
                    const double v = o[r] * amp;
                    out[(size_t)sr * NP + p] = v;
                    if (mirror_n > 0 && sr > 0 && 2 * sr < mirror_n) out[(size_t)(mirror_n - sr) * NP + p] = v;
                }
            }
        }
    }
    RM_TRACE_MARK(2, 3);
}

// The same products for LARGE levels (4K x 512, skip 2: 518 400 pixels per frame, 2.1 GB in, 1.07 GB out): throughput, not
// latency, is what counts there, and k_temporal_sym's K-split costs it an LDS exchange plus a barrier per 16 pixels and a
// frontier of only 128 contiguous bytes per frame and workgroup in DRAM.  Here a WAVE owns 16 pixel columns for the whole
// contraction (the four waves of a workgroup sit on adjacent columns: 512 contiguous bytes per frame) and x streams through two
// register buffers of TP_XC K-steps (the next chunk is requested before the current one is multiplied).
// Round 5: the operator fragments travel through LDS, fetched ONCE per workgroup.  Every wave needs all of Rf and Cf (272 KB at T = 512)
// for its 16 pixels; with each wave loading them itself the CU's vector memory path moved 3 KB per K-step and wave against 256 MFMA
// cycles per SIMD -- PMC at 4K x 512: TA busy 69 %, MFMA pipe 39 %, waves 70 % in issue stalls, 14 % in s_waitcnt.  Now the workgroup's 256
// threads copy a chunk of TP_XC K-steps (stage 2: one output tile) into one of two LDS buffers with 16-byte loads while the previous
// chunk is multiplied, one barrier per chunk, and the waves read their A operands with conflict-free ds_read_b64.
// The products into each accumulator happen in the same order as in k_temporal_sym?  No: there the partial sums of the four K-phases
// are added in wave order -- here K runs straight through.  The two kernels agree to rounding (~1e-16), and a given (T, level size)
// always takes the same one.
template <int NH>
__global__ __launch_bounds__(256, 2) void k_temporal_sym_px(const double *__restrict__ x, int T, size_t NP, const double *__restrict__ Rf,
                                                            const double *__restrict__ Cf, double amp, double *__restrict__ out, int mirror_n,
                                                            struct CollapseState *st_init)
{
    if (st_init && blockIdx.x == 0 && threadIdx.x < 64) state_init_lane(st_init, (int)threadIdx.x);
    constexpr int NT = 2 * NH;
    constexpr int TP_XC = 8;                       // K-steps per chunk (x registers and operator fragments alike)
    constexpr int RCH = TP_XC * NT * 64;           // doubles of one stage-1 fragment chunk (NH = 2: 16 KB)
    constexpr int CCH = 4 * NT * 64;               // doubles of one output tile's stage-2 fragments (NH = 2: 8 KB)
    constexpr int RL = RCH / 512, CL = CCH / 512;  // 16-byte loads per thread and chunk
    __shared__ double s_frag[2][RCH];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6)), lo = lane & 15, hi = lane >> 4;
    const size_t p = ((size_t)blockIdx.x * 4 + wave) * 16 + lo;   // (waves past NP multiply a clamped column and store nothing: they keep the barriers)
    const size_t pc = p < NP ? p : NP - 1;
    const int Th = sym_frames(T), nks = (Th + 3) >> 2;
    const int nchunks = (nks + TP_XC - 1) / TP_XC;
    v4f64 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double xa[2][TP_XC], xb[2][TP_XC];
    auto load_x = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TP_XC; ++i) {
            const int ks = k0 + i;
            const int t = 4 * (ks < nks ? ks : nks - 1) + hi, tc = t < Th ? t : Th - 1, tp = tc == 0 ? 0 : T - tc;
            xa[buf][i] = x[(size_t)tc * NP + pc];
            xb[buf][i] = x[(size_t)tp * NP + pc];
        }
    };
    // this thread's share of a fragment chunk: global -> registers (in flight while the previous chunk is multiplied) -> LDS
    typedef RM_VEC(double, 2) v2f64;
    const size_t r_last = (size_t)nks * NT * 64 - 2;              // (chunks are whole TP_XC K-steps: the last one reads clamped, unused values)
    v2f64 gl[RL];
    auto fetch_r = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            size_t i = (size_t)c * RCH + 2 * tid + 512 * j;
            i = i < r_last ? i : r_last;
            gl[j] = *reinterpret_cast<const v2f64 *>(Rf + i);
        }
    };
    auto stash_r = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RL; ++j) *reinterpret_cast<v2f64 *>(&s_frag[buf][2 * tid + 512 * j]) = gl[j];
    };
    auto products = [&](int xbuf, int c) __attribute__((always_inline)) {
        const double *sr = &s_frag[c & 1][lane];
#pragma unroll
        for (int u = 0; u < TP_XC; ++u) {
            const int ks = c * TP_XC + u;
            if (ks < nks) {   // (uniform)
                double rr[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) rr[q] = sr[(u * NT + q) * 64];
                const int t = 4 * ks + hi;
                const bool self = t == 0 || 2 * t == T, valid = t < Th;
                double e = self ? xa[xbuf][u] : xa[xbuf][u] + xb[xbuf][u], o = self ? 0.0 : xa[xbuf][u] - xb[xbuf][u];
                if (!valid) { e = 0.0; o = 0.0; }
#pragma unroll
                for (int q = 0; q < NH; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[q], e, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = NH; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[q], o, acc[q], 0, 0, 0);
            }
        }
    };
    fetch_r(0);
    load_x(0, 0);
    stash_r(0);
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += 2) {      // two chunks per trip: the x register buffers are static
        if (c0 + 1 < nchunks) fetch_r(c0 + 1);
        load_x(1, (c0 + 1) * TP_XC);
        products(0, c0);
        if (c0 + 1 < nchunks) stash_r(1);          // (every wave left buffer 1 before the barrier that ended the previous chunk)
        __syncthreads();
        if (c0 + 1 >= nchunks) break;              // (uniform)
        if (c0 + 2 < nchunks) fetch_r(c0 + 2);
        load_x(0, (c0 + 2) * TP_XC);
        products(1, c0 + 1);
        if (c0 + 2 < nchunks) stash_r(0);
        __syncthreads();
    }
    // stage 2: out tile m (16 unique frames) = Cz[m] z, the A operands of a tile through the same two LDS buffers
    const int mt = (Th + 15) >> 4;
    v2f64 gc[CL];
    auto fetch_c = [&](int m) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CL; ++j) gc[j] = *reinterpret_cast<const v2f64 *>(Cf + (size_t)m * CCH + 2 * tid + 512 * j);
    };
    auto stash_c = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CL; ++j) *reinterpret_cast<v2f64 *>(&s_frag[buf][2 * tid + 512 * j]) = gc[j];
    };
    fetch_c(0);
    stash_c(0);                                    // (the barrier that ended the last chunk of stage 1 freed both buffers)
    __syncthreads();
    for (int m = 0; m < mt; ++m) {
        if (m + 1 < mt) fetch_c(m + 1);            // the next tile's operands travel while this one is multiplied and stored
        const double *sc = &s_frag[m & 1][lane];
        const int s0 = 16 * m;
        v4f64 o = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f64_16x16x4f64(sc[(4 * q + r) * 64], acc[q][r], o, 0, 0, 0);
        }
        if (p < NP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sr = s0 + hi + 4 * r;
                if (sr < Th) {
                    const double v = o[r] * amp;
                    out[(size_t)sr * NP + p] = v;
                    if (mirror_n > 0 && sr > 0 && 2 * sr < mirror_n) out[(size_t)(mirror_n - sr) * NP + p] = v;
                }
            }
        }
        if (m + 1 < mt) stash_c((m + 1) & 1);
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------
// transforms.py:72-79 temporal_bandpass_filter (the IIR alternative to the FFT filter, selectable through
// eulerian_magnification_bandpass(temporal_filter_function=...)): scipy.signal.lfilter(b, a, data, axis=0) * amp.
// One lane per pixel column, the recurrence runs sequentially in t in scipy's transposed direct form II
//   y = z[0] + b[0] x ;  z[i] = z[i+1] + b[i+1] x - a[i+1] y ;  z[n-2] = b[n-1] x - a[n-1] y
// (b, a already divided by a[0], as scipy does), so every value is the same sequence of float64 operations.
// Coefficients are wave-uniform (constant memory through the kernel argument), loads are coalesced across pixels.
// ----------------------------------------------------------------------------------------
constexpr int IIR_MAX = 16;  // coefficients per polynomial (a band-pass of order 6 has 13)
struct IirCoef { double b[IIR_MAX], a[IIR_MAX]; int n; };

RM_KERNEL __launch_bounds__(64) void k_lfilter(const double *x, int T, size_t NP, IirCoef c, double scale, double *y)
{
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= NP) return;
    double z[IIR_MAX];
#pragma unroll
    for (int i = 0; i < IIR_MAX; ++i) z[i] = 0.0;
    double nxt = x[p];
    for (int t = 0; t < T; ++t) {
        const double v = nxt;
        if (t + 1 < T) nxt = x[(size_t)(t + 1) * NP + p];   // one sample ahead of the recurrence
        const double out = z[0] + c.b[0] * v;
#pragma unroll
        for (int i = 0; i < IIR_MAX - 2; ++i)
            if (i < c.n - 2) z[i] = (z[i + 1] + c.b[i + 1] * v) - c.a[i + 1] * out;
        if (c.n >= 2) {
#pragma unroll
            for (int i = 0; i < IIR_MAX - 1; ++i)
                if (i == c.n - 2) z[i] = c.b[i + 1] * v - c.a[i + 1] * out;
        }
        y[(size_t)t * NP + p] = out * scale;
    }
}

// ----------------------------------------------------------------------------------------
// Small pyramid, one workgroup per frame, everything in LDS (levels S..L-1 of a 1080p frame are 87 KB):
//   k_small_pyramid : G_S[t] -> G_{S+1..L-1} (cv2.pyrDown, pyramid.py:14) -> L_l = G_l - pyrUp(G_{l+1})
//                     for l = L-2..S (pyramid.py:23-26), written side by side into lap_all[t, :]
//   k_small_collapse: band-passed levels bp_all[t, :] -> c = bp_{L-2}; c = pyrUp(c) + bp_l for l = L-3..S
//                     (pyramid.py:51-57) -> C_S[t]
// Same per-pixel arithmetic as k_pyr_down / k_pyr_up (bit-identical); they replace ~13 tiny launches.
// ----------------------------------------------------------------------------------------
constexpr int SMALL_MAX_LEVELS = 16;
constexpr int SMALL_NT = 1024;     // one workgroup per frame and per CU: 16 waves hide the LDS / global latencies
struct SmallGeom {
    int S, L;                        // levels S .. L-1 take part
    int h[SMALL_MAX_LEVELS], w[SMALL_MAX_LEVELS];
    int g_off[SMALL_MAX_LEVELS];     // LDS offset (doubles) of Gaussian level l
    int np_off[SMALL_MAX_LEVELS];    // offset of level l inside a [NP] frame of lap_all / bp_all
    int NP;
};

struct CollapseState;
__device__ void state_init_lane(CollapseState *st, int i);   // defined with the state, below

// ---- pyrUp inside LDS, separable and branch-free -------------------------------------------------------------
// up_h() above chooses between five differently shaped expressions per column (interior / left / right border,
// even / odd), which on a wavefront means divergent branches and every shape executed.  The same values, bit for
// bit, come out of ONE shape with per-column operands and weights fixed once per level:
//     h(x) = (A*wa + B*wb) + C*wc
//   even interior  A=s[j-1] B=s[j] C=s[j+1]  w = 1,6,1    == s[j-1] + s[j]*6 + s[j+1]
//   even left      B=s[0]   C=s[1]           w = 0,6,2    == s[0]*6 + s[1]*2          (0 + x is exact)
//   even right     A=s[j-1] B=s[j]           w = 1,7,0    == s[j-1] + s[j]*7          (x + 0 is exact)
//   odd  interior  B=s[j]   C=s[j+1]         w = 0,4,4    == (s[j] + s[j+1])*4        (scaling by 4 commutes with rounding)
//   odd  right / single column  B=C=s[j]     w = 0,4,4    == s[j]*8
// (only the sign of an exact zero can differ, which no later operation observes).  The vertical pass needs no such
// trick: OpenCV's border rules are index clamps there, and row parity is uniform across a wavefront.
struct HTap { int ia, ib, ic; double wa, wb, wc; };

__device__ __forceinline__ HTap make_htap(int x, int sw)   // destination column x of a pyrUp from a source row of width sw
{
    HTap t;
    const int j = x >> 1;
    const bool odd = (x & 1) != 0, single = sw == 1, left = j == 0, right = j == sw - 1;
    const bool four = odd || single;                      // the (B + C) * 4 shapes
    t.ib = j;
    t.ia = (four || left) ? j : j - 1;                   // unused (weight 0) in those shapes: any valid index
    t.ic = (single || right) ? j : j + 1;
    t.wa = (four || left) ? 0.0 : 1.0;
    t.wb = four ? 4.0 : (right ? 7.0 : 6.0);
    t.wc = four ? 4.0 : (left ? 2.0 : (right ? 0.0 : 1.0));
    return t;
}

// whole-image pyrUp inside LDS for the one-workgroup-per-frame kernels: wave = destination row (parity and the
// border rules of the row index are wave-uniform), lane = destination column (its taps and weights fixed once per
// level, make_htap), so there is no index division and no divergent shape.  Same values as up_at(), bit for bit.
// sink(i, v) receives destination element i = y * dw + x.
template <typename Sink>
__device__ __forceinline__ void small_up_level(const double *src, int sh, int sw, int dh, int dw, int tid, Sink &&sink, int y_begin = 0, int y_end = 0x7fffffff)
{
    const int lane = tid & 63, wave = tid >> 6;
    if (y_end > dh) y_end = dh;   // destination rows [y_begin, y_end): all of them by default
    for (int x = lane; x < dw; x += 64) {
        const HTap t = make_htap(x, sw);
        for (int y = y_begin + wave; y < y_end; y += SMALL_NT / 64) {
            const int i = y >> 1;
            const double *ri = src + i * sw, *r2 = src + ((i == sh - 1) ? i : i + 1) * sw;
            const double hi_ = (ri[t.ia] * t.wa + ri[t.ib] * t.wb) + ri[t.ic] * t.wc;
            const double h2 = (r2[t.ia] * t.wa + r2[t.ib] * t.wb) + r2[t.ic] * t.wc;
            double v;
            if (y & 1) {
                v = ((hi_ + h2) * 4) * (1.0 / 64);
            } else {
                const double *r0 = src + ((i == 0) ? (sh > 1 ? 1 : 0) : i - 1) * sw;
                const double h0 = (r0[t.ia] * t.wa + r0[t.ib] * t.wb) + r0[t.ic] * t.wc;
                v = (h0 + hi_ * 6 + h2) * (1.0 / 64);
            }
            sink(y * dw + x, v);
        }
    }
}

// global -> LDS copy by one SMALL_NT-thread workgroup with 8 loads in flight per lane: a plain
// `for (i) lds[i] = src[i]` compiles to load / s_waitcnt vmcnt(0) / ds_write per iteration, i.e. one HBM round trip
// per 8 KB of a frame -- most of the run time of the one-workgroup-per-frame kernels below
__device__ __forceinline__ void fill_lds(double *dst, const double *src, int n, int tid)
{
    constexpr int U = 8;
    for (int base = 0; base < n; base += U * SMALL_NT) {
        double v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + tid + k * SMALL_NT;
            v[k] = (i < n) ? src[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = base + tid + k * SMALL_NT;
            if (i < n) dst[i] = v[k];
        }
    }
}

// st_init (nullable): workgroup 0 also resets the reduction state of the collapse passes that follow on the stream
RM_KERNEL __launch_bounds__(SMALL_NT) void k_small_pyramid(const double *gS, SmallGeom g, double *lap_all, CollapseState *st_init)
{
    RM_TRACE_SCOPE(1);
    if (st_init && blockIdx.x == 0 && threadIdx.x < 64) state_init_lane(st_init, (int)threadIdx.x);
    HIP_DYNAMIC_SHARED(double, lds)
    const int t = blockIdx.x, tid = threadIdx.x;
    const int S = g.S, L = g.L;
    RM_TRACE_MARK(1, 0);
    fill_lds(lds + g.g_off[S], gS + (size_t)t * (g.h[S] * g.w[S]), g.h[S] * g.w[S], tid);
    __syncthreads();
    RM_TRACE_MARK(1, 1);
    for (int l = S + 1; l < L; ++l) {
        const int sh = g.h[l - 1], sw = g.w[l - 1], dh = g.h[l], dw = g.w[l];
        const double *s = lds + g.g_off[l - 1];
        double *d = lds + g.g_off[l];
        for (int x = (tid & 63); x < dw; x += 64) {          // lane = column, wave = row: no index division
            const int c0 = reflect101(2 * x - 2, sw), c1 = reflect101(2 * x - 1, sw), c2 = reflect101(2 * x, sw);
            const int c3 = reflect101(2 * x + 1, sw), c4 = reflect101(2 * x + 2, sw);
            for (int y = (tid >> 6); y < dh; y += SMALL_NT / 64) {
                double r[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const double *row = s + reflect101(2 * y - 2 + k, sh) * sw;
                    r[k] = row[c2] * 6 + (row[c1] + row[c3]) * 4 + row[c0] + row[c4];
                }
                d[y * dw + x] = (r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]) * (1.0 / 256);
            }
        }
        __syncthreads();
        RM_TRACE_MARK(1, 2 + (l - S - 1));
    }
    double *out = lap_all + (size_t)t * g.NP;
    for (int l = L - 2; l >= S; --l) {
        const int dh = g.h[l], dw = g.w[l], sh = g.h[l + 1], sw = g.w[l + 1];
        const double *base = lds + g.g_off[l];
        double *o = out + g.np_off[l];
        small_up_level(lds + g.g_off[l + 1], sh, sw, dh, dw, tid, [&](int i, double v) { o[i] = base[i] - v; });
        RM_TRACE_MARK(1, 8 + (L - 2 - l));
    }
}

// st_init (nullable): workgroup 0 also resets the reduction state of the collapse passes that follow on the stream
// (k_state_init's job: one tiny launch less on the critical path)
RM_KERNEL __launch_bounds__(SMALL_NT) void k_small_collapse(const double *bp_all, SmallGeom g, double *cS, CollapseState *st_init)
{
    HIP_DYNAMIC_SHARED(double, lds)   // the [NP] frame, levels laid out as in bp_all
    const int t = blockIdx.x, tid = threadIdx.x;
    if (st_init && t == 0 && tid < 64) state_init_lane(st_init, tid);
    const int S = g.S, L = g.L;
    fill_lds(lds, bp_all + (size_t)t * g.NP, g.NP, tid);
    __syncthreads();
    for (int l = L - 3; l >= S; --l) {
        const int dh = g.h[l], dw = g.w[l], sh = g.h[l + 1], sw = g.w[l + 1];
        double *d = lds + g.np_off[l];
        small_up_level(lds + g.np_off[l + 1], sh, sw, dh, dw, tid, [&](int i, double v) { d[i] = v + d[i]; });
        __syncthreads();
    }
    const int n = g.h[S] * g.w[S];
    double *o = cS + (size_t)t * n;
    for (int i = tid; i < n; i += SMALL_NT) o[i] = lds[g.np_off[S] + i];
}

// ----------------------------------------------------------------------------------------
// Fused collapse from level S to full resolution (pyramid.py:51-69 for the levels below
// skip_levels_at_top, which are all-zero in the band-passed pyramid: transforms.py:150-160).
//
// One single-wave workgroup owns a CT_W x CT_H tile of the full-resolution frame.  For a frame
// t it stages the level-S footprint of the tile in LDS, runs the pyrUp chain S..1 inside LDS
// and evaluates level 0 in registers (thread = column, marching down CT_H rows so every
// horizontal value is computed once).  No [T,H,W] array is ever written.
// ----------------------------------------------------------------------------------------
constexpr int CT_W = 64, CT_H = 16;
constexpr int MAX_CHAIN = 8;

struct ChainGeom {
    int S;                       // number of pyrUp steps (== skip_levels_at_top), 1..MAX_CHAIN-1
    int h[MAX_CHAIN], w[MAX_CHAIN];  // level sizes, index 0 = full resolution
    int lds_off[MAX_CHAIN];      // offset (doubles) of level k's tile buffer in LDS, k = 1..S
    int lds_hb[MAX_CHAIN];       // offset of the scratch buffer of the horizontal pass of step k -> k-1 (chain_step), k = 2..S
    int lds_total;               // doubles
    int tiles_x, tiles_y;
    double lat_a, lat_b;         // raw[t, y << S, x << S] == lat_a * (c[y-1] + c[y+1]) + lat_b * c[y] per axis (lattice_sample)
};

struct Region { int y0, y1, x0, x1; };  // inclusive

__host__ __device__ __forceinline__ int floordiv2(int a) { return a >> 1; }  // arithmetic shift == floor for negatives

// footprint of `tile` at level k (0 = the tile itself): scalars only, so nothing lands in scratch
__host__ __device__ __forceinline__ Region tile_region(const ChainGeom &g, int tile, int k)
{
    int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    Region R;
    R.y0 = ty * CT_H; R.y1 = min(R.y0 + CT_H, g.h[0]) - 1;
    R.x0 = tx * CT_W; R.x1 = min(R.x0 + CT_W, g.w[0]) - 1;
    for (int i = 1; i <= k; ++i) {
        R.y0 = max(0, floordiv2(R.y0) - 1);
        R.y1 = min(g.h[i] - 1, floordiv2(R.y1) + 1);
        R.x0 = max(0, floordiv2(R.x0) - 1);
        R.x1 = min(g.w[i] - 1, floordiv2(R.x1) + 1);
    }
    return R;
}

// worst-case tile-buffer extent of level k (host + device agree on the LDS layout)
__host__ __device__ inline int chain_extent(int base, int k)
{
    int lo = 0, hi = base - 1;  // worst case is an interior tile starting at a multiple of `base`
    for (int i = 0; i < k; ++i) { lo = (lo >> 1) - 1; hi = (hi >> 1) + 1; }
    return hi - lo + 1;
}

struct LdsImg {  // a level's tile buffer: absolute coordinates -> LDS
    const double *p; int y0, x0, pitch;
    __device__ __forceinline__ double operator()(int r, int c) const { return p[(r - y0) * pitch + (c - x0)]; }
};

// block-wide min / max: wave shuffles, then one LDS hop across the block's waves (blockDim.x <= 256)
__device__ __forceinline__ void block_minmax(double &mn, double &mx)
{
    __shared__ double s_mn[4], s_mx[4];
    mn = wave_min(mn); mx = wave_max(mx);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    mn = s_mn[0]; mx = s_mx[0];
    for (int i = 1; i < nw; ++i) { mn = (s_mn[i] < mn) ? s_mn[i] : mn; mx = (s_mx[i] > mx) ? s_mx[i] : mx; }
    __syncthreads();
}

// i -> (i / nw, i % nw) for the small element counts of a tile buffer (i < 2^16, 1 <= nw <= 2^10): one multiply
// by the reciprocal instead of the ~35-instruction integer division; (i + 0.5) / nw is at least 0.5 / nw away
// from an integer, far more than the float rounding error at these magnitudes, so the floor is exact
__device__ __forceinline__ void split_rc(int i, int nw, float inv_nw, int &r, int &c)
{
    r = (int)(((float)i + 0.5f) * inv_nw);
    c = i - r * nw;
}

// LDS hand-off between the lanes of ONE wave: the LDS executes a wave's DS instructions in order, so a
// read issued after a write sees it; only the compiler must be kept from reordering them.  (The host
// emulation models lanes as fibers and needs a real barrier.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform value into a scalar register (index math derived from it then runs on the scalar unit)
__device__ __forceinline__ int uniform(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// per (frame, tile) bounds of the level-S footprint: every full-resolution value of the tile
// is a convex combination of these, so  lo - margin <= raw <= hi + margin.
// Layout [t][tile]: consecutive lanes take consecutive tiles of one frame (overlapping, x-contiguous
// footprints -> coalesced reads).
// (k_tile_bounds follows the CollapseState it reduces into)
// device-side scalars shared by the collapse passes
// Thousands of workgroups finish at about the same time and all want to fold their extremum into ONE word:
// same-address atomics serialise at ~5-12 ns each (100+ us for the evaluation pass).  Each reduction target is
// therefore striped over NSTRIPE words (workgroup b uses stripe b % NSTRIPE); readers fold the stripes with one
// load per lane and a wave reduction.
constexpr int NSTRIPE = 64;

struct CollapseState {
    unsigned long long lb_max_key;  // max over pairs of lo  (lower bound of raw.max())
    unsigned long long ub_min_key;  // min over pairs of hi  (upper bound of raw.min())
    unsigned long long ub_max_key;  // max over pairs of hi  (upper bound of raw.max())
    unsigned long long lb_min_key;  // min over pairs of lo  (lower bound of raw.min())
    unsigned long long min_key, max_key;  // exact raw.min() / raw.max()
    unsigned long long lb_max_keys[NSTRIPE], ub_min_keys[NSTRIPE], ub_max_keys[NSTRIPE], lb_min_keys[NSTRIPE];
    unsigned long long min_keys[NSTRIPE], max_keys[NSTRIPE];  // stripes of the six words above
    unsigned long long heat_min_keys[NSTRIPE], heat_max_keys[NSTRIPE];  // stripes of heat_min_key / heat_max_key
    unsigned long long smp_min_keys[NSTRIPE], smp_max_keys[NSTRIPE];    // extrema of the lattice samples (true raw values)
    unsigned int n_list_a;          // (frame, tile) pairs that may hold raw.min() / raw.max(): always evaluated (k_select_pairs -> list_a)
    unsigned int n_list_b;          // pairs kept for the masked time sum that are not in list_a: evaluated on the sparse path only
    unsigned int n_slots;           // pairs whose values are kept for the masked time sum = slots of the value store handed out
    unsigned int n_heavy;           // tiles with at least one such pair among this rank's frames (k_select_pairs -> heavy[])
    double margin;                  // absolute safety margin of the bounds
    double top_ub;                  // upper bound of `top`, from the bounds alone
    double min_val, max_val, top;   // transforms.py:185-189, decoded by k_finish_minmax
    unsigned long long heat_min_key, heat_max_key;
    double sp_bg;                   // sparse merge: rank-ordered sum of the packets' backgrounds (k_sparse_index)
    unsigned long long xs_next;     // exception store (rm_xstore.h): words handed out so far
    unsigned int xs_overflow;       // ... a record did not fit: the store-less kernel behind k_xs_sum takes the sum
};

__device__ void state_init_lane(CollapseState *st, int i)   // lanes 0 .. NSTRIPE-1 of one wavefront
{
    if (i >= NSTRIPE) return;
    st->lb_max_keys[i] = 0ull; st->ub_min_keys[i] = ~0ull; st->ub_max_keys[i] = 0ull; st->lb_min_keys[i] = ~0ull;
    st->min_keys[i] = ~0ull; st->max_keys[i] = 0ull;
    st->heat_min_keys[i] = ~0ull; st->heat_max_keys[i] = 0ull;
    st->smp_min_keys[i] = ~0ull; st->smp_max_keys[i] = 0ull;
    if (i != 0) return;
    st->lb_max_key = 0ull; st->ub_min_key = ~0ull; st->ub_max_key = 0ull; st->lb_min_key = ~0ull;
    st->min_key = ~0ull; st->max_key = 0ull; st->n_list_a = 0; st->n_list_b = 0; st->n_slots = 0; st->n_heavy = 0;
    st->margin = 0; st->top_ub = 0; st->min_val = 0; st->max_val = 0; st->top = 0;
    st->heat_min_key = ~0ull; st->heat_max_key = 0ull;
    st->xs_next = 0ull; st->xs_overflow = 0u;
}

RM_KERNEL __launch_bounds__(NSTRIPE) void k_state_init(CollapseState *st) { state_init_lane(st, (int)threadIdx.x); }

// min / max of a pair of striped targets: skipped when they cannot change the result.  Both current words are requested before either is
// compared -- a load-compare-atomic at a time is a round trip per target at the end of a wave's life (rm_bounds_l1.h)
__device__ __forceinline__ void striped_min_max(unsigned long long *mins, unsigned long long *maxs, int sp, unsigned long long kmn, unsigned long long kmx)
{
    const unsigned long long cmn = *(volatile unsigned long long *)&mins[sp], cmx = *(volatile unsigned long long *)&maxs[sp];
    if (kmn < cmn) atomicMin(&mins[sp], kmn);
    if (kmx > cmx) atomicMax(&maxs[sp], kmx);
}

// fold the stripes of one target (plus its unstriped word); every lane of the wave gets the result
__device__ __forceinline__ unsigned long long fold_min_keys(const unsigned long long *stripes, unsigned long long word)
{
    unsigned long long v = stripes[threadIdx.x & (NSTRIPE - 1)];
    v = wave_reduce(v, [](unsigned long long o, unsigned long long w) { return (o < w) ? o : w; });
    return (word < v) ? word : v;
}
__device__ __forceinline__ unsigned long long fold_max_keys(const unsigned long long *stripes, unsigned long long word)
{
    unsigned long long v = stripes[threadIdx.x & (NSTRIPE - 1)];
    v = wave_reduce(v, [](unsigned long long o, unsigned long long w) { return (o > w) ? o : w; });
    return (word > v) ? word : v;
}

// Lattice samples.  The tile bounds say where raw.min() / raw.max() CAN be; how low `top` can be -- and with it how many
// pairs must be evaluated -- hangs on an UPPER bound of raw.min() and a LOWER bound of raw.max(), and the bounds alone
// give poor ones (min over pairs of hi, max over pairs of lo: -39 / +40 against the true -50 / +51 on the synthetic
// 1080p stream, so top_ub = -12 instead of -19.7 and 9 066 pairs kept instead of ~5 000).  Any true value of raw
// bounds them far better, and some come almost for free: the full-resolution pixel (y << S, x << S) of an interior
// level-S pixel is, through every pyrUp step, the even-even sample of its 3 x 3 level-S neighbourhood -- per axis
// lat_a * (c[y-1] + c[y+1]) + lat_b * c[y] with dyadic weights (S = 4: 85/512, 342/512) -- because position p << k at level
// S-k only ever draws on positions (p << (k-1)) - 1 .. + 1 one level up, none of which touches a border rule for 1 <= p <=
// size - 2.  The weights are applied directly (a few roundings, ~3e-15 relative to max|c|, against the chain's own few), so
// the samples enter the selection with twice the pruning margin (1e-12 relative).  Pruning stays exact: the evaluated
// pairs still yield the exact extrema, only fewer pairs need evaluating.
__device__ __forceinline__ double lattice_sample(const double *r0, const double *r1, const double *r2, int x, double a, double b)
{
    const double h0 = (r0[x - 1] + r0[x + 1]) * a + r0[x] * b;
    const double h1 = (r1[x - 1] + r1[x + 1]) * a + r1[x] * b;
    const double h2 = (r2[x - 1] + r2[x + 1]) * a + r2[x] * b;
    return (h0 + h2) * a + h1 * b;
}

// The four extrema of the bounds (over ALL pairs) are reduced here as well: block-level min/max, then striped
// atomics that are skipped when they cannot change the result.
RM_KERNEL __launch_bounds__(256) void k_tile_bounds(const double *cS, ChainGeom g, int T, int ntiles,
                                                     double *lo, double *hi, CollapseState *st, int *sel_cnt)
{
    const double inf = __builtin_huge_val();
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < ntiles; i += 256) sel_cnt[i] = 0;   // k_select_pairs counts into it
    int idx = blockIdx.x * 256 + threadIdx.x;
    double mn = inf, mx = -inf;
    if (idx < ntiles * T) {
        int t = idx / ntiles, tile = idx - t * ntiles;
        const int S = g.S;
        const Region R = tile_region(g, tile, S);
        const int wS = g.w[S];
        const double *p = cS + (size_t)t * g.h[S] * wS;
        mn = p[(size_t)R.y0 * wS + R.x0]; mx = mn;
        for (int y = R.y0; y <= R.y1; ++y)
            for (int x = R.x0; x <= R.x1; ++x) {
                double v = p[(size_t)y * wS + x];
                mn = (v < mn) ? v : mn;
                mx = (v > mx) ? v : mx;
            }
        lo[idx] = mn; hi[idx] = mx;
    }
    // lanes past the end hold (+inf, -inf): neutral for min-of-lo / max-of-hi; for max-of-lo / min-of-hi they must
    // not take part, so those two use the swapped neutral elements
    double lo_mn = mn, lo_mx = (idx < ntiles * T) ? mn : -inf;
    double hi_mx = mx, hi_mn = (idx < ntiles * T) ? mx : inf;
    block_minmax(lo_mn, lo_mx);
    block_minmax(hi_mn, hi_mx);
    if (threadIdx.x == 0) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->lb_min_keys, st->lb_max_keys, sp, k_lo_mn, k_lo_mx);
        striped_min_max(st->ub_min_keys, st->ub_max_keys, sp, k_hi_mn, k_hi_mx);
    }
}

// The same bounds, one workgroup per frame: the level-S footprint of a tile is a rectangle, so its min / max is
// the min / max over the footprint rows of per-row extrema over the footprint columns (exact: min and max are
// associative).  Row extrema for every (row, tile column) go to LDS first; ~3.6x fewer loads than k_tile_bounds
// and no per-thread 2-D loop over global memory.  Used when the [h_S][tiles_x] x 2 table fits LDS.
// blockIdx.y selects a band of `band` tile rows (large levels: the row-extrema table of a whole frame would not fit LDS);
// the table then holds only the level-S rows [y_lo, y_hi] that band's footprints touch (at most `tbl_rows` of them).
RM_KERNEL __launch_bounds__(256) void k_frame_bounds(const double *cS, ChainGeom g, int ntiles, double *lo, double *hi,
                                                      CollapseState *st, int band, int tbl_rows, int *sel_cnt)
{
    HIP_DYNAMIC_SHARED(double, lds)
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < ntiles; i += 256) sel_cnt[i] = 0;   // k_select_pairs counts into it
    const double inf = __builtin_huge_val();
    const int S = g.S, hS = g.h[S], wS = g.w[S], ntx = g.tiles_x;
    const int t = blockIdx.x;
    const int ty0 = blockIdx.y * band, ty1 = min(ty0 + band, g.tiles_y) - 1;             // tile rows of this workgroup
    const int y_lo = tile_region(g, ty0 * ntx, S).y0, y_hi = tile_region(g, ty1 * ntx, S).y1;   // level-S rows they touch
    const int nrows = y_hi - y_lo + 1;
    double *rmin = lds, *rmax = lds + (size_t)tbl_rows * ntx;
    const double *p = cS + (size_t)t * hS * wS;
    const float inv_ntx = 1.0f / (float)ntx;
    double t_mn = inf, t_mx = -inf;
    int p_mn = -1, p_mx = -1;
    for (int i = threadIdx.x; i < nrows * ntx; i += 256) {
        int y, tx;
        split_rc(i, ntx, inv_ntx, y, tx);
        const Region R = tile_region(g, tx, S);   // tile tx of the first tile row: same column range as every tile below it
        const double *row = p + (size_t)(y_lo + y) * wS;
        double mn = row[R.x0], mx = mn;
        int xn = R.x0, xx = R.x0;
        // FB_CHUNK loads in flight per thread (the plain loop waited for every element in turn: ~20 dependent round trips per
        // footprint row at skip 2).  Columns past the footprint repeat its last one: a repeated value changes neither the
        // extrema nor the first position they were met at, so the result is that of the element-by-element scan.
        constexpr int FB_CHUNK = 10;
        for (int x = R.x0 + 1; x <= R.x1; x += FB_CHUNK) {
            double v[FB_CHUNK];
#pragma unroll
            for (int j = 0; j < FB_CHUNK; ++j) v[j] = row[min(x + j, R.x1)];
#pragma unroll
            for (int j = 0; j < FB_CHUNK; ++j) {
                const int xj = min(x + j, R.x1);
                if (v[j] < mn) { mn = v[j]; xn = xj; }
                if (v[j] > mx) { mx = v[j]; xx = xj; }
            }
        }
        rmin[i] = mn; rmax[i] = mx;
        // (where this thread has seen the lowest / highest C_S so far: its lattice samples are taken there)
        if (mn < t_mn) { t_mn = mn; p_mn = (y_lo + y) * wS + xn; }
        if (mx > t_mx) { t_mx = mx; p_mx = (y_lo + y) * wS + xx; }
    }
    __syncthreads();
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    const int tile_begin = ty0 * ntx, tile_end = (ty1 + 1) * ntx;
    for (int tile = tile_begin + threadIdx.x; tile < tile_end; tile += 256) {
        const int tx = tile % ntx;
        const Region R = tile_region(g, tile, S);
        double mn = rmin[(R.y0 - y_lo) * ntx + tx], mx = rmax[(R.y0 - y_lo) * ntx + tx];
        for (int y = R.y0 + 1; y <= R.y1; ++y) {
            const double a = rmin[(y - y_lo) * ntx + tx], b = rmax[(y - y_lo) * ntx + tx];
            mn = (a < mn) ? a : mn;
            mx = (b > mx) ? b : mx;
        }
        lo[(size_t)t * ntiles + tile] = mn; hi[(size_t)t * ntiles + tile] = mx;
        lo_mn = (mn < lo_mn) ? mn : lo_mn; lo_mx = (mn > lo_mx) ? mn : lo_mx;
        hi_mn = (mx < hi_mn) ? mx : hi_mn; hi_mx = (mx > hi_mx) ? mx : hi_mx;
    }
    // lattice samples (true raw values: see lattice_sample) at the interior pixels nearest to the lowest / highest C_S each
    // thread met in the first loop: two per thread bound the extrema as well as sampling every pixel would (a sample per
    // pixel -- nine global loads each -- made this kernel 6x slower on the 180 x 320 level of the 720p configuration)
    double sm_mn = inf, sm_mx = -inf;
    if (hS >= 3 && wS >= 3) {
        const int cand[2] = {p_mn, p_mx};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cand[k] < 0) continue;
            int y = cand[k] / wS, x = cand[k] - y * wS;
            y = min(max(y, 1), hS - 2); x = min(max(x, 1), wS - 2);
            const double *r1 = p + (size_t)y * wS;
            const double v = lattice_sample(r1 - wS, r1, r1 + wS, x, g.lat_a, g.lat_b);
            sm_mn = (v < sm_mn) ? v : sm_mn; sm_mx = (v > sm_mx) ? v : sm_mx;
        }
    }
    block_minmax(lo_mn, lo_mx);
    block_minmax(hi_mn, hi_mx);
    block_minmax(sm_mn, sm_mx);
    if (threadIdx.x == 0) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = (blockIdx.x + blockIdx.y * 7) & (NSTRIPE - 1);
        striped_min_max(st->lb_min_keys, st->lb_max_keys, sp, k_lo_mn, k_lo_mx);
        striped_min_max(st->ub_min_keys, st->ub_max_keys, sp, k_hi_mn, k_hi_mx);
        if (sm_mn <= sm_mx) {
            const unsigned long long k_mn = f64_key(sm_mn), k_mx = f64_key(sm_mx);
            striped_min_max(st->smp_min_keys, st->smp_max_keys, sp, k_mn, k_mx);
        }
    }
}

// k_frame_bounds for wide levels (4K, skip 2), streaming: no row-extrema table, no workgroup barrier.  In k_frame_bounds a
// thread per (row, tile column) reads its footprint straight from memory, lanes 64 >> S columns apart -- every load instruction
// touches 64 cache lines -- and a band's table (100 KB at 4K) leaves one workgroup per CU.  Here a WAVE owns FB_TR consecutive tile
// rows of a frame and walks down the level-S rows their footprints touch: lanes load consecutive columns (FB_MAXNL loads in flight,
// the next row requested before this one is scanned), park the row in a skewed LDS buffer (index i + (i >> 4): the scans of
// neighbouring tile columns hit different banks), lane tx takes the extrema of tile column tx's footprint columns from there and
// folds them into the running extrema of the (at most two) tile rows whose footprint holds this row.  8 KB of LDS per wave.
// Same bounds (min / max are exact in any order); the lattice samples are taken where a LANE met its extreme values, so the
// sample set -- and with it how many pairs the selection keeps, never the result -- differs from k_frame_bounds'.
constexpr int FB_MAXNL = 16;   // row length <= 64 * FB_MAXNL level-S columns
#define RM_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // register budget of 512 / n per lane
#ifndef RM_HIPEMU
#define RM_WAVES_PER_EU_IF(cond, a, b) __attribute__((amdgpu_waves_per_eu((cond) ? (a) : (b), (cond) ? (a) : (b))))   // ... chosen by a template parameter
#else
#define RM_WAVES_PER_EU_IF(cond, a, b)   // (g++ does not parse an expression inside an attribute it does not know)
#endif
// FB_TR (template): tile rows per wave -- 8 where that still gives every SIMD a few waves (halo rows: 12 %), 2 for small images
__host__ __device__ __forceinline__ int fb_row_pitch(int wS) { return wS + (wS >> 4) + 2; }

template <int FB_TR>
__global__ __launch_bounds__(256) void k_frame_bounds_rows(const double *cS, ChainGeom g, int ntiles, double *lo, double *hi,
                                                           CollapseState *st, int *sel_cnt)
{
    HIP_DYNAMIC_SHARED(double, lds)
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < ntiles; i += 256) sel_cnt[i] = 0;   // k_select_pairs counts into it
    const double inf = __builtin_huge_val();
    const int S = g.S, hS = g.h[S], wS = g.w[S], ntx = g.tiles_x;
    const int t = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ty_first = (blockIdx.y * 4 + wave) * FB_TR;
    const int ty_last = min(ty_first + FB_TR, g.tiles_y) - 1;
    const bool have = ty_first <= ty_last;               // (wave-uniform; waves past the last tile row only keep the barriers company)
    const int y_lo = have ? tile_region(g, ty_first * ntx, S).y0 : 0, y_hi = have ? tile_region(g, ty_last * ntx, S).y1 : -1;
    const int nrows = y_hi - y_lo + 1;
    int nrows_max = 0;                                   // trips every wave of the workgroup makes (host emulation: wave_sync() is a barrier)
    for (int w = 0; w < 4; ++w) {
        const int a = (blockIdx.y * 4 + w) * FB_TR, b = min(a + FB_TR, g.tiles_y) - 1;
        if (a <= b) nrows_max = max(nrows_max, tile_region(g, b * ntx, S).y1 - tile_region(g, a * ntx, S).y0 + 1);
    }
    double *rowbuf = lds + (size_t)wave * fb_row_pitch(wS);
    const double *p = cS + (size_t)t * hS * wS;
    const int nl = (wS + 63) >> 6;
    const int tx = min(lane, ntx - 1);
    const Region Rx = tile_region(g, tx, S);             // column range of tile column tx (the same in every tile row)
    double amn[FB_TR], amx[FB_TR];
#pragma unroll
    for (int k = 0; k < FB_TR; ++k) { amn[k] = inf; amx[k] = -inf; }
    double t_mn = inf, t_mx = -inf;
    int p_mn = -1, p_mx = -1;
    double nxt[FB_MAXNL];
    auto fetch = [&](int y) __attribute__((always_inline)) {
        const double *row = p + (size_t)min(y_lo + max(min(y, nrows - 1), 0), hS - 1) * wS;
#pragma unroll
        for (int j = 0; j < FB_MAXNL; ++j)
            if (j < nl) nxt[j] = row[min(lane + 64 * j, wS - 1)];
    };
    fetch(0);
    for (int y = 0; y < nrows_max; ++y) {
#pragma unroll
        for (int j = 0; j < FB_MAXNL; ++j) {
            const int x = lane + 64 * j;
            if (j < nl && x < wS) rowbuf[x + (x >> 4)] = nxt[j];
        }
        fetch(y + 1);
        wave_sync();
        if (y < nrows) {
            double mn = rowbuf[Rx.x0 + (Rx.x0 >> 4)], mx = mn;
            for (int x = Rx.x0 + 1; x <= Rx.x1; ++x) {
                const double v = rowbuf[x + (x >> 4)];
                mn = (v < mn) ? v : mn; mx = (v > mx) ? v : mx;
            }
            const int ya = y_lo + y;
#pragma unroll
            for (int k = 0; k < FB_TR; ++k) {
                const int ty = ty_first + k;
                if (ty <= ty_last) {
                    const Region R = tile_region(g, ty * ntx, S);
                    if (ya >= R.y0 && ya <= R.y1) { amn[k] = (mn < amn[k]) ? mn : amn[k]; amx[k] = (mx > amx[k]) ? mx : amx[k]; }   // (uniform)
                }
            }
            // the ROW in which this lane met its lowest / highest C_S so far; the column is looked up once, at the end
            if (mn < t_mn) { t_mn = mn; p_mn = ya; }
            if (mx > t_mx) { t_mx = mx; p_mx = ya; }
        }
        wave_sync();
    }
    if (have && lane < ntx) {   // first column of the extreme value inside its row's footprint
        if (p_mn >= 0) { const double *row = p + (size_t)p_mn * wS; int x = Rx.x0; while (x < Rx.x1 && row[x] != t_mn) ++x; p_mn = p_mn * wS + x; }
        if (p_mx >= 0) { const double *row = p + (size_t)p_mx * wS; int x = Rx.x0; while (x < Rx.x1 && row[x] != t_mx) ++x; p_mx = p_mx * wS + x; }
    }
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    if (have && lane < ntx) {
#pragma unroll
        for (int k = 0; k < FB_TR; ++k) {
            const int ty = ty_first + k;
            if (ty <= ty_last) {
                const size_t o = (size_t)t * ntiles + (size_t)ty * ntx + lane;
                lo[o] = amn[k]; hi[o] = amx[k];
                lo_mn = (amn[k] < lo_mn) ? amn[k] : lo_mn; lo_mx = (amn[k] > lo_mx) ? amn[k] : lo_mx;
                hi_mn = (amx[k] < hi_mn) ? amx[k] : hi_mn; hi_mx = (amx[k] > hi_mx) ? amx[k] : hi_mx;
            }
        }
    }
    double sm_mn = inf, sm_mx = -inf;
    if (have && lane < ntx && hS >= 3 && wS >= 3) {
        const int cand[2] = {p_mn, p_mx};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cand[k] < 0) continue;
            int y = cand[k] / wS, x = cand[k] - y * wS;
            y = min(max(y, 1), hS - 2); x = min(max(x, 1), wS - 2);
            const double *r1 = p + (size_t)y * wS;
            const double v = lattice_sample(r1 - wS, r1, r1 + wS, x, g.lat_a, g.lat_b);
            sm_mn = (v < sm_mn) ? v : sm_mn; sm_mx = (v > sm_mx) ? v : sm_mx;
        }
    }
    lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
    sm_mn = wave_min(sm_mn); sm_mx = wave_max(sm_mx);
    if (lane == 0 && have) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = (blockIdx.x + (blockIdx.y * 4 + wave) * 7) & (NSTRIPE - 1);
        striped_min_max(st->lb_min_keys, st->lb_max_keys, sp, k_lo_mn, k_lo_mx);
        striped_min_max(st->ub_min_keys, st->ub_max_keys, sp, k_hi_mn, k_hi_mx);
        if (sm_mn <= sm_mx) {
            const unsigned long long k_mn = f64_key(sm_mn), k_mx = f64_key(sm_mx);
            striped_min_max(st->smp_min_keys, st->smp_max_keys, sp, k_mn, k_mx);
        }
    }
}

// k_small_collapse and k_frame_bounds in one: the collapsed level S of a frame is still in LDS when its tile bounds are
// wanted, so they are taken from there (no second pass over C_S in memory, one kernel boundary less).  Used when the
// row-extrema table of a whole frame fits beside the frame's small pyramid.
// What follows the collapse of a frame, with C_S of the frame in LDS at `c`: copy-out to cS[t], tile bounds (per-row extrema
// over the footprint columns, then extrema over the footprint rows) from the LDS copy, their extrema and the lattice samples
// into the striped state.  rmin / rmax: the row-extrema table, 2 x hS x tiles_x doubles of LDS.  One SMALL_NT-thread workgroup.
// A workgroup may own only PART of the frame (k_small_filter_first puts two workgroups on a frame): tile rows [ty_a, ty_b), whose
// footprints touch the level-S rows [y_lo, y_hi] (valid in `c`), and the rows [y_out_a, y_out_b) it copies out.
__device__ __forceinline__ void frame_bounds_from_lds(const double *c, double *rmin_base, const SmallGeom &sg, const ChainGeom &g, int ntiles, int t,
                                                      double *cS, double *lo, double *hi, CollapseState *st, double (*s_red)[SMALL_NT / 64],
                                                      int (*s_arg)[SMALL_NT / 64], int mark_kid, int ty_a = 0, int ty_b = 0x7fffffff,
                                                      int y_lo = 0, int y_hi = 0x7fffffff, int y_out_a = 0, int y_out_b = 0x7fffffff)
{
    const int tid = threadIdx.x;
    const int S = sg.S;
    const int hS = sg.h[S], wS = sg.w[S], ntx = g.tiles_x;
    if (ty_b > g.tiles_y) ty_b = g.tiles_y;
    if (y_hi > hS - 1) y_hi = hS - 1;
    if (y_out_b > hS) y_out_b = hS;
    double *o = cS + (size_t)t * (hS * wS);
    // (the copy-out loop also finds where this part of the frame's C_S is lowest / highest: the lattice samples are taken there)
    const double inf = __builtin_huge_val();
    double c_mn = inf, c_mx = -inf;
    int i_mn = y_out_a * wS, i_mx = y_out_a * wS;
    for (int i = y_out_a * wS + tid; i < y_out_b * wS; i += SMALL_NT) {
        const double v = c[i];
        o[i] = v;
        if (v < c_mn) { c_mn = v; i_mn = i; }
        if (v > c_mx) { c_mx = v; i_mx = i; }
    }
    RM_TRACE_MARK(mark_kid, 8);
    // tile bounds from the LDS copy: per-row extrema over the footprint columns, then extrema over the footprint rows
    double *rmin = rmin_base, *rmax = rmin + (size_t)hS * ntx;
    const float inv_ntx = 1.0f / (float)ntx;
    for (int i = y_lo * ntx + tid; i < (y_hi + 1) * ntx; i += SMALL_NT) {
        int y, tx;
        split_rc(i, ntx, inv_ntx, y, tx);
        const Region R = tile_region(g, tx, S);
        const double *row = c + y * wS;
        double mn = row[R.x0], mx = mn;
        for (int x = R.x0 + 1; x <= R.x1; ++x) {
            const double v = row[x];
            mn = (v < mn) ? v : mn;
            mx = (v > mx) ? v : mx;
        }
        rmin[i] = mn; rmax[i] = mx;
    }
    __syncthreads();
    RM_TRACE_MARK(mark_kid, 9);
    
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    (void)ntiles;
    for (int tile = ty_a * ntx + tid; tile < ty_b * ntx; tile += SMALL_NT) {
        const int tx = tile % ntx;
        const Region R = tile_region(g, tile, S);
        double mn = rmin[R.y0 * ntx + tx], mx = rmax[R.y0 * ntx + tx];
        for (int y = R.y0 + 1; y <= R.y1; ++y) {
            const double a = rmin[y * ntx + tx], b = rmax[y * ntx + tx];
            mn = (a < mn) ? a : mn;
            mx = (b > mx) ? b : mx;
        }
        lo[(size_t)t * ntiles + tile] = mn; hi[(size_t)t * ntiles + tile] = mx;
        lo_mn = (mn < lo_mn) ? mn : lo_mn; lo_mx = (mn > lo_mx) ? mn : lo_mx;
        hi_mn = (mx < hi_mn) ? mx : hi_mn; hi_mx = (mx > hi_mx) ? mx : hi_mx;
    }
    lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
    // wave-level arg-min / arg-max of C_S (value and position travel together)
    wave_arg_reduce(c_mn, i_mn, [](double o, double w) { return o < w; });
    wave_arg_reduce(c_mx, i_mx, [](double o, double w) { return o > w; });
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        s_red[0][wave] = lo_mn; s_red[1][wave] = lo_mx; s_red[2][wave] = hi_mn; s_red[3][wave] = hi_mx;
        s_red[4][wave] = c_mn; s_red[5][wave] = c_mx; s_arg[0][wave] = i_mn; s_arg[1][wave] = i_mx;
    }
    __syncthreads();
    RM_TRACE_MARK(mark_kid, 11);
    if (wave != 0) return;
    {   // wave 0 folds the per-wave partials: lane w takes wave w's
        const bool have = tid < SMALL_NT / 64;
        lo_mn = have ? s_red[0][tid] : inf; lo_mx = have ? s_red[1][tid] : -inf;
        hi_mn = have ? s_red[2][tid] : inf; hi_mx = have ? s_red[3][tid] : -inf;
        c_mn = have ? s_red[4][tid] : inf; c_mx = have ? s_red[5][tid] : -inf;
        i_mn = have ? s_arg[0][tid] : 0; i_mx = have ? s_arg[1][tid] : 0;
        lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
        wave_arg_reduce(c_mn, i_mn, [](double o, double w) { return o < w; });
        wave_arg_reduce(c_mx, i_mx, [](double o, double w) { return o > w; });
    }
    if (tid == 0) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = blockIdx.x & (NSTRIPE - 1);
        atomicMax(&st->lb_max_keys[sp], k_lo_mx);
        atomicMin(&st->lb_min_keys[sp], k_lo_mn);
        atomicMax(&st->ub_max_keys[sp], k_hi_mx);
        atomicMin(&st->ub_min_keys[sp], k_hi_mn);
        // lattice samples (true raw values: see lattice_sample) at the interior pixels nearest to this frame's lowest and
        // highest C_S: on the synthetic 1080p stream they bound the extrema as tightly as sampling every pixel would
        if (hS >= 3 && wS >= 3) {
            int ya = i_mn / wS, xa = i_mn - ya * wS, yb = i_mx / wS, xb = i_mx - yb * wS;
            const int y_first = max(1, y_lo + 1), y_last = max(y_first, min(hS - 2, y_hi - 1));   // rows whose 3 x 3 neighbourhood is valid in `c`
            ya = min(max(ya, y_first), y_last); xa = min(max(xa, 1), wS - 2);
            yb = min(max(yb, y_first), y_last); xb = min(max(xb, 1), wS - 2);
            const double *ra = c + ya * wS, *rb = c + yb * wS;
            const double va = lattice_sample(ra - wS, ra, ra + wS, xa, g.lat_a, g.lat_b);
            const double vb = lattice_sample(rb - wS, rb, rb + wS, xb, g.lat_a, g.lat_b);
            atomicMin(&st->smp_min_keys[sp], f64_key(va < vb ? va : vb));
            atomicMax(&st->smp_max_keys[sp], f64_key(va > vb ? va : vb));
        }
    }
    RM_TRACE_MARK(mark_kid, 12);
}

RM_KERNEL __launch_bounds__(SMALL_NT) void k_small_collapse_bounds(const double *bp_all, SmallGeom sg, double *cS, CollapseState *st,
                                                                     ChainGeom g, int ntiles, double *lo, double *hi, int *sel_cnt)
{
    RM_TRACE_SCOPE(3);
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < ntiles; i += SMALL_NT) sel_cnt[i] = 0;   // k_select_pairs counts into it
    HIP_DYNAMIC_SHARED(double, lds)   // [NP] frame (levels laid out as in bp_all), then the row-extrema table
    __shared__ double s_red[6][SMALL_NT / 64];
    __shared__ int s_arg[2][SMALL_NT / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int S = sg.S, L = sg.L;
    // (st was reset by an EARLIER kernel on the stream -- k_small_pyramid or k_state_init: the atomics at the end of this
    //  kernel must not race with a reset inside it)
    RM_TRACE_MARK(3, 0);
    fill_lds(lds, bp_all + (size_t)t * sg.NP, sg.NP, tid);
    __syncthreads();
    RM_TRACE_MARK(3, 1);
    for (int l = L - 3; l >= S; --l) {
        const int dh = sg.h[l], dw = sg.w[l], sh = sg.h[l + 1], sw = sg.w[l + 1];
        double *d = lds + sg.np_off[l];
        small_up_level(lds + sg.np_off[l + 1], sh, sw, dh, dw, tid, [&](int i, double v) { d[i] = v + d[i]; });
        __syncthreads();
        RM_TRACE_MARK(3, 2 + (L - 3 - l));
    }
    frame_bounds_from_lds(lds + sg.np_off[S], lds + sg.NP, sg, g, ntiles, t, cS, lo, hi, st, s_red, s_arg, 3);
}

// ---- filter-first form of the small pyramid (round 2) ---------------------------------------------------------------------
// The temporal band-pass is linear and acts per pixel, the pyramid steps are linear and act per frame: they commute.  With
// X_l = B(G_l) = pyrDown^(l-S)(X_S) the band-passed Laplacians are L_l = X_l - pyrUp(X_{l+1}) (pyramid.py:23-26), and the collapse
// (pyramid.py:51-57: img = pyrUp(img) + L_l from a zero coarsest level) telescopes:
//     C_{L-2} = X_{L-2} - pyrUp(X_{L-1}),   C_l = pyrUp(C_{l+1}) + X_l - pyrUp(X_{l+1}) = X_l - pyrUp^(L-1-l)(X_{L-1})
// so  C_S = X_S - pyrUp^(L-1-S)(pyrDown^(L-1-S)(X_S)):  filter G_S once ([T, h_S w_S]), then ONE per-frame kernel walks down to
// the coarsest level, back up, and subtracts.  k_small_pyramid, its [T, NP] Laplacian / band-passed arrays, a quarter of the
// filter's pixels and half of the collapse's pyrUp work go.
// The price is the rounding ORDER: the reference filters the Laplacians and adds them up, this filters their common source, so
// C_S agrees with the per-level path to ~1e-15 relative instead of bit for bit (the ROI and the uint8 heatmap are unaffected
// except on exact ties of the mask threshold -- the same class of event the explicit filter operator already belongs to).
// RM_FLAG_FILTER_LAPLACIANS selects the reference's order (k_small_pyramid / k_small_collapse_bounds above).
// LDS: the levels S .. L-1 (sg.g_off; levels S+1 .. L-2 are overwritten on the way up), then the bounds table.
RM_KERNEL __launch_bounds__(SMALL_NT) void k_small_filter_first(const double *xg, SmallGeom sg, int lds_levels, double *cS, CollapseState *st,
                                                                  ChainGeom g, int ntiles, double *lo, double *hi, int *sel_cnt, int parts)
{
    RM_TRACE_SCOPE(3);
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < ntiles; i += SMALL_NT) sel_cnt[i] = 0;   // k_select_pairs counts into it
    HIP_DYNAMIC_SHARED(double, lds)
    __shared__ double s_red[6][SMALL_NT / 64];
    __shared__ int s_arg[2][SMALL_NT / 64];
    // `parts` workgroups per frame (gridDim.x = frames * parts): each walks the whole way down and back up to level S + 1 (those
    // levels are a quarter of the frame and less), then takes the last step up, the subtraction, the copy-out and the bounds for ITS
    // band of tile rows only -- with one workgroup per unique frame half of the chip's CUs had nothing to do (129 frames at T = 256)
    const int t = blockIdx.x / parts, part = blockIdx.x - t * parts, tid = threadIdx.x;
    const int S = sg.S, L = sg.L;
    const int nS = sg.h[S] * sg.w[S];
    const int ty_a = (int)((long long)g.tiles_y * part / parts), ty_b = (int)((long long)g.tiles_y * (part + 1) / parts);
    const int y_lo = parts == 1 ? 0 : tile_region(g, ty_a * g.tiles_x, S).y0;
    const int y_hi = parts == 1 ? sg.h[S] - 1 : tile_region(g, (ty_b - 1) * g.tiles_x, S).y1;
    // rows this part copies out: the frame's rows cut where the parts' tile rows are cut (inside both neighbours' computed ranges)
    const int y_out_a = part == 0 ? 0 : min(sg.h[S], (ty_a * CT_H) >> S), y_out_b = part == parts - 1 ? sg.h[S] : min(sg.h[S], (ty_b * CT_H) >> S);
    RM_TRACE_MARK(3, 0);
    fill_lds(lds + sg.g_off[S], xg + (size_t)t * nS, nS, tid);
    __syncthreads();
    RM_TRACE_MARK(3, 1);
    for (int l = S + 1; l < L; ++l) {   // cv2.pyrDown chain of the filtered level (pyramid.py:14)
        const int sh = sg.h[l - 1], sw = sg.w[l - 1], dh = sg.h[l], dw = sg.w[l];
        const double *sp = lds + sg.g_off[l - 1];
        double *d = lds + sg.g_off[l];
        for (int x = (tid & 63); x < dw; x += 64) {
            const int c0 = reflect101(2 * x - 2, sw), c1 = reflect101(2 * x - 1, sw), c2 = reflect101(2 * x, sw);
            const int c3 = reflect101(2 * x + 1, sw), c4 = reflect101(2 * x + 2, sw);
            for (int y = (tid >> 6); y < dh; y += SMALL_NT / 64) {
                double r[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const double *row = sp + reflect101(2 * y - 2 + k, sh) * sw;
                    r[k] = row[c2] * 6 + (row[c1] + row[c3]) * 4 + row[c0] + row[c4];
                }
                d[y * dw + x] = (r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]) * (1.0 / 256);
            }
        }
        __syncthreads();
    }
    RM_TRACE_MARK(3, 2);
    for (int l = L - 2; l >= S; --l) {   // back up: U_l = pyrUp(U_{l+1}) over the dead X_l, and C_S = X_S - U_S in place
        const int dh = sg.h[l], dw = sg.w[l], sh = sg.h[l + 1], sw = sg.w[l + 1];
        double *d = lds + sg.g_off[l];
        const bool last = l == S;
        small_up_level(lds + sg.g_off[l + 1], sh, sw, dh, dw, tid, [&](int i, double v) { d[i] = last ? d[i] - v : v; }, last ? y_lo : 0,
                       last ? y_hi + 1 : 0x7fffffff);
        __syncthreads();
    }
    RM_TRACE_MARK(3, 4);
    frame_bounds_from_lds(lds + sg.g_off[S], lds + lds_levels, sg, g, ntiles, t, cS, lo, hi, st, s_red, s_arg, 3, ty_a, ty_b, y_lo, y_hi, y_out_a, y_out_b);
}

constexpr double PRUNE_REL_MARGIN = 1e-12;  // >> the ~1e-14 relative rounding of the S-level chain

constexpr int SLOT_PRUNED = -1;    // every value of the pair is provably >= top: contributes `min`
// slot >= 0: the pair's 16 x 64 values are parked in slot `slot` of the value store for the masked time sum.  k_select_pairs hands
// the slots out so that the kept frames of a tile are NEIGHBOURS in the store (one contiguous run per tile and frame chunk): the
// sum pass walks a tile's frames, and slots scattered over the store cost it a TLB / DRAM-page miss per frame.
//
// Sparse or dense sum?  (rm_dense_sum.h)  Decided ON THE DEVICE from what this call's own selection kept -- every kernel that
// cares evaluates sum_is_dense() on the counters k_select_pairs left in the state, so the first call of a geometry behaves like
// the hundredth and nothing is remembered between calls:
//   * more kept pairs than the value store has slots -> dense (the store is capped: rm_collapse_eval.hip collapse_eval);
//   * skip <= 2 and more than one pair in DENSE_ONE_IN kept -> dense (measured per pair of the geometry on MI355X: sparse 3.7-4.8 ns
//     per KEPT pair; dense 1.6 ns at 4K x 512 skip 2, 3.6 ns at 720p x 128 skip 2, 4.6 ns at 1080p x 256 skip 4 -- slower than
//     sparse even with everything kept, so deeper chains go dense only on overflow);
//   * RM_FLAG_DENSE_SUM / RM_FLAG_SPARSE_SUM force it (an overflowing store still goes dense).
constexpr unsigned long long DENSE_ONE_IN = 2;
struct SumPlan {
    int mode;                  // 0 automatic, 1 dense, 2 sparse
    unsigned int cap_slots;    // slots of the value store
    unsigned int npairs_mine;  // unique (tile, frame) pairs among this rank's frames
    int auto_dense_ok;         // the automatic rule may choose the dense kernel (skip <= 2)
};
__device__ __forceinline__ bool sum_is_dense(const CollapseState *st, const SumPlan &sp)
{
    const unsigned int kept = st->n_slots;
    if (sp.mode == 1 || kept > sp.cap_slots) return true;
    if (sp.mode == 2) return false;
    return sp.auto_dense_ok && (unsigned long long)kept * DENSE_ONE_IN > (unsigned long long)sp.npairs_mine;
}

// exclusive prefix sum over the 256 threads of a workgroup (thread order); s_wave: 4 words of LDS.  total = sum over all threads.
__device__ __forceinline__ unsigned long long block_excl_scan_256(unsigned long long v, unsigned long long *s_wave, unsigned long long &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const unsigned long long c = s_wave[w]; base += (w < wave) ? c : 0; tot += c; }
    total = tot;
    return base + inc - v;
}

// which pairs need their full-resolution values:
//   C: may hold raw.max() or raw.min()                        -> evaluated for the exact min/max      (list_a)
//   D: may hold a value below top (lo - margin < top_ub)      -> values kept for the masked sum       (list_b unless also C)
// Pairs are [u][tile] over the UNIQUE frames u < Th (sym_frames).  A workgroup takes SEL_TILES adjacent tiles (16 lanes = one
// 128-byte row of bounds) and a chunk of SEL_PH x SEL_U unique frames; thread (tile, phase) owns the frames phase, phase + 16, ...
// of its tile, whose bound loads are issued together (the kernel is latency bound).  Everything the workgroup hands out -- value
// store slots, list positions -- is counted in LDS (one block-wide prefix sum in (tile, phase) order over three packed 20-bit
// counts) and reserved with ONE returning atomic per counter, so a tile's kept frames get consecutive slots.
// sel_cnt[tile] counts the kept pairs of a tile among this rank's frames (zeroed by the bounds kernel that ran before); the
// chunk that adds the first ones appends the tile to heavy[]: k_masked_sum_tiles gives those tiles to its worker workgroups and
// finishes every other tile with a constant fill.
// A frame shard [t0, t1) of the T-frame buffer owns unique frame u when it holds t = u or t = T - u (sym_in_range).
constexpr int SEL_TILES = 16, SEL_PH = 16, SEL_U = 9;   // 16 x 9 = 144 unique frames per chunk: one chunk at T = 256
RM_KERNEL __launch_bounds__(256) void k_select_pairs(const double *lo, const double *hi, int ntiles, int Th, int T, int t0, int t1,
                                                      CollapseState *st, unsigned int *list_a, unsigned int *list_b, int *slot_of,
                                                      int no_prune, double thr, int *sel_cnt, unsigned int *heavy, unsigned long long *xs_tab)
{
    RM_TRACE_SCOPE(4);
    __shared__ unsigned long long s_cnt[256], s_off[257], s_wave[4];
    __shared__ unsigned int s_base[3];
    const int ti = threadIdx.x & (SEL_TILES - 1), ph = threadIdx.x / SEL_TILES;
    const int tile = blockIdx.x * SEL_TILES + ti;
    const int u0 = blockIdx.y * (SEL_PH * SEL_U) + ph;
    // the pairs' bounds first: nothing below depends on them until the comparisons
    double l[SEL_U], h[SEL_U];
    bool mine[SEL_U];
#pragma unroll
    for (int k = 0; k < SEL_U; ++k) {
        const int u = u0 + SEL_PH * k;
        mine[k] = tile < ntiles && u < Th && sym_in_range(u, T, t0, t1);
        const size_t i = (size_t)u * ntiles + tile;
        l[k] = mine[k] ? lo[i] : 0.0;
        h[k] = mine[k] ? hi[i] : 0.0;
    }
    // margin and the bounds-only upper bound of top = max - (max - min) * thr (increasing in max and min
    // for 0 <= thr <= 1); every thread derives them from the reduced bounds
    const unsigned long long k_lb_max = fold_max_keys(st->lb_max_keys, st->lb_max_key), k_ub_min = fold_min_keys(st->ub_min_keys, st->ub_min_key);
    const unsigned long long k_ub_max = fold_max_keys(st->ub_max_keys, st->ub_max_key), k_lb_min = fold_min_keys(st->lb_min_keys, st->lb_min_key);
    double lb_max = f64_unkey(k_lb_max), ub_min = f64_unkey(k_ub_min);
    const double ub_max = f64_unkey(k_ub_max), lb_min = f64_unkey(k_lb_min);
    const double aa = ub_max < 0 ? -ub_max : ub_max, bb = lb_min < 0 ? -lb_min : lb_min;
    const double m = PRUNE_REL_MARGIN * (aa > bb ? aa : bb);
    {   // true raw values (lattice samples) bound raw.min() from above and raw.max() from below far better than the tile bounds
        const unsigned long long k_smn = fold_min_keys(st->smp_min_keys, ~0ull), k_smx = fold_max_keys(st->smp_max_keys, 0ull);
        if (k_smn != ~0ull) {
            const double s_mn = f64_unkey(k_smn) + 2 * m, s_mx = f64_unkey(k_smx) - 2 * m;
            ub_min = (s_mn < ub_min) ? s_mn : ub_min;
            lb_max = (s_mx > lb_max) ? s_mx : lb_max;
        }
    }
    const double mx_ = ub_max + m, mn_ = ub_min + m;
    const double top_ub = (mx_ - (mx_ - mn_) * thr) + m;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { st->margin = m; st->top_ub = top_ub; }
    unsigned int fC = 0, fD = 0;   // bit k: pair k is C / D
    unsigned long long cnt = 0;    // D | A << 20 | B << 40
#pragma unroll
    for (int k = 0; k < SEL_U; ++k) {
        if (!mine[k]) continue;
        const bool isC = no_prune || !(h[k] + m < lb_max - m) || !(l[k] - m > ub_min + m);
        const bool isD = no_prune || (l[k] - m < top_ub);
        fC |= (isC ? 1u : 0u) << k; fD |= (isD ? 1u : 0u) << k;
        cnt += (isD ? 1ull : 0ull) + (isC ? 1ull << 20 : 0ull) + ((isD && !isC) ? 1ull << 40 : 0ull);
    }
    // prefix sums in (tile, phase) order: thread j of the scan stands for tile j / 16, phase j % 16
    s_cnt[ti * SEL_PH + ph] = cnt;
    __syncthreads();
    unsigned long long total = 0;
    const unsigned long long ex = block_excl_scan_256(s_cnt[threadIdx.x], s_wave, total);
    s_off[threadIdx.x] = ex;
    if (threadIdx.x == 0) s_off[256] = total;
    if (threadIdx.x < 3) {   // the three reservations by three lanes: ONE round trip instead of three in a row (every thread holds `total`)
        const unsigned int tot = threadIdx.x == 0 ? (unsigned)(total & 0xfffffu) : threadIdx.x == 1 ? (unsigned)((total >> 20) & 0xfffffu) : (unsigned)(total >> 40);
        unsigned int *ctr = threadIdx.x == 0 ? &st->n_slots : threadIdx.x == 1 ? &st->n_list_a : &st->n_list_b;
        s_base[threadIdx.x] = tot ? atomicAdd(ctr, tot) : 0u;
    }
    __syncthreads();
    const unsigned long long mo = s_off[ti * SEL_PH + ph];
    unsigned int oD = s_base[0] + (unsigned)(mo & 0xfffffu), oA = s_base[1] + (unsigned)((mo >> 20) & 0xfffffu), oB = s_base[2] + (unsigned)(mo >> 40);
    bool new_heavy = false;           // this chunk adds the tile's first kept pairs
    if (ph == 0 && tile < ntiles) {   // kept pairs of this tile in this chunk
        const unsigned int tot = (unsigned)((s_off[(ti + 1) * SEL_PH] - s_off[ti * SEL_PH]) & 0xfffffu);
        new_heavy = tot && atomicAdd(&sel_cnt[tile], (int)tot) == 0;
    }
    if (threadIdx.x < 64) {   // (phase 0 = lanes 0 .. 15 of wave 0) ONE reservation on n_heavy for the workgroup's new tiles: at 4K x 512 every
                              // tile of a noisy stream is heavy, and 8 100 returning atomics on one address were most of the kernel's 58 us
        const unsigned long long mk = __ballot(new_heavy);
        if (mk) {
            const int lane = threadIdx.x, first = __builtin_ctzll(mk);
            unsigned int base = 0;
            if (lane == first) base = atomicAdd(&st->n_heavy, (unsigned)__popcll(mk));
            base = (unsigned)__shfl((int)base, first);
            if (new_heavy) heavy[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned)tile;
        }
    }
#pragma unroll
    for (int k = 0; k < SEL_U; ++k) {
        if (!mine[k]) continue;
        const int u = u0 + SEL_PH * k;
        const unsigned int i = (unsigned)u * (unsigned)ntiles + (unsigned)tile;
        const bool isC = (fC >> k) & 1u, isD = (fD >> k) & 1u;
        slot_of[slot_index(u, tile, Th)] = isD ? (int)oD : SLOT_PRUNED;
        if (xs_tab) xs_tab[slot_index(u, tile, Th)] = 0x00000000ffffffffull;   // XsEntry{XS_NONE, 0}: no exception record yet (rm_xstore.h)
        if (isD) ++oD;
        if (isC) list_a[oA++] = i;
        else if (isD) list_b[oB++] = i;
    }
}

// stage the level-S footprint of `tile` for frame t
__device__ __forceinline__ Region chain_stage(const ChainGeom &g, int tile, const double *cS_t, double *lds)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int S = g.S;
    const Region Rk = tile_region(g, tile, S);
    double *d = lds + g.lds_off[S];
    const int nw = Rk.x1 - Rk.x0 + 1, n = (Rk.y1 - Rk.y0 + 1) * nw, wS = g.w[S];
    const float inv_nw = 1.0f / (float)nw;
    for (int i = tid; i < n; i += nthr) {
        int r, c;
        split_rc(i, nw, inv_nw, r, c);
        d[i] = cS_t[(size_t)(Rk.y0 + r) * wS + Rk.x0 + c];
    }
    __syncthreads();
    return Rk;
}

// one pyrUp step inside LDS, level k (footprint Rk) -> level k-1: horizontal pass into the scratch buffer (every
// source row at the destination columns), then the vertical pass.  Wave w takes rows w, w + nwaves, ...; lane = column.
__device__ __forceinline__ Region chain_step(const ChainGeom &g, int tile, double *lds, int k, const Region &Rk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    const Region Rd = tile_region(g, tile, k - 1);
    const double *src = lds + g.lds_off[k];
    double *hb = lds + g.lds_hb[k], *dst = lds + g.lds_off[k - 1];
    const int sp = Rk.x1 - Rk.x0 + 1, srows = Rk.y1 - Rk.y0 + 1;      // source pitch / rows
    const int dw = Rd.x1 - Rd.x0 + 1, drows = Rd.y1 - Rd.y0 + 1;
    const int sh = g.h[k], sw = g.w[k];
    for (int c = lane; c < dw; c += 64) {
        const HTap t = make_htap(Rd.x0 + c, sw);
        const int oa = t.ia - Rk.x0, ob = t.ib - Rk.x0, oc = t.ic - Rk.x0;
        for (int r = wave; r < srows; r += nwaves) {
            const double *row = src + r * sp;
            hb[r * dw + c] = (row[oa] * t.wa + row[ob] * t.wb) + row[oc] * t.wc;
        }
    }
    __syncthreads();
    for (int r = wave; r < drows; r += nwaves) {
        const int y = Rd.y0 + r, i = y >> 1;                          // uniform per wave: no divergence
        const int r2 = ((i == sh - 1) ? i : i + 1) - Rk.y0;
        const int r1 = i - Rk.y0;
        if (y & 1) {
            for (int c = lane; c < dw; c += 64) dst[r * dw + c] = ((hb[r1 * dw + c] + hb[r2 * dw + c]) * 4) * (1.0 / 64);
        } else {
            const int r0 = ((i == 0) ? (sh > 1 ? 1 : 0) : i - 1) - Rk.y0;
            for (int c = lane; c < dw; c += 64)
                dst[r * dw + c] = (hb[r0 * dw + c] + hb[r1 * dw + c] * 6 + hb[r2 * dw + c]) * (1.0 / 64);
        }
    }
    __syncthreads();
    return Rd;
}

// stage the level-S footprint of `tile` for frame t, then run the chain S -> 1 inside LDS
__device__ __forceinline__ void chain_to_level1(const ChainGeom &g, int tile, const double *cS_t, double *lds)
{
    Region Rk = chain_stage(g, tile, cS_t, lds);
    RM_TRACE_MARK(5, 2);
    for (int k = g.S; k >= 2; --k) { Rk = chain_step(g, tile, lds, k, Rk); RM_TRACE_MARK(5, 3 + (g.S - k)); }
}

// level 1 (LDS) -> level 0 for this lane's column: out[j] = raw[t, y0 + j0 + j, x], j < NR (j0, NR even)
template <int NR>
__device__ __forceinline__ void level0_rows(const ChainGeom &g, const Region &R0, const Region &R1, const double *lds, int x, int j0,
                                            double (&out)[NR])
{
    const double *src = lds + g.lds_off[1];
    const int sp = R1.x1 - R1.x0 + 1;
    const int sh = g.h[1], sw = g.w[1];
    const HTap t = make_htap(x, sw);
    const int oa = t.ia - R1.x0, ob = t.ib - R1.x0, oc = t.ic - R1.x0;
    // horizontal values of source rows i0-1 .. i0+NR/2 (border rules applied by row index)
    const int i0 = (R0.y0 + j0) >> 1;  // R0.y0 is a multiple of CT_H, j0 is even
    double hv[NR / 2 + 2];
#pragma unroll
    for (int k = 0; k < NR / 2 + 2; ++k) {
        int i = i0 - 1 + k;
        int r = (i < 0) ? (sh > 1 ? 1 : 0) : (i > sh - 1 ? sh - 1 : i);
        const double *row = src + (r - R1.y0) * sp;
        hv[k] = (row[oa] * t.wa + row[ob] * t.wb) + row[oc] * t.wc;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        int k = (j >> 1) + 1;  // hv index of source row i = (y0+j0+j)/2
        if (j & 1) out[j] = ((hv[k] + hv[k + 1]) * 4) * (1.0 / 64);
        else out[j] = (hv[k - 1] + hv[k] * 6 + hv[k + 1]) * (1.0 / 64);
    }
}

// Filter-first collapse of levels too large for LDS (4K, skip 2; rm_front.hip front_filter): with X_l the band-passed Gaussian levels,
//     C_S = X_S - pyrUp^n(X_{L-1}),  n = L - 1 - S        (the telescoped collapse, see k_small_filter_first)
// for one 64 x 16 tile of level S per work item: the footprint of the tile at the coarsest level is staged in LDS, the pyrUp chain
// runs there exactly as in k_eval_pairs (`g` describes levels S .. L-1 as its levels 0 .. n), the last step lands in registers
// (lane = column, 16 rows) and is subtracted from the tile of X_S, requested before the chain starts.  X_S is read once, C_S
// written once, the intermediate levels U_l never exist (three k_pyr_up launches over the 2.1 GB level at 4K x 512 did 1.05 ms).
// Same per-pixel arithmetic as k_pyr_up (modes 0 and 1): bit-identical.
RM_KERNEL __launch_bounds__(64) void k_ff_collapse(const double *xS, const double *xL, ChainGeom g, int ntiles, int nitems, double *cS)
{
    HIP_DYNAMIC_SHARED(double, lds)
    const int lane = threadIdx.x;
    const int w0 = g.w[0];
    const size_t fs0 = (size_t)g.h[0] * w0, fsL = (size_t)g.h[g.S] * g.w[g.S];
    for (int c = blockIdx.x; c < nitems; c += gridDim.x) {
        const int u = c / ntiles, tile = c - u * ntiles;   // (wave-uniform)
        const Region R0 = tile_region(g, tile, 0), R1 = tile_region(g, tile, 1);
        const int x = R0.x0 + lane, rows = R0.y1 - R0.y0 + 1;
        const bool col = x <= R0.x1;
        const double *src = xS + (size_t)u * fs0 + (size_t)R0.y0 * w0 + x;
        double xs[CT_H];
#pragma unroll
        for (int j = 0; j < CT_H; ++j) xs[j] = (col && j < rows) ? src[(size_t)j * w0] : 0.0;
        chain_to_level1(g, tile, xL + (size_t)u * fsL, lds);
        if (col) {
            double v[CT_H];
            level0_rows<CT_H>(g, R0, R1, lds, x, 0, v);
            double *dst = cS + (size_t)u * fs0 + (size_t)R0.y0 * w0 + x;
#pragma unroll
            for (int j = 0; j < CT_H; ++j)
                if (j < rows) dst[(size_t)j * w0] = xs[j] - v[j];
        }
        __syncthreads();
    }
}

// the one evaluation pass: full-resolution values of every listed (unique frame, tile) pair, once.
// Exact raw.min()/raw.max() (transforms.py:185,187) come from here (list_a); on the sparse path the values of the pairs that can
// fall below `top` (list_a's kept pairs and all of list_b) are parked in their slot of `store` ([slot][row][lane], coalesced) for
// the masked time sum.  On the dense path (sum_is_dense) list_b is not touched and nothing is stored.
RM_KERNEL __launch_bounds__(64) void k_eval_pairs(const double *cS, ChainGeom g, int ntiles, const unsigned int *list_a, const unsigned int *list_b,
                                                   int *slot_of, CollapseState *st, double *store, SumPlan sp, int Th)
{
    RM_TRACE_SCOPE(5);
    HIP_DYNAMIC_SHARED(double, lds)
    // the first list entry is requested together with the list lengths (the list buffer is valid memory whatever they turn out
    // to be): one memory round trip less at the head of every workgroup's dependent chain
    const unsigned first_idx = list_a[blockIdx.x];
    const unsigned nA = st->n_list_a, nB = st->n_list_b;
    const bool dense = sum_is_dense(st, sp);
    const unsigned n = nA + (dense ? 0u : nB);
    const int lane = threadIdx.x;
    const double inf = __builtin_huge_val();
    const double top_ub = st->top_ub;   // upper bound of `top` from the tile bounds (k_select_pairs)
    double mn = inf, mx = -inf;
    for (unsigned c = blockIdx.x; c < n; c += gridDim.x) {
        RM_TRACE_MARK(5, 0);
        const unsigned raw_idx = c < nA ? (c == blockIdx.x ? first_idx : list_a[c]) : list_b[c - nA];
        const unsigned idx = (unsigned)uniform((int)raw_idx);   // wave-uniform: the tile geometry stays in scalar registers
        const int u = idx / ntiles, tile = idx - u * ntiles;
        const int slot = dense ? SLOT_PRUNED : uniform(slot_of[slot_index(u, tile, Th)]);   // (needed after the chain: requested now)
        const Region R0 = tile_region(g, tile, 0), R1 = tile_region(g, tile, 1);
        RM_TRACE_MARK(5, 1);
        chain_to_level1(g, tile, cS + (size_t)u * g.h[g.S] * g.w[g.S], lds);
        RM_TRACE_MARK(5, 6);
        int x = R0.x0 + lane;
        double pmn = inf;   // minimum of this pair's tile
        double v[CT_H];
        const int rows = R0.y1 - R0.y0 + 1;
        if (x <= R0.x1) {
            level0_rows<CT_H>(g, R0, R1, lds, x, 0, v);
#pragma unroll
            for (int j = 0; j < CT_H; ++j)
                if (j < rows) { pmn = (v[j] < pmn) ? v[j] : pmn; mx = (v[j] > mx) ? v[j] : mx; }
        }
        mn = (pmn < mn) ? pmn : mn;
        RM_TRACE_MARK(5, 7);
        if (slot != SLOT_PRUNED) {   // wave-uniform
            pmn = wave_min(pmn);
            // nothing of this tile can fall below top (top <= top_ub): every pixel adds `min`, exactly like a pruned pair --
            // no values to park, and the sum pass never sees the frame
            if (pmn >= top_ub) {
                if (lane == 0) slot_of[slot_index(u, tile, Th)] = SLOT_PRUNED;
            } else if (x <= R0.x1) {
                double *d = store + (size_t)slot * (CT_H * CT_W) + lane;
#pragma unroll
                for (int j = 0; j < CT_H; ++j) d[j * CT_W] = v[j];
            }
        }
        RM_TRACE_MARK(5, 8);
        __syncthreads();
    }
    mn = wave_min(mn); mx = wave_max(mx);
    RM_TRACE_MARK(5, 9);
    if (lane == 0 && blockIdx.x < n) {
        // striped, and skipped when they cannot change the result
        const unsigned long long kmn = f64_key(mn), kmx = f64_key(mx);
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->min_keys, st->max_keys, sp_, kmn, kmx);
    }
}

// transforms.py:184-189: min, max, top = max - (max - min) * threshold
RM_KERNEL __launch_bounds__(NSTRIPE) void k_finish_minmax(CollapseState *st, double threshold)
{
    const unsigned long long kmn = fold_min_keys(st->min_keys, st->min_key), kmx = fold_max_keys(st->max_keys, st->max_key);
    if (threadIdx.x != 0) return;
    double mn = f64_unkey(kmn), mx = f64_unkey(kmx);
    st->min_val = mn; st->max_val = mx;
    st->top = mx - (mx - mn) * threshold;
}

// frame-sharded calibration: the exact extrema of this rank's frames leave as {-min, max} (one all-reduce(MAX)
// serves both) and the global pair comes back the same way
RM_KERNEL __launch_bounds__(NSTRIPE) void k_export_minmax(const CollapseState *st, double *negmin_max)
{
    const double inf = __builtin_huge_val();
    const unsigned long long kmn = fold_min_keys(st->min_keys, st->min_key), kmx = fold_max_keys(st->max_keys, st->max_key);
    if (threadIdx.x != 0) return;
    negmin_max[0] = (kmn == ~0ull) ? -inf : -f64_unkey(kmn);
    negmin_max[1] = (kmx == 0ull) ? -inf : f64_unkey(kmx);
}
RM_KERNEL __launch_bounds__(NSTRIPE) void k_import_minmax(CollapseState *st, const double *negmin_max)
{
    st->min_keys[threadIdx.x] = ~0ull; st->max_keys[threadIdx.x] = 0ull;   // the global pair replaces this rank's stripes
    if (threadIdx.x != 0) return;
    st->min_key = f64_key(-negmin_max[0]);
    st->max_key = f64_key(negmin_max[1]);
}

// pass D: heat_sum[y,x] = sum_t (raw >= top ? min : raw), sequential in t (np.average order, base.py:562).
// Pruned pairs add `min`; kept pairs read their values back from `store`.
//
// One launch of `nworkers` 256-thread workgroups:
//   * WORKER items.  Item i is (heavy[i / MS_Q], row group i % MS_Q): the tile's ordered list of kept frames is compacted by
//     ballot / popcount (every worker of the tile repeats that cheap, parallel step), then thread (wave, lane) owns pixel
//     (row MS_RQ * q + wave, column lane) and walks the kept frames in batches of MS_B loads issued one batch ahead.  The
//     longest dependent chain of the launch is therefore ceil(kept / MS_B) round trips of ONE tile row group, not
//     256 / 6 of a whole tile (the earlier form: one 256-thread workgroup per tile, 4 rows per lane, 6-frame batches --
//     its heaviest tile alone took 29 us and every empty tile 8-10 us in three rounds).
//   * FILL.  A tile without a kept pair (sel_cnt[tile] == 0: 94 % of the tiles of the synthetic 1080p stream) is one
//     constant -- T sequential additions of `min` -- computed once per workgroup and stored into every such tile of its
//     share.  The workgroups that found no worker item do the filling (they are free at once; separate fill workgroups
//     queued behind the workers' registers and started 5-14 us late); when every workgroup has items, all of them fill
//     after their items.
// Dynamic LDS: s_kt[T] and s_ks[T], the tile's kept frames in order and their value store slots.
// Frames are walked in time order t = t_first .. t_end - 1 (np.average's order); frame t's pair is that of its unique frame
// sym_frame(t, T).  On the dense path (sum_is_dense) the kernel returns at once: k_dense_sum takes the sum.
constexpr int MAX_T = 4096;
constexpr int MS_Q = 4;              // row groups (worker items) per heavy tile
constexpr int MS_RQ = CT_H / MS_Q;   // rows per worker == waves per workgroup
#ifndef RM_MS_B
#define RM_MS_B 16
#endif
constexpr int MS_B = RM_MS_B;        // kept frames per batch (32: 220 VGPRs, two waves per SIMD -- measured 32 us against 20)

RM_KERNEL __launch_bounds__(64 * MS_RQ) void k_masked_sum_tiles(int t_first, int t_end, int T, int ntiles, int W0, int H0,
                                                          const int *slot_of, const double *store,
                                                          CollapseState *st, double threshold, double *heat_sum, int avg_T,
                                                          int *tile_nkept, const int *sel_cnt, const unsigned int *heavy, int nworkers,
                                                          SumPlan sp, int *unserved_host)
{
    RM_TRACE_SCOPE(6);
    HIP_DYNAMIC_SHARED(int, s_kt)     // kept frames of the tile, in order; then their slots
    int *s_ks = s_kt + T;
    __shared__ int s_wcnt[MS_RQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W0 + CT_W - 1) / CT_W;
    // everything that does not depend on the state is requested before it: the tile of this workgroup's first item and its
    // first slot_of column (speculatively: heavy[] and slot_of[] are valid memory whatever n_heavy turns out to be)
    const int tile0 = (int)(heavy[blockIdx.x / MS_Q] % (unsigned)ntiles);
    int slot0 = SLOT_PRUNED;
    if (t_first + tid < t_end) slot0 = slot_of[slot_index(sym_frame(t_first + tid, T), tile0, sym_frames(T))];
    const int nitems = (int)st->n_heavy * MS_Q;
    if (unserved_host && blockIdx.x == 0 && threadIdx.x == 0) unserved_host[1] = (int)st->n_slots;   // (pinned: how many pairs this call's selection kept -- rm_locate's refine_hint)
    if (sum_is_dense(st, sp)) {   // (uniform over the grid: k_dense_sum takes the sum)
        // unserved_host (pinned, nullable): no dense kernel follows on the stream -- the caller synchronises anyway and enqueues it
        // itself when it finds this word set (rm_locate: the rare value-store overflow costs the common case no launch)
        if (unserved_host && blockIdx.x == 0 && tid == 0) *unserved_host = 1;
        return;
    }
    // transforms.py:184-189: min, max, top = max - (max - min) * threshold
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;
    RM_TRACE_MARK(6, 0);
    if (blockIdx.x == 0 && tid == 0) {
        st->min_val = min_val; st->max_val = max_val; st->top = top;
    }
    // avg_T > 0 (the whole buffer is summed here): write np.average = sum / T (base.py:562) and reduce the
    // heatmap's min / max for the normalisation (base.py:563) on the way out
    const double cnt = (double)avg_T;
    for (int item = (int)blockIdx.x; item < nitems; item += nworkers) {
        const bool first = item == (int)blockIdx.x;
        const int tile = first ? tile0 : (int)heavy[item / MS_Q], q = item % MS_Q;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        // frames [t_first, t_end): the whole buffer, or this rank's frame shard (partial time sums add up across ranks).
        // Ordered compaction of the frames that are not pruned (by the selection, or by the evaluation pass when the whole
        // tile turned out >= top_ub): ballot + prefix popcount, 256 frames per round.
        int nkept = 0;
        for (int c0 = t_first; c0 < t_end; c0 += 64 * MS_RQ) {
            const int t = c0 + tid;
            int slot = SLOT_PRUNED;
            if (first && c0 == t_first) slot = slot0;
            else if (t < t_end) slot = slot_of[slot_index(sym_frame(t, T), tile, sym_frames(T))];
            const bool kept = slot != SLOT_PRUNED;
            const unsigned long long m = __ballot(kept);
            if (lane == 0) s_wcnt[wave] = __popcll(m);
            __syncthreads();
            int off = nkept, tot = 0;
#pragma unroll
            for (int w = 0; w < MS_RQ; ++w) { const int c = s_wcnt[w]; off += (w < wave) ? c : 0; tot += c; }
            if (kept) { const int pos = off + __popcll(m & ((1ull << lane) - 1ull)); s_kt[pos] = t; s_ks[pos] = slot; }
            nkept += tot;
            __syncthreads();
        }
        if (tid == 0 && q == 0 && tile_nkept) tile_nkept[tile] = nkept;   // 0: every pixel of the tile ends up as the same constant
        RM_TRACE_MARK(6, 1);
        const int x = tx * CT_W + lane;
        const int row = q * MS_RQ + wave, y = ty * CT_H + row;
        const bool active = x < W0 && y < H0;
        const double *mine = store + (size_t)row * CT_W + lane;   // + slot * 1024: this pixel in the pair parked in `slot`
        double acc = 0.0;
        // a batch = MS_B kept frames: their frame numbers and (one batch ahead) values sit in registers, so the serial
        // part below touches neither LDS nor memory (per-frame LDS look-ups were 2/3 of the heaviest worker's time)
        double nxt[MS_B];
        int ktn[MS_B];
#pragma unroll
        for (int b = 0; b < MS_B; ++b) nxt[b] = 0.0;
        auto fetch = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < MS_B; ++b) {
                const int i = ib + b;
                const bool ok = i < nkept;
                ktn[b] = ok ? s_kt[i] : t_end;
                if (ok && active) nxt[b] = mine[(size_t)s_ks[i] * (CT_H * CT_W)];
            }
        };
        fetch(0);
        RM_TRACE_MARK(6, 2);
        int t_done = t_first;
        for (int ib = 0; ib < nkept; ib += MS_B) {
            double cur[MS_B];
            int kt[MS_B];
#pragma unroll
            for (int b = 0; b < MS_B; ++b) { cur[b] = nxt[b]; kt[b] = ktn[b]; }
            fetch(ib + MS_B);
#pragma unroll
            for (int b = 0; b < MS_B; ++b) {
                if (ib + b < nkept) {
                    const int t_stop = uniform(kt[b]);     // frames [t_done, t_stop) are pruned
                    for (int t = t_done; t < t_stop; ++t) acc = acc + min_val;
                    if (active) acc = acc + ((cur[b] >= top) ? min_val : cur[b]);
                    t_done = t_stop + 1;
                }
            }
            RM_TRACE_MARK(6, 3 + ib / MS_B);
        }
        for (int t = t_done; t < t_end; ++t) acc = acc + min_val;
        RM_TRACE_MARK(6, 12);
        double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
        if (active) {
            const double v = avg_T > 0 ? acc / cnt : acc;
            heat_sum[(size_t)y * W0 + x] = v;
            hmn = v; hmx = v;
        }
        if (avg_T > 0) {
            block_minmax(hmn, hmx);
            if (tid == 0) {
                const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
                const int sp = blockIdx.x & (NSTRIPE - 1);
                striped_min_max(st->heat_min_keys, st->heat_max_keys, sp, kmn, kmx);
            }
        }
        RM_TRACE_MARK(6, 13);
        __syncthreads();   // s_kt is rewritten by the next item
    }
    // FILL: by the workgroups without items when there are any, by every workgroup otherwise
    const int idle = nworkers - min(nitems, nworkers);
    const int nfill = idle > 0 ? idle : nworkers;
    const int fid = idle > 0 ? (int)blockIdx.x - nitems : (int)blockIdx.x;
    if (fid < 0) return;
    double lead = 0.0;
    for (int t = t_first; t < t_end; ++t) lead = lead + min_val;
    const double v = avg_T > 0 ? lead / cnt : lead;
    bool any = false;
    constexpr int FU = 4;    // tiles whose kept-pair counts are requested together
    for (int base = fid; base < ntiles; base += FU * nfill) {
        int cntk[FU];
#pragma unroll
        for (int k = 0; k < FU; ++k) { const int tile = base + k * nfill; cntk[k] = tile < ntiles ? sel_cnt[tile] : 1; }
#pragma unroll
        for (int k = 0; k < FU; ++k) {
            const int tile = base + k * nfill;
            if (cntk[k] != 0) continue;               // past the end, or a worker sums this tile
            any = true;
            const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
            const int x = tx * CT_W + lane, y0 = ty * CT_H;
            if (x < W0) {
#pragma unroll
                for (int j = 0; j < CT_H / MS_RQ; ++j) {
                    const int y = y0 + wave * (CT_H / MS_RQ) + j;
                    if (y < H0) heat_sum[(size_t)y * W0 + x] = v;
                }
            }
            if (tid == 0 && tile_nkept) tile_nkept[tile] = 0;     // 0: every pixel of the tile is the same constant
        }
    }
    if (any && tid == 0 && avg_T > 0) {
        const unsigned long long kv = f64_key(v);
        const int sp = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->heat_min_keys, st->heat_max_keys, sp, kv, kv);
    }
}

// ----------------------------------------------------------------------------------------
// plain (materialised) forms: global min/max, mask, time sum  -- transforms.py:184-192, base.py:562
// ----------------------------------------------------------------------------------------
RM_KERNEL __launch_bounds__(256) void k_minmax_plain(const double *a, size_t n, CollapseState *st)
{
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = a[i];
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
    block_minmax(mn, mx);
    if (threadIdx.x == 0) {
        atomicMin(&st->min_key, f64_key(mn));
        atomicMax(&st->max_key, f64_key(mx));
    }
}

RM_KERNEL __launch_bounds__(256) void k_mask_plain(const double *raw, size_t n, const CollapseState *st, double *masked)
{
    const double top = st->top, mn = st->min_val;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = raw[i];
        masked[i] = (v >= top) ? mn : v;
    }
}

// heat_sum[p] = sum_t (raw[t,p] >= top ? min : raw[t,p])   (sequential in t); raw holds the sym_frames(T) unique frames
RM_KERNEL __launch_bounds__(256) void k_masked_sum_plain(const double *raw, int T, size_t npix, const CollapseState *st,
                                                          double *heat_sum)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const double top = st->top, mn = st->min_val;
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
        double v = raw[(size_t)sym_frame(t, T) * npix + p];
        acc = acc + ((v >= top) ? mn : v);
    }
    heat_sum[p] = acc;
}

// frames T/2+1 .. T-1 of a [T, npix] array from their mirror images (sym_frame): dst[t] = dst[T - t]
RM_KERNEL __launch_bounds__(256) void k_mirror_frames(double *a, int T, size_t npix)
{
    const int t = sym_frames(T) + (int)blockIdx.y;   // t in (T/2, T)
    const double *src = a + (size_t)(T - t) * npix;
    double *dst = a + (size_t)t * npix;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// np.average(video, axis=0) of a [T, npix] array of any frame dtype (base.py:562, 579, 587, 589): float64 sum in t
// order, then / T -- the order numpy's pairwise-free axis-0 reduction uses.
template <typename Tin>
__global__ __launch_bounds__(256) void k_time_average(const Tin *v, int T, size_t npix, double *out)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    double acc = 0.0;
    for (int t = 0; t < T; ++t) acc = acc + load_px(v, (size_t)t * npix + p);
    out[p] = acc / (double)T;
}

// ----------------------------------------------------------------------------------------
// base.py:562-566: avg = sum / T ; normalise ; float_to_uint8 ; threshold
// ----------------------------------------------------------------------------------------
RM_KERNEL __launch_bounds__(256) void k_heat_avg_minmax(const double *heat_sum, size_t npix, int T, double *heat,
                                                         CollapseState *st)
{
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    const double cnt = (double)T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        double v = heat_sum[i] / cnt;
        heat[i] = v;
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
    block_minmax(mn, mx);
    if (threadIdx.x == 0) {
        atomicMin(&st->heat_min_key, f64_key(mn));
        atomicMax(&st->heat_max_key, f64_key(mx));
    }
}

// min/max of an existing heatmap (rm_heatmap_to_roi entry point)
// reset of the heatmap extrema in the state (in front of k_heat_minmax / k_heat_avg_minmax / a sum kernel that reduces them)
RM_KERNEL __launch_bounds__(NSTRIPE) void k_heat_state_init(CollapseState *st)
{
    st->heat_min_keys[threadIdx.x] = ~0ull; st->heat_max_keys[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) { st->heat_min_key = ~0ull; st->heat_max_key = 0ull; }
}

RM_KERNEL __launch_bounds__(256) void k_heat_minmax(const double *heat, size_t npix, CollapseState *st)
{
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        double v = heat[i];
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
    block_minmax(mn, mx);
    if (threadIdx.x == 0) {
        atomicMin(&st->heat_min_key, f64_key(mn));
        atomicMax(&st->heat_max_key, f64_key(mx));
    }
}

// `bits` receives the thresholded image bit-packed (bit p & 63 of word p >> 6 = pixel p, row-major): 1/8 of a byte per
// pixel, stored straight into pinned, device-mapped host memory.  `row_any[y]` (pinned bytes) is set for every row that holds
// foreground: the host contour stage then reads only those rows of `bits` -- the breathing region covers ~1/5 of a 1080p frame,
// and reading memory the device has just written (lines no host cache holds) was most of that stage.  The host zeroes the
// flags AND the rows it read after use, so the image is all-zero between calls and the kernel stores only the words that
// have a bit set (~12 KB instead of 259 KB over PCIe: the launch's end-of-kernel flush of host-memory writes shrinks with it).  (Tried and dropped: a sparse list of the non-zero words with a `done` word the host spins on instead of the
// runtime's completion query -- the kernel's own hand-off cost 14 us more, and a stream the runtime never sees complete
// makes the NEXT launch ~100 us slower.)
struct alignas(16) CclBox { int minx, maxx, maxy, cnt; };   // bounding box of a labelled component (rm_ccl.h), indexed by its root; cnt: 2 * pixels - cracks (rm_ccl.h ccl_piece_2n_minus_p)

// tile_const (nullable; needs W % 64 == 0): tile_nkept of the sum kernel that wrote `heat` -- 0 for a 64 x 16 tile every pixel of which
// is the same constant (96 % of the tiles of the synthetic 1080p stream): such a word takes its ONE value from a wave-uniform load
// and the 16.6 MB heatmap is read only where it varies
RM_KERNEL __launch_bounds__(256) void k_heat_to_u8(const double *heat, size_t npix, int W, const CollapseState *st,
                                                    int threshold, uint8_t *avg_u8, uint8_t *binary,
                                                    unsigned long long *bits, uint8_t *row_any,
                                                    unsigned long long *bits_dev, int *ccl_label, CclBox *ccl_box,
                                                    unsigned int *ccl_counters, const int *tile_const = nullptr)
{
    RM_TRACE_SCOPE(7);
    if (ccl_counters && blockIdx.x == 0 && threadIdx.x == 0) ccl_counters[0] = 0;   // k_ccl_bbox reserves the root list's slots there
    const int lane = threadIdx.x & 63;
    // `base` is the first pixel of this wave's 64-pixel group: the same for all lanes, so the ballot is complete.
    // HU groups per trip: their heat values are requested together and BEFORE the extrema are folded from the state
    constexpr int HU = 4;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t first = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63u);
    double hv[HU];
    const int tiles_x = (W + CT_W - 1) / CT_W;
    auto fetch = [&](size_t base0) __attribute__((always_inline)) {
        if (tile_const) {
            // the word's tile flag and its first value (both wave-uniform), then the 64 values only where the tile is not a constant
            int cst[HU];
            double h0[HU];
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                const size_t base = base0 + k * stride;
                const size_t bc = base < npix ? base : 0;
                const int y = (int)(bc / (size_t)W), x = (int)(bc - (size_t)y * W);
                cst[k] = tile_const[(y / CT_H) * tiles_x + x / CT_W];
                h0[k] = heat[bc];
            }
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                const size_t i = base0 + k * stride + lane;
                hv[k] = (cst[k] != 0 && i < npix) ? heat[i] : h0[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < HU; ++k) { const size_t i = base0 + k * stride + lane; hv[k] = i < npix ? heat[i] : 0.0; }
        }
    };
    fetch(first);
    const double mn = f64_unkey(fold_min_keys(st->heat_min_keys, st->heat_min_key));
    const double mx = f64_unkey(fold_max_keys(st->heat_max_keys, st->heat_max_key));
    const double range = mx - mn;
    for (size_t base0 = first; base0 < npix; base0 += HU * stride) {
        if (base0 != first) fetch(base0);
#pragma unroll
        for (int k = 0; k < HU; ++k) {
            const size_t base = base0 + k * stride, i = base + lane;
            if (base >= npix) break;                          // wave-uniform
            uint8_t b = 0;
            if (i < npix) {
                double nrm = (hv[k] - mn) / range;            // base.py:563 (NaN when the heatmap is flat)
                uint8_t u = f64_to_u8_trunc(nrm * 255);       // transforms.py:26-29
                b = (u > threshold) ? 255 : 0;                // cv2.threshold THRESH_BINARY, base.py:566
                if (avg_u8) avg_u8[i] = u;
                if (binary) binary[i] = b;
            }
            const unsigned long long m = __ballot(b != 0);
            if (bits_dev) {   // device labelling of the components (rm_ccl.h) follows: every word, and the start state of its
                              // union-find -- the ballot IS the pixel's word, so no separate pass has to read it back
                if (lane == 0) bits_dev[base >> 6] = m;
                if (b && ccl_label) {   // (null: k_ccl_tile builds the start state itself, in LDS)
                    // label = first pixel of the run of ones that ends here (inside this word, not crossing the row start).  Only
                    // such a first pixel can end up a root, and only it carries a box: that of its piece of the run
                    const unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
                    int run0 = zeros_below ? 64 - __builtin_clzll(zeros_below) : 0;
                    const unsigned int y = (unsigned int)i / (unsigned int)W, x = (unsigned int)i - y * (unsigned int)W;
                    if (lane - run0 > (int)x) run0 = lane - (int)x;
                    ccl_label[i] = (int)i - (lane - run0);
                    if (run0 == lane) {
                        const unsigned long long zeros_above = ~(m >> lane);              // bit k: pixel i + k is background (or past the word)
                        int len = zeros_above ? __builtin_ctzll(zeros_above) : 64;        // (lane 0 of a full word: 64 ones)
                        if (len > 64 - lane) len = 64 - lane;
                        if (len > W - (int)x) len = W - (int)x;                           // the row ends inside the word
                        CclBox e; e.minx = (int)x; e.maxx = (int)x + len - 1; e.maxy = (int)y; e.cnt = 0;
                        ccl_box[i] = e;
                    }
                }
            }
            if (lane == 0 && bits && m) {   // the host keeps the image all-zero between calls: only set words travel
                bits[base >> 6] = m;
                if (row_any) {   // the group may straddle row ends: flag every row it touches (a superset is fine)
                    const size_t last = (base + 63 < npix ? base + 63 : npix - 1);
                    for (size_t y = base / (size_t)W; y <= last / (size_t)W; ++y) row_any[y] = 1;
                }
            }
        }
    }
}

// The same for images whose rows are whole 64-pixel words (W % 64 == 0: 1080p, 720p, 4K), one workgroup per image row.  Beside the
// packed image the host gets ONE 8-byte record per row that holds foreground,
//     rec[y] = first | last << 16 | min(runs, 0xffff) << 32 | 1 << 48        (first / last foreground column, runs of foreground)
// so the host's one-blob rule (rm_contour.cpp simple_shape_row_records: one run per row, neighbouring runs touching => one hole-free
// 8-connected component => the ROI is the bounding box of the runs, base.py:568-575) reads H x 8 bytes instead of hunting through
// the image rows the device has just written (lines no host cache holds: ~10 us of the 48 us the GPU idles between two synchronous
// locate() calls at 1080p).  The image words still travel for the images the rule does not settle (the host then follows the
// borders as before).  Wave w of the row takes the words w, w + 4, ...; the words meet in LDS, wave 0 folds them.
constexpr int HR_MAXW = 512;   // words per row the row kernel takes (W <= 32768)
RM_KERNEL __launch_bounds__(256) void k_heat_rows_u8(const double *heat, int H, int W, const CollapseState *st, int threshold, uint8_t *avg_u8,
                                                      uint8_t *binary, unsigned long long *bits, unsigned long long *rec, const int *tile_const)
{
    RM_TRACE_SCOPE(7);
    __shared__ unsigned long long s_words[HR_MAXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = blockIdx.x, nw = W >> 6;
    const int tiles_x = (W + CT_W - 1) / CT_W;
    const double *row = heat + (size_t)y * W;
    constexpr int HU = 4;
    const int trow = (y / CT_H) * tiles_x;
    double hv[HU];
    auto fetch = [&](int j0) __attribute__((always_inline)) {   // words j0, j0 + 4, .. of this wave, requested together
        if (tile_const) {
            int cst[HU];
            double h0[HU];
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                const int j = j0 + 4 * k, jc = j < nw ? j : 0;
                cst[k] = tile_const[trow + (jc * 64) / CT_W];
                h0[k] = row[jc * 64];
            }
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                const int j = j0 + 4 * k;
                hv[k] = (cst[k] != 0 && j < nw) ? row[j * 64 + lane] : h0[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < HU; ++k) { const int j = j0 + 4 * k; hv[k] = j < nw ? row[j * 64 + lane] : 0.0; }
        }
    };
    fetch(wave);
    const double mn = f64_unkey(fold_min_keys(st->heat_min_keys, st->heat_min_key));
    const double mx = f64_unkey(fold_max_keys(st->heat_max_keys, st->heat_max_key));
    const double range = mx - mn;
    for (int j0 = wave; j0 < nw; j0 += 4 * HU) {
        if (j0 != wave) fetch(j0);
#pragma unroll
        for (int k = 0; k < HU; ++k) {
            const int j = j0 + 4 * k;
            if (j >= nw) break;                               // wave-uniform
            const size_t i = (size_t)y * W + (size_t)j * 64 + lane;
            const double nrm = (hv[k] - mn) / range;          // base.py:563 (NaN when the heatmap is flat)
            const uint8_t u = f64_to_u8_trunc(nrm * 255);     // transforms.py:26-29
            const uint8_t b = (u > threshold) ? 255 : 0;      // cv2.threshold THRESH_BINARY, base.py:566
            if (avg_u8) avg_u8[i] = u;
            if (binary) binary[i] = b;
            const unsigned long long m = __ballot(b != 0);
            if (lane == 0) {
                s_words[j] = m;
                if (m) bits[i >> 6] = m;                      // the host keeps the image all-zero between calls: only set words travel
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    int first = 0x7fffffff, last = -1, runs = 0;
    for (int c = 0; c < nw; c += 64) {
        const int j = c + lane;
        const unsigned long long m = j < nw ? s_words[j] : 0ull;
        const unsigned long long prev = (j > 0 && j < nw) ? (s_words[j - 1] >> 63) : 0ull;
        if (m) {
            const int a = j * 64 + __builtin_ctzll(m), b = j * 64 + 63 - __builtin_clzll(m);
            first = a < first ? a : first;
            last = b > last ? b : last;
            runs += __popcll(m & ~((m << 1) | prev));
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int f2 = __shfl_xor(first, d), l2 = __shfl_xor(last, d), r2 = __shfl_xor(runs, d);
        first = f2 < first ? f2 : first; last = l2 > last ? l2 : last; runs += r2;
    }
    if (lane == 0 && runs > 0)
        rec[y] = (unsigned long long)first | ((unsigned long long)last << 16) | ((unsigned long long)(runs > 0xffff ? 0xffff : runs) << 32) | (1ull << 48);
    (void)H;
}

// ----------------------------------------------------------------------------------------
// Sparse heatmap exchange between GPUs (one stream per GPU, dist.locate_streams).  A stream's heatmap is ONE
// constant -- the time average of `min` -- in every tile none of whose frames survived the pruning (98 % of the
// tiles on the synthetic video), so instead of all-reducing 16.6 MB per GPU over xGMI each rank sends a packet
//   header { u32 count, u32 reserved, f64 background, 2 x f64 reserved } , f64 tile index [cap] , f64 values [cap][16][64]
// (0.5 MB at cap = 64) through ONE all-gather, and every rank rebuilds  sum_r heat_r  in rank order.
// count > cap (or no pruning information) makes every rank fall back to the dense all-reduce.
// ----------------------------------------------------------------------------------------
constexpr int SP_HDR = 4;  // doubles

// background constant = the heatmap value of the first tile without kept frames (header double 1); none -> overflow.
// One wave, 64 tiles per ballot (the first tile is almost always one of them).
struct alignas(16) F64Pair { double a, b; };
constexpr unsigned int SP_DENSE_ONLY = 0xffffffffu;   // header count: this rank has no sparse form, use the dense exchange
RM_KERNEL __launch_bounds__(64) void k_sparse_background(const double *heat, int W, int tiles_x, int ntiles, const int *tile_nkept,
                                                          int cap, double *packet)
{
    const int lane = threadIdx.x;
    int first = ntiles;
    for (int base = 0; base < ntiles && first == ntiles; base += 64) {
        const int i = base + lane;
        const unsigned long long m = __ballot(i < ntiles && tile_nkept[i] == 0);
        if (m) first = base + __builtin_ctzll(m);
    }
    if (lane != 0) return;
    packet[0] = 0.0; packet[1] = 0.0; packet[2] = 0.0; packet[3] = 0.0;   // header: count = 0 before k_sparse_pack counts
    if (first >= ntiles) { *reinterpret_cast<unsigned int *>(packet) = SP_DENSE_ONLY; return; }
    const int ty = first / tiles_x, tx = first - ty * tiles_x;
    packet[1] = heat[(size_t)ty * CT_H * W + (size_t)tx * CT_W];
}

// a tile travels only if one of its pixels differs from the background (a tile with kept frames whose values were
// all masked ends up as the same constant, bit for bit: the same sequence of additions of `min`)
RM_KERNEL __launch_bounds__(256) void k_sparse_pack(const double *heat, int H, int W, int tiles_x, const int *tile_nkept, int cap,
                                                     double *packet)
{
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * CT_H, x0 = tx * CT_W;
    if (tile_nkept[tile] == 0) return;
    unsigned int *count = reinterpret_cast<unsigned int *>(packet);
    if (*(volatile unsigned int *)count == SP_DENSE_ONLY) return;
    const double c = packet[1];
    double v[CT_H * CT_W / 256];
    bool differs = false;
#pragma unroll
    for (int k = 0; k < CT_H * CT_W / 256; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int y = y0 + i / CT_W, x = x0 + (i & (CT_W - 1));
        const bool in = y < H && x < W;
        v[k] = in ? heat[(size_t)y * W + x] : c;
        differs = differs || (in && v[k] != c);
    }
    __shared__ unsigned int s_slot;
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if (__ballot(differs) != 0ull && (threadIdx.x & 63) == 0) s_any = 1;
    __syncthreads();
    if (!s_any) return;
    if (threadIdx.x == 0) s_slot = atomicAdd(count, 1u);
    __syncthreads();
    const unsigned int slot = s_slot;
    if (slot >= (unsigned)cap) return;   // overflow: count says so, the receiver falls back
    if (threadIdx.x == 0) packet[SP_HDR + slot] = (double)tile;
    double *dst = packet + SP_HDR + cap + (size_t)slot * (CT_H * CT_W);
#pragma unroll
    for (int k = 0; k < CT_H * CT_W / 256; ++k) dst[threadIdx.x + 256 * k] = v[k];
}

// ONE workgroup prepares the merge: map[r][tile] = slot of `tile` in rank r's packet or -1, any[tile] = 1 when some rank
// sent the tile, flag_host[0] = 1 when some rank overflowed, flag_host[1] = the largest tile count a rank needed
// (pinned host words: the caller reads them after the ROI stage's synchronisation), and the stripes the merge
// kernel reduces the fused heatmap's extrema into
RM_KERNEL __launch_bounds__(256) void k_sparse_index(const double *packets, size_t packet_doubles, int world, int cap, int ntiles,
                                                      int *map, int *any, int *flag_host, CollapseState *st, int avg_T)
{
    for (int i = threadIdx.x; i < world * ntiles; i += 256) map[i] = -1;
    for (int i = threadIdx.x; i < ntiles; i += 256) any[i] = 0;
    if (threadIdx.x < NSTRIPE) { st->heat_min_keys[threadIdx.x] = ~0ull; st->heat_max_keys[threadIdx.x] = 0ull; }
    if (threadIdx.x == 0) { st->heat_min_key = ~0ull; st->heat_max_key = 0ull; }
    __syncthreads();
    int over = 0;
    unsigned int need = 0;
    for (int r = 0; r < world; ++r) {
        const double *pk = packets + (size_t)r * packet_doubles;
        const unsigned int count = *reinterpret_cast<const unsigned int *>(pk);
        if (count != SP_DENSE_ONLY && count > need) need = count;
        if (count > (unsigned)cap) { over = 1; continue; }
        for (unsigned int j = threadIdx.x; j < count; j += 256) {
            const int tile = (int)pk[SP_HDR + j];
            if (tile >= 0 && tile < ntiles) { map[(size_t)r * ntiles + tile] = (int)j; any[tile] = 1; }
        }
    }
    if (threadIdx.x == 0) { flag_host[0] = over; flag_host[1] = (int)need; }
    // the tiles nobody sent are one constant: the backgrounds summed in rank order (the per-pixel arithmetic, done once)
    double bg = 0.0;
    for (int r = 0; r < world; ++r) {
        const double v = packets[(size_t)r * packet_doubles + 1];
        bg = (r == 0) ? v : bg + v;
    }
    if (avg_T > 0) bg = bg / (double)avg_T;
    __shared__ int s_const;
    if (threadIdx.x == 0) s_const = 0;
    __syncthreads();   // also orders the any[] writes above before the reads below
    int mine = 0;
    for (int i = threadIdx.x; i < ntiles; i += 256) mine |= (any[i] == 0);
    if (mine) s_const = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        st->sp_bg = bg;
        if (s_const) { st->heat_min_keys[0] = f64_key(bg); st->heat_max_keys[0] = f64_key(bg); }
    }
}

// fused[p] = sum over ranks (in rank order) of heat_r[p]; also the fused heatmap's min / max (striped)
// avg_T > 0: the packets hold partial time SUMS of a frame-sharded buffer; the fused value is their sum / avg_T.
// A tile no rank sent is the constant k_sparse_index prepared (already in the extrema), stored 16 bytes per lane.
RM_KERNEL __launch_bounds__(256) void k_sparse_merge(const double *packets, size_t packet_doubles, int world, int cap, int H, int W,
                                                      int tiles_x, int ntiles, const int *map, const int *any, double *fused,
                                                      CollapseState *st, int avg_T)
{
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * CT_H, x0 = tx * CT_W;
    if (!any[tile]) {   // workgroup-uniform
        const double acc = st->sp_bg;
        const int x = x0 + 2 * (threadIdx.x & 31);
        const bool pair = ((W & 1) == 0) && x + 1 < W;   // even W: every row starts 16-byte aligned (x is even)
#pragma unroll
        for (int k = 0; k < CT_H / 8; ++k) {
            const int y = y0 + (threadIdx.x >> 5) + 8 * k;
            if (y >= H) continue;
            double *dst = fused + (size_t)y * W + x;
            if (pair) {
                *reinterpret_cast<F64Pair *>(dst) = F64Pair{acc, acc};
            } else {
                if (x < W) dst[0] = acc;
                if (x + 1 < W) dst[1] = acc;
            }
        }
        return;
    }
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    for (int i = threadIdx.x; i < CT_H * CT_W; i += 256) {
        const int y = y0 + i / CT_W, x = x0 + (i & (CT_W - 1));
        if (y >= H || x >= W) continue;
        double acc = 0.0;
        for (int r = 0; r < world; ++r) {
            const double *pk = packets + (size_t)r * packet_doubles;
            const int slot = map[(size_t)r * ntiles + tile];
            const double v = slot >= 0 ? pk[SP_HDR + cap + (size_t)slot * (CT_H * CT_W) + i] : pk[1];
            acc = (r == 0) ? v : acc + v;
        }
        if (avg_T > 0) acc = acc / (double)avg_T;   // np.average = sum / T (base.py:562)
        fused[(size_t)y * W + x] = acc;
        mn = (acc < mn) ? acc : mn;
        mx = (acc > mx) ? acc : mx;
    }
    block_minmax(mn, mx);
    if (threadIdx.x == 0) {
        const unsigned long long kmn = f64_key(mn), kmx = f64_key(mx);
        const int sp = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->heat_min_keys, st->heat_max_keys, sp, kmn, kmx);
    }
}

// ----------------------------------------------------------------------------------------
// dtype helpers and ROI reductions
// ----------------------------------------------------------------------------------------
RM_KERNEL __launch_bounds__(256) void k_u8_to_f64(const uint8_t *src, double *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (double)src[i] * (1.0 / 255);
}

RM_KERNEL __launch_bounds__(256) void k_f64_to_u8(const double *src, uint8_t *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = f64_to_u8_trunc(src[i] * 255);
}

// np.average(frame[y:y+h, x:x+w]) (base.py:357): numpy's pairwise order is not reproduced; any
// float64 order is within ~1e-13 relative of it.  One block; wave partials summed in lane order.
template <typename Tin>
__global__ __launch_bounds__(256) void k_roi_mean(const Tin *frame, int W, int x, int y, int w, int h, double *out)
{
    __shared__ double s_part[4];
    double acc = 0.0;
    int n = w * h;
    for (int i = threadIdx.x; i < n; i += 256) {
        int r = i / w, c = i - r * w;
        acc = acc + load_px(frame, (size_t)(y + r) * W + x + c);
    }
    for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]) / (double)n;
}

template <typename Tin>
__global__ __launch_bounds__(256) void k_roi_to_u8(const Tin *frame, int W, int x, int y, int w, int h, uint8_t *dst)
{
    int n = w * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int r = i / w, c = i - r * w;
        dst[i] = f64_to_u8_trunc(load_px(frame, (size_t)(y + r) * W + x + c) * 255);
    }
}

// cv2.cvtColor(BGR2GRAY), base.py:230: Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14
RM_KERNEL __launch_bounds__(256) void k_bgr_to_gray(const uint8_t *bgr, size_t npix, uint8_t *gray)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
}

// The same sum on whole words.  The weights do not fit a byte, so it is two v_dot4_u32_u8 (low and high bytes of the weights; the
// fourth byte of the word meets a zero weight) joined by one v_lshl_add_u32; a pixel whose three bytes straddle two words is brought
// together by v_alignbyte_b32 first.
constexpr unsigned BGR_W_LO = 0x0023914Cu, BGR_W_HI = 0x00132507u;   // 1868 = 0x074C, 9617 = 0x2591, 4899 = 0x1323: byte 0 = B, 1 = G, 2 = R

// gray value of the pixel whose B sits in byte SH of word d0 (its G / R may continue in d1), times 8: the byte offset into the table
template <int SH> __device__ __forceinline__ unsigned bgr_gray_x8(unsigned d0, unsigned d1)
{
    unsigned x = d0, wlo = BGR_W_LO, whi = BGR_W_HI;
    if constexpr (SH == 1) { wlo = BGR_W_LO << 8; whi = BGR_W_HI << 8; }        // bytes 1..3 of d0: move the weights, not the pixel
    else if constexpr (SH >= 2) x = __builtin_amdgcn_alignbyte(d1, d0, SH);     // {d1, d0} >> 8 SH
    const unsigned lo = __builtin_amdgcn_udot4(x, wlo, 8192u, false), hi = __builtin_amdgcn_udot4(x, whi, 0u, false);
    return (((hi << 8) + lo) >> 11) & 0x7f8u;                                  // (sum >> 14) << 3
}

// four pixels (three words) per thread and trip -> one word of gray; `nquads` = npix / 4, both pointers 4-byte aligned
RM_KERNEL __launch_bounds__(256) void k_bgr_to_gray_quads(const unsigned *bgr, size_t nquads, unsigned *gray)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (size_t)gridDim.x * 256) {
        const unsigned d0 = bgr[3 * i], d1 = bgr[3 * i + 1], d2 = bgr[3 * i + 2];
        const unsigned g0 = bgr_gray_x8<0>(d0, d1), g1 = bgr_gray_x8<3>(d0, d1), g2 = bgr_gray_x8<2>(d1, d2), g3 = bgr_gray_x8<1>(d2, d2);
        gray[i] = (g0 >> 3) | (g1 << 5) | (g2 << 13) | (g3 << 21);
    }
}

}  // namespace rm
