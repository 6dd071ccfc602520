// respmon_amd/csrc/rm_down_chain.h -- the roofline kernel: fused Gaussian pyramid chain
// frames[T,H,W] (u8 / f16 / f32 / f64)  ->  G_S[T,h_S,w_S] float64   (S x cv2.pyrDown, pyramid.py:9-17)
//
// The frame buffer is read from HBM exactly once; levels 1..S-1 never leave the CU.
//
// Decomposition: one single-wave workgroup owns (frame, strip of SW level-S columns, segment of
// level-S rows) and MARCHES down the input rows.  Per level k it keeps one row buffer in LDS (for the
// horizontal taps, which cross lanes) and the 5 most recent horizontally filtered rows in REGISTERS
// (each lane owns fixed columns, so the vertical pass needs no memory at all); whenever rows
// 2y-2..2y+2 of level k are in the ring, row y of level k+1 is produced by the vertical pass and
// pushed down the cascade (template recursion over levels).  Nothing is recomputed vertically; horizontally a strip re-filters its halo
// (2^(S+1)-2 input columns per side).  Operation order per output is OpenCV's (horizontal 5-tap then
// vertical 5-tap): bit-identical to the per-level kernel and to the CPU oracle.
//
// Leanness (the kernel is issue-bound before it is HBM-bound): the LDS layout is a compile-time
// function of (S, SW), every lane owns fixed elements of every level, row buffers are
// de-interleaved (even | odd columns) so the stride-2 taps are unit-stride conflict-free
// ds_read_b64 at immediate offsets from one per-lane base, and image borders are handled by writing
// the two reflected columns next to the row (as OpenCV does) so the tap code has no border branch.
// Global loads are 16 B per lane, coalesced, software-prefetched DC_PREFETCH rows ahead in registers.
// Workgroups of one frame are placed on one XCD (blockIdx % 8) so halo lines are shared in its L2.
#pragma once
#include "rm_kernels.h"
#include <cstdlib>

namespace rm {

#ifndef RM_DC_PREFETCH
#define RM_DC_PREFETCH 2
#endif
constexpr int DC_PREFETCH = RM_DC_PREFETCH;  // input rows in flight per wave

struct DownGeom {
    int S;
    int h[MAX_CHAIN], w[MAX_CHAIN];  // level sizes, 0 = input
    int seg_h;                       // segment size in level-S rows
    int seg_split;                   // > 0: TWO uneven segments, [y_begin, seg_split) and [seg_split, y_end) (see k_down_chain)
    int y_begin, y_end;              // level-S rows [y_begin, y_end) covered by this launch
    int strips, segs;                // per frame
    int sw;                          // level-S columns per strip: the row split evenly over `strips` (<= StripWidth<S>::SW)
    int wpg;                         // waves (= adjacent strips of one segment) per workgroup, marching in lockstep
    int T;
    int vec;                         // 1: 16-byte aligned vector loads are legal for this buffer
    int prio;                        // issue priority of the waves (s_setprio), see DownChain::set_prio: 2 (default) rotating every 2^prio_shift input rows;
                                     // developer settings: 0 none, 1 static (younger workgroups higher), 3 static (older higher)
    int prio_shift;                  // log2 of the rotation period in input rows (4: measured 16 / 32 / 64 / 128 rows in one process, tools/ab_inproc.py -- 0.6817 / 0.6837 / 0.6848 / 0.6883 ms against 0.7028 without priorities)
    int prio_rank;                   // dispatch-order quartile of this workgroup (0 = oldest), set by the kernel
};

// compile-time strip width (level-S columns) per chain depth: level-0 span stays <= ~380 pixels
#ifndef RM_DC_SW4
#define RM_DC_SW4 20
#endif
template <int S> struct StripWidth;
template <> struct StripWidth<1> { static constexpr int SW = 160; };
#ifndef RM_DC_SW2
#define RM_DC_SW2 88
#endif
template <> struct StripWidth<2> { static constexpr int SW = RM_DC_SW2; };
template <> struct StripWidth<3> { static constexpr int SW = 44; };
template <> struct StripWidth<4> { static constexpr int SW = RM_DC_SW4; };
template <> struct StripWidth<5> { static constexpr int SW = 8; };

constexpr int dc_max(int a, int b) { return a > b ? a : b; }

template <int S, int SW, int V> struct DCLayout {
    static constexpr int width(int k) { int w = SW; for (int i = S; i > k; --i) w = 2 * w + 3; return w; }
    static constexpr int nq(int k) { return (width(k) + 63) / 64; }  // elements per lane at level k
    static constexpr int nl() { return (width(0) + 2 * (V - 1) + V - 1) / V / 64 + 1; }  // 16-byte lane-loads per lane per row
    // de-interleaved row buffer of level k: columns [c0-2, c0-2+2*half) ; even | odd halves.  Sized so that
    // every lane may run every tap / store unmasked (lanes beyond the strip compute garbage nobody reads).
    static constexpr int half(int k)
    {
        return dc_max(dc_max((width(k) + 8) / 2 + 2, 64 * nq(k + 1) + 4), k == 0 ? 32 * nl() * V + 4 : 32 * nq(k) + 4);
    }
    static constexpr int rowbuf_size(int k) { return 2 * half(k); }
    static constexpr int rowbuf_off(int k) { int o = 0; for (int i = 0; i < k; ++i) o += rowbuf_size(i); return o; }
    static constexpr int total() { return rowbuf_off(S); }
};

// whole-wave shift by one lane (DPP wave_shr:1 / wave_shl:1): lane i receives lane i-1 / i+1;
// the end lane keeps its own value (callers never use it)
__device__ __forceinline__ double wave_from_prev(double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_from_next(double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// lane 0 receives lane 2's value, every other lane keeps its own (DPP quad_perm [2,1,2,3] on row 0 / bank 0 only): the reflected
// column -2 -> 2 of a row whose first lane-load sits at the left image border (run_dpp)
__device__ __forceinline__ double lane0_from_lane2(double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0xE6, 0x1, 0x1, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0xE6, 0x1, 0x1, false);
    return __hiloint2double(hi, lo);
}

// The SIMD's issue arbiter prefers the OLDER of its resident waves (MI355X_MICROARCH.md, "two waves per SIMD"): with one resident
// round of workgroups the first-dispatched quarter of a launch finishes ~8 % earlier than the last (workgroup timelines), and the
// chip spends the end of the kernel with too few waves to keep HBM busy.  Explicit priorities outrank age: the frame-buffer kernels
// rotate theirs every few input rows, starting from the workgroup's dispatch-order quartile, so every wave spends the same share
// of its life at every level.
__device__ __forceinline__ void dc_set_prio(int v)
{
    switch (v & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
__device__ __forceinline__ int dc_dispatch_quartile() { return (int)(((blockIdx.x >> 3) * 4u) / max(1u, gridDim.x >> 3)); }   // inside this workgroup's XCD

template <typename Tin> struct VecTraits;
template <> struct VecTraits<double> { static constexpr int V = 2; };
template <> struct VecTraits<float> { static constexpr int V = 4; };
template <> struct VecTraits<__half> { static constexpr int V = 8; };
template <> struct VecTraits<uint8_t> { static constexpr int V = 16; };

struct alignas(16) Raw16 { unsigned int x, y, z, w; };

// 16-byte lane-loads of the frame buffer.  NON-TEMPORAL (global_load_dwordx4 ... nt) for everything only this strip reads: every
// such byte is used once, and without the hint the 4.25 GB stream allocates in L2 / Infinity Cache like data that will be re-read.
// PLAIN for the first / last 128-pixel chunk of a row when a neighbouring strip (the other strip of the lockstep pair, or the next
// workgroup of the frame on the same XCD) reads ~60 of its columns as well: those lines are wanted twice.  (A per-LANE choice --
// two exec-masked loads per chunk -- destroyed the counted-prefetch pipeline: 1.30 ms.)  Measured (tools/ab_lib.py, alternating processes on one box; PMC traffic):
//   all plain 0.777 ms, 1.091x algorithmic | all nt 0.759 ms, 1.154x (the shared columns travel twice) | split: DESIGN 4.1.
#ifndef RM_DC_PLAIN_LOADS
#define RM_DC_NT_LOADS 1
#endif
__device__ __forceinline__ Raw16 load_raw16(const void *p) { return *reinterpret_cast<const Raw16 *>(p); }
__device__ __forceinline__ Raw16 load_raw16_stream(const void *p)
{
#if defined(RM_DC_NT_LOADS)
    typedef RM_VEC(unsigned int, 4) u32x4;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    Raw16 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; r.w = v[3];
    return r;
#else
    return *reinterpret_cast<const Raw16 *>(p);
#endif
}

template <typename Tin> __device__ __forceinline__ double unpack_px(const Raw16 &r, int e);
template <> __device__ __forceinline__ double unpack_px<double>(const Raw16 &r, int e)
{
    unsigned long long b = e ? (((unsigned long long)r.w << 32) | r.z) : (((unsigned long long)r.y << 32) | r.x);
    return __longlong_as_double((long long)b);
}
template <> __device__ __forceinline__ double unpack_px<float>(const Raw16 &r, int e)
{
    unsigned int b = e == 0 ? r.x : e == 1 ? r.y : e == 2 ? r.z : r.w;
    return (double)__int_as_float((int)b);
}
template <> __device__ __forceinline__ double unpack_px<__half>(const Raw16 &r, int e)
{
    unsigned int wd = (e >> 1) == 0 ? r.x : (e >> 1) == 1 ? r.y : (e >> 1) == 2 ? r.z : r.w;
    unsigned short hb = (unsigned short)((e & 1) ? (wd >> 16) : (wd & 0xffffu));
    __half hv;
    __builtin_memcpy(&hv, &hb, 2);
    return (double)__half2float(hv);
}
template <> __device__ __forceinline__ double unpack_px<uint8_t>(const Raw16 &r, int e)
{
    unsigned int wd = (e >> 2) == 0 ? r.x : (e >> 2) == 1 ? r.y : (e >> 2) == 2 ? r.z : r.w;
    return (double)((wd >> (8 * (e & 3))) & 0xffu) * (1.0 / 255);  // uint8_to_float, transforms.py:20-23
}

// Vertical-pass state of level K for this lane's columns, in REGISTERS.  The 5-tap vertical filter
//   out[y] = ((r[2y]*6 + (r[2y-1] + r[2y+1])*4) + r[2y-2]) + r[2y+2]      (OpenCV's operation order)
// is evaluated as rows stream by: on odd row 2y+1 the partial t = (c*6 + (b + n)*4) + a is formed, on
// even row 2y+2 the output is t + n.  Only (a, b, c, t) are live per column -- no 5-row ring, no shifting.
template <typename LL, int S, int K> struct VState : VState<LL, S, K + 1> {
    double a[LL::nq(K + 1)], b[LL::nq(K + 1)], c[LL::nq(K + 1)], t[LL::nq(K + 1)];
};
template <typename LL, int S> struct VState<LL, S, S> {};

// VB = false: hot instantiation, valid when every level has >= 3 rows: steady-state vertical pass plus two
// uniform special cases at the image top and virtual (replayed) rows at the image bottom;
// VB = true: fully generic form (any size, incl. 1- and 2-row levels), used for tiny images only.
template <typename Tin, int S, bool VB>
struct DownChain {
    static constexpr int SW = StripWidth<S>::SW;
    static constexpr int V = VecTraits<Tin>::V;
    using L = DCLayout<S, SW, V>;
    static constexpr int NL = L::nl();

    const DownGeom &g;
    double *lds;
    const int lane;
    int cx0[S + 1], cx1[S + 1];  // column range of each level held by this strip (inclusive, clamped)
    int c0[S];                   // even (level 0: vector-aligned) base column of row buffer k
    int next[S + 1], last[S + 1];
    const double *tap[S];        // per-lane LDS address of the centre tap of this lane's first column, level k
    double *rowdst[S];           // per-lane LDS address of this lane's first column in row buffer k (k >= 1)
    int border_src[S], border_dst[S];  // lanes 0..3: reflected-column copy of level k (-1: not this lane)
    bool has_border[S];
    double *out_frame;           // G_S of this frame
    int col1_lane;               // level-1 column of this lane's q = 0 element
    int q1;                      // level-1 columns between consecutive q (64; 62 on the register/DPP front end)
    bool lane_ok1;               // this lane's level-1 elements are real (DPP front end: lanes 1..62)
    VState<L, S, 0> vs;

    __device__ __forceinline__ DownChain(const DownGeom &g_, double *lds_) : g(g_), lds(lds_), lane(threadIdx.x & 63) {}

    // The waves of a workgroup are the adjacent strips of one (frame, segment): a barrier per prefetch round
    // keeps them on the same input rows, so the columns two strips share are fetched from HBM once and hit
    // in L2 for the neighbour.  Raw s_barrier: no waitcnt, the prefetched rows stay in flight across it.
    __device__ __forceinline__ void lockstep() const
    {
        if (g.wpg > 1) __builtin_amdgcn_s_barrier();
    }

    // index of column c inside row buffer K (columns c0-2 .. are stored de-interleaved)
    template <int K> __device__ __forceinline__ int rb_index(int c) const
    {
        int i = c - c0[K] + 2;
        return L::rowbuf_off(K) + (i >> 1) + (i & 1) * L::half(K);
    }

    // OpenCV-style border: write the two reflected columns beside the row so taps never branch
    template <int K> __device__ __forceinline__ void make_border()
    {
        if (has_border[K]) {
            if (border_dst[K] >= 0) lds[border_dst[K]] = lds[border_src[K]];
        }
    }

    // horizontal 5-tap of the row in buffer K for this lane's columns (unnormalised, like OpenCV's row pass)
    template <int K> __device__ __forceinline__ void taps(double (&n)[L::nq(K + 1)])
    {
        const double *ev = tap[K];
#pragma unroll
        for (int q = 0; q < L::nq(K + 1); ++q) {
            const double *e = ev + 64 * q;
            const double *o = e + L::half(K);
            n[q] = e[0] * 6 + (o[-1] + o[0]) * 4 + e[-1] + e[1];
        }
    }

    // row y of level K+1 is complete: store it (G_S) or hand it to the next level of the cascade
    template <int K> __device__ __forceinline__ void emit(int y, const double (&v)[L::nq(K + 1)])
    {
        next[K + 1] = y + 1;
        const int qs = (K == 0) ? q1 : 64;                       // columns between this lane's consecutive elements
        const bool ok = (K == 0) ? lane_ok1 : true;
        if constexpr (K + 1 == S) {
            const int col = ((K == 0) ? col1_lane : cx0[S] + lane);
#pragma unroll
            for (int q = 0; q < L::nq(K + 1); ++q) {
                const int cq = col + qs * q;
                if (ok && cq >= cx0[S] && cq <= cx1[S]) out_frame[(size_t)y * g.w[S] + cq] = v[q] * (1.0 / 256);
            }
        } else {
            if (ok) {
#pragma unroll
                for (int q = 0; q < L::nq(K + 1); ++q) rowdst[K + 1][(qs >> 1) * q] = v[q] * (1.0 / 256);  // +qs columns: same parity
            }
            wave_sync();
            make_border<K + 1>();
            wave_sync();
            double n[L::nq(K + 2)];
            taps<K + 1>(n);
            wave_sync();  // row buffer K+1 is free again
            feed<K + 1>(y, n);
        }
    }

    // number of VIRTUAL rows that follow the last real row of level K in the hot form: the rows that
    // BORDER_REFLECT_101 maps back into the image are replayed from the vertical state, so the image bottom
    // runs through the same single step() site as every other row
    template <int K> __device__ __forceinline__ int virtual_rows(int p) const
    {
        if constexpr (VB) return 0;
        return (p == g.h[K] - 1) ? ((g.h[K] & 1) ? 2 : 1) : 0;
    }

    // real row p of level K plus, after the last real row, its virtual successors (one step() site).
    // virtual row h: even h: row h -> h-2 (= c);  odd h: row h -> h-2 (= b), then row h+1 -> h-3 (= a before).
    // The next iteration's values are formed with value selects only (no control flow around the arrays).
    template <int K> __device__ __forceinline__ void feed(int p, const double (&n)[L::nq(K + 1)])
    {
        constexpr int NQ = L::nq(K + 1);
        VState<L, S, K> &st = vs;
        const int nv = virtual_rows<K>(p);
        const bool odd_h = (g.h[K] & 1) != 0;
        double cur[NQ], held_a[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { cur[q] = n[q]; held_a[q] = 0.0; }
#pragma nounroll
        for (int rep = 0; rep <= nv; ++rep) {
            step<K>(p + rep, cur);
            if (rep < nv) {  // only the last real row of a level has successors to prepare
                const bool first = rep == 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double from_state = odd_h ? st.b[q] : st.c[q];
                    cur[q] = first ? from_state : held_a[q];
                    held_a[q] = first ? st.a[q] : held_a[q];
                }
            }
        }
    }

    // feed the horizontally filtered row p of level K (values n) into the streaming vertical pass.
    // There is exactly ONE emit site per level (the cascade below it is inlined once).
    template <int K> __device__ __forceinline__ void step(int p, const double (&n)[L::nq(K + 1)])
    {
        constexpr int NQ = L::nq(K + 1);
        VState<L, S, K> &st = vs;
        if constexpr (!VB) {
            // hot form (every level has >= 3 rows).  The image top needs two uniform special cases
            // (rows -2, -1 reflect onto 2, 1); the image bottom arrives as virtual rows (feed()).
            if (p & 1) {
                // row 1 of the image: rows -1, -2 reflect onto 1, 2, i.e. b := n and the `+ a` term moves to row 2.
                // Written as value selects (x + -0.0 == x bit for bit), not as control flow, so the state stays
                // in registers.
                const bool top = p == 1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double be = top ? n[q] : st.b[q], ae = top ? -0.0 : st.a[q];
                    st.t[q] = (st.c[q] * 6 + (be + n[q]) * 4) + ae;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) { st.a[q] = st.c[q]; st.b[q] = n[q]; }
            } else {
                const int y = (p >> 1) - 1;
                if (y == next[K + 1] && y <= last[K + 1]) {
                    double v[NQ];
                    const bool top = p == 2;  // row 2 of the image also stands for row -2
#pragma unroll
                    for (int q = 0; q < NQ; ++q) v[q] = (st.t[q] + n[q]) + (top ? n[q] : -0.0);
                    emit<K>(y, v);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) st.c[q] = n[q];
            }
            return;
        }
        const int hk = g.h[K];
        const bool odd = (p & 1) != 0;
        const bool lastrow = p == hk - 1;
        if (hk > 2 && odd) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) st.t[q] = (st.c[q] * 6 + (st.b[q] + n[q]) * 4) + st.a[q];
        }
        if (hk <= 2) {
            // degenerate level: the single output row reflects everything onto rows 0 (and 1)
            if (p == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { st.a[q] = n[q]; st.b[q] = n[q]; st.c[q] = n[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) st.b[q] = n[q];
            }
        }
        // which output rows complete with this input row?  (at most two: the odd-height bottom case)
        //   kind 0: steady state   v = t + n               (even p, y = p/2 - 1 >= 1)
        //   kind 1: image top      v = ((a*6+(b+b)*4)+n)+n (even p, y = 0)
        //   kind 2: even-h bottom  v = t + c               (odd  p = h-1, y = (p-1)/2)
        //   kind 3: odd-h bottom   v = ((n*6+(b+b)*4)+a)+a (even p = h-1, y = p/2), after kind 0/1 of the same row
        //   kind 4: degenerate     v = ((a*6+(b+b)*4)+c)+c (h <= 2, p = h-1, y = 0)
#pragma nounroll
        for (int rep = 0; rep < 2; ++rep) {
            int kind = -1, y = 0;
            if (hk <= 2) { if (rep == 0 && lastrow) { kind = 4; y = 0; } }
            else if (odd) { if (rep == 0 && lastrow) { kind = 2; y = (p - 1) >> 1; } }
            else if (rep == 0) { if (p >= 2) { y = (p >> 1) - 1; kind = (y == 0) ? 1 : 0; } }
            else if (lastrow) { kind = 3; y = p >> 1; }
            if (kind < 0 || y != next[K + 1] || y > last[K + 1]) continue;
            double v[NQ];
            if (kind == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) v[q] = st.t[q] + n[q];
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double t = st.t[q], a = st.a[q], b = st.b[q], c = st.c[q], m = n[q];
                    double r = ((a * 6 + (b + b) * 4) + c) + c;             // kind 4
                    if (kind == 1) r = ((a * 6 + (b + b) * 4) + m) + m;
                    if (kind == 2) r = t + c;
                    if (kind == 3) r = ((m * 6 + (b + b) * 4) + a) + a;
                    v[q] = r;
                }
            }
            emit<K>(y, v);
        }
        if (hk > 2) {
            if (odd) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { st.a[q] = st.c[q]; st.b[q] = n[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) st.c[q] = n[q];
            }
        }
    }

    template <int K> __device__ __forceinline__ void setup_level()
    {
        const int wk = g.w[K];
        tap[K] = lds + L::rowbuf_off(K) + (((2 * cx0[K + 1] - c0[K] + 2) >> 1) + lane);
        if constexpr (K >= 1) rowdst[K] = lds + rb_index<K>(K == 1 ? col1_lane : cx0[K] + lane);
        const bool left = cx0[K] == 0, right = cx1[K] == wk - 1;
        has_border[K] = left || right;
        border_src[K] = 0; border_dst[K] = -1;
        if (lane < 4) {
            const bool is_left = lane < 2;
            const int c = is_left ? lane - 2 : wk + lane - 2;
            if (is_left ? left : right) { border_dst[K] = rb_index<K>(c); border_src[K] = rb_index<K>(reflect101(c, wk)); }
        }
        if constexpr (K + 1 < S) setup_level<K + 1>();
    }

    // ---- level-0 front end A (float64, 16-byte aligned rows): horizontal pass in REGISTERS ---------------
    // A lane-load is two adjacent pixels (a = s[c], b = s[c+1], c even); the 5 taps of output column c/2 are
    //   s[c]*6 + (s[c-1] + s[c+1])*4 + s[c-2] + s[c+2] = a*6 + (b_prev + b)*4 + a_prev + a_next
    // with a_prev, b_prev, a_next taken from the neighbouring lanes by DPP wave shifts: no LDS, no waits.
    // Chunk q covers columns P + 124*q + [0,128): consecutive chunks overlap by two lanes so that lanes
    // 1..62 of every chunk own 62 consecutive level-1 columns and lanes 0 / 63 only provide halo.
    // PL / PR: the first / last chunk of a row holds columns a neighbouring strip reads too -> plain (cache-allocating) loads there
    template <bool PL, bool PR>
    __device__ __forceinline__ void run_dpp(const Tin *frame, int p_first, int p_last)
    {
        constexpr int NQ1 = L::nq(1);
        const int W = g.w[0];
        const int P = 2 * cx0[1] - 2;  // first column of chunk 0 (-2 at the left image border)
        const bool fix_l = cx0[0] == 0, fix_r = cx1[0] == W - 1;
        const Tin *lane_src[NQ1];
#pragma unroll
        for (int q = 0; q < NQ1; ++q) lane_src[q] = frame + min(max(P + 124 * q + 2 * lane, 0), W - 2);
        auto issue = [&](int row, Raw16 (&r)[NQ1]) __attribute__((always_inline)) {
#ifdef RM_DC_SAMEROW  // developer experiment: every load hits the same (cached) row
            row = p_first;
#endif
            const size_t ro = (size_t)row * W;
#pragma unroll
            for (int q = 0; q < NQ1; ++q) {
                if ((q == 0 && PL) || (q == NQ1 - 1 && PR)) r[q] = load_raw16(lane_src[q] + ro);
                else r[q] = load_raw16_stream(lane_src[q] + ro);
            }
        };
        // BORDER_REFLECT_101 on the column index: only chunks that contain slots outside the image need it
        // (chunk 0 at the left image edge, the chunk(s) holding columns W, W+1 at the right edge).  Per lane
        // and flagged chunk one packed word is precomputed: source lane and component for a and for b.
        // The two common cases need no table: (left) chunk 0 starts at column -2, so only lane 0 holds out-of-image columns (-2, -1);
        // its clamped load brought columns (0, 1): b = s[1] is already the reflection of -1, a must become s[2] = lane 2's a -- one
        // DPP move; (right, even W) the only out-of-image column a tap reaches is W, the first of its lane-load, and the clamped load
        // of that lane brought columns (W-2, W-1): a = s[W-2] is its reflection already.  (The shuffles of the table form cost the
        // two outer strips of a row ~6 % -- and their lockstep partners with them: workgroup timelines, round 5.)
        const bool fast_l = fix_l && P == -2, fast_r = fix_r && (W & 1) == 0;
        unsigned fixq = 0;       // bit q: chunk q has out-of-image slots (wave-uniform)
        int fixw[NQ1];           // per lane: bits 0-5 src lane of a, 6 src comp, 7 fix a; 8-13 / 14 / 15 the same for b
#pragma unroll
        for (int q = 0; q < NQ1; ++q) {
            const int base = P + 124 * q;
            fixw[q] = 0;
            if ((fix_l && base < 0 && !fast_l) || (fix_r && base + 127 >= W && !fast_r)) {
                fixq |= 1u << q;
                const int ca = base + 2 * lane, cb = ca + 1;
                const int ra = reflect101(ca, W) - base, rb = reflect101(cb, W) - base;
                fixw[q] = ((ra >> 1) & 63) | ((ra & 1) << 6) | ((ca < 0 || ca >= W) ? 0x80 : 0) |
                          (((rb >> 1) & 63) << 8) | ((rb & 1) << 14) | ((cb < 0 || cb >= W) ? 0x8000 : 0);
            }
        }
        auto hrow = [&](const Raw16 (&r)[NQ1], double (&n)[NQ1]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NQ1; ++q) {
                double a = unpack_px<Tin>(r[q], 0), b = unpack_px<Tin>(r[q], 1);
                if (fixq & (1u << q)) {
                    const int fw = fixw[q];
                    const double a_from_a = __shfl(a, fw & 63), a_from_b = __shfl(b, fw & 63);
                    const double b_from_a = __shfl(a, (fw >> 8) & 63), b_from_b = __shfl(b, (fw >> 8) & 63);
                    if (fw & 0x80) a = (fw & 0x40) ? a_from_b : a_from_a;
                    if (fw & 0x8000) b = (fw & 0x4000) ? b_from_b : b_from_a;
                }
                if (q == 0 && fast_l) a = lane0_from_lane2(a);
                const double a_prev = wave_from_prev(a), b_prev = wave_from_prev(b), a_next = wave_from_next(a);
                n[q] = a * 6 + (b_prev + b) * 4 + a_prev + a_next;
            }
        };
        Raw16 regs[DC_PREFETCH][NQ1];
#pragma unroll
        for (int i = 0; i < DC_PREFETCH; ++i) issue(min(p_first + i, p_last), regs[i]);
        if (g.prio == 1) dc_set_prio(g.prio_rank);
        else if (g.prio == 3) dc_set_prio(3 - g.prio_rank);
        for (int base = p_first; base <= p_last; base += DC_PREFETCH) {
            if (g.prio == 2 && ((base - p_first) & ((1 << g.prio_shift) - 1)) == 0) dc_set_prio(g.prio_rank + ((base - p_first) >> g.prio_shift));
            lockstep();
#pragma unroll
            for (int i = 0; i < DC_PREFETCH; ++i) {
                const int p = base + i;
                if (p <= p_last) {
                    double n[NQ1];
                    hrow(regs[i], n);
                    issue(min(p + DC_PREFETCH, p_last), regs[i]);
                    feed<0>(p, n);
                }
            }
        }
    }

    // ---- level-0 front end B (any dtype / alignment): rows staged through the LDS row buffer ---------------
    __device__ __forceinline__ void run_lds(const Tin *frame, int p_first, int p_last, bool vec)
    {
        const int W = g.w[0];
        auto consume = [&](int p) __attribute__((always_inline)) {
            wave_sync();
            make_border<0>();
            wave_sync();
            double n[L::nq(1)];
            taps<0>(n);
            wave_sync();  // the row buffer is free again
            feed<0>(p, n);
        };
        if (vec) {
            // lane-load j covers columns c0 + j*V ..; lanes past the strip re-load its last chunk (no exec masking)
            const int nload = (cx1[0] - c0[0] + V) / V;
            const Tin *lane_src[NL];
#pragma unroll
            for (int q = 0; q < NL; ++q) lane_src[q] = frame + c0[0] + (size_t)min(lane + 64 * q, nload - 1) * V;
            auto issue = [&](int row, Raw16 (&r)[NL]) __attribute__((always_inline)) {
                const size_t ro = (size_t)row * W;
#pragma unroll
                for (int q = 0; q < NL; ++q) r[q] = load_raw16(lane_src[q] + ro);
            };
            // column c0 + j*V + e sits at buffer index j*V + e + 2: even half (j*V)/2 + e/2 + 1
            double *ev0 = lds + L::rowbuf_off(0) + (lane * V) / 2 + 1;
            auto stash = [&](const Raw16 (&r)[NL]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    double *ev = ev0 + (64 * q * V) / 2;
                    double *od = ev + L::half(0);
#pragma unroll
                    for (int e = 0; e < V; e += 2) {
                        ev[e >> 1] = unpack_px<Tin>(r[q], e);
                        od[e >> 1] = unpack_px<Tin>(r[q], e + 1);
                    }
                }
            };
            Raw16 regs[DC_PREFETCH][NL];
#pragma unroll
            for (int i = 0; i < DC_PREFETCH; ++i) issue(min(p_first + i, p_last), regs[i]);
            for (int base = p_first; base <= p_last; base += DC_PREFETCH) {
                lockstep();
#pragma unroll
                for (int i = 0; i < DC_PREFETCH; ++i) {
                    const int p = base + i;
                    if (p <= p_last) {
                        stash(regs[i]);
                        issue(min(p + DC_PREFETCH, p_last), regs[i]);
                        consume(p);
                    }
                }
            }
        } else {
            for (int p = p_first; p <= p_last; ++p) {
                const Tin *src = frame + (size_t)p * W;
                for (int c = cx0[0] + lane; c <= cx1[0]; c += 64) lds[rb_index<0>(c)] = load_px(src, (size_t)c);
                consume(p);
            }
        }
    }

    __device__ __forceinline__ void run(const Tin *frame, double *out_t, int strip, int seg)
    {
        // ranges, from level S back to 0
        cx0[S] = strip * g.sw; cx1[S] = min(cx0[S] + g.sw, g.w[S]) - 1;
        if (g.seg_split > 0) { next[S] = seg ? g.seg_split : g.y_begin; last[S] = (seg ? g.y_end : g.seg_split) - 1; }
        else { next[S] = g.y_begin + seg * g.seg_h; last[S] = min(next[S] + g.seg_h, g.y_end) - 1; }
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {
            cx0[k] = max(0, 2 * cx0[k + 1] - 2); cx1[k] = min(g.w[k] - 1, 2 * cx1[k + 1] + 2);
            next[k] = max(0, 2 * next[k + 1] - 2); last[k] = min(g.h[k] - 1, 2 * last[k + 1] + 2);
            c0[k] = cx0[k] & ~1;
        }
        const bool vec = g.vec != 0;
        constexpr bool kF64 = (V == 2);
        const bool dpp = kF64 && vec;
        if (vec) c0[0] = cx0[0] & ~(V - 1);
        q1 = dpp ? 62 : 64;
        col1_lane = dpp ? (cx0[1] - 1 + lane) : (cx0[1] + lane);
        lane_ok1 = dpp ? (lane >= 1 && lane <= 62) : true;
        out_frame = out_t;
        setup_level<0>();
        if constexpr (kF64) {
            if (dpp) {
#ifdef RM_DC_ALL_STREAM
                run_dpp<false, false>(frame, next[0], last[0]);
#else
                // wave-uniform choice, made once: inside the march the loads are straight-line code
#if defined(RM_DC_POLICY_LEFT)
                const bool pl = cx0[S] > 0, pr = false;
#elif defined(RM_DC_POLICY_RIGHT)
                const bool pl = false, pr = cx1[S] < g.w[S] - 1;
#else
                const bool pl = cx0[S] > 0, pr = cx1[S] < g.w[S] - 1;
#endif
                if (pl && pr) run_dpp<true, true>(frame, next[0], last[0]);
                else if (pl) run_dpp<true, false>(frame, next[0], last[0]);
                else if (pr) run_dpp<false, true>(frame, next[0], last[0]);
                else run_dpp<false, false>(frame, next[0], last[0]);
#endif
                return;
            }
        }
        run_lds(frame, next[0], last[0], vec);
    }
};

template <typename Tin, int S> __host__ __device__ constexpr int down_chain_lds_doubles() { return DCLayout<S, StripWidth<S>::SW, VecTraits<Tin>::V>::total(); }

constexpr int DC_MAX_WPG = 2;  // waves per workgroup (launch bound).  Measured in bench.py, 1080p x 256 f64: 1 -> 0.816 ms, 2 -> 0.788 ms, 3 -> 1.06 ms (a stalled wave stalls its whole group: fewer independent contexts per CU)

template <typename Tin, int S, bool VB>
__global__ __launch_bounds__(64 * DC_MAX_WPG) void k_down_chain(const Tin *frames, size_t frame_stride, DownGeom g, double *out)
{
    RM_TRACE_SCOPE(0);
    HIP_DYNAMIC_SHARED(double, lds_all)
    // XCD-aware mapping: block b runs on XCD b % 8; give each XCD whole frames so the strips and
    // segments of a frame share halo lines in one L2.  A workgroup = g.wpg adjacent strips of one segment,
    // one wave each (private LDS slice, no data exchanged between the waves).
    const int groups = (g.strips + g.wpg - 1) / g.wpg;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int t, seg, grp;
    if (g.seg_split > 0) {
        // Two uneven segments per frame, the upper (larger) one on an even XCD and the lower one on its odd neighbour: the
        // odd XCDs of an MI355X stream ~4-5 % slower than the even ones (workgroup timelines, tools/trace_tail.py, on every
        // box measured), and with equal shares the launch waits for them while the even XCDs sit idle.
        const int pair = xcd >> 1;
        seg = xcd & 1;
        t = (j / groups) * 4 + pair;
        grp = j - (j / groups) * groups;
    } else {
        const int per_frame = groups * g.segs;
        t = (j / per_frame) * 8 + xcd;
        const int inner = j % per_frame;
        seg = inner / groups; grp = inner - seg * groups;
    }
    if (t >= g.T) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep the geometry in SGPRs
    const int strip = grp * g.wpg + wave;
    if (strip >= g.strips) return;  // a terminated wave no longer counts at s_barrier
    double *lds = lds_all + wave * down_chain_lds_doubles<Tin, S>();
    g.prio_rank = dc_dispatch_quartile();
    DownChain<Tin, S, VB> dc(g, lds);
    dc.run(frames + (size_t)t * frame_stride, out + (size_t)t * g.h[S] * g.w[S], strip, seg);
}

// host-side geometry

inline unsigned down_chain_grid(const DownGeom &g)
{
    const int groups = (g.strips + g.wpg - 1) / g.wpg;
    if (g.seg_split > 0) return (unsigned)(((g.T + 3) / 4) * 8 * groups);   // 4 XCD pairs, one frame segment per XCD
    return (unsigned)(((g.T + 7) / 8) * 8 * groups * g.segs);
}
inline unsigned down_chain_block(const DownGeom &g) { return 64u * (unsigned)g.wpg; }

// level-S row range [y0, y1) whose dependency cone needs no vertical border handling at any level
inline void down_chain_interior(int S, const int *h, int *y0, int *y1)
{
    *y0 = 2; *y1 = 2;
    for (int Y1 = h[S] - 1; Y1 >= 2; --Y1) {
        bool ok = true;
        long long last = Y1;
        for (int k = S - 1; k >= 0 && ok; --k) { last = 2 * last + 2; if (last > h[k] - 1) ok = false; }
        if (ok) { *y1 = Y1 + 1; break; }
    }
    if (*y1 < *y0) *y1 = *y0;
}

// the hot instantiation needs >= 3 rows at every level it filters
inline bool down_chain_hot_ok(int S, const int *h)
{
    for (int k = 0; k < S; ++k) if (h[k] < 3) return false;
    return true;
}

// geometry of one launch over level-S rows [y_begin, y_end)
// force_segs / force_wpg / split_permille > 0: developer overrides (rm_debug_set "dc_segs" / "dc_wpg" / "dc_split")
inline bool make_down_geom(int S, const int *h, const int *w, int T, int vec_ok, int y_begin, int y_end, DownGeom &g, bool tiny = false,
                           int force_segs = 0, int force_wpg = 0, int split_permille = 0)
{
    if (S < 1 || S > 5 || y_end <= y_begin) return false;
    g.S = S; g.T = T; g.vec = vec_ok; g.y_begin = y_begin; g.y_end = y_end; g.prio = 0; g.prio_rank = 0; g.prio_shift = 4;
    for (int k = 0; k <= S; ++k) { g.h[k] = h[k]; g.w[k] = w[k]; }
    const int SW = S == 1 ? StripWidth<1>::SW : S == 2 ? StripWidth<2>::SW : S == 3 ? StripWidth<3>::SW
                 : S == 4 ? StripWidth<4>::SW : StripWidth<5>::SW;
    g.strips = (g.w[S] + SW - 1) / SW;
    g.sw = (g.w[S] + g.strips - 1) / g.strips;   // even split: 320 columns = 4 x 80, not 88 + 88 + 88 + 56 (the lockstep group waits for its widest strip)
    // segments: ONE resident round of waves (12 per CU at ~147 VGPRs = 3072 on the chip).  Every extra segment
    // re-reads 2*(2^(S+1)-2) input rows from HBM (measured: 4 segments = 1.23x the algorithmic bytes at 1080p),
    // so take the fewest segments that fill the machine, and never let the halo exceed ~25 % of a segment.
    const int rows = y_end - y_begin;
    const int halo0 = (1 << (S + 1)) - 2;
    const long long per_seg = (long long)T * g.strips;
    int segs = (int)((3072 + per_seg / 2) / per_seg);
    if (segs < 1) segs = 1;
    if (segs > rows) segs = rows;
    while (segs > 1 && ((((rows + segs - 1) / segs) << S) < 8 * halo0)) --segs;
#ifdef RM_DC_SEGS  // developer experiment
    segs = RM_DC_SEGS;
#endif
    if (force_segs >= 1 && force_segs <= rows) segs = force_segs;
    g.seg_h = (rows + segs - 1) / segs;
    if (tiny) g.seg_h = rows < 2 ? rows : 2;  // test hook: many small segments
    g.segs = (rows + g.seg_h - 1) / g.seg_h;
    // exactly two segments: split them 51.3 : 48.7 between an even XCD and its (slower) odd neighbour, see k_down_chain
    g.seg_split = 0;
    if (g.segs == 2 && !tiny && rows >= 8) {
        int upper = (int)(rows * (split_permille > 0 ? split_permille * 0.001 : 0.513) + 0.5);
        if (upper >= rows) upper = rows - 1;
        if (upper > g.seg_h) g.seg_split = y_begin + upper;
    }
    // workgroup = up to DC_MAX_WPG adjacent strips in lockstep, split evenly when a row has more strips
    const int ngroups = (g.strips + DC_MAX_WPG - 1) / DC_MAX_WPG;
    g.wpg = (g.strips + ngroups - 1) / ngroups;
#ifdef RM_DC_WPG  // developer experiment
    g.wpg = RM_DC_WPG;
#endif
    if (force_wpg >= 1 && force_wpg <= DC_MAX_WPG) g.wpg = force_wpg;
    return true;
}

}  // namespace rm
