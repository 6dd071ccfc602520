// respmon_amd/csrc/rm_down_chain_u8.h -- fused Gaussian pyramid chain for NARROW frame buffers (uint8, float16, float32)
// frames[T,H,W] (W % 16 == 0) -> G_S[T,h_S,w_S] float64, S <= 4                   (pyramid.py:9-17)
//
// Written for uint8 first (the text below); float16 / float32 buffers use the same register-resident chain with
// 2 / 4 sixteen-byte loads per lane and row instead of one (a lane still owns 16 adjacent pixels), and their
// values widen exactly to float64.  float64 buffers keep the 2-pixels-per-lane DPP front end of rm_down_chain.h:
// 128 contiguous bytes per lane would make every load instruction touch 64 cache lines.
//
// uint8 is what a camera delivers and 1/8 of the HBM bytes of the reference's float64 buffer; the kernels
// apply uint8_to_float's  k * (1./255)  (transforms.py:20-23) on the fly, so results are bit-identical to
// the float64 path.  With 8x fewer bytes the chain is compute bound, and this variant removes the LDS
// entirely: a lane-load is 16 adjacent pixels, and the lane keeps owning that block of columns at every
// level (16 -> 8 -> 4 -> 2 -> 1).  A 5-tap horizontal stencil then needs only the previous lane's last two
// columns and the next lane's first column -- three DPP wave shifts per level -- and the vertical pass is
// the same streaming (a, b, c, t) state as rm_down_chain.h, one set per owned column.  A wave covers
// 1024 input columns of which lanes 2..61 (960 columns) are exact; lanes 0,1,62,63 supply the halo.
// Image borders: the left image edge is lane 2 of strip 0 and the right edge is the end of a lane at every
// level (W % 16 == 0), so BORDER_REFLECT_101 is two value selects.
#pragma once
#include "rm_down_chain.h"

namespace rm {

constexpr int U8_VALID_LANES = 60;            // lanes 2..61
constexpr int U8_STRIP_PX = 16 * U8_VALID_LANES;  // 960 exact input columns per wave
#ifndef RM_U8_PREFETCH
#define RM_U8_PREFETCH 4
#endif

template <int S, int K> struct VStateU8 : VStateU8<S, K + 1> {
    double a[(16 >> K) / 2], b[(16 >> K) / 2], c[(16 >> K) / 2], t[(16 >> K) / 2];
};
template <int S> struct VStateU8<S, S> {};

template <typename Tin> struct RegTraits;   // NLD: 16-byte loads per lane and row; PF: rows in flight
template <> struct RegTraits<uint8_t> { static constexpr int NLD = 1, PF = RM_U8_PREFETCH; };
template <> struct RegTraits<__half> { static constexpr int NLD = 2, PF = 3; };
template <> struct RegTraits<float> { static constexpr int NLD = 4, PF = 2; };

template <int S, typename Tin = uint8_t>
struct RegChain {
    static_assert(S >= 1 && S <= 4, "a lane owns 16 >> K columns of level K");
    static constexpr int NLD = RegTraits<Tin>::NLD, PF = RegTraits<Tin>::PF, VPER = 16 / NLD;  // VPER pixels per 16-byte load
    const DownGeom &g;
    const int lane;
    int next[S + 1], last[S + 1];
    bool left_lane, last_lane;   // this lane holds column 0 / the last column of every level
    int col_S;                   // level-S column of this lane's first owned column
    bool store_ok;               // lane is exact (2..61)
    double *out_frame;
    VStateU8<S, 0> vs;

    __device__ __forceinline__ RegChain(const DownGeom &g_) : g(g_), lane(threadIdx.x) {}

    // horizontal 5-tap of a level-K row held as B = 16>>K columns per lane -> B/2 columns of level K+1
    template <int K> __device__ __forceinline__ void hfilter(const double (&v)[16 >> K], double (&n)[(16 >> K) / 2])
    {
        constexpr int B = 16 >> K;
        double pm2 = wave_from_prev(v[B - 2]), pm1 = wave_from_prev(v[B - 1]), nx = wave_from_next(v[0]);
        // BORDER_REFLECT_101: columns -2, -1 -> 2, 1 ; column w -> w-2
        const double l2 = (B >= 4) ? v[B >= 4 ? 2 : 0] : nx;
        pm2 = left_lane ? l2 : pm2;
        pm1 = left_lane ? v[1] : pm1;
        nx = last_lane ? v[B - 2] : nx;
#pragma unroll
        for (int j = 0; j < B / 2; ++j) {
            const double m2 = (j == 0) ? pm2 : v[2 * j - 2], m1 = (j == 0) ? pm1 : v[2 * j - 1];
            const double p2 = (2 * j + 2 < B) ? v[(2 * j + 2 < B) ? 2 * j + 2 : 0] : nx;
            n[j] = v[2 * j] * 6 + (m1 + v[2 * j + 1]) * 4 + m2 + p2;
        }
    }

    template <int K> __device__ __forceinline__ void emit(int y, const double (&v)[(16 >> K) / 2])
    {
        constexpr int NO = (16 >> K) / 2;
        next[K + 1] = y + 1;
        if constexpr (K + 1 == S) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const int col = col_S + j;
                if (store_ok && col >= 0 && col < g.w[S]) out_frame[(size_t)y * g.w[S] + col] = v[j] * (1.0 / 256);
            }
        } else {
            double row[NO], n[NO / 2];
#pragma unroll
            for (int j = 0; j < NO; ++j) row[j] = v[j] * (1.0 / 256);
            hfilter<K + 1>(row, n);
            feed<K + 1>(y, n);
        }
    }

    template <int K> __device__ __forceinline__ void feed(int p, const double (&n)[(16 >> K) / 2])
    {
        constexpr int NO = (16 >> K) / 2;
        VStateU8<S, K> &st = vs;
        const int hk = g.h[K];
        const int nv = (p == hk - 1) ? ((hk & 1) ? 2 : 1) : 0;   // virtual rows replay the reflected bottom rows
        const bool odd_h = (hk & 1) != 0;
        double cur[NO], held_a[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) { cur[j] = n[j]; held_a[j] = 0.0; }
#pragma nounroll
        for (int rep = 0; rep <= nv; ++rep) {
            step<K>(p + rep, cur);
            if (rep < nv) {  // only the last real row of a level has successors to prepare
                const bool first = rep == 0;
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    const double from_state = odd_h ? st.b[j] : st.c[j];
                    cur[j] = first ? from_state : held_a[j];
                    held_a[j] = first ? st.a[j] : held_a[j];
                }
            }
        }
    }

    // streaming vertical pass, identical to DownChain::step (hot form: every level has >= 3 rows)
    template <int K> __device__ __forceinline__ void step(int p, const double (&n)[(16 >> K) / 2])
    {
        constexpr int NO = (16 >> K) / 2;
        VStateU8<S, K> &st = vs;
        if (p & 1) {
            // row 1 of the image: rows -1, -2 reflect onto 1, 2, i.e. b := n and the `+ a` term moves to row 2
            // (x + -0.0 == x bit for bit)
            double be[NO], ae[NO];
#pragma unroll
            for (int j = 0; j < NO; ++j) { be[j] = st.b[j]; ae[j] = st.a[j]; }
            if (p == 1) {
#pragma unroll
                for (int j = 0; j < NO; ++j) { be[j] = n[j]; ae[j] = -0.0; }
            }
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                st.t[j] = (st.c[j] * 6 + (be[j] + n[j]) * 4) + ae[j];
                st.a[j] = st.c[j]; st.b[j] = n[j];
            }
        } else {
            const int y = (p >> 1) - 1;
            if (y == next[K + 1] && y <= last[K + 1]) {
                double v[NO];
#pragma unroll
                for (int j = 0; j < NO; ++j) v[j] = st.t[j] + n[j];
                if (p == 2) {  // row 2 of the image also stands for row -2
#pragma unroll
                    for (int j = 0; j < NO; ++j) v[j] = v[j] + n[j];
                }
                emit<K>(y, v);
            }
#pragma unroll
            for (int j = 0; j < NO; ++j) st.c[j] = n[j];
        }
    }

    __device__ __forceinline__ void run(const Tin *frame, double *out_t, int strip, int seg)
    {
        out_frame = out_t;
        const int W = g.w[0];
        next[S] = g.y_begin + seg * g.seg_h; last[S] = min(next[S] + g.seg_h, g.y_end) - 1;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) { next[k] = max(0, 2 * next[k + 1] - 2); last[k] = min(g.h[k] - 1, 2 * last[k + 1] + 2); }
        const int P = strip * U8_STRIP_PX - 32;               // first column of lane 0 (two halo lanes)
        const int c_first = P + 16 * lane;                     // this lane's first input column
        left_lane = (c_first == 0);
        last_lane = (c_first + 15 == W - 1);
        store_ok = lane >= 2 && lane <= 61;
        col_S = c_first >> S;                                  // exact: P and 16*lane are multiples of 16 >= 2^S
        const Tin *src = frame + min(max(c_first, 0), W - 16);
        const int p_first = next[0], p_last = last[0];
        Raw16 regs[PF][NLD];
        auto issue = [&](int row, Raw16 (&r)[NLD]) __attribute__((always_inline)) {
            const Tin *rp = src + (size_t)row * W;
#pragma unroll
            for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const Raw16 *>(rp + j * VPER);   // (plain: this kernel lives on cache hits for its strip halos -- non-temporal loads measured 0.43 -> 0.70 ms on the float32 buffer)
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) issue(min(p_first + i, p_last), regs[i]);
        for (int base = p_first; base <= p_last; base += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int p = base + i;
                if (p <= p_last) {
                    double v[16], n[8];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = unpack_px<Tin>(regs[i][e / VPER], e % VPER);
                    issue(min(p + PF, p_last), regs[i]);
                    hfilter<0>(v, n);
                    feed<0>(p, n);
                }
            }
        }
    }
};

template <int S, typename Tin = uint8_t>
__global__ __launch_bounds__(64) void k_down_chain_u8(const Tin *frames, size_t frame_stride, DownGeom g, double *out)
{
    const int per_frame = g.strips * g.segs;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = (j / per_frame) * 8 + xcd;
    if (t >= g.T) return;
    const int inner = j % per_frame;
    const int seg = inner / g.strips, strip = inner - seg * g.strips;
    RegChain<S, Tin> rc(g);
    rc.run(frames + (size_t)t * frame_stride, out + (size_t)t * g.h[S] * g.w[S], strip, seg);
}

// geometry: strips of 960 exact columns; enough segments for ~8 waves per CU
inline bool make_down_geom_u8(int S, const int *h, const int *w, int T, DownGeom &g, bool tiny = false)
{
    if (S < 1 || S > 4 || (w[0] % 16) != 0) return false;
    // every filtered level needs >= 3 rows (streaming vertical pass) and >= 3 columns: the in-lane border selects of
    // hfilter() realise BORDER_REFLECT_101 as -2 -> 2, -1 -> 1, w -> w-2, which is only what OpenCV does for w >= 3
    // (found by tools/fuzz_parity.py: 16-pixel-wide frames with 4 levels reach a 2-column level)
    for (int k = 0; k < S; ++k) if (h[k] < 3 || w[k] < 3) return false;
    g.S = S; g.T = T; g.vec = 1; g.y_begin = 0; g.y_end = h[S]; g.seg_split = 0; g.wpg = 1;
    for (int k = 0; k <= S; ++k) { g.h[k] = h[k]; g.w[k] = w[k]; }
    g.strips = (w[0] + U8_STRIP_PX - 1) / U8_STRIP_PX;
    const int rows = h[S];
    const int halo0 = (1 << (S + 1)) - 2;
    const long long per_seg = (long long)T * g.strips;
    int segs = (int)((2048 + per_seg / 2) / per_seg);
    if (segs < 1) segs = 1;
    if (segs > rows) segs = rows;
    while (segs > 1 && ((((rows + segs - 1) / segs) << S) < 8 * halo0)) --segs;
    g.seg_h = (rows + segs - 1) / segs;
    if (tiny) g.seg_h = rows < 2 ? rows : 2;
    g.segs = (rows + g.seg_h - 1) / g.seg_h;
    return true;
}

}  // namespace rm
