// respmon_amd/csrc/rm_down_chain.h -- the roofline kernel: fused Gaussian pyramid chain
// frames[T,H,W] (u8 / f16 / f32 / f64)  ->  G_S[T,h_S,w_S] float64   (S x cv2.pyrDown, pyramid.py:9-17)
//
// The frame buffer is read from HBM exactly once; levels 1..S-1 never leave the CU.
//
// Decomposition: one single-wave workgroup owns (frame, strip of level-S columns, segment of level-S
// rows) and MARCHES down the input rows.  Per level k it keeps in LDS one row buffer and a 5-row ring
// of horizontally filtered rows; whenever rows 2y-2..2y+2 of level k are in the ring, row y of level
// k+1 is produced by the vertical pass and pushed down the cascade (template recursion over levels).
// Nothing is recomputed vertically; horizontally a strip re-filters its halo (2^(S+1)-2 columns/side).
// Operation order per output is OpenCV's (horizontal 5-tap then vertical 5-tap), so
// the result is bit-identical to the per-level kernel and to the CPU oracle.
//
// LDS layout: row buffers are de-interleaved (even columns | odd columns) so the stride-2 taps of the
// horizontal pass are unit-stride, bank-conflict-free ds_read_b64.  Global loads are 16 B per lane,
// coalesced, software-prefetched DC_PREFETCH rows ahead in registers.  Workgroups are mapped so that
// all strips/segments of a frame run on one XCD (blockIdx % 8), sharing halo lines in that XCD's L2.
#pragma once
#include "rm_kernels.h"

namespace rm {

constexpr int DC_PREFETCH = 4;   // rows in flight per wave
constexpr int DC_MAX_LOADS = 3;  // 16-byte lane-loads per lane per row (strip width <= 64*3*V pixels)

struct DownGeom {
    int S;
    int h[MAX_CHAIN], w[MAX_CHAIN];  // level sizes, 0 = input
    int strip_w, seg_h;              // strip / segment size in level-S columns / rows
    int strips, segs;                // per frame
    int half[MAX_CHAIN];             // half-length (doubles) of the de-interleaved row buffer of level k
    int rowbuf_off[MAX_CHAIN];       // LDS offsets (doubles), k = 0..S-1
    int ring_off[MAX_CHAIN];
    int ring_pitch[MAX_CHAIN];       // >= width of the level k+1 column range
    int lds_total;
    int T;
    int vec;                         // 1: 16-byte aligned vector loads are legal for this buffer
};

template <typename Tin> struct VecTraits;
template <> struct VecTraits<double> { static constexpr int V = 2; };
template <> struct VecTraits<float> { static constexpr int V = 4; };
template <> struct VecTraits<__half> { static constexpr int V = 8; };
template <> struct VecTraits<uint8_t> { static constexpr int V = 16; };

struct alignas(16) Raw16 { unsigned int x, y, z, w; };

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

template <typename Tin, int S>
struct DownChain {
    const DownGeom &g;
    double *lds;
    const int lane;
    int cx0[S + 1], cx1[S + 1];  // column range of each level held by this strip (inclusive, clamped)
    int c0[S];                   // even-aligned base column of row buffer k
    int next[S + 1], last[S + 1];
    int slot[S];                 // ring slot of the row being pushed at level k (row % 5)
    double *out_frame;           // G_S of this frame

    __device__ __forceinline__ DownChain(const DownGeom &g_, double *lds_) : g(g_), lds(lds_), lane(threadIdx.x) {}

    __device__ __forceinline__ int rb_index(int k, int c) const
    {
        int i = c - c0[k];
        return g.rowbuf_off[k] + (i >> 1) + (i & 1) * g.half[k];
    }

    // horizontal 5-tap at level K for output column x of level K+1 (unnormalised)
    template <int K> __device__ __forceinline__ double hfilter(int x) const
    {
        const int c = 2 * x, wk = g.w[K];
        if (c - 2 >= 0 && c + 2 < wk) {
            const double *ev = lds + g.rowbuf_off[K] + ((c - c0[K]) >> 1);
            const double *od = ev + g.half[K];
            return ev[0] * 6 + (od[-1] + od[0]) * 4 + ev[-1] + ev[1];
        }
        double m2 = lds[rb_index(K, reflect101(c - 2, wk))], m1 = lds[rb_index(K, reflect101(c - 1, wk))];
        double p1 = lds[rb_index(K, reflect101(c + 1, wk))], p2 = lds[rb_index(K, reflect101(c + 2, wk))];
        return lds[rb_index(K, reflect101(c, wk))] * 6 + (m1 + p1) * 4 + m2 + p2;
    }

    // level-K row p sits in row buffer K: filter it into the ring, fire every level K+1 row that became
    // computable, and cascade
    template <int K> __device__ __forceinline__ void push(int p)
    {
        const int sl = slot[K];
        slot[K] = (sl == 4) ? 0 : sl + 1;
        {
            double *ring = lds + g.ring_off[K] + sl * g.ring_pitch[K];
            for (int x = cx0[K + 1] + lane; x <= cx1[K + 1]; x += 64) ring[x - cx0[K + 1]] = hfilter<K>(x);
        }
        __syncthreads();
        const int hk = g.h[K];
        while (next[K + 1] <= last[K + 1]) {
            const int y = next[K + 1];
            const int need = (2 * y + 2 < hk - 1) ? 2 * y + 2 : hk - 1;
            if (need > p) break;
            next[K + 1] = y + 1;
            // ring slot of physical row r: rows are pushed consecutively, row p is in slot sl
            int s0, s1, s2, s3, s4;
            {
                auto slot_of_row = [&](int r) { int d = (p - r) % 5; int q = sl - d; return q < 0 ? q + 5 : q; };
                s0 = slot_of_row(reflect101(2 * y - 2, hk));
                s1 = slot_of_row(reflect101(2 * y - 1, hk));
                s2 = slot_of_row(reflect101(2 * y, hk));
                s3 = slot_of_row(reflect101(2 * y + 1, hk));
                s4 = slot_of_row(reflect101(2 * y + 2, hk));
            }
            const double *rbase = lds + g.ring_off[K];
            const int pitch = g.ring_pitch[K];
            for (int x = cx0[K + 1] + lane; x <= cx1[K + 1]; x += 64) {
                const int i = x - cx0[K + 1];
                double v = (rbase[s2 * pitch + i] * 6 + (rbase[s1 * pitch + i] + rbase[s3 * pitch + i]) * 4 +
                            rbase[s0 * pitch + i] + rbase[s4 * pitch + i]) * (1.0 / 256);
                if constexpr (K + 1 == S) out_frame[(size_t)y * g.w[S] + x] = v;
                else lds[rb_index(K + 1, x)] = v;
            }
            __syncthreads();
            if constexpr (K + 1 < S) push<K + 1>(y);
        }
    }

    __device__ __forceinline__ void run(const Tin *frame, double *out_t, int strip, int seg)
    {
        constexpr int V = VecTraits<Tin>::V;
        out_frame = out_t;
        // ranges, from level S back to 0
        cx0[S] = strip * g.strip_w; cx1[S] = min(cx0[S] + g.strip_w, g.w[S]) - 1;
        next[S] = seg * g.seg_h; last[S] = min(next[S] + g.seg_h, g.h[S]) - 1;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {
            cx0[k] = max(0, 2 * cx0[k + 1] - 2); cx1[k] = min(g.w[k] - 1, 2 * cx1[k + 1] + 2);
            next[k] = max(0, 2 * next[k + 1] - 2); last[k] = min(g.h[k] - 1, 2 * last[k + 1] + 2);
            c0[k] = cx0[k] & ~1;
            slot[k] = 0;
        }
        const bool vec = g.vec != 0;
        if (vec) c0[0] = cx0[0] & ~(V - 1);
        const int W = g.w[0];
        const int nload = vec ? (cx1[0] - c0[0] + V) / V : 0;  // lane-loads per row (vector path)

        auto issue = [&](int row, Raw16 (&r)[DC_MAX_LOADS]) {
            const Tin *src = frame + (size_t)row * W + c0[0];
#pragma unroll
            for (int q = 0; q < DC_MAX_LOADS; ++q) {
                int j = lane + 64 * q;
                if (j < nload) r[q] = *reinterpret_cast<const Raw16 *>(src + (size_t)j * V);
            }
        };
        auto stash = [&](const Raw16 (&r)[DC_MAX_LOADS]) {
#pragma unroll
            for (int q = 0; q < DC_MAX_LOADS; ++q) {
                int j = lane + 64 * q;
                if (j < nload) {
                    double *ev = lds + g.rowbuf_off[0] + (j * V) / 2;
                    double *od = ev + g.half[0];
#pragma unroll
                    for (int e = 0; e < V; e += 2) {
                        ev[e >> 1] = unpack_px<Tin>(r[q], e);
                        od[e >> 1] = unpack_px<Tin>(r[q], e + 1);
                    }
                }
            }
        };

        const int p_first = next[0], p_last = last[0];
        if (vec) {
            Raw16 regs[DC_PREFETCH][DC_MAX_LOADS];
#pragma unroll
            for (int i = 0; i < DC_PREFETCH; ++i)
                if (p_first + i <= p_last) issue(p_first + i, regs[i]);
            for (int base = p_first; base <= p_last; base += DC_PREFETCH) {
#pragma unroll
                for (int i = 0; i < DC_PREFETCH; ++i) {
                    const int p = base + i;
                    if (p <= p_last) {
                        stash(regs[i]);
                        if (p + DC_PREFETCH <= p_last) issue(p + DC_PREFETCH, regs[i]);
                        __syncthreads();
                        push<0>(p);
                    }
                }
            }
        } else {
            for (int p = p_first; p <= p_last; ++p) {
                const Tin *src = frame + (size_t)p * W;
                for (int c = cx0[0] + lane; c <= cx1[0]; c += 64) lds[rb_index(0, c)] = load_px(src, (size_t)c);
                __syncthreads();
                push<0>(p);
            }
        }
    }
};

template <typename Tin, int S>
__global__ __launch_bounds__(64) void k_down_chain(const Tin *frames, size_t frame_stride, DownGeom g, double *out)
{
    HIP_DYNAMIC_SHARED(double, lds)
    // XCD-aware mapping: block b runs on XCD b % 8; give each XCD whole frames so the strips and
    // segments of a frame share halo lines in one L2
    const int per_frame = g.strips * g.segs;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = (j / per_frame) * 8 + xcd;
    if (t >= g.T) return;
    const int inner = j % per_frame;
    const int seg = inner / g.strips, strip = inner - seg * g.strips;
    DownChain<Tin, S> dc(g, lds);
    dc.run(frames + (size_t)t * frame_stride, out + (size_t)t * g.h[S] * g.w[S], strip, seg);
}

// host-side geometry
inline int down_chain_level0_width(int S, int strip_w)
{
    int w = strip_w;
    for (int k = 0; k < S; ++k) w = 2 * w + 3;  // 2*(x1)+2 - (2*x0-2) + 1
    return w;
}

constexpr int DC_LDS_BUDGET = 2304;  // doubles per wave (18 KB): >= 8 resident waves per CU

inline void down_chain_layout(DownGeom &g, int V)
{
    const int S = g.S;
    int widths[MAX_CHAIN];
    widths[S] = g.strip_w;
    for (int k = S - 1; k >= 0; --k) widths[k] = 2 * widths[k + 1] + 3;
    int off = 0;
    for (int k = 0; k < S; ++k) {
        int span = widths[k] + 2 + (k == 0 ? 2 * (V < 2 ? 2 : V) : 2);
        g.half[k] = (span + 1) / 2 + 1;
        g.rowbuf_off[k] = off; off += 2 * g.half[k];
        g.ring_pitch[k] = widths[k + 1] + 1;
        g.ring_off[k] = off; off += 5 * g.ring_pitch[k];
    }
    g.lds_total = off;
}

inline bool make_down_geom(int S, const int *h, const int *w, int T, int vec_ok, int V, DownGeom &g, bool tiny = false)
{
    if (S < 1 || S >= MAX_CHAIN) return false;
    g.S = S; g.T = T; g.vec = vec_ok;
    for (int k = 0; k <= S; ++k) { g.h[k] = h[k]; g.w[k] = w[k]; }
    // widest strip whose level-0 span fits DC_MAX_LOADS vector loads per lane and the LDS budget
    const int max_px0 = 64 * DC_MAX_LOADS * (V < 2 ? 2 : V) - 2 * V;
    int sw = g.w[S];
    for (;; --sw) {
        g.strip_w = sw;
        down_chain_layout(g, V);
        if (sw == 1) break;
        if (down_chain_level0_width(S, sw) <= max_px0 && g.lds_total <= DC_LDS_BUDGET) break;
    }
    int strips = (g.w[S] + sw - 1) / sw;
    sw = (g.w[S] + strips - 1) / strips;  // balance the strips
    g.strip_w = sw; g.strips = strips;
    down_chain_layout(g, V);
    // segments: enough workgroups to fill the chip (~24 waves per CU over the launch) while the vertical
    // halo (2^(S+1)-2 input rows per side) stays below ~25 % of a segment
    int segs = 1;
    const int halo0 = (1 << (S + 1)) - 2;
    while ((long long)T * strips * segs < 256LL * 24 && segs < g.h[S]) {
        int seg_h = (g.h[S] + segs) / (segs + 1);
        if ((seg_h << S) < 8 * halo0) break;
        ++segs;
    }
    g.seg_h = (g.h[S] + segs - 1) / segs;
    g.segs = (g.h[S] + g.seg_h - 1) / g.seg_h;
    if (tiny) {  // test hook: many small strips and segments
        g.strip_w = g.w[S] < 3 ? g.w[S] : 3; g.strips = (g.w[S] + g.strip_w - 1) / g.strip_w;
        g.seg_h = g.h[S] < 2 ? g.h[S] : 2; g.segs = (g.h[S] + g.seg_h - 1) / g.seg_h;
        down_chain_layout(g, V);
    }
    return true;
}

}  // namespace rm
