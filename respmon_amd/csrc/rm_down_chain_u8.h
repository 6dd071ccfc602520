// respmon_amd/csrc/rm_down_chain_u8.h -- fused Gaussian pyramid chain for NARROW frame buffers (uint8, float16, float32)
// frames[T,H,W] (W % 16 == 0) -> G_S[T,h_S,w_S] float64, S <= 4                   (pyramid.py:9-17)
// ... and for [T,H,W,3] uint8 BGR frame buffers (bgr8_t below: base.py:230's cvtColor fused into the first row pass)
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
// developer knobs (bench_micro/dc8_bench.hip): register budget of the kernel, hot blocks on / off, scheduling fence between rows
#ifndef RM_NARROW_WAVES
#define RM_NARROW_WAVES 2
#endif
#if RM_NARROW_WAVES > 0
#define RM_NARROW_OCC __attribute__((amdgpu_waves_per_eu(RM_NARROW_WAVES, RM_NARROW_WAVES)))
#else
#define RM_NARROW_OCC
#endif
#ifndef RM_NARROW_HOT
#define RM_NARROW_HOT 1
#endif
#ifndef RM_NARROW_FENCE
#define RM_NARROW_FENCE 1
#endif
#ifndef RM_U8_LUT
#define RM_U8_LUT 1
#endif

template <int S, int K> struct VStateU8 : VStateU8<S, K + 1> {
    double a[(16 >> K) / 2], b[(16 >> K) / 2], c[(16 >> K) / 2], t[(16 >> K) / 2];
};
template <int S> struct VStateU8<S, S> {};

// [H,W,3] uint8 pixels as cv2.VideoCapture.read() delivers them (base.py:229) -- RM_BGR8, north_star's [T,H,W,C] frame buffer.  The
// chain applies base.py:230-231 while it unpacks a row: cv2.cvtColor(BGR2GRAY) is OpenCV's 14-bit fixed point
//     Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14
// and uint8_to_float is the table look-up of the gray path.  A lane still owns 16 adjacent pixels: 48 bytes, three 16-byte loads.
// The weights do not fit a byte, so the sum is two v_dot4_u32_u8 (low and high bytes of the weights; the fourth byte of the word
// meets a zero weight) joined by one v_lshl_add_u32; pixels whose three bytes straddle two words are brought together by
// v_alignbyte_b32 first: 5.5 integer instructions per pixel instead of the 1 of a gray row, bit-identical to k_bgr_to_gray.
struct bgr8_t { uint8_t b, g, r; };
static_assert(sizeof(bgr8_t) == 3, "packed BGR pixel");
template <typename Tin> struct IsBgr { static constexpr bool value = false; };
template <> struct IsBgr<bgr8_t> { static constexpr bool value = true; };
template <typename Tin> struct RegTraits;   // NLD: 16-byte loads per lane and row; PF: rows in flight (a divisor of the hot block, 4)
// HOT: the static steady-state blocks of RegChain pay (measured, bench_micro/dc8_bench.hip, kernel ms old -> new): uint8 1080p x 256
// 0.365 -> 0.30, float16 4K x 512 2.19 -> 2.03; the float32 chain is bound by bytes in flight, not by instruction issue, and
// keeps the row-at-a-time form without a register cap (0.42 ms; 0.47 with hot blocks, 1.1 under a 256-register cap)
// DMA (developer option, off): rows travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of RD rows per wave, RD - 1
// of them in flight, and cost no registers until they are unpacked.  Written because the float16 / float32 chains spend half of
// their wave cycles in s_waitcnt with 2 rows per wave in flight (profiles/r03/narrow_chain_sq_counters.txt: 55 % / 50 %) -- and
// measured no faster (bench_micro/dc8_bench.hip, profiles/r03/narrow_chain_sweep.txt: float16 4K 2.13 -> 2.10 ms, float32 1080p
// 0.45 -> 0.50 ms, bit-identical output): the reads of the chain's access pattern ALONE take 1.45 ms / 0.435 ms there, and the
// float16 kernel also writes its 2.1 GB level-2 array -- both kernels already run at the bandwidth their traffic allows.
#ifndef RM_NARROW_DMA
#define RM_NARROW_DMA 0
#endif
#ifdef RM_HIPEMU
#define RM_NARROW_DMA_ON 0
#else
#define RM_NARROW_DMA_ON RM_NARROW_DMA
#endif
template <> struct RegTraits<uint8_t> { static constexpr int NLD = 1, PF = RM_U8_PREFETCH, RD = 8; static constexpr bool HOT = RM_NARROW_HOT != 0, DMA = false; };
#ifndef RM_F16_PREFETCH
#define RM_F16_PREFETCH 2
#endif
#ifndef RM_F16_RING
#define RM_F16_RING 8
#endif
#ifndef RM_F32_RING
#define RM_F32_RING 4
#endif
#ifndef RM_F16_HOT
#define RM_F16_HOT RM_NARROW_HOT
#endif
template <> struct RegTraits<__half> { static constexpr int NLD = 2, PF = RM_F16_PREFETCH, RD = RM_F16_RING; static constexpr bool HOT = RM_F16_HOT != 0, DMA = RM_NARROW_DMA_ON != 0; };
#ifndef RM_F32_HOT
#define RM_F32_HOT 0
#endif
#ifndef RM_F32_PREFETCH
#define RM_F32_PREFETCH 4   // (1080p x 256, same box, alternating processes: 2 rows in flight / 3 segments 0.4466-0.4473 ms, 4 rows / 2 segments 0.4325-0.4335: tools/r05_f32_pf.sh)
#endif
#ifndef RM_BGR_PREFETCH
#define RM_BGR_PREFETCH 4   // (measured at 1080p x 256, depth 4: 2 rows in flight 0.364-0.370 ms, 4 rows and HALF the waves -- two row segments per frame instead of four -- 0.343-0.347)
#endif
template <> struct RegTraits<bgr8_t> { static constexpr int NLD = 3, PF = RM_BGR_PREFETCH, RD = 8; static constexpr bool HOT = RM_NARROW_HOT != 0, DMA = false; };
template <> struct RegTraits<float> { static constexpr int NLD = 4, PF = RM_F32_PREFETCH, RD = RM_F32_RING; static constexpr bool HOT = RM_F32_HOT != 0, DMA = RM_NARROW_DMA_ON != 0; };
// LDS bytes per wave of the row ring (0: rows in registers)
template <typename Tin> constexpr size_t narrow_ring_bytes() { return RegTraits<Tin>::DMA ? (size_t)RegTraits<Tin>::RD * RegTraits<Tin>::NLD * 1024 : 0; }

template <int S, typename Tin = uint8_t>
struct RegChain {
    static_assert(S >= 1 && S <= 4, "a lane owns 16 >> K columns of level K");
    static constexpr int NLD = RegTraits<Tin>::NLD, VPER = 16 / NLD;  // VPER pixels per 16-byte load (BGR: pixels straddle the loads)
    static constexpr bool BGR = IsBgr<Tin>::value, BYTES = sizeof(Tin) == 1 || BGR;   // 8-bit samples: table look-up, deep hot blocks
    // Levels 0 .. D-1 have a STATIC steady state ("hot blocks" of B = 2^D input rows): away from the image top / bottom and
    // once a level's (a, b, c, t) state is warm, the even / odd role of every row of these levels inside an aligned block is
    // known at compile time, so the block is straight-line code -- no parity test, no border select, no state copies (the
    // generic row-at-a-time form below spends more instructions on those than on arithmetic).  Levels D .. S-1 (1/16 of the
    // pixels and less) and every row near a segment start, the image top or the image bottom take the generic form.
    static constexpr bool HOT = RegTraits<Tin>::HOT && (BYTES || S <= 2);   // (float16 at depth 3 / 4: the hot blocks do not fit 256 registers -- compiler resource report)
#ifndef RM_NARROW_D
#define RM_NARROW_D 2
#endif
    static constexpr int D = S < RM_NARROW_D ? S : RM_NARROW_D;
    static constexpr int B = 1 << D;
    static constexpr int PRIO_BLOCKS_SHIFT = D >= 4 ? 0 : 4 - D;   // the issue priority rotates every 16 input rows (rm_down_chain.h dc_set_prio)
    // (BGR at depth 2 -- the instantiation with the most live state, 250 registers for gray uint8 -- keeps ONE row in flight: a second
    //  one spills 48 bytes per lane)
    static constexpr int PF_T = (BGR && S == 2) ? 1 : RegTraits<Tin>::PF;
    static constexpr int PF = !HOT ? PF_T : PF_T < B ? PF_T : B;
    static_assert(!HOT || B % PF == 0, "a row's prefetch slot must be static inside a hot block");
    // Every pyrDown ends with an exact scaling by 1/256 (pyramid.py:14 -> cv2.pyrDown); it commutes with the roundings of the
    // levels above it (powers of two, magnitudes nowhere near the exponent limits for uint8 / float16 / float32 data), so the
    // chain carries UNSCALED values and the store applies 2^(-8 S) once.
    static constexpr double OUT_SCALE = S == 1 ? 1.0 / 256 : S == 2 ? 1.0 / 65536 : S == 3 ? 1.0 / 16777216 : 1.0 / 4294967296.0;
    const DownGeom &g;
    int prio_rank_ = 0;
    const int lane;
    int next[S + 1], last[S + 1];
    int p_first, p_last, W;
    const Tin *src;
    bool left_lane, last_lane;   // this lane holds column 0 / the last column of every level
    int col_S;                   // level-S column of this lane's first owned column
    bool store_ok;               // lane is exact (2..61)
    double *out_frame;
    VStateU8<S, 0> vs;

    __device__ __forceinline__ RegChain(const DownGeom &g_) : g(g_), lane(threadIdx.x) {}

    // horizontal 5-tap of a level-K row held as B = 16>>K columns per lane -> B/2 columns of level K+1
    template <int K> __device__ __forceinline__ void hfilter(const double (&v)[16 >> K], double (&n)[(16 >> K) / 2])
    {
        constexpr int BW = 16 >> K;
        double pm2 = wave_from_prev(v[BW - 2]), pm1 = wave_from_prev(v[BW - 1]), nx = wave_from_next(v[0]);
        // BORDER_REFLECT_101: columns -2, -1 -> 2, 1 ; column w -> w-2
        const double l2 = (BW >= 4) ? v[BW >= 4 ? 2 : 0] : nx;
        pm2 = left_lane ? l2 : pm2;
        pm1 = left_lane ? v[1] : pm1;
        nx = last_lane ? v[BW - 2] : nx;
#pragma unroll
        for (int j = 0; j < BW / 2; ++j) {
            const double m2 = (j == 0) ? pm2 : v[2 * j - 2], m1 = (j == 0) ? pm1 : v[2 * j - 1];
            const double p2 = (2 * j + 2 < BW) ? v[(2 * j + 2 < BW) ? 2 * j + 2 : 0] : nx;
            // ((v*6 + (m1 + p1)*4) + m2) + p2: the product by 4 is exact, so it rides an FMA (same rounding, one instruction less)
            n[j] = __builtin_fma(m1 + v[2 * j + 1], 4.0, v[2 * j] * 6) + m2 + p2;
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
                if (store_ok && col >= 0 && col < g.w[S]) out_frame[(size_t)y * g.w[S] + col] = v[j] * OUT_SCALE;
            }
        } else {
            double n[NO / 2];
            hfilter<K + 1>(v, n);
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
                st.t[j] = __builtin_fma(be[j] + n[j], 4.0, st.c[j] * 6) + ae[j];
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

    // ---- hot blocks ------------------------------------------------------------------------------------------------------
    // Row J (compile-time) of level K inside a block that starts on an aligned row: even J emits row J / 2 of level K + 1.
    template <int K, int J> __device__ __forceinline__ void hot_row(const double (&n)[(16 >> K) / 2])
    {
        constexpr int NO = (16 >> K) / 2;
        VStateU8<S, K> &st = vs;
        if constexpr (J & 1) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                st.t[j] = __builtin_fma(st.b[j] + n[j], 4.0, st.c[j] * 6) + st.a[j];
                st.a[j] = st.c[j]; st.b[j] = n[j];
            }
        } else {
            double v[NO];
#pragma unroll
            for (int j = 0; j < NO; ++j) { v[j] = st.t[j] + n[j]; st.c[j] = n[j]; }
            if constexpr (K + 1 == D) {
                emit<K>(next[D], v);   // level D (or the store, D == S) in the generic form; hot_ok() vouches for the row index
            } else {
                double n2[NO / 2];
                hfilter<K + 1>(v, n2);
                hot_row<K + 1, J / 2>(n2);
            }
        }
    }

    // may rows [base, base + B) run as a hot block?  (wave-uniform)
    __device__ __forceinline__ bool hot_ok(int base) const
    {
        bool ok = base >= p_first && base + B - 1 <= p_last;
        int y = base;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int cnt = B >> k;
            // no top-of-image row forms (rows 1 and 2), the last real row (its virtual successors) is not in the block
            ok = ok && y >= 3 && y + cnt - 1 < g.h[k] - 1;
            const int y2 = (y >> 1) - 1;                      // first row of level k + 1 the block emits
            ok = ok && y2 == next[k + 1] && y2 + (cnt >> 1) - 1 <= last[k + 1];   // level k is warm, and every emission is wanted
            y = y2;
        }
        return ok;
    }

    // rows are independent instruction streams over the same (a, b, c, t) registers: without a fence between them the scheduler
    // hoists the unpacking of all B rows of a block to the front (128 more live registers, one wave per SIMD)
    static __device__ __forceinline__ void row_fence()
    {
#if RM_NARROW_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
    }

    // ---- rows through LDS (RegTraits::DMA) -------------------------------------------------------------------------------
    static constexpr bool DMA = RegTraits<Tin>::DMA;
    static constexpr int RD = RegTraits<Tin>::RD;
    static_assert((RD & (RD - 1)) == 0 && RD >= 4, "ring depth: a power of two");
    unsigned ring_lds = 0;          // LDS byte address of this wave's ring (wave-uniform)
    const char *ring_ptr = nullptr; // ... the same as a pointer, plus this lane's 16 bytes
    int base0_ = 0;

    __device__ __forceinline__ int slot_of(int row) const { return (row - base0_) & (RD - 1); }

    // one row: NLD pieces of 1 KB (lane l's 16 bytes at + 16 l).  The statement sets M0 (the DMA's LDS base) itself and hides
    // the load from the compiler, which would otherwise wait vmcnt(0) in front of every LDS read.
    __device__ __forceinline__ void dma_issue(int row, int slot) const
    {
#if !defined(RM_HIPEMU)
        const Tin *rp = src + (size_t)row * W;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            unsigned keep;
            const unsigned dst = ring_lds + (unsigned)(slot * NLD + j) * 1024u;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(rp + j * VPER), "s"(dst) : "memory");
        }
#else
        (void)row; (void)slot;
#endif
    }
    // the row in `slot` has landed once at most (RD - 2) NLD younger DMA pieces are outstanding (loads retire in order; the
    // chain's own stores only raise the count, i.e. make this wait longer)
    __device__ __forceinline__ void dma_fetch(int slot, Raw16 (&r)[NLD]) const
    {
#if !defined(RM_HIPEMU)
        constexpr int K = (RD - 2) * NLD;
        __builtin_amdgcn_s_waitcnt((K & 0xF) | ((K >> 4) << 14) | 0x0F70);   // vmcnt(K) only
        asm volatile("" ::: "memory");
#endif
#pragma unroll
        for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const Raw16 *>(ring_ptr + (size_t)(slot * NLD + j) * 1024);
    }

    template <int I> __device__ __forceinline__ void hot_rows_dma(int base)
    {
        if constexpr (I < B) {
            double v[16], n[8];
            Raw16 cur[NLD];
            dma_fetch(slot_of(base + I), cur);
            unpack_row(cur, v);
            dma_issue(min(base + I + RD - 1, p_last), slot_of(base + I + RD - 1));   // (= the slot of row base + I - 1, read in the previous trip)
            hfilter<0>(v, n);
            hot_row<0, I>(n);
            row_fence();
            hot_rows_dma<I + 1>(base);
        }
    }

    __device__ __forceinline__ void issue(int row, Raw16 (&r)[NLD]) const
    {
        const char *rp = reinterpret_cast<const char *>(src + (size_t)row * W);
#pragma unroll
        for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const Raw16 *>(rp + 16 * j);   // (plain: this kernel lives on cache hits for its strip halos -- non-temporal loads measured 0.43 -> 0.70 ms on the float32 buffer)
    }

    // uint8 with RM_U8_LUT: uint8_to_float's 256 possible values k * (1./255) come from a 2 KB table in LDS -- one SDWA shift
    // (byte k of the word, times 8) and one ds_read_b64 per pixel instead of bit-field extract + convert + multiply (three
    // instructions on the units this kernel is bound by; the LDS reads run beside them)
    const double *lut = nullptr;
    template <int K> static __device__ __forceinline__ unsigned lut_offset(unsigned w)
    {
#ifdef RM_HIPEMU
        return ((w >> (8 * K)) & 0xffu) << 3;
#else
        unsigned off;
        const unsigned three = 3;
        if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(three), "v"(w));
        else if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(three), "v"(w));
        else if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(three), "v"(w));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(three), "v"(w));
        return off;
#endif
    }
    template <int E> __device__ __forceinline__ void unpack_lut(const Raw16 &r, double (&v)[16]) const
    {
        if constexpr (E < 16) {
            const unsigned w = (E >> 2) == 0 ? r.x : (E >> 2) == 1 ? r.y : (E >> 2) == 2 ? r.z : r.w;
            v[E] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lut) + lut_offset<E & 3>(w));
            unpack_lut<E + 1>(r, v);
        }
    }

    // BGR: pixel E starts at byte 3 E of the lane's 48
    template <int Q> static __device__ __forceinline__ unsigned word_of(const Raw16 (&r)[NLD])
    {
        constexpr int J = Q >> 2 < NLD ? Q >> 2 : NLD - 1;
        return (Q & 3) == 0 ? r[J].x : (Q & 3) == 1 ? r[J].y : (Q & 3) == 2 ? r[J].z : r[J].w;
    }
    template <int E> __device__ __forceinline__ void unpack_bgr(const Raw16 (&r)[NLD], double (&v)[16]) const
    {
        if constexpr (E < 16) {
            constexpr int Q = (3 * E) >> 2, SH = (3 * E) & 3;
            const unsigned off = bgr_gray_x8<SH>(word_of<Q>(r), word_of<(Q + 1 < 4 * NLD ? Q + 1 : Q)>(r));
            if constexpr (RM_U8_LUT) v[E] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lut) + off);
            else v[E] = (double)(off >> 3) * (1.0 / 255);
            unpack_bgr<E + 1>(r, v);
        }
    }

    __device__ __forceinline__ void unpack_row(const Raw16 (&r)[NLD], double (&v)[16]) const
    {
        if constexpr (BGR) {
            unpack_bgr<0>(r, v);
        } else if constexpr (sizeof(Tin) == 1 && RM_U8_LUT) {
            unpack_lut<0>(r[0], v);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = unpack_px<Tin>(r[e / VPER], e % VPER);
        }
    }

    template <int I> __device__ __forceinline__ void hot_rows(int base, Raw16 (&regs)[PF][NLD])
    {
        if constexpr (I < B) {
            double v[16], n[8];
            unpack_row(regs[I % PF], v);
            issue(min(base + I + PF, p_last), regs[I % PF]);
            hfilter<0>(v, n);
            hot_row<0, I>(n);
            row_fence();
            hot_rows<I + 1>(base, regs);
        }
    }

    __device__ __forceinline__ void run(const Tin *frame, double *out_t, int strip, int seg)
    {
        out_frame = out_t;
        W = g.w[0];
        next[S] = g.y_begin + seg * g.seg_h; last[S] = min(next[S] + g.seg_h, g.y_end) - 1;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) { next[k] = max(0, 2 * next[k + 1] - 2); last[k] = min(g.h[k] - 1, 2 * last[k + 1] + 2); }
        const int P = strip * U8_STRIP_PX - 32;               // first column of lane 0 (two halo lanes)
        const int c_first = P + 16 * lane;                     // this lane's first input column
        left_lane = (c_first == 0);
        last_lane = (c_first + 15 == W - 1);
        store_ok = lane >= 2 && lane <= 61;
        col_S = c_first >> S;                                  // exact: P and 16*lane are multiples of 16 >= 2^S
        src = frame + min(max(c_first, 0), W - 16);
        p_first = next[0]; p_last = last[0];
        if constexpr (DMA) {
            base0_ = p_first - ((p_first + 2) & (B - 1));
            for (int i = 0; i < RD - 1; ++i) dma_issue(min(p_first + i, p_last), slot_of(p_first + i));
            int base = base0_;
            while (base <= p_last) {
                if constexpr (HOT) {
                    if (hot_ok(base)) {
                        do {
                            hot_rows_dma<0>(base);
#pragma unroll
                            for (int k = 1; k < D; ++k) next[k] += B >> k;
                            base += B;
                        } while (hot_ok(base));
                        continue;
                    }
                }
                const int p_end = min(base + B - 1, p_last);
#pragma nounroll
                for (int p = max(base, p_first); p <= p_end; ++p) {
                    Raw16 cur[NLD];
                    double v[16], n[8];
                    dma_fetch(slot_of(p), cur);
                    unpack_row(cur, v);
                    dma_issue(min(p + RD - 1, p_last), slot_of(p + RD - 1));
                    hfilter<0>(v, n);
                    feed<0>(p, n);
                    row_fence();
                }
                base += B;
            }
            return;
        }
        if constexpr (!HOT) {
            // row at a time, PF rows in flight in static register sets
            Raw16 regs[PF][NLD];
#pragma unroll
            for (int i = 0; i < PF; ++i) issue(min(p_first + i, p_last), regs[i]);
            for (int base = p_first; base <= p_last; base += PF) {
                if (g.prio == 2 && ((base - p_first) & 15) < PF) dc_set_prio(prio_rank_ + ((base - p_first) >> 4));   // every 16 input rows (rm_down_chain.h dc_set_prio)
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int p = base + i;
                    if (p <= p_last) {
                        double v[16], n[8];
                        unpack_row(regs[i], v);
                        issue(min(p + PF, p_last), regs[i]);
                        hfilter<0>(v, n);
                        feed<0>(p, n);
                    }
                }
            }
            return;
        }
        // rows are taken in chunks of B whose first row is = -2 (mod B): the alignment at which an even row of level 0 emits an
        // even row of level 1 (hot_row); a row's prefetch slot is its offset from base0 modulo PF (PF divides B)
        const int base0 = p_first - ((p_first + 2) & (B - 1));
        Raw16 regs[PF][NLD];
#pragma unroll
        for (int i = 0; i < PF; ++i)   // slot i <- the first row >= p_first that belongs to it
            issue(min(p_first + ((i - (p_first - base0)) & (PF - 1)), p_last), regs[i]);
        // Three loops one after the other -- generic chunks until the levels are warm, ONE run of hot blocks, generic chunks to the
        // end of the segment -- instead of one loop that picks the form per chunk: with the forms alternating inside a loop the
        // register allocator moved the (a, b, c, t) state through scratch at every change of form (72 spilled registers, 1.6x the
        // algorithmic traffic: VERDICT r3).  A segment is warm-up, steady state, bottom; a chunk that is not hot after the run of hot
        // blocks (the image bottom, the segment's last rows) is never followed by a hot one that matters: the generic form is
        // always valid.
        int base = base0;
        auto generic_chunk = [&](int b0) __attribute__((always_inline)) {
            const int p_end = min(b0 + B - 1, p_last);
#pragma nounroll
            for (int p = max(b0, p_first); p <= p_end; ++p) {
                const int slot = (p - base0) & (PF - 1);
                Raw16 cur[NLD];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (slot == i) {
#pragma unroll
                        for (int j = 0; j < NLD; ++j) cur[j] = regs[i][j];
                        issue(min(p + PF, p_last), regs[i]);
                    }
                double v[16], n[8];
                unpack_row(cur, v);
                hfilter<0>(v, n);
                feed<0>(p, n);
                row_fence();
            }
        };
#pragma nounroll
        while (base <= p_last && !hot_ok(base)) { generic_chunk(base); base += B; }
        int nb = 0;
#pragma nounroll
        while (base <= p_last && hot_ok(base)) {
            if (g.prio == 2 && (nb & ((1 << PRIO_BLOCKS_SHIFT) - 1)) == 0) dc_set_prio(prio_rank_ + (nb >> PRIO_BLOCKS_SHIFT));   // (rm_down_chain.h dc_set_prio)
            ++nb;
            hot_rows<0>(base, regs);
#pragma unroll
            for (int k = 1; k < D; ++k) next[k] += B >> k;   // (next[D] was set by emit)
            base += B;
        }
#pragma nounroll
        while (base <= p_last) { generic_chunk(base); base += B; }
    }
};

template <int S, typename Tin> __device__ __forceinline__ void down_chain_narrow_body(const Tin *frames, size_t frame_stride, const DownGeom &g, double *out);

// float32 frame buffers: no register cap (RegTraits<float>::HOT == false)
template <int S, typename Tin = float>
__global__ __launch_bounds__(64) void k_down_chain_narrow(const Tin *frames, size_t frame_stride, DownGeom g, double *out)
{
    down_chain_narrow_body<S, Tin>(frames, frame_stride, g, out);
}

// uint8 / float16 frame buffers: two waves per SIMD, i.e. a cap of 256 registers.  Every instantiation holds its state in registers
// (156-214 VGPRs, no scratch: respmon_amd/csrc/build_resources.txt; tools/check_resources.py fails the build otherwise)
template <int S, typename Tin = uint8_t>
__global__ __launch_bounds__(64) RM_NARROW_OCC void k_down_chain_u8(const Tin *frames, size_t frame_stride, DownGeom g, double *out)
{
    down_chain_narrow_body<S, Tin>(frames, frame_stride, g, out);
}

template <int S, typename Tin> __device__ __forceinline__ void down_chain_narrow_body(const Tin *frames, size_t frame_stride, const DownGeom &g, double *out)
{
    const int per_frame = g.strips * g.segs;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = (j / per_frame) * 8 + xcd;
    if (t >= g.T) return;
    const int inner = j % per_frame;
    const int seg = inner / g.strips, strip = inner - seg * g.strips;
    RegChain<S, Tin> rc(g);
    rc.prio_rank_ = dc_dispatch_quartile();
    if constexpr (RegTraits<Tin>::DMA) {
        HIP_DYNAMIC_SHARED(char, ring)
#if !defined(RM_HIPEMU)
        rc.ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char *)ring);
#endif
        rc.ring_ptr = ring + 16 * threadIdx.x;
    }
    if constexpr (RegChain<S, Tin>::BYTES && RM_U8_LUT) {
        __shared__ double s_lut[256];
        for (int i = threadIdx.x; i < 256; i += 64) s_lut[i] = (double)i * (1.0 / 255);   // uint8_to_float, transforms.py:20-23
        wave_sync();
        rc.lut = s_lut;
    }
    rc.run(frames + (size_t)t * frame_stride, out + (size_t)t * g.h[S] * g.w[S], strip, seg);
}

// geometry: strips of 960 exact columns; enough segments for ~8 waves per CU (target_waves over the chip)
inline bool make_down_geom_u8(int S, const int *h, const int *w, int T, DownGeom &g, bool tiny = false, int force_segs = 0, int target_waves = 2048)
{
    if (S < 1 || S > 4 || (w[0] % 16) != 0) return false;
    // every filtered level needs >= 3 rows (streaming vertical pass) and >= 3 columns: the in-lane border selects of
    // hfilter() realise BORDER_REFLECT_101 as -2 -> 2, -1 -> 1, w -> w-2, which is only what OpenCV does for w >= 3
    // (found by tools/fuzz_parity.py: 16-pixel-wide frames with 4 levels reach a 2-column level)
    for (int k = 0; k < S; ++k) if (h[k] < 3 || w[k] < 3) return false;
    g.S = S; g.T = T; g.vec = 1; g.y_begin = 0; g.y_end = h[S]; g.seg_split = 0; g.wpg = 1; g.prio = 0; g.prio_rank = 0; g.prio_shift = 4;
    for (int k = 0; k <= S; ++k) { g.h[k] = h[k]; g.w[k] = w[k]; }
    g.strips = (w[0] + U8_STRIP_PX - 1) / U8_STRIP_PX;
    const int rows = h[S];
    const int halo0 = (1 << (S + 1)) - 2;
    const long long per_seg = (long long)T * g.strips;
    // (target_waves by dtype and depth: rm_down_narrow.hip / rm_down_bgr.hip; the uint8 kernel, bound by its arithmetic, loses 20 % below four
    //  segments per 1080p frame)
    int segs = (int)((target_waves + per_seg / 2) / per_seg);
    if (force_segs > 0) segs = force_segs;   // (rm_debug_set "dc_segs")
    if (segs < 1) segs = 1;
    if (segs > rows) segs = rows;
    while (segs > 1 && ((((rows + segs - 1) / segs) << S) < 8 * halo0)) --segs;
    g.seg_h = (rows + segs - 1) / segs;
    if (tiny) g.seg_h = rows < 2 ? rows : 2;
    g.segs = (rows + g.seg_h - 1) / g.seg_h;
    return true;
}

}  // namespace rm
