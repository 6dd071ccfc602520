// respmon_amd/csrc/rm_bounds_l1.h -- tile bounds of skip_levels_at_top = 2 taken ONE LEVEL FURTHER DOWN (round 6)
//
//   raw[t] = pyrUp(pyrUp(C_2[t]))                  (pyramid.py:51-57 below `skip`; transforms.py:150-160 leaves levels 0, 1 zero)
//
// The selection (k_select_pairs) and the store-less sum (rm_tile_eval.h k_dense_sum_t) prune a (tile, frame) pair when a LOWER bound
// of its full-resolution values clears `top`.  k_frame_bounds / k_frame_bounds_rows take that bound from the pair's level-2
// footprint -- the band-passed level itself, speckled at the scale of one pixel -- and on the streams that matter it prunes
// nothing: 4K x 512 keeps 98 % of the pairs, 720p x 128 99.8 %, although 22 % / 45 % of them hold a value below `top` at all.
// One pyrUp step smooths the speckle away: the extrema of the LEVEL-1 footprint (rows 8 ty - 1 .. 8 ty + 8 by columns
// 32 tx - 1 .. 32 tx + 32: every level-0 value of the tile is a convex combination of it, pyrUp's weights are positive and sum to
// one) keep 27 % / 65 %.  k_dense_sum_t found that out pair by pair, inside its frame loop, at ~145 instructions a visit with a
// third of the lanes idle and a 2x halo; here the level-1 values of a whole frame are formed ONCE, streaming, with every lane busy:
//
//   * a wave owns three tile columns and a band of tile rows and marches down the level-2 rows the band's footprints touch.  Lane l
//     stands for level-2 column j = j0 + l (lanes 0 .. 49; lane 63 = j0 - 1) and for the level-1 column PAIR (2 j - 1, 2 j): a tile's
//     34 footprint columns 32 tx - 1 .. 32 tx + 32 are then exactly the pairs of 17 neighbouring lanes, so ONE running minimum and
//     maximum per lane serve both columns and both tiles a pair may belong to.  ONE 8-byte load per lane and row (four rows in
//     flight), the horizontal neighbours by DPP wave rotation, the horizontal values of rows i - 1, i, i + 1 in registers, level-1
//     rows 2 i and 2 i + 1 out of them -- te_step()'s expressions exactly, so lo / hi are the extrema of the very values the
//     evaluation forms; v_min_f64 / v_max_f64 (a NaN is dropped, as the comparison folds of the other bounds kernels drop it);
//   * at every eighth level-1 row the three tiles' extrema are folded over their 17 lanes with DPP row shifts and written --
//     [unique frame][tile], as the other bounds kernels write them;
//   * the extrema of the bounds go to the striped state as in k_frame_bounds_rows, and two lattice samples per wave (rm_kernels.h
//     lattice_sample: true raw values) taken where the wave met its lowest / highest C_2.
// No LDS, no barrier.  C_2 is read once.
#pragma once

namespace rm {

// can the level-1 bounds kernel take this geometry?  (two pyrUp steps, at least two rows and columns at levels 1 and 2: the virtual
// borders of te_step())
inline bool bounds_l1_ok(const ChainGeom &g)
{
    return g.S == 2 && g.h[1] >= 2 && g.w[1] >= 2 && g.h[2] >= 2 && g.w[2] >= 2;
}

// wave rotation of a double: every lane has a source, so there is no old value to keep (no register copy in front of the DPP move)
template <int CTRL> __device__ __forceinline__ double bl1_rot(double v)
{
    const unsigned long long b = bits_of(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    double o;
    from_bits(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo, o);
    return o;
}

constexpr int BL1_TILES = 3;                 // tile columns per wave
constexpr int BL1_COLS = 16 * BL1_TILES;     // level-2 columns they own
constexpr int BL1_PF = 4;                    // level-2 rows in flight per lane (= rows per tile row: the ring is indexed statically)

RM_KERNEL __launch_bounds__(256) void k_frame_bounds_l1(const double *cS, ChainGeom g, int ntiles, double *lo, double *hi, CollapseState *st,
                                                         int *sel_cnt, int nchunks, int nbands, int trb)
{
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < ntiles; i += 256) sel_cnt[i] = 0;   // k_select_pairs counts into it
    const double inf = __builtin_huge_val();
    const int h2 = g.h[2], w2 = g.w[2], h1 = g.h[1], w1 = g.w[1], ntx = g.tiles_x, nty = g.tiles_y;
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = uniform((int)blockIdx.y * 4 + wave);
    if (wid >= nchunks * nbands) return;   // (whole waves; nothing below synchronises across waves)
    const int c = wid % nchunks, b = wid / nchunks;
    const int ty_first = b * trb, ty_last = min(ty_first + trb, nty) - 1;
    const int j0 = BL1_COLS * c;
    // lane -> level-2 column: 0 .. 49 -> j0 + lane, 63 -> j0 - 1 (the others load nothing anyone reads)
    const int jv = lane == 63 ? j0 - 1 : j0 + lane;
    const int ja = min(max(jv, 0), w2 - 1);
    // this lane's level-1 column pair: 2 jv (even: taps jv - 1, jv, jv + 1) and 2 jv - 1 (odd: taps jv - 1, jv).  Lanes 0 .. 48 count.
    const bool ok_e = lane <= BL1_COLS && jv <= w2 - 1;
    const bool ok_o = lane <= BL1_COLS && jv >= 1 && 2 * jv - 1 <= w1 - 1;   // (jv == w2 when w1 is even: the last column, 8 s[w2 - 1] -- the clamped load)
    const bool ok_any = ok_e || ok_o;
    const bool edge = uniform((int)(__ballot(lane <= BL1_COLS && ok_e != ok_o) != 0ull)) != 0;   // (uniform) a lane of this wave owns only one column of its pair
    // make_htap()'s shapes with the three neighbours as operands (rm_dense_sum.h k_dense_sum_w): (L wa + C wb) + R wc; the clamped
    // loads supply C for the missing neighbour where its weight is not zero anyway
    const bool left = jv <= 0, right = jv >= w2 - 1;
    const double wa = left ? 0.0 : 1.0, wb = right ? 7.0 : 6.0, wc = left ? 2.0 : (right ? 0.0 : 1.0);
    const double *p = cS + (size_t)u * h2 * w2 + ja;
    auto row_ptr = [&](int i) __attribute__((always_inline)) { return p + (size_t)min(max(i, 0), h2 - 1) * w2; };
    // extrema of C_2 this lane met, and the tile row whose close saw each of them move last (the rows are looked up once, at the end,
    // among the nine level-2 rows that close covers: lattice samples)
    double t_mn = inf, t_mx = -inf, seen_mn = inf, seen_mx = -inf;
    int ty_mn = 0, ty_mx = 0;
    // extrema over the pairs this wave writes (lanes 15, 31, 47 only)
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    // (EDGE: a lane of this wave owns only one column of its pair -- the image's left / right edge; the other waves run without the selects)
    const int i_first = 4 * ty_first;
    auto march = [&](auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        // horizontal values of one level-2 row at this lane's column pair
        auto hvals = [&](double s, double &he, double &ho) __attribute__((always_inline)) {
            const double L = bl1_rot<0x13C>(s);           // wave_ror:1 -- the value of lane - 1 (lane 0: lane 63 = column j0 - 1)
            const double R = bl1_rot<0x134>(s);           // wave_rol:1 -- the value of lane + 1
            he = dw_tap3(L, s, R, wa, wb, wc);
            ho = __builtin_fma(s, 4.0, L * 4.0);           // both products exact
            if (EDGE) { ho = ok_o ? ho : he; he = ok_e ? he : ho; }   // (a pair with one column: that column twice)
            t_mn = f64_min(t_mn, s); t_mx = f64_max(t_mx, s);
        };
        // running extrema of the tile row being collected; the level-1 row in front of it (row 8 ty - 1 opens tile row ty)
        double amn = inf, amx = -inf, last_mn = inf, last_mx = -inf;
        auto take = [&](double ve, double vo) __attribute__((always_inline)) {
            amn = f64_min(amn, f64_min(ve, vo)); amx = f64_max(amx, f64_max(ve, vo));
        };
        const bool writer = lane < BL1_COLS && (lane & 15) == 15 && BL1_TILES * c + (lane >> 4) < ntx;
        auto finalize = [&](int ty) __attribute__((always_inline)) {
            // (rows consumed since the previous close: 4 ty + 2 .. 4 ty + 5; the band's first close starts at 4 ty - 1)
            if (t_mn < seen_mn) { seen_mn = t_mn; ty_mn = ty; }
            if (t_mx > seen_mx) { seen_mx = t_mx; ty_mx = ty; }
            double mn = ok_any ? amn : inf, mx = ok_any ? amx : -inf;
            // the pair of the lane to the right closes a tile's footprint (and opens the next tile's)
            const double r_mn = bl1_rot<0x134>(mn), r_mx = bl1_rot<0x134>(mx);
            if ((lane & 15) == 15) { mn = f64_min(mn, r_mn); mx = f64_max(mx, r_mx); }
            // fold each 16-lane row into its last lane (a lane without a source keeps its own value)
#define RM_BL1_FOLD(CTRL)                                                             \
            {                                                                            \
                const double a_ = dpp_get<CTRL, 0xF>(mn), b_ = dpp_get<CTRL, 0xF>(mx);   \
                mn = f64_min(mn, a_); mx = f64_max(mx, b_);                \
            }
            RM_BL1_FOLD(0x111) RM_BL1_FOLD(0x112) RM_BL1_FOLD(0x114) RM_BL1_FOLD(0x118)
#undef RM_BL1_FOLD
            if (writer) {
                const size_t o = (size_t)u * ntiles + (size_t)ty * ntx + BL1_TILES * c + (lane >> 4);
                lo[o] = mn; hi[o] = mx;
                lo_mn = f64_min(lo_mn, mn); lo_mx = f64_max(lo_mx, mn);
                hi_mn = f64_min(hi_mn, mx); hi_mx = f64_max(hi_mx, mx);
            }
        };
        // ---- prologue: horizontal values of rows 4 ty_first - 1 (hp) and 4 ty_first (hc); the four rows behind them requested
        double hp_e, hp_o, hc_e, hc_o;
        double q[BL1_PF];
        {
            const double s_p = *row_ptr(i_first - 1), s_c = *row_ptr(i_first);
#pragma unroll
            for (int k = 0; k < BL1_PF; ++k) q[k] = *row_ptr(i_first + 1 + k);
            hvals(s_p, hp_e, hp_o);
            hvals(s_c, hc_e, hc_o);
        }
        bool have_last = false;
        if (ty_first > 0) {   // level-1 row 8 ty_first - 1 (odd row of level-2 row i_first - 1)
            const double ve = (hp_e + hc_e) * (1.0 / 16), vo = (hp_o + hc_o) * (1.0 / 16);
            last_mn = f64_min(ve, vo); last_mx = f64_max(ve, vo);
            have_last = true;
        }
        // ---- tile rows: level-2 rows i = 4 ty + k give level-1 rows 8 ty + 2 k (even) and 8 ty + 2 k + 1 (odd).  FAST: a tile row
        // inside the band and the image -- every row exists, tile row ty - 1 is waiting for its last footprint row: straight-line code
        auto tile_row = [&](int ty, auto fast_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int k = 0; k < BL1_PF; ++k) {
                const int i = 4 * ty + k;                       // (uniform)
                if (!FAST && i > h2 - 1) break;                 // the image ends
                if (!FAST && ty > ty_last && k > 0) break;      // (below the band only level-1 row 8 (ty_last + 1) counts)
                double hn_e = hc_e, hn_o = hc_o;                // bottom: the last row again (up_at()'s r2)
                if (FAST || i < h2 - 1) hvals(q[k], hn_e, hn_o);
                q[k] = *row_ptr(i + 1 + BL1_PF);
                if (!FAST && i == 0) { hp_e = hn_e; hp_o = hn_o; }   // top: row -1 := row 1
                {   // level-1 row 2 i
                    const double ve = (hp_e + hc_e * 6 + hn_e) * (1.0 / 64), vo = (hp_o + hc_o * 6 + hn_o) * (1.0 / 64);
                    if (k == 0) {
                        // row 8 ty: the row below tile row ty - 1 -- its footprint is complete --, and the second row of tile row ty's
                        if (FAST || ty > ty_first) { take(ve, vo); finalize(ty - 1); }
                        amn = f64_min(ve, vo); amx = f64_max(ve, vo);
                        if (FAST || have_last) { amn = f64_min(amn, last_mn); amx = f64_max(amx, last_mx); }
                    } else take(ve, vo);
                }
                if (FAST || (ty <= ty_last && 2 * i + 1 <= h1 - 1)) {   // level-1 row 2 i + 1
                    const double ve = (hc_e + hn_e) * (1.0 / 16), vo = (hc_o + hn_o) * (1.0 / 16);
                    take(ve, vo);
                    if (k == BL1_PF - 1) { last_mn = f64_min(ve, vo); last_mx = f64_max(ve, vo); have_last = true; }
                }
                hp_e = hc_e; hp_o = hc_o; hc_e = hn_e; hc_o = hn_o;
            }
        };
        for (int ty = ty_first; ty <= ty_last + 1; ++ty) {
            if (ty > ty_first && ty <= ty_last && 4 * ty + 4 <= h2 - 1) tile_row(ty, std::true_type{});
            else tile_row(ty, std::false_type{});
            if (4 * ty > h2 - 1) break;
        }
        // the image ended inside the last tile row's footprint (no level-1 row 8 (ty_last + 1)): that tile row is still open
        if (8 * (ty_last + 1) > h1 - 1) finalize(ty_last);
    };
    if (edge) march(std::true_type{}); else march(std::false_type{});
    // lattice samples (true raw values) where this wave met its lowest / highest C_2: the lane that holds the extreme value, then
    // the row of it in that lane's column (every lane looks at one row)
    double sm_mn = inf, sm_mx = -inf;
    if (h2 >= 3 && w2 >= 3) {
        const bool cnt = lane < BL1_COLS && jv <= w2 - 1;
        const double w_mn = wave_min(cnt ? t_mn : inf), w_mx = wave_max(cnt ? t_mx : -inf);
        const double *f = cS + (size_t)u * h2 * w2;
        const int ia = max(i_first - 1, 0), ib = min(4 * ty_last + 5, h2 - 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double target = k ? w_mx : w_mn;
            const unsigned long long who = __ballot(cnt && (k ? t_mx : t_mn) == target);
            if (who == 0ull) continue;                                        // (uniform; a NaN extreme)
            const int src = (int)__builtin_ctzll(who), x = j0 + src;
            const int tyw = __builtin_amdgcn_readlane(k ? ty_mx : ty_mn, src);
            const int ra = max(4 * tyw - 3, ia), rb = min(4 * tyw + 5, ib), r = ra + lane;   // (nine rows, one per lane)
            const unsigned long long hit = __ballot(r <= rb && f[(size_t)min(r, rb) * w2 + x] == target);
            if (hit == 0ull) continue;
            const int y = ra + (int)__builtin_ctzll(hit);
            const int ys = min(max(y, 1), h2 - 2), xs = min(max(x, 1), w2 - 2);
            const double *r1 = f + (size_t)ys * w2;
            const double v = lattice_sample(r1 - w2, r1, r1 + w2, xs, g.lat_a, g.lat_b);
            sm_mn = f64_min(sm_mn, v); sm_mx = f64_max(sm_mx, v);
        }
    }
    lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
    if (lane == 0 && lo_mn <= lo_mx) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = (u + wid * 7) & (NSTRIPE - 1);
        // (the six current values requested together, then the comparisons: a load-compare-atomic at a time was six round trips in a row
        //  at the end of every wave's life)
        const unsigned long long c_lb_mx = *(volatile unsigned long long *)&st->lb_max_keys[sp], c_lb_mn = *(volatile unsigned long long *)&st->lb_min_keys[sp];
        const unsigned long long c_ub_mx = *(volatile unsigned long long *)&st->ub_max_keys[sp], c_ub_mn = *(volatile unsigned long long *)&st->ub_min_keys[sp];
        const unsigned long long c_s_mn = *(volatile unsigned long long *)&st->smp_min_keys[sp], c_s_mx = *(volatile unsigned long long *)&st->smp_max_keys[sp];
        if (k_lo_mx > c_lb_mx) atomicMax(&st->lb_max_keys[sp], k_lo_mx);
        if (k_lo_mn < c_lb_mn) atomicMin(&st->lb_min_keys[sp], k_lo_mn);
        if (k_hi_mx > c_ub_mx) atomicMax(&st->ub_max_keys[sp], k_hi_mx);
        if (k_hi_mn < c_ub_mn) atomicMin(&st->ub_min_keys[sp], k_hi_mn);
        if (sm_mn <= sm_mx) {
            const unsigned long long k_mn = f64_key(sm_mn), k_mx = f64_key(sm_mx);
            if (k_mn < c_s_mn) atomicMin(&st->smp_min_keys[sp], k_mn);
            if (k_mx > c_s_mx) atomicMax(&st->smp_max_keys[sp], k_mx);
        }
    }
}

// ---- the same bounds in single precision, with a margin (round 6) -------------------------------------------------------------------
// A bound only has to be SOUND: lo <= every level-1 value of the footprint <= hi.  k_frame_bounds_l1 forms the level-1 values in the
// chain's own float64 operations (lo / hi are then the exact extrema) at ~60 instructions per level-2 row and lane -- the kernel is
// bound by instruction issue, not by the 1 GB it reads.  Here the values are formed in float32 -- both columns of a lane's pair in ONE
// packed instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), extrema with v_min3_f32 / v_max3_f32, 32-bit DPP moves -- and
// the bounds are widened by what float32 can have lost:
//     every level-1 value is a positive combination (weights summing to one) of the 3 x 3 level-2 values around it; rounding the
//     inputs to float32 and each of the <= 6 operations of a path (products with 1, 2, 4 and 1/16, 1/64 are exact) moves it by less
//     than 7 u M, u = 2^-24, M = the largest |C_2| of the support (standard running error bound of a sum of non-negative terms:
//     every intermediate is bounded by M); eps = 2^-20 M + 2^-140 (16 u M: twice the bound and more; the constant covers float32
//     underflow) with M = the largest |C_2| the lanes of the tile AND their neighbours have met so far in the band.
//     lo = float(min) - eps, hi = float(max) + eps, formed in float64 (both conversions are exact).
// The selection's own margin (PRUNE_REL_MARGIN) comes on top as before.  ~1e-6 of the value range: no measurable loss of pruning.
typedef RM_VEC(float, 2) bl1_v2f;

__device__ __forceinline__ bl1_v2f bl1_fma2(bl1_v2f a, bl1_v2f b, bl1_v2f c)
{
#ifndef RM_HIPEMU
    return __builtin_elementwise_fma(a, b, c);
#else
    bl1_v2f r;
    r[0] = __builtin_fmaf(a[0], b[0], c[0]); r[1] = __builtin_fmaf(a[1], b[1], c[1]);
    return r;
#endif
}
__device__ __forceinline__ float bl1_min3(float a, float b, float c)
{
#ifndef RM_HIPEMU
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return __builtin_fminf(a, __builtin_fminf(b, c));
#endif
}
__device__ __forceinline__ float bl1_max3(float a, float b, float c)
{
#ifndef RM_HIPEMU
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return __builtin_fmaxf(a, __builtin_fmaxf(b, c));
#endif
}
// min / max of two floats as ONE instruction (f64_min's reason: no canonicalising v_max_f32 x, x in front of the operands)
__device__ __forceinline__ float bl1_minf(float a, float b)
{
#ifndef RM_HIPEMU
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fminf(a, b);
#endif
}
__device__ __forceinline__ float bl1_maxf(float a, float b)
{
#ifndef RM_HIPEMU
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fmaxf(a, b);
#endif
}
// v = min / max(v, v of the lane N to the left inside its 16-lane row) as ONE instruction: the DPP operand of v_min_f32 / v_max_f32
// itself; a lane without a source keeps its value (the instruction does not write it).  (s_nop 1: the two wait states between
// a VALU write of a register and a DPP read of it, which the compiler cannot see inside the asm)
#ifndef RM_HIPEMU
#define RM_BL1_FOLD_MIN(v, N) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
#define RM_BL1_FOLD_MAX(v, N) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
#else
#define RM_BL1_FOLD_MIN(v, N) v = __builtin_fminf(v, bl1_shrf<0x110 + N>(v))
#define RM_BL1_FOLD_MAX(v, N) v = __builtin_fmaxf(v, bl1_shrf<0x110 + N>(v))
#endif
template <int CTRL> __device__ __forceinline__ float bl1_shrf(float v);
template <int CTRL> __device__ __forceinline__ float bl1_rotf(float v)   // wave rotation: every lane has a source
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ float bl1_shrf(float v)   // row shift: a lane without a source keeps its own value
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}

RM_KERNEL __launch_bounds__(256) void k_frame_bounds_l1f(const double *cS, ChainGeom g, int ntiles, double *lo, double *hi, CollapseState *st,
                                                          int *sel_cnt, int nchunks, int nbands, int trb)
{
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < ntiles; i += 256) sel_cnt[i] = 0;   // k_select_pairs counts into it
    const double inf = __builtin_huge_val();
    const float finf = __builtin_huge_valf();
    const int h2 = g.h[2], w2 = g.w[2], h1 = g.h[1], w1 = g.w[1], ntx = g.tiles_x, nty = g.tiles_y;
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = uniform((int)blockIdx.y * 4 + wave);
    if (wid >= nchunks * nbands) return;   // (whole waves; nothing below synchronises across waves)
    const int c = wid % nchunks, b = wid / nchunks;
    const int ty_first = b * trb, ty_last = min(ty_first + trb, nty) - 1;
    const int j0 = BL1_COLS * c;
    const int jv = lane == 63 ? j0 - 1 : j0 + lane;          // (k_frame_bounds_l1's lane -> column map and column pairs)
    const int ja = min(max(jv, 0), w2 - 1);
    const bool ok_e = lane <= BL1_COLS && jv <= w2 - 1;
    const bool ok_o = lane <= BL1_COLS && jv >= 1 && 2 * jv - 1 <= w1 - 1;
    const bool ok_any = ok_e || ok_o;
    const bool edge = uniform((int)(__ballot(lane <= BL1_COLS && ok_e != ok_o) != 0ull)) != 0;
    const bool left = jv <= 0, right = jv >= w2 - 1;
    // the pair (even column, odd column): (L wa + C wb + R wc, L 4 + C 4)
    bl1_v2f wL, wC, wR;
    wL[0] = left ? 0.0f : 1.0f; wL[1] = 4.0f;
    wC[0] = right ? 7.0f : 6.0f; wC[1] = 4.0f;
    wR[0] = left ? 2.0f : (right ? 0.0f : 1.0f); wR[1] = 0.0f;
    const double *p = cS + (size_t)u * h2 * w2 + ja;
    auto row_ptr = [&](int i) __attribute__((always_inline)) { return p + (size_t)min(max(i, 0), h2 - 1) * w2; };
    float t_mn = finf, t_mx = -finf, seen_mn = finf, seen_mx = -finf;     // extrema of float(C_2) this lane met (lattice samples)
    int ty_mn = 0, ty_mx = 0;                                              // ... and the tile row whose close saw them move last
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    const int i_first = 4 * ty_first;
    auto march = [&](auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        auto hvals = [&](double sd) __attribute__((always_inline)) -> bl1_v2f {
            const float s = (float)sd;
            const float L = bl1_rotf<0x13C>(s), R = bl1_rotf<0x134>(s);
            bl1_v2f vL, vC, vR;
            vL[0] = L; vL[1] = L; vC[0] = s; vC[1] = s; vR[0] = R; vR[1] = R;
            bl1_v2f h = bl1_fma2(vR, wR, bl1_fma2(vL, wL, vC * wC));
            if (EDGE) { h[1] = ok_o ? h[1] : h[0]; h[0] = ok_e ? h[0] : h[1]; }
            t_mn = bl1_minf(t_mn, s); t_mx = bl1_maxf(t_mx, s);
            return h;
        };
        float amn = finf, amx = -finf, last_mn = finf, last_mx = -finf;
        auto take = [&](bl1_v2f v) __attribute__((always_inline)) { amn = bl1_min3(amn, v[0], v[1]); amx = bl1_max3(amx, v[0], v[1]); };
        const bool writer = lane < BL1_COLS && (lane & 15) == 15 && BL1_TILES * c + (lane >> 4) < ntx;
        auto finalize = [&](int ty) __attribute__((always_inline)) {
            if (t_mn < seen_mn) { seen_mn = t_mn; ty_mn = ty; }
            if (t_mx > seen_mx) { seen_mx = t_mx; ty_mx = ty; }
            // M of the margin: the largest |C_2| this lane has met in the band so far (its running extrema say it) -- and its two
            // neighbours', whose values its taps read: the fold below then covers every column the tile's level-1 values draw on
            float m = bl1_maxf(__builtin_fabsf(t_mn), __builtin_fabsf(t_mx));
            m = bl1_max3(m, bl1_rotf<0x13C>(m), bl1_rotf<0x134>(m));
            float mn = ok_any ? amn : finf, mx = ok_any ? amx : -finf;
            const float r_mn = bl1_rotf<0x134>(mn), r_mx = bl1_rotf<0x134>(mx), r_m = bl1_rotf<0x134>(m);
            if ((lane & 15) == 15) { mn = bl1_minf(mn, r_mn); mx = bl1_maxf(mx, r_mx); m = bl1_maxf(m, r_m); }
            RM_BL1_FOLD_MIN(mn, 1); RM_BL1_FOLD_MAX(mx, 1); RM_BL1_FOLD_MAX(m, 1);
            RM_BL1_FOLD_MIN(mn, 2); RM_BL1_FOLD_MAX(mx, 2); RM_BL1_FOLD_MAX(m, 2);
            RM_BL1_FOLD_MIN(mn, 4); RM_BL1_FOLD_MAX(mx, 4); RM_BL1_FOLD_MAX(m, 4);
            RM_BL1_FOLD_MIN(mn, 8); RM_BL1_FOLD_MAX(mx, 8); RM_BL1_FOLD_MAX(m, 8);
            if (writer) {
                const double eps = (double)m * (1.0 / 1048576.0) + 0x1p-140;
                const double dlo = (double)mn - eps, dhi = (double)mx + eps;
                const size_t o = (size_t)u * ntiles + (size_t)ty * ntx + BL1_TILES * c + (lane >> 4);
                lo[o] = dlo; hi[o] = dhi;
                lo_mn = f64_min(lo_mn, dlo); lo_mx = f64_max(lo_mx, dlo);
                hi_mn = f64_min(hi_mn, dhi); hi_mx = f64_max(hi_mx, dhi);
            }
        };
        bl1_v2f hp, hc;
        double q[BL1_PF];
        {
            const double s_p = *row_ptr(i_first - 1), s_c = *row_ptr(i_first);
#pragma unroll
            for (int k = 0; k < BL1_PF; ++k) q[k] = *row_ptr(i_first + 1 + k);
            hp = hvals(s_p);
            hc = hvals(s_c);
        }
        bool have_last = false;
        if (ty_first > 0) {
            const bl1_v2f v = (hp + hc) * (1.0f / 16);
            last_mn = bl1_minf(v[0], v[1]); last_mx = bl1_maxf(v[0], v[1]);
            have_last = true;
        }
        auto tile_row = [&](int ty, auto fast_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int k = 0; k < BL1_PF; ++k) {
                const int i = 4 * ty + k;                       // (uniform)
                if (!FAST && i > h2 - 1) break;
                if (!FAST && ty > ty_last && k > 0) break;
                bl1_v2f hn = hc;
                if (FAST || i < h2 - 1) hn = hvals(q[k]);
                q[k] = *row_ptr(i + 1 + BL1_PF);
                if (!FAST && i == 0) hp = hn;
                {   // level-1 row 2 i
                    bl1_v2f six; six[0] = 6.0f; six[1] = 6.0f;
                    const bl1_v2f v = (bl1_fma2(hc, six, hp) + hn) * (1.0f / 64);
                    if (k == 0) {
                        if (FAST || ty > ty_first) { take(v); finalize(ty - 1); }
                        amn = bl1_minf(v[0], v[1]); amx = bl1_maxf(v[0], v[1]);
                        if (FAST || have_last) { amn = bl1_minf(amn, last_mn); amx = bl1_maxf(amx, last_mx); }
                    } else take(v);
                }
                if (FAST || (ty <= ty_last && 2 * i + 1 <= h1 - 1)) {   // level-1 row 2 i + 1
                    const bl1_v2f v = (hc + hn) * (1.0f / 16);
                    take(v);
                    if (k == BL1_PF - 1) { last_mn = bl1_minf(v[0], v[1]); last_mx = bl1_maxf(v[0], v[1]); have_last = true; }
                }
                hp = hc; hc = hn;
            }
        };
        for (int ty = ty_first; ty <= ty_last + 1; ++ty) {
            if (ty > ty_first && ty <= ty_last && 4 * ty + 4 <= h2 - 1) tile_row(ty, std::true_type{});
            else tile_row(ty, std::false_type{});
            if (4 * ty > h2 - 1) break;
        }
        if (8 * (ty_last + 1) > h1 - 1) finalize(ty_last);
    };
    if (edge) march(std::true_type{}); else march(std::false_type{});
    // lattice samples where this wave met its lowest / highest float(C_2)  (k_frame_bounds_l1)
    double sm_mn = inf, sm_mx = -inf;
    if (h2 >= 3 && w2 >= 3) {
        const bool cnt = lane < BL1_COLS && jv <= w2 - 1;
        const float w_mn = (float)wave_min(cnt ? (double)t_mn : inf), w_mx = (float)wave_max(cnt ? (double)t_mx : -inf);
        const double *f = cS + (size_t)u * h2 * w2;
        const int ia = max(i_first - 1, 0), ib = min(4 * ty_last + 5, h2 - 1);
        // both look-ups side by side, without a branch between their loads (two round trips at the end of the wave's life instead of four):
        // an extreme nobody holds (NaN) looks at valid addresses and is dropped at the end
        const unsigned long long who0 = __ballot(cnt && t_mn == w_mn), who1 = __ballot(cnt && t_mx == w_mx);
        const int src0 = who0 ? (int)__builtin_ctzll(who0) : 0, src1 = who1 ? (int)__builtin_ctzll(who1) : 0;
        const int x0 = j0 + src0, x1 = j0 + src1;
        const int tyw0 = __builtin_amdgcn_readlane(ty_mn, src0), tyw1 = __builtin_amdgcn_readlane(ty_mx, src1);
        const int ra0 = min(max(4 * tyw0 - 3, ia), ib), rb0 = max(min(4 * tyw0 + 5, ib), ra0);
        const int ra1 = min(max(4 * tyw1 - 3, ia), ib), rb1 = max(min(4 * tyw1 + 5, ib), ra1);
        const float c0 = (float)f[(size_t)min(ra0 + lane, rb0) * w2 + x0], c1 = (float)f[(size_t)min(ra1 + lane, rb1) * w2 + x1];
        const unsigned long long hit0 = __ballot(ra0 + lane <= rb0 && c0 == w_mn), hit1 = __ballot(ra1 + lane <= rb1 && c1 == w_mx);
        const int y0 = ra0 + (hit0 ? (int)__builtin_ctzll(hit0) : 0), y1 = ra1 + (hit1 ? (int)__builtin_ctzll(hit1) : 0);
        const double *p0 = f + (size_t)min(max(y0, 1), h2 - 2) * w2, *p1 = f + (size_t)min(max(y1, 1), h2 - 2) * w2;
        const double v0 = lattice_sample(p0 - w2, p0, p0 + w2, min(max(x0, 1), w2 - 2), g.lat_a, g.lat_b);
        const double v1 = lattice_sample(p1 - w2, p1, p1 + w2, min(max(x1, 1), w2 - 2), g.lat_a, g.lat_b);
        if (who0 && hit0) { sm_mn = f64_min(sm_mn, v0); sm_mx = f64_max(sm_mx, v0); }
        if (who1 && hit1) { sm_mn = f64_min(sm_mn, v1); sm_mx = f64_max(sm_mx, v1); }
    }
    lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
    if (lane == 0 && lo_mn <= lo_mx) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = (u + wid * 7) & (NSTRIPE - 1);
        // (the six current values requested together, then the comparisons: a load-compare-atomic at a time was six round trips in a row
        //  at the end of every wave's life)
        const unsigned long long c_lb_mx = *(volatile unsigned long long *)&st->lb_max_keys[sp], c_lb_mn = *(volatile unsigned long long *)&st->lb_min_keys[sp];
        const unsigned long long c_ub_mx = *(volatile unsigned long long *)&st->ub_max_keys[sp], c_ub_mn = *(volatile unsigned long long *)&st->ub_min_keys[sp];
        const unsigned long long c_s_mn = *(volatile unsigned long long *)&st->smp_min_keys[sp], c_s_mx = *(volatile unsigned long long *)&st->smp_max_keys[sp];
        if (k_lo_mx > c_lb_mx) atomicMax(&st->lb_max_keys[sp], k_lo_mx);
        if (k_lo_mn < c_lb_mn) atomicMin(&st->lb_min_keys[sp], k_lo_mn);
        if (k_hi_mx > c_ub_mx) atomicMax(&st->ub_max_keys[sp], k_hi_mx);
        if (k_hi_mn < c_ub_mn) atomicMin(&st->ub_min_keys[sp], k_hi_mn);
        if (sm_mn <= sm_mx) {
            const unsigned long long k_mn = f64_key(sm_mn), k_mx = f64_key(sm_mx);
            if (k_mn < c_s_mn) atomicMin(&st->smp_min_keys[sp], k_mn);
            if (k_mx > c_s_mx) atomicMax(&st->smp_max_keys[sp], k_mx);
        }
    }
}

// ---- skip 3 / 4: the bounds of a DENSE stream refined one level down (round 6) -------------------------------------------------------------
// At skip >= 3 the tile bounds come from the level-S footprint inside the per-frame kernel that builds the small pyramid
// (k_small_filter_first): the 1080p headline stream keeps 1 % of its pairs with them and nothing more is needed.  Sensor noise in every
// pixel is another matter -- CPU study with the oracle, 1080p, skip 4 (tools/r06_l1_study.py noise): 8.3 % of the pairs hold a value
// below `top`; the level-4 footprint bound keeps 56.7 %, the level-3 bound 17.5 %, level 2 11.6 %, level 1 9.3 %.  One pyrUp step
// does most of it, as at skip 2.  This kernel overwrites lo / hi with the extrema of the level-(S - 1) footprint -- a thread per
// (tile, unique frame): the tile's level-S block (4 x 7 values at skip 4, 5 x 11 at skip 3) into registers, the level-(S - 1) values
// column by column (up_at()'s expressions: horizontal 3-tap with make_htap()'s weights, then the even / odd row forms), extrema on the
// fly; every index compile-time.  The level is small (8 MB at 1080p x 256): ~10 us -- too much for the headline path, so rm_locate asks
// for it only behind a call that kept many pairs (rm_collapse_eval.hip, ctx->refine_hint).  The extrema of the bounds in the state stay
// those of the level-S bounds (looser, still valid).
template <int S>
__global__ __launch_bounds__(256) void k_bounds_up1(const double *cS, ChainGeom g, int ntiles, double *lo, double *hi)
{
    static_assert(S == 3 || S == 4, "skip 3 / 4 (skip 2 has the streaming kernels above)");
    using F = TileFoot<S, false>;
    constexpr int K = S - 1;
    constexpr int NRD = F::nr(K), NCD = F::nc(K);   // the footprint at level S - 1: 5 x 11 (skip 4), 7 x 19 (skip 3)
    constexpr int NRS = F::nr(S), NCS = F::nc(S);   // the level-S block it is formed from: 4 x 7, 5 x 11
    static_assert(NCS >= (NCD - 1) / 2 + 2 && NRS >= (NRD - 1) / 2 + 2, "the block holds every tap");
    const int u = blockIdx.y;
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= ntiles) return;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const int hs = g.h[S], ws = g.w[S], hd = g.h[K], wd = g.w[K];
    const int Y = (16 * ty) >> S, X = (64 * tx) >> S;   // the block starts at virtual (Y - 1, X - 1), the footprint at (2 Y - 1, 2 X - 1)
    const double *src = cS + (size_t)u * hs * ws;
    double s[NRS][NCS];
#pragma unroll
    for (int r = 0; r < NRS; ++r) {
        const int yv = Y - 1 + r;
        const int ya = yv < 0 ? min(1, hs - 1) : min(yv, hs - 1);     // pyrUp's rows: -1 := 1, past the bottom the last one again (up_at()'s r0 / r2)
#pragma unroll
        for (int c = 0; c < NCS; ++c) s[r][c] = src[(size_t)ya * ws + min(max(X - 1 + c, 0), ws - 1)];
    }
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
#pragma unroll
    for (int c = 0; c < NCD; ++c) {
        const int xv = 2 * X - 1 + c;                 // level-(S - 1) column; odd for even c
        if (xv < 0 || xv > wd - 1) continue;
        // horizontal values of the block's rows at this column (make_htap()'s shapes; the clamped loads supply the repeated column)
        double h[NRS];
        if ((c & 1) == 0) {                           // odd column: (s[j] + s[j + 1]) * 4, j = X - 1 + c / 2
#pragma unroll
            for (int r = 0; r < NRS; ++r) h[r] = __builtin_fma(s[r][c / 2 + 1], 4.0, s[r][c / 2] * 4.0);
        } else {                                      // even column: taps j - 1, j, j + 1, j = X + (c - 1) / 2
            const int j = X + (c - 1) / 2;
            const bool left = j == 0, right = j == ws - 1;
            const double wa = left ? 0.0 : 1.0, wb = right ? 7.0 : 6.0, wc = left ? 2.0 : (right ? 0.0 : 1.0);
#pragma unroll
            for (int r = 0; r < NRS; ++r) h[r] = dw_tap3(s[r][(c - 1) / 2], s[r][(c - 1) / 2 + 1], s[r][(c - 1) / 2 + 2], wa, wb, wc);
        }
#pragma unroll
        for (int p = 0; p < NRD; ++p) {
            const int yv = 2 * Y - 1 + p;             // level-(S - 1) row; odd for even p
            if (yv < 0 || yv > hd - 1) continue;
            const double v = (p & 1) ? (h[(p - 1) / 2] + h[(p - 1) / 2 + 1] * 6 + h[(p - 1) / 2 + 2]) * (1.0 / 64)
                                     : (h[p / 2] + h[p / 2 + 1]) * (1.0 / 16);
            mn = (v < mn) ? v : mn; mx = (v > mx) ? v : mx;
        }
    }
    const size_t o = (size_t)u * ntiles + tile;
    lo[o] = mn; hi[o] = mx;
}

}  // namespace rm
