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
//   * a wave owns three tile columns (48 level-2 columns: lanes 0 .. 47; lane 48 = the column to their right, lane 63 = the one to
//     their left, lane 49 feeds lane 48's right tap) and a band of tile rows, and marches down the level-2 rows the band's
//     footprints touch: ONE 8-byte load per lane and row (up to four rows in flight), the horizontal neighbours by DPP wave
//     rotation, the horizontal values of rows i - 1, i, i + 1 in registers, level-1 rows 2 i and 2 i + 1 (two columns each) out of
//     them -- te_step()'s expressions exactly, so lo / hi are the extrema of the very values the evaluation forms;
//   * running extrema per lane and column parity; at every eighth level-1 row the three tiles' extrema are folded over their 16 lanes
//     (+ the edge column of either neighbour) with DPP row shifts and written -- [unique frame][tile], as the other bounds kernels;
//   * the extrema of the bounds and the lattice samples (rm_kernels.h lattice_sample) go to the striped state as in
//     k_frame_bounds_rows.
// No LDS, no barrier, ~45 VGPRs.  C_2 is read once.
#pragma once

namespace rm {

// can the level-1 bounds kernel take this geometry?  (two pyrUp steps, at least two rows and columns at levels 1 and 2: the virtual
// borders of te_step())
inline bool bounds_l1_ok(const ChainGeom &g)
{
    return g.S == 2 && g.h[1] >= 2 && g.w[1] >= 2 && g.h[2] >= 2 && g.w[2] >= 2;
}

constexpr int BL1_TILES = 3;                 // tile columns per wave
constexpr int BL1_COLS = 16 * BL1_TILES;     // level-2 columns they own
constexpr int BL1_PF = 4;                    // level-2 rows in flight per lane

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
    const bool own = lane < BL1_COLS;
    // this lane's two level-1 columns 2 jv (even) and 2 jv + 1 (odd): inside the image?
    const bool in2 = jv >= 0 && jv <= w2 - 1;
    const bool ok_e = in2 && (own || lane == BL1_COLS), ok_o = in2 && 2 * jv + 1 < w1 && (own || lane == 63);
    // make_htap()'s shapes with the three neighbours as operands (rm_dense_sum.h k_dense_sum_w): even column (L wa + C wb) + R wc,
    // odd column C 4 + R 4 with R := C at the right edge
    const bool left = jv <= 0, right = jv >= w2 - 1;
    const double wa = left ? 0.0 : 1.0, wb = right ? 7.0 : 6.0, wc = left ? 2.0 : (right ? 0.0 : 1.0);
    const double *p = cS + (size_t)u * h2 * w2 + ja;
    // level-1 rows whose extrema this wave needs, and the level-2 rows they are formed from
    const int r_lo = max(8 * ty_first - 1, 0), r_hi = min(8 * ty_last + 8, h1 - 1);
    const int i_lo = r_lo >> 1, i_hi = r_hi >> 1;
    const int ia = max(i_lo - 1, 0), ib = min(i_hi + 1, h2 - 1);
    double q[BL1_PF];
    int inext = ia;
#pragma unroll
    for (int k = 0; k < BL1_PF; ++k) q[k] = p[(size_t)min(ia + k, ib) * w2];
    // where this lane met its lowest / highest C_2 (rows only; the column is the lane's): the lattice samples are taken there
    double t_mn = inf, t_mx = -inf;
    int p_mn = -1, p_mx = -1;
    auto next_h = [&](double &he, double &ho) __attribute__((always_inline)) {
        const double s = q[0];
#pragma unroll
        for (int k = 0; k + 1 < BL1_PF; ++k) q[k] = q[k + 1];
        q[BL1_PF - 1] = p[(size_t)min(inext + BL1_PF, ib) * w2];
        const double L = dpp_get<0x13C, 0xF>(s);      // wave_ror:1 -- the value of lane - 1 (lane 0: lane 63 = column j0 - 1)
        const double R0 = dpp_get<0x134, 0xF>(s);     // wave_rol:1 -- the value of lane + 1
        const double R = right ? s : R0;
        he = dw_tap3(L, s, R, wa, wb, wc);
        ho = __builtin_fma(R, 4.0, s * 4.0);           // both products exact
        if (own && in2) {
            if (s < t_mn) { t_mn = s; p_mn = inext; }
            if (s > t_mx) { t_mx = s; p_mx = inext; }
        }
        ++inext;
    };
    double hp_e = 0.0, hp_o = 0.0, hc_e, hc_o, hn_e, hn_o;
    if (i_lo > 0) next_h(hp_e, hp_o);
    next_h(hc_e, hc_o);
    // running extrema of the tile row being collected, per column parity; the last level-1 row seen (row 8 ty - 1 opens tile row ty)
    double amn_e = inf, amx_e = -inf, amn_o = inf, amx_o = -inf;
    double last_e = 0.0, last_o = 0.0;
    int cur_ty = ty_first;
    // extrema over the pairs this wave writes (lanes 15, 31, 47 only)
    double lo_mn = inf, lo_mx = -inf, hi_mn = inf, hi_mx = -inf;
    const bool writer = own && (lane & 15) == 15 && BL1_TILES * c + (lane >> 4) < ntx;
    auto finalize = [&](int ty) __attribute__((always_inline)) {
        double mn = inf, mx = -inf;
        if (own && ok_e) { mn = amn_e; mx = amx_e; }
        if (own && ok_o) { mn = (amn_o < mn) ? amn_o : mn; mx = (amx_o > mx) ? amx_o : mx; }
        // the even column of the lane to the right closes a tile's footprint, the odd column of the lane to the left opens it
        const double r_mn = dpp_get<0x134, 0xF>(ok_e ? amn_e : inf), r_mx = dpp_get<0x134, 0xF>(ok_e ? amx_e : -inf);
        const double l_mn = dpp_get<0x13C, 0xF>(ok_o ? amn_o : inf), l_mx = dpp_get<0x13C, 0xF>(ok_o ? amx_o : -inf);
        if (own && (lane & 15) == 15) { mn = (r_mn < mn) ? r_mn : mn; mx = (r_mx > mx) ? r_mx : mx; }
        if (own && (lane & 15) == 0) { mn = (l_mn < mn) ? l_mn : mn; mx = (l_mx > mx) ? l_mx : mx; }
        // fold each 16-lane row into its last lane (a lane without a source keeps its own value)
#define RM_BL1_FOLD(CTRL)                                                             \
        {                                                                            \
            const double a_ = dpp_get<CTRL, 0xF>(mn), b_ = dpp_get<CTRL, 0xF>(mx);   \
            mn = (a_ < mn) ? a_ : mn; mx = (b_ > mx) ? b_ : mx;                      \
        }
        RM_BL1_FOLD(0x111) RM_BL1_FOLD(0x112) RM_BL1_FOLD(0x114) RM_BL1_FOLD(0x118)
#undef RM_BL1_FOLD
        if (writer) {
            const size_t o = (size_t)u * ntiles + (size_t)ty * ntx + BL1_TILES * c + (lane >> 4);
            lo[o] = mn; hi[o] = mx;
            lo_mn = (mn < lo_mn) ? mn : lo_mn; lo_mx = (mn > lo_mx) ? mn : lo_mx;
            hi_mn = (mx < hi_mn) ? mx : hi_mn; hi_mx = (mx > hi_mx) ? mx : hi_mx;
        }
    };
    // level-1 row r (values ve / vo of this lane's two columns); returns false once the band's last tile row is written
    auto emit = [&](int r, double ve, double vo) __attribute__((always_inline)) -> bool {
        amn_e = (ve < amn_e) ? ve : amn_e; amx_e = (ve > amx_e) ? ve : amx_e;
        amn_o = (vo < amn_o) ? vo : amn_o; amx_o = (vo > amx_o) ? vo : amx_o;
        if ((r >> 3) > cur_ty) {   // (uniform) r = 8 (cur_ty + 1): the row below tile row cur_ty -- its footprint is complete
            finalize(cur_ty);
            ++cur_ty;
            if (cur_ty > ty_last) return false;
            // tile row cur_ty opens with rows r - 1 and r
            amn_e = (last_e < ve) ? last_e : ve; amx_e = (last_e > ve) ? last_e : ve;
            amn_o = (last_o < vo) ? last_o : vo; amx_o = (last_o > vo) ? last_o : vo;
        }
        last_e = ve; last_o = vo;
        return true;
    };
    bool open = true;
    for (int i = i_lo; i <= i_hi && open; ++i) {
        if (i < h2 - 1) next_h(hn_e, hn_o); else { hn_e = hc_e; hn_o = hc_o; }   // (uniform) bottom: the last row again (up_at()'s r2)
        if (i == 0) { hp_e = hn_e; hp_o = hn_o; }                                // (uniform) top: row -1 := row 1
        if (2 * i >= r_lo) open = emit(2 * i, (hp_e + hc_e * 6 + hn_e) * (1.0 / 64), (hp_o + hc_o * 6 + hn_o) * (1.0 / 64));
        if (open && 2 * i + 1 <= r_hi) open = emit(2 * i + 1, (hc_e + hn_e) * (1.0 / 16), (hc_o + hn_o) * (1.0 / 16));
        hp_e = hc_e; hp_o = hc_o; hc_e = hn_e; hc_o = hn_o;
    }
    if (open && cur_ty <= ty_last) finalize(cur_ty);   // the image ends inside the last tile row's footprint
    // lattice samples (true raw values) at the interior pixels nearest to where this lane met its extreme C_2
    double sm_mn = inf, sm_mx = -inf;
    if (own && in2 && h2 >= 3 && w2 >= 3) {
        const double *f = cS + (size_t)u * h2 * w2;
        const int cand[2] = {p_mn, p_mx};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (cand[k] < 0) continue;
            const int y = min(max(cand[k], 1), h2 - 2), x = min(max(jv, 1), w2 - 2);
            const double *r1 = f + (size_t)y * w2;
            const double v = lattice_sample(r1 - w2, r1, r1 + w2, x, g.lat_a, g.lat_b);
            sm_mn = (v < sm_mn) ? v : sm_mn; sm_mx = (v > sm_mx) ? v : sm_mx;
        }
    }
    lo_mn = wave_min(lo_mn); lo_mx = wave_max(lo_mx); hi_mn = wave_min(hi_mn); hi_mx = wave_max(hi_mx);
    sm_mn = wave_min(sm_mn); sm_mx = wave_max(sm_mx);
    if (lane == 0 && lo_mn <= lo_mx) {
        const unsigned long long k_lo_mx = f64_key(lo_mx), k_lo_mn = f64_key(lo_mn), k_hi_mx = f64_key(hi_mx), k_hi_mn = f64_key(hi_mn);
        const int sp = (u + wid * 7) & (NSTRIPE - 1);
        if (k_lo_mx > *(volatile unsigned long long *)&st->lb_max_keys[sp]) atomicMax(&st->lb_max_keys[sp], k_lo_mx);
        if (k_lo_mn < *(volatile unsigned long long *)&st->lb_min_keys[sp]) atomicMin(&st->lb_min_keys[sp], k_lo_mn);
        if (k_hi_mx > *(volatile unsigned long long *)&st->ub_max_keys[sp]) atomicMax(&st->ub_max_keys[sp], k_hi_mx);
        if (k_hi_mn < *(volatile unsigned long long *)&st->ub_min_keys[sp]) atomicMin(&st->ub_min_keys[sp], k_hi_mn);
        if (sm_mn <= sm_mx) {
            const unsigned long long k_mn = f64_key(sm_mn), k_mx = f64_key(sm_mx);
            if (k_mn < *(volatile unsigned long long *)&st->smp_min_keys[sp]) atomicMin(&st->smp_min_keys[sp], k_mn);
            if (k_mx > *(volatile unsigned long long *)&st->smp_max_keys[sp]) atomicMax(&st->smp_max_keys[sp], k_mx);
        }
    }
}

}  // namespace rm
