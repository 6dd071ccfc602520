// respmon_amd/csrc/rm_tile_eval.h -- the collapse passes of skip_levels_at_top = 3, 4 without a value store (round 4)
//
//   raw[t] = pyrUp^S(C_S[t])                        (pyramid.py:51-57 below `skip`: transforms.py:150-160 leaves those levels zero)
//   heat   = (1 / T) sum_t (raw[t] >= top ? min : raw[t])      (transforms.py:184-192, base.py:562), sequentially in t
//
// Rounds 1-3 evaluated every kept (tile, frame) pair in one flat pass (k_eval_pairs: a single-wave workgroup runs the generic pyrUp
// chain in LDS, ~2 800 instructions per pair, 20+ us of latency), parked the 8 KB of values of each pair in a value store and summed
// them in a second pass (k_masked_sum_tiles): 28 + 21 us per step at 1080p x 256 for 2 600 pairs, 21 MB written and read back.
// Here:
//   * TileEval<S, HALF>: ONE wave evaluates a 64 x 16 tile (or its upper / lower 8 rows) of a frame from the tile's level-S footprint
//     with everything frame-invariant settled before the frame loop, as rm_dense_sum.h's DenseW does for S <= 2: the footprint of the
//     tile at level k is the fixed VIRTUAL rectangle rows (16 ty >> k) - 1 ..  by columns (64 tx >> k) - 1 .. (10 x 34, 7 x 19,
//     5 x 11, 4 x 7 at k = 1 .. 4); virtual rows outside the image are materialised as the rows OpenCV's border rules substitute
//     (row -1 := row 1 falls out of the arithmetic, rows past the bottom repeat the last one), columns outside the image only ever
//     meet a zero weight; lane = destination column, the horizontal values of a step stay in registers and the row structure (which
//     rows are even, which three values meet) is compile time.  ~300 instructions per (tile, frame) instead of ~2 800.
//     Same expressions per value as up_at() / chain_step() / level0_rows() (commuted additions and merged power-of-two scalings at
//     most): bit-identical to the generic chain.
//   * k_eval_c<S>: the exact raw.min() / raw.max() from the C pairs alone (k_select_pairs' list_a), one wave per pair.
//   * k_tile_sum<S>: one workgroup of 16 waves per heavy tile (or half tile): round r evaluates the tile's next 16 kept frames
//     -- in TIME order; wave w takes frame r * 16 + w -- masks them with the exact `top` and parks them in LDS; after a barrier the
//     waves add the 16 frames, in frame order and with the pruned frames' `min` in between, to the running sums they own.  Same values,
//     same order of additions as k_masked_sum_tiles / k_dense_sum: bit-identical.  Nothing but C_S is read, nothing but the heatmap
//     written, and the cost of a tile grows with ITS kept frames only -- from the sparse synthetic stream (86 heavy tiles) to a stream
//     that keeps every pair.
#pragma once

namespace rm {

// geometry of the virtual footprints; HALF: the upper or lower 8 rows of the tile only
template <int S, bool HALF> struct TileFoot {
    static_assert(S >= 1 && S <= 4, "TileEval covers skip_levels_at_top 1 .. 4");
    static constexpr int nc(int k) { return k == 1 ? 34 : k == 2 ? 19 : k == 3 ? 11 : 7; }
    // rows of level k the evaluation holds in its buffer
    static constexpr int nr(int k)
    {
        if (!HALF) return k == 1 ? 10 : k == 2 ? 7 : k == 3 ? 5 : 4;
        return k == 1 ? 6 : k == 2 ? 5 : k == 3 ? (S == 4 ? 5 : 4) : 4;   // (S = 4: level 3 is computed whole, its needed rows start at an odd index)
    }
    // rows of level k a step k -> k - 1 reads
    static constexpr int nr_src(int k)
    {
        if (!HALF) return nr(k);
        return k == 2 ? 5 : 4;
    }
    static constexpr int size(int k) { return nr(k) * nc(k); }
    static constexpr int off(int k) { int o = 0; for (int i = 1; i < k; ++i) o += size(i); return o; }   // level 1 first
    static constexpr int TOTAL = off(S) + size(S);    // doubles of LDS per wave
    static constexpr int NST = size(S);               // staged elements
    static constexpr int PF = (NST + 63) / 64;        // ... per lane
    static constexpr int NV = HALF ? 8 : 16;          // level-0 values per lane
};

// first row index (inside the FULL virtual footprint of level k) of the rows a half-tile evaluation holds of level k
template <int S, bool HALF> __host__ __device__ __forceinline__ int te_row_start(int k, int hsel)
{
    if (!HALF) return 0;
    return k == 1 ? 4 * hsel : k == 2 ? 2 * hsel : (k == 3 && S == 3) ? hsel : 0;
}

inline bool tile_eval_ok(const ChainGeom &g)
{
    if (g.S < 1 || g.S > 4) return false;
    for (int k = 1; k <= g.S; ++k) if (g.h[k] < 2 || g.w[k] < 2) return false;
    return true;
}

// everything about a tile that does not depend on the frame
template <int S, bool HALF> struct TileSetup {
    using F = TileFoot<S, HALF>;
    int off_g[F::PF];             // staged element lane + 64 p: offset inside a frame of C_S (virtual rows / columns resolved)
    int ha[S + 1], hb[S + 1], hc[S + 1];   // step k -> k - 1 (k = 2 .. S), lane < nc(k - 1): element offsets of the three column taps in a source row
    double wa[S + 1], wb[S + 1], wc[S + 1];
    int lastrow[S + 1];           // level k (k = 1 .. S - 1): last buffer row that lies inside the image; later rows repeat it
    int src_row0[S + 1];          // step k -> k - 1: first buffer row of level k the step reads
    double we_a, we_b, we_c, wo_b, wo_c;   // level 1 -> 0, this lane's column pair
    int l0off;                    // ... its taps of source row k: slice[l0off + k * nc(1) + {0, 1, 2}]
    int X, Y0;                    // ... its pixels: columns X, X + 1, rows Y0 .. Y0 + NV / 2 - 1
};

template <int S, bool HALF>
__device__ __forceinline__ void tile_setup(const ChainGeom &g, int tx, int ty, int hsel, int lane, TileSetup<S, HALF> &ts)
{
    using F = TileFoot<S, HALF>;
    const int hS = g.h[S], wS = g.w[S];
    {   // staging: virtual rows / columns of the level-S footprint resolved to addresses
        const int fy = ((16 * ty) >> S) - 1 + te_row_start<S, HALF>(S, hsel), fx = ((64 * tx) >> S) - 1;
#pragma unroll
        for (int p = 0; p < F::PF; ++p) {
            const int i = min(lane + 64 * p, F::NST - 1);
            const int r = i / F::nc(S), c = i - r * F::nc(S);
            const int yv = fy + r, xv = fx + c;
            const int ya = yv < 0 ? 1 : (yv > hS - 1 ? hS - 1 : yv), xa = min(max(xv, 0), wS - 1);
            ts.off_g[p] = ya * wS + xa;
        }
    }
#pragma unroll
    for (int k = 2; k <= S; ++k) {
        // lane c owns destination column xv of level k - 1; its taps j - 1, j, j + 1 of level k (make_htap's five shapes)
        const int fxd = ((64 * tx) >> (k - 1)) - 1, fxs = ((64 * tx) >> k) - 1;
        const int xv = fxd + lane;
        const int sw = g.w[k], dw = g.w[k - 1];
        ts.ha[k] = ts.hb[k] = ts.hc[k] = 0; ts.wa[k] = ts.wb[k] = ts.wc[k] = 0.0;   // (outside the image / beyond the footprint: a finite value nobody reads with a non-zero weight)
        if (lane < F::nc(k - 1) && xv >= 0 && xv < dw) {
            const HTap t = make_htap(xv, sw);
            ts.ha[k] = t.ia - fxs; ts.hb[k] = t.ib - fxs; ts.hc[k] = t.ic - fxs;
            ts.wa[k] = t.wa; ts.wb[k] = t.wb; ts.wc[k] = t.wc;
        }
        ts.src_row0[k] = (te_row_start<S, HALF>(k - 1, hsel) >> 1) - te_row_start<S, HALF>(k, hsel);
    }
#pragma unroll
    for (int k = 1; k < S; ++k) ts.lastrow[k] = (g.h[k] - 1) - (((16 * ty) >> k) - 1 + te_row_start<S, HALF>(k, hsel));
    // level 1 -> 0: lane = (column pair cp, row group rg): columns X, X + 1; full tile: rows 16 ty + 8 rg .. + 7, half: 16 ty + 8 hsel + 4 rg .. + 3
    const int cp = lane & 31, rg = lane >> 5;
    const int sw1 = g.w[1];
    ts.X = 64 * tx + 2 * cp;
    ts.Y0 = 16 * ty + (HALF ? 8 * hsel + 4 * rg : 8 * rg);
    {
        const int j = ts.X >> 1;
        const bool left = j == 0, right = j >= sw1 - 1;
        ts.we_a = left ? 0.0 : 1.0; ts.we_b = right ? 7.0 : 6.0; ts.we_c = left ? 2.0 : (right ? 0.0 : 1.0);
        ts.wo_b = right ? 8.0 : 4.0; ts.wo_c = right ? 0.0 : 4.0;
    }
    // source rows (Y0 >> 1) - 1 .. of level 1; the buffer's first row is virtual row 8 ty - 1 + te_row_start(1)
    ts.l0off = F::off(1) + ((HALF ? 2 * rg : 4 * rg)) * F::nc(1) + cp;
}

// one pyrUp step inside the wave's slice, level K -> K - 1 (K >= 2)
// lmin (nullable): running minimum of the destination values this lane forms (lanes beyond the footprint leave it alone)
template <int S, bool HALF, int K>
__device__ __forceinline__ void te_step(const TileSetup<S, HALF> &ts, double *sl, int lane, double *lmin = nullptr)
{
    using F = TileFoot<S, HALF>;
    constexpr int NRS = F::nr_src(K), PS = F::nc(K), NRD = F::nr(K - 1), PD = F::nc(K - 1);
    static_assert(((NRD - 1) >> 1) + ((NRD - 1) & 1 ? 2 : 1) <= NRS - 1, "source rows of the last destination row");
    const double *src = sl + F::off(K) + ts.src_row0[K] * PS;
    double *dst = sl + F::off(K - 1);
    const int oa = ts.ha[K], ob = ts.hb[K], oc = ts.hc[K];
    const double wa = ts.wa[K], wb = ts.wb[K], wc = ts.wc[K];
    double hq[NRS];
#pragma unroll
    for (int q = 0; q < NRS; ++q) hq[q] = dw_tap3(src[q * PS + oa], src[q * PS + ob], src[q * PS + oc], wa, wb, wc);
    // buffer row p of level K - 1 <-> an odd virtual row for even p (the values of rows q, q + 1 of level K), an even one for odd p
    // (rows q, q + 1, q + 2): up_at() with the exact power-of-two scalings merged
    const int last = ts.lastrow[K - 1];
    if (last >= NRD - 1) {   // (uniform) every buffer row lies inside the image: the common case, no selects
#pragma unroll
        for (int p = 0; p < NRD; ++p) {
            const int q = p >> 1;
            const double v = (p & 1) ? (hq[q] + hq[q + 1] * 6 + hq[q + 2]) * (1.0 / 64) : (hq[q] + hq[q + 1]) * (1.0 / 16);
            if (lane < PD) dst[p * PD + lane] = v;
            if (lmin && lane < PD) *lmin = (v < *lmin) ? v : *lmin;
        }
        return;
    }
    double prev = 0.0;
#pragma unroll
    for (int p = 0; p < NRD; ++p) {
        const int q = p >> 1;
        double v = (p & 1) ? (hq[q] + hq[q + 1] * 6 + hq[q + 2]) * (1.0 / 64) : (hq[q] + hq[q + 1]) * (1.0 / 16);
        if (p > last) v = prev;   // (uniform) virtual row past the bottom of the image: the last row again (up_at()'s r2)
        prev = v;
        if (lane < PD) dst[p * PD + lane] = v;
        if (lmin && lane < PD) *lmin = (v < *lmin) ? v : *lmin;
    }
}

// lmin1 (nullable): the step that forms level 1 tracks this lane's minimum of it
template <int S, bool HALF, int K> struct TeChain {
    static __device__ __forceinline__ void run(const TileSetup<S, HALF> &ts, double *sl, int lane, double *lmin1 = nullptr)
    {
        te_step<S, HALF, K>(ts, sl, lane, K == 2 ? lmin1 : nullptr);
        wave_sync();
        TeChain<S, HALF, K - 1>::run(ts, sl, lane, lmin1);
    }
};
template <int S, bool HALF> struct TeChain<S, HALF, 1> {
    static __device__ __forceinline__ void run(const TileSetup<S, HALF> &, double *, int, double * = nullptr) {}
};

// the staged level S of the frame is in the slice (and visible): run the chain; out[8 o + r] (full tile) / out[4 o + r] (half) =
// raw[t, Y0 + r, X + o]
// the last step, level 1 (in the slice) -> level 0 (registers)
template <int S, bool HALF>
__device__ __forceinline__ void tile_eval_level0(const TileSetup<S, HALF> &ts, double *sl, double (&out)[TileFoot<S, HALF>::NV]);

// tile_eval() that may stop at level 1: every level-0 value is a convex combination of the tile's level-1 footprint (pyrUp's weights
// are positive and sum to one), so when the minimum of that footprint clears `top` by the pruning margin every pixel of the tile is
// masked and the last step need not run.  Returns false in that case (wave-uniform; `out` is not written).  The convexity bound of level 1 is far tighter than the level-S footprint bound the selection works with: on a frame
// of sensor noise (1080p, skip 4) 16 % of the pairs pass it against 39 %, at 4K skip 2 27 % against 98 %.
template <int S, bool HALF>
__device__ __forceinline__ bool tile_eval_below(const TileSetup<S, HALF> &ts, double *sl, int lane, double top_plus_margin, double (&out)[TileFoot<S, HALF>::NV])
{
    static_assert(S >= 2, "level 1 is staged, not computed, at skip 1");
    double lmin1 = __builtin_huge_val();
    TeChain<S, HALF, S>::run(ts, sl, lane, &lmin1);
    if (wave_min(lmin1) >= top_plus_margin) return false;
    tile_eval_level0<S, HALF>(ts, sl, out);
    return true;
}

template <int S, bool HALF>
__device__ __forceinline__ void tile_eval(const TileSetup<S, HALF> &ts, double *sl, int lane, double (&out)[TileFoot<S, HALF>::NV])
{
    TeChain<S, HALF, S>::run(ts, sl, lane);
    tile_eval_level0<S, HALF>(ts, sl, out);
}

template <int S, bool HALF>
__device__ __forceinline__ void tile_eval_level0(const TileSetup<S, HALF> &ts, double *sl, double (&out)[TileFoot<S, HALF>::NV])
{
    using F = TileFoot<S, HALF>;
    constexpr int P1 = F::nc(1), NM = F::NV / 4, NK = NM + 2;   // NM source rows own an (even, odd) output row pair
    const double *l0src = sl + ts.l0off;
    double hve[NK], hvo[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const double *row = l0src + k * P1;
        const double a = row[0], b = row[1], c = row[2];
        hve[k] = dw_tap3(a, b, c, ts.we_a, ts.we_b, ts.we_c);
        hvo[k] = __builtin_fma(c, ts.wo_c, b * ts.wo_b);   // b * 4 + c * 4 (or b * 8 + c * 0): both products exact
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        out[2 * m] = (hve[m] + hve[m + 1] * 6 + hve[m + 2]) * (1.0 / 64);
        out[2 * m + 1] = (hve[m + 1] + hve[m + 2]) * (1.0 / 16);
        out[F::NV / 2 + 2 * m] = (hvo[m] + hvo[m + 1] * 6 + hvo[m + 2]) * (1.0 / 64);
        out[F::NV / 2 + 2 * m + 1] = (hvo[m + 1] + hvo[m + 2]) * (1.0 / 16);
    }
}

// ---- exact raw.min() / raw.max() (transforms.py:185, 187) from the C pairs: one wave per listed pair -------------------------------
template <int S>
__global__ __launch_bounds__(64) void k_eval_c(const double *cS, ChainGeom g, int ntiles, const unsigned int *list_a, CollapseState *st)
{
    RM_TRACE_SCOPE(5);
    using F = TileFoot<S, false>;
    HIP_DYNAMIC_SHARED(double, lds)
    const int lane = threadIdx.x;
    // the first list entry is requested together with the list length (the list buffer is valid memory whatever it turns out to be)
    const unsigned first_idx = list_a[blockIdx.x];
    const unsigned nA = st->n_list_a;
    const double inf = __builtin_huge_val();
    const size_t fs = (size_t)g.h[S] * g.w[S];
    const int H0 = g.h[0], W0 = g.w[0];
    double mn = inf, mx = -inf;
    for (unsigned c = blockIdx.x; c < nA; c += gridDim.x) {
        const unsigned idx = (unsigned)uniform((int)(c == blockIdx.x ? first_idx : list_a[c]));
        const int u = idx / ntiles, tile = idx - u * ntiles;
        const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
        TileSetup<S, false> ts;
        tile_setup<S, false>(g, tx, ty, 0, lane, ts);
        const double *src = cS + (size_t)u * fs;
        double stg[F::PF];
#pragma unroll
        for (int p = 0; p < F::PF; ++p) stg[p] = src[ts.off_g[p]];
        wave_sync();   // the previous pair's reads of the slice are behind us
#pragma unroll
        for (int p = 0; p < F::PF; ++p) if (lane + 64 * p < F::NST) lds[F::off(S) + lane + 64 * p] = stg[p];
        wave_sync();
        double v[16];
        tile_eval<S, false>(ts, lds, lane, v);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (ts.Y0 + r < H0 && ts.X + o < W0) { const double x = v[8 * o + r]; mn = (x < mn) ? x : mn; mx = (x > mx) ? x : mx; }
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane == 0 && blockIdx.x < nA) {
        const unsigned long long kmn = f64_key(mn), kmx = f64_key(mx);
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->min_keys, st->max_keys, sp_, kmn, kmx);
    }
}

// ---- the flat evaluation pass of the sparse path with the wave-private evaluator (k_eval_pairs' job, rm_kernels.h) ---------------------
// One wave per listed pair, every SIMD of the chip busy whatever tile the pairs belong to: exact min / max from all evaluated pairs,
// the values of the kept ones parked in their slot of the value store ([slot][row][column], 16 bytes per lane and row) for
// k_masked_sum_tiles.  ~500 instructions per pair instead of ~2 800.
template <int S>
__global__ __launch_bounds__(64) void k_eval_pairs_fast(const double *cS, ChainGeom g, int ntiles, const unsigned int *list_a, const unsigned int *list_b,
                                                        int *slot_of, CollapseState *st, double *store, SumPlan sp, int Th)
{
    RM_TRACE_SCOPE(5);
    using F = TileFoot<S, false>;
    HIP_DYNAMIC_SHARED(double, lds)
    const int lane = threadIdx.x;
    const unsigned first_idx = list_a[blockIdx.x];
    const unsigned nA = st->n_list_a, nB = st->n_list_b;
    const bool dense = sum_is_dense(st, sp);
    const unsigned n = nA + (dense ? 0u : nB);
    const double inf = __builtin_huge_val();
    const double top_ub = st->top_ub;   // upper bound of `top` from the tile bounds (k_select_pairs)
    const size_t fs = (size_t)g.h[S] * g.w[S];
    const int H0 = g.h[0], W0 = g.w[0];
    double mn = inf, mx = -inf;
    for (unsigned c = blockIdx.x; c < n; c += gridDim.x) {
        RM_TRACE_MARK(5, 0);
        const unsigned raw_idx = c < nA ? (c == blockIdx.x ? first_idx : list_a[c]) : list_b[c - nA];
        const unsigned idx = (unsigned)uniform((int)raw_idx);
        const int u = idx / ntiles, tile = idx - u * ntiles;
        const int slot = dense ? SLOT_PRUNED : uniform(slot_of[slot_index(u, tile, Th)]);   // (needed after the chain: requested now)
        const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
        RM_TRACE_MARK(5, 1);
        TileSetup<S, false> ts;
        tile_setup<S, false>(g, tx, ty, 0, lane, ts);
        const double *src = cS + (size_t)u * fs;
        double stg[F::PF];
#pragma unroll
        for (int p = 0; p < F::PF; ++p) stg[p] = src[ts.off_g[p]];
        RM_TRACE_MARK(5, 2);
        wave_sync();   // the previous pair's reads of the slice are behind us
#pragma unroll
        for (int p = 0; p < F::PF; ++p) if (lane + 64 * p < F::NST) lds[F::off(S) + lane + 64 * p] = stg[p];
        wave_sync();
        RM_TRACE_MARK(5, 3);
        double v[16];
        tile_eval<S, false>(ts, lds, lane, v);
        RM_TRACE_MARK(5, 4);
        double pmn = inf;   // minimum of this pair's tile
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (ts.Y0 + r < H0 && ts.X + o < W0) { const double x = v[8 * o + r]; pmn = (x < pmn) ? x : pmn; mx = (x > mx) ? x : mx; }
        mn = (pmn < mn) ? pmn : mn;
        if (slot != SLOT_PRUNED) {   // wave-uniform
            pmn = wave_min(pmn);
            // nothing of this tile can fall below top (top <= top_ub): every pixel adds `min`, exactly like a pruned pair --
            // no values to park, and the sum pass never sees the frame
            if (pmn >= top_ub) {
                if (lane == 0) slot_of[slot_index(u, tile, Th)] = SLOT_PRUNED;
            } else {
                // lane (column pair cp, row half rg): rows 8 rg .. 8 rg + 7 of the tile, columns 2 cp, 2 cp + 1: 16 bytes per row
                F64Pair *d = reinterpret_cast<F64Pair *>(store + (size_t)slot * (CT_H * CT_W) + (size_t)(8 * (lane >> 5)) * CT_W + 2 * (lane & 31));
#pragma unroll
                for (int r = 0; r < 8; ++r) d[r * (CT_W / 2)] = F64Pair{v[r], v[8 + r]};
            }
        }
        RM_TRACE_MARK(5, 5);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    RM_TRACE_MARK(5, 6);
    if (lane == 0 && blockIdx.x < n) {
        // non-returning atomics and NO load in front of them: a load here would wait (vmcnt) for the acknowledgement of the value
        // stores above, 4-15 us at 1080p x 256 (workgroup timelines, profiles/r04) -- the wave may end while its stores are in flight
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        atomicMin(&st->min_keys[sp_], f64_key(mn));
        atomicMax(&st->max_keys[sp_], f64_key(mx));
    }
}

// ---- the masked time sum of the sparse path over the WHOLE buffer, unique frames loaded once (k_masked_sum_tiles' job) ------------------
//   heat[y, x] = (1 / T) sum_t (raw[t] >= top ? min : raw[t]),  t = 0 .. T - 1 in order            (transforms.py:184-192, base.py:562)
// The band-passed signal is even in time (rm_kernels.h sym_frame): frame t > T / 2 is frame T - t again, so the time-ordered walk of a
// pixel visits the tile's kept UNIQUE frames twice -- ascending, then descending.  k_masked_sum_tiles fetched every visit (16 loads per
// batch, one memory round trip per batch: 4-7 dependent round trips for the heaviest tile of the synthetic stream = 21 us).  Here a
// thread requests the values of up to MS2_B unique frames of its pixel TOGETHER (one round trip), adds them on the way up and again,
// from the same registers, on the way down; frame numbers and slots travel as one value per LANE and are read with v_readlane at
// compile-time positions.  More than MS2_B kept unique frames: batches (the last batch of the way up is the first of the way down).
// Same additions in the same order as k_masked_sum_tiles: bit-identical.  Work items, the constant fill and the heatmap's extrema as
// there.  Dynamic LDS: s_ku[Th], s_ks[Th].
constexpr int MS2_B = 48;

__device__ __forceinline__ int lane_value(int v, int b)   // v of lane b (b: compile-time after unrolling), wave-uniform
{
    return __builtin_amdgcn_readlane(v, b);
}

RM_KERNEL __launch_bounds__(64 * MS_RQ, 2) void k_masked_sum_sym(int T, int ntiles, int W0, int H0, const int *slot_of, const double *store, CollapseState *st,
                                                               double threshold, double *heat, int *tile_nkept, const int *sel_cnt,
                                                               const unsigned int *heavy, int nworkers, SumPlan sp, int *unserved_host)
{
    RM_TRACE_SCOPE(6);
    HIP_DYNAMIC_SHARED(int, s_ku)     // kept unique frames of the tile, ascending; then their slots
    const int Th = sym_frames(T);
    int *s_ks = s_ku + Th;
    __shared__ int s_wcnt[MS_RQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int tiles_x = (W0 + CT_W - 1) / CT_W;
    // requested before the state: the tile of this workgroup's first item and its first slot_of column
    const int tile0 = (int)(heavy[blockIdx.x / MS_Q] % (unsigned)ntiles);
    int slot0 = SLOT_PRUNED;
    if (tid < Th) slot0 = slot_of[slot_index(tid, tile0, Th)];
    const int nitems = (int)st->n_heavy * MS_Q;
    if (sum_is_dense(st, sp)) {   // (uniform over the grid) the value store overflowed: the caller takes the sum another way
        if (unserved_host && blockIdx.x == 0 && tid == 0) *unserved_host = 1;
        return;
    }
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && tid == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    const double cnt = (double)T;
    const int t_up_end = T / 2 + 1;              // the way up: t = u = 0 .. T / 2
    const int u_down = (T + 1) / 2 - 1;          // the way down starts at t = T / 2 + 1, i.e. u = T - t = u_down, and ends at u = 1
    RM_TRACE_MARK(6, 0);
    for (int item = (int)blockIdx.x; item < nitems; item += nworkers) {
        const bool first = item == (int)blockIdx.x;
        const int tile = first ? tile0 : (int)heavy[item / MS_Q], q = item % MS_Q;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        // the tile's kept unique frames, ascending (ballot + prefix popcount, 256 frames per round)
        int m = 0;
        for (int c0 = 0; c0 < Th; c0 += 64 * MS_RQ) {
            const int u = c0 + tid;
            int slot = SLOT_PRUNED;
            if (first && c0 == 0) slot = slot0;
            else if (u < Th) slot = slot_of[slot_index(u, tile, Th)];
            const bool kept = slot != SLOT_PRUNED;
            const unsigned long long mk = __ballot(kept);
            if (lane == 0) s_wcnt[wave] = __popcll(mk);
            __syncthreads();
            int off = m, tot = 0;
#pragma unroll
            for (int w = 0; w < MS_RQ; ++w) { const int c = s_wcnt[w]; off += (w < wave) ? c : 0; tot += c; }
            if (kept) { const int pos = off + __popcll(mk & ((1ull << lane) - 1ull)); s_ku[pos] = u; s_ks[pos] = slot; }
            m += tot;
            __syncthreads();
        }
        RM_TRACE_MARK(6, 1);
        if (tid == 0 && q == 0 && tile_nkept) {   // kept frames in time order (0: every pixel of the tile ends up as the same constant)
            int n_t = 0;
            for (int i = 0; i < m; ++i) { const int u = s_ku[i]; n_t += 1 + ((u >= 1 && u <= u_down) ? 1 : 0); }
            tile_nkept[tile] = n_t;
        }
        const int x = tx * CT_W + lane;
        const int row = q * MS_RQ + wave, y = ty * CT_H + row;
        const bool active = x < W0 && y < H0;
        const double *mine = store + (size_t)row * CT_W + lane;   // + slot * 1024: this pixel in the pair parked in `slot`
        const int nb = (m + MS2_B - 1) / MS2_B;
        double acc = 0.0;
        int t_done = 0;
        double v[MS2_B];
        int ku = 0x7fffffff, ks = 0;     // lane l: frame number and slot of the batch's l-th kept frame
        auto load_batch = [&](int bi) __attribute__((always_inline)) {
            const int i = bi * MS2_B + lane;
            ku = i < m ? s_ku[i] : 0x7fffffff;
            ks = i < m ? s_ks[i] : 0;
#pragma unroll
            for (int b = 0; b < MS2_B; ++b) {
                const int slot = lane_value(ks, b);
                v[b] = (bi * MS2_B + b < m && active) ? mine[(size_t)slot * (CT_H * CT_W)] : 0.0;
            }
        };
        // the way up: t = u
        for (int bi = 0; bi < nb; ++bi) {
            load_batch(bi);
            RM_TRACE_MARK(6, 2);
#pragma unroll
            for (int b = 0; b < MS2_B; ++b) {
                if (bi * MS2_B + b < m) {   // (uniform)
                    const int t_stop = lane_value(ku, b);               // frames [t_done, t_stop) are pruned
                    for (int t = t_done; t < t_stop; ++t) acc = acc + min_val;
                    acc = acc + ((v[b] >= top) ? min_val : v[b]);
                    t_done = t_stop + 1;
                }
            }
        }
        for (int t = t_done; t < t_up_end; ++t) acc = acc + min_val;
        t_done = t_up_end;
        RM_TRACE_MARK(6, 3);
        // the way down: t = T - u for the kept u in [1, u_down], largest first (the batch in registers is the last one of the way up)
        for (int bi = nb - 1; bi >= 0; --bi) {
            if (bi != nb - 1) load_batch(bi);
#pragma unroll
            for (int b = MS2_B - 1; b >= 0; --b) {
                if (bi * MS2_B + b < m) {   // (uniform)
                    const int u = lane_value(ku, b);
                    if (u >= 1 && u <= u_down) {
                        const int t_stop = T - u;
                        for (int t = t_done; t < t_stop; ++t) acc = acc + min_val;
                        acc = acc + ((v[b] >= top) ? min_val : v[b]);
                        t_done = t_stop + 1;
                    }
                }
            }
        }
        for (int t = t_done; t < T; ++t) acc = acc + min_val;
        RM_TRACE_MARK(6, 12);
        double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
        if (active) {
            const double a = acc / cnt;          // base.py:562: np.average = sum / T
            heat[(size_t)y * W0 + x] = a;
            hmn = a; hmx = a;
        }
        block_minmax(hmn, hmx);                  // the heatmap's extrema for base.py:563
        if (tid == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp_ = blockIdx.x & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
        }
        RM_TRACE_MARK(6, 13);
        __syncthreads();   // s_ku is rewritten by the next item
    }
    // FILL: by the workgroups without items when there are any, by every workgroup otherwise (as in k_masked_sum_tiles)
    const int idle = nworkers - min(nitems, nworkers);
    const int nfill = idle > 0 ? idle : nworkers;
    const int fid = idle > 0 ? (int)blockIdx.x - nitems : (int)blockIdx.x;
    if (fid < 0) return;
    double lead = 0.0;
    for (int t = 0; t < T; ++t) lead = lead + min_val;
    const double fv = lead / cnt;
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
                    if (y < H0) heat[(size_t)y * W0 + x] = fv;
                }
            }
            if (tid == 0 && tile_nkept) tile_nkept[tile] = 0;     // 0: every pixel of the tile is the same constant
        }
    }
    if (any && tid == 0) {
        const unsigned long long kv = f64_key(fv);
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kv, kv);
    }
}

// ---- the masked time sum of the sparse path, one wave per (heavy tile, row), values staged by LDS-DMA ---------------------------------
// k_masked_sum_tiles walks a pixel's kept frames in batches of 16 register loads, one memory round trip per batch: 4-7 dependent round
// trips for the heaviest tiles of the synthetic stream (21 us).  Here ONE wave owns one row of a heavy tile (lane = column):
//   1. the tile's kept UNIQUE frames (the band-passed signal is even in time: rm_kernels.h sym_frame), compacted by ballot;
//   2. their values of this row travel store -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, two rows of 512 bytes per
//      instruction, no registers, ALL requests in flight together: one round trip however many frames);
//   3. the additions run from LDS: on the way up (t = u) and, from the same LDS copy, on the way down (t = T - u) -- every stored value
//      is fetched once.
// More kept unique frames than MSR_CHUNK: chunks (the last chunk of the way up is the first of the way down).  Same additions in the
// same order as k_masked_sum_tiles: bit-identical.  Whole-buffer sums only (frame shards keep k_masked_sum_tiles).
// Dynamic LDS: val[MSR_CHUNK][64] doubles, then s_ku[Th], s_ks[Th].
constexpr int MSR_CHUNK = 40;
constexpr int MSR_PRE = 3;     // trips of the frame compaction whose slot_of entries the first item requests up front (T <= 382)

__device__ __forceinline__ double masked_gap(double acc, int n, double min_val)   // n sequential additions of `min`
{
    for (; n >= 8; n -= 8) {
        acc = acc + min_val; acc = acc + min_val; acc = acc + min_val; acc = acc + min_val;
        acc = acc + min_val; acc = acc + min_val; acc = acc + min_val; acc = acc + min_val;
    }
    switch (n) {   // (one jump instead of a loop of taken branches)
    case 7: acc = acc + min_val; [[fallthrough]];
    case 6: acc = acc + min_val; [[fallthrough]];
    case 5: acc = acc + min_val; [[fallthrough]];
    case 4: acc = acc + min_val; [[fallthrough]];
    case 3: acc = acc + min_val; [[fallthrough]];
    case 2: acc = acc + min_val; [[fallthrough]];
    case 1: acc = acc + min_val; [[fallthrough]];
    default: break;
    }
    return acc;
}

__device__ __forceinline__ int lane_value_dyn(int v, int b)   // v of lane b (b wave-uniform, not compile-time)
{
    return __builtin_amdgcn_readlane(v, b);
}

RM_KERNEL __launch_bounds__(64) void k_masked_sum_rows(int T, int ntiles, int W0, int H0, const int *slot_of, const double *store, CollapseState *st,
                                                        double threshold, double *heat, int *tile_nkept, const int *sel_cnt,
                                                        const unsigned int *heavy, int nworkers, SumPlan sp, int *unserved_host)
{
    RM_TRACE_SCOPE(6);
    HIP_DYNAMIC_SHARED(double, val)    // [MSR_CHUNK][64]
    const int Th = sym_frames(T);
    int *s_ku = reinterpret_cast<int *>(val + MSR_CHUNK * 64);   // kept unique frames of the tile, ascending; then their slots
    int *s_ks = s_ku + Th;
    const int lane = threadIdx.x;
    const int tiles_x = (W0 + CT_W - 1) / CT_W;
    // requested before the state: the tile of this wave's first item and its first slot_of entries
    const int tile0 = (int)(heavy[blockIdx.x / CT_H] % (unsigned)ntiles);
    int slot0[MSR_PRE];
#pragma unroll
    for (int k = 0; k < MSR_PRE; ++k) slot0[k] = (lane + 64 * k < Th) ? slot_of[slot_index(lane + 64 * k, tile0, Th)] : SLOT_PRUNED;
    const int nitems = (int)st->n_heavy * CT_H;
    if (sum_is_dense(st, sp)) {   // (uniform over the grid) the value store overflowed: the sum is taken another way
        if (unserved_host && blockIdx.x == 0 && lane == 0) *unserved_host = 1;
        return;
    }
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && lane == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    const double cnt = (double)T;
    const int t_up_end = T / 2 + 1;              // the way up: t = u = 0 .. T / 2
    const int u_down = (T + 1) / 2 - 1;          // the way down starts at t = T / 2 + 1, i.e. u = T - t = u_down, and ends at u = 1
#ifndef RM_HIPEMU
    const unsigned val_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char *)val);
#endif
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
    RM_TRACE_MARK(6, 0);
    for (int item = (int)blockIdx.x; item < nitems; item += nworkers) {
        const bool first = item == (int)blockIdx.x;
        const int tile = first ? tile0 : (int)heavy[item / CT_H], row = item % CT_H;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        // the tile's kept unique frames, ascending (ballot + prefix popcount, 64 frames per trip)
        wave_sync();   // (the previous item's reads of the lists are behind us)
        int m = 0;
        for (int c0 = 0; c0 < Th; c0 += 64) {
            const int u = c0 + lane;
            int slot = SLOT_PRUNED;
            if (first && c0 < 64 * MSR_PRE) {
#pragma unroll
                for (int k = 0; k < MSR_PRE; ++k) if (c0 == 64 * k) slot = slot0[k];
            } else if (u < Th) slot = slot_of[slot_index(u, tile, Th)];
            const bool kept = slot != SLOT_PRUNED;
            const unsigned long long mk = __ballot(kept);
            if (kept) { const int pos = m + __popcll(mk & ((1ull << lane) - 1ull)); s_ku[pos] = u; s_ks[pos] = slot; }
            m += __popcll(mk);
        }
        wave_sync();
        RM_TRACE_MARK(6, 1);
        if (row == 0 && tile_nkept) {   // kept frames in time order (0: every pixel of the tile ends up as the same constant)
            int n_t = 0;
            for (int i = lane; i < m; i += 64) { const int u = s_ku[i]; n_t += 1 + ((u >= 1 && u <= u_down) ? 1 : 0); }
            for (int d = 32; d >= 1; d >>= 1) n_t += __shfl_xor(n_t, d);
            if (lane == 0) tile_nkept[tile] = n_t;
        }
        const int x = tx * CT_W + lane, y = ty * CT_H + row;
        const bool active = x < W0 && y < H0;
        // chunk ci of the kept frames -> val[j][*]: lanes 0 .. 31 fetch frame 2 i, lanes 32 .. 63 frame 2 i + 1 of the pair i
        auto stage = [&](int ci) __attribute__((always_inline)) {
            const int i0 = ci * MSR_CHUNK, n = min(m - i0, MSR_CHUNK);
            wave_sync();   // the previous chunk's reads of val are behind us
#ifndef RM_HIPEMU
            const int half = lane >> 5, l32 = lane & 31;
            for (int j = 0; j < n; j += 2) {
                const int jj = min(j + half, n - 1);     // (an odd count: the upper half repeats the last frame into a row nobody reads)
                const int slot = s_ks[i0 + jj];
                const double *gp = store + (size_t)slot * (CT_H * CT_W) + (size_t)row * CT_W + 2 * l32;
                unsigned keep;
                const unsigned dst = val_lds + (unsigned)j * 512u;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gp), "s"(dst) : "memory");
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only: every piece has landed
            asm volatile("" ::: "memory");
#else
            for (int j = 0; j < n; ++j) val[j * 64 + lane] = store[(size_t)s_ks[i0 + j] * (CT_H * CT_W) + (size_t)row * CT_W + lane];
#endif
            wave_sync();
            return n;
        };
        const int nc = (m + MSR_CHUNK - 1) / MSR_CHUNK;
        double acc = 0.0;
        int t_done = 0;
        int n_last = 0;
        // the way up: t = u.  The chunk's frame numbers travel as one value per lane (read with v_readlane), its values come from
        // LDS four at a time: the chain of additions never waits for a look-up of its own
        int kuv = 0;
        for (int ci = 0; ci < nc; ++ci) {
            const int n = stage(ci);
            n_last = n;
            RM_TRACE_MARK(6, 2);
            const int i0 = ci * MSR_CHUNK;
            kuv = lane < n ? s_ku[i0 + lane] : 0;
            for (int j0 = 0; j0 < n; j0 += 4) {
                double v4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v4[k] = val[min(j0 + k, n - 1) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 + k < n) {   // (uniform)
                        const int t_stop = lane_value_dyn(kuv, j0 + k);     // frames [t_done, t_stop) are pruned
                        acc = masked_gap(acc, t_stop - t_done, min_val);
                        acc = acc + ((v4[k] >= top) ? min_val : v4[k]);
                        t_done = t_stop + 1;
                    }
                }
            }
        }
        acc = masked_gap(acc, t_up_end - t_done, min_val);
        t_done = t_up_end;
        RM_TRACE_MARK(6, 3);
        // the way down: t = T - u for the kept u in [1, u_down], largest first (the chunk in LDS is the last one of the way up)
        for (int ci = nc - 1; ci >= 0; --ci) {
            const int n = ci == nc - 1 ? n_last : stage(ci);
            const int i0 = ci * MSR_CHUNK;
            if (ci != nc - 1) kuv = lane < n ? s_ku[i0 + lane] : 0;
            for (int j0 = n - 1; j0 >= 0; j0 -= 4) {
                double v4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v4[k] = val[max(j0 - k, 0) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 - k >= 0) {   // (uniform)
                        const int u = lane_value_dyn(kuv, j0 - k);
                        if (u >= 1 && u <= u_down) {
                            const int t_stop = T - u;
                            acc = masked_gap(acc, t_stop - t_done, min_val);
                            acc = acc + ((v4[k] >= top) ? min_val : v4[k]);
                            t_done = t_stop + 1;
                        }
                    }
                }
            }
        }
        acc = masked_gap(acc, T - t_done, min_val);
        RM_TRACE_MARK(6, 12);
        if (active) {
            const double a = acc / cnt;          // base.py:562: np.average = sum / T
            heat[(size_t)y * W0 + x] = a;
            hmn = (a < hmn) ? a : hmn; hmx = (a > hmx) ? a : hmx;
        }
    }
    // FILL: by the waves without items when there are any, by every wave otherwise
    const int idle = nworkers - min(nitems, nworkers);
    const int nfill = idle > 0 ? idle : nworkers;
    const int fid = idle > 0 ? (int)blockIdx.x - nitems : (int)blockIdx.x;
    if (fid >= 0) {
        const double fv = masked_gap(0.0, T, min_val) / cnt;
        constexpr int FU = 4;    // tiles whose kept-pair counts are requested together
        bool any = false;
        for (int base = fid; base < ntiles; base += FU * nfill) {
            int cntk[FU];
#pragma unroll
            for (int k = 0; k < FU; ++k) { const int tile = base + k * nfill; cntk[k] = tile < ntiles ? sel_cnt[tile] : 1; }
#pragma unroll
            for (int k = 0; k < FU; ++k) {
                const int tile = base + k * nfill;
                if (cntk[k] != 0) continue;               // past the end, or workers sum this tile
                any = true;
                const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
                const int x = tx * CT_W + lane, y0 = ty * CT_H;
                if (x < W0) {
#pragma unroll
                    for (int j = 0; j < CT_H; ++j) if (y0 + j < H0) heat[(size_t)(y0 + j) * W0 + x] = fv;
                }
                if (lane == 0 && tile_nkept) tile_nkept[tile] = 0;     // 0: every pixel of the tile is the same constant
            }
        }
        if (any) { hmn = (fv < hmn) ? fv : hmn; hmx = (fv > hmx) ? fv : hmx; }
    }
    // the heatmap's extrema for base.py:563
    hmn = wave_min(hmn); hmx = wave_max(hmx);
    if (lane == 0 && hmn <= hmx) {
        const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
    }
}

// ---- the masked time sum of a DENSE selection at skip 3 / 4: one wave per tile, frame after frame (k_dense_sum_w's form, rm_dense_sum.h) -----
// When (nearly) every (tile, frame) pair is kept -- sensor noise in every pixel, bench.py `worst_case` -- the value store is the
// materialised video in disguise (2.1 GB written and read back at 1080p x 256) and k_tile_sum's rounds of sixteen waves keep one CU
// per tile busy behind two barriers a round (1.85 ms).  Here a wave owns a 64 x 16 tile of the heatmap for ALL frames: TileEval of frame t
// from a private 4.4 KB slice of LDS, `raw >= top ? min : raw` added to 16 running sums per lane in frame order; a pair the selection
// pruned adds `min` to every pixel without being evaluated; no barrier, no store, no separate constant fill, every SIMD of the chip
// busy with two or three independent waves.  Same values, same order of additions as every other sum kernel: bit-identical.
constexpr int DST_MAXW = (MAX_T / 2 + 1 + 63) / 64;   // 64-frame words of a tile's kept mask

template <int S>
__global__ __launch_bounds__(64) RM_WAVES_PER_EU_IF(S <= 2, 4, 3) void k_dense_sum_t(const double *cS, ChainGeom g, int t_first, int t_end, int T, int ntiles, const int *slot_of,
                                                    CollapseState *st, double threshold, double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp,
                                                    int only_if_dense, int *ran_host, const double *lo, int xs_standin, int l1_stop)
{
    using F = TileFoot<S, false>;
    HIP_DYNAMIC_SHARED(double, lds)                 // the wave's footprint slice, the kept mask (DST_MAXW words), the kept frames in time order (T 16-bit entries)
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(lds + F::TOTAL);
    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_mask + DST_MAXW);
    const int lane = threadIdx.x;
    if (only_if_dense && !sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse kernel in front took the sum)
    if (xs_standin && !st->xs_overflow) return;           // (uniform over the grid: the exception store held everything, k_xs_sum took the sum -- rm_xstore.h)
    if (ran_host && blockIdx.x == 0 && lane == 0) *ran_host = 2;   // (pinned: tells rm_locate that the stand-in it enqueued on a hint was needed)
    const int tile = dense_tile_of_block((int)blockIdx.x, ntiles);   // XCD x takes the x-th eighth of the tiles (rm_dense_sum.h)
    if (tile >= ntiles) return;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const int Th = sym_frames(T);
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && lane == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    // Which unique frames of this tile can hold a value below `top`?  The selection (k_select_pairs) had to decide with an UPPER
    // bound of top -- the exact extrema did not exist yet -- and with a loose one it keeps (nearly) every pair of a noisy stream.  Here
    // the exact top is known: a pair whose lower bound lo (minimum of its level-S footprint; every pyrUp output is a convex
    // combination of it) clears top by the pruning margin adds `min` to every pixel, evaluated or not.  Effect of this test alone (same
    // process, switch on / off): 1080p full-frame noise, skip 4: step 1.44 -> 1.20 ms; 720p skip 2 with this kernel forced: 0.567 ->
    // 0.552 (its level-2 bounds are loose -- the same test in k_dense_sum_wf, the kernel 720p takes, with a compacted frame list to keep
    // its four waves balanced, measured 115.3 against 115.0 us and was dropped).  (tile-major slot_of: a few cache lines; lo is
    // [unique frame][tile].)
    const double margin = st->margin;
    // (the exhaustive baseline and the developer switch evaluate every kept pair to the end: no level-1 minimum reaches +inf)
    const double top_m = lo != nullptr ? top + margin : __builtin_huge_val();
    for (int c0 = 0; c0 < Th; c0 += 64) {
        const int u = c0 + lane;
        bool kept = u < Th && slot_of[slot_index(u, tile, Th)] != SLOT_PRUNED;
        if (kept && lo) kept = !(lo[(size_t)u * ntiles + tile] - margin >= top);   // (a NaN bound keeps the pair: NaN must reach the sum)
        const unsigned long long mk = __ballot(kept);
        if (lane == 0) s_mask[c0 >> 6] = mk;
    }
    TileSetup<S, false> ts;
    tile_setup<S, false>(g, tx, ty, 0, lane, ts);
    const size_t fs = (size_t)g.h[S] * g.w[S];
    const int H0 = g.h[0], W0 = g.w[0];
    wave_sync();
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    // The kept frames of [t_first, t_end) in TIME order (round 6): the frame loop used to walk every t -- two LDS look-ups of the mask,
    // the register copies of the prefetch ring and a branch per frame, kept or not: ~25 instructions x 3 M frames that add `min`
    // anyway at 4K x 512, a sixth of the kernel's issue slots.  Now it walks the list; the frames between two entries are a count.
    int nlist = 0;
    for (int c0 = t_first; c0 < t_end; c0 += 64) {
        const int t = c0 + lane;
        bool kept = false;
        if (t < t_end) { const int u = sym_frame(t, T); kept = ((s_mask[u >> 6] >> (u & 63)) & 1ull) != 0; }
        const unsigned long long mk = __ballot(kept);
        if (kept) s_list[nlist + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)t;
        nlist += __popcll(mk);
    }
    wave_sync();
    // The footprints of the next two kept frames travel while this one is evaluated.  A frame that adds `min` to every pixel --
    // stopped at level 1 by tile_eval_below() -- only counts up `gap`; the additions happen, in order, in front of the next frame that
    // has real values (and at the end): ONE site with the masked additions, one with the plain ones.  A frame that stopped at level 1
    // also clears its bit in the kept mask: the band-passed signal is even in time, frame T - u is frame u again, and the second
    // visit of the pair (its list entry is looked at, its footprint not used) then costs sixteen additions.
    // l1_stop == 0: the bounds the mask was built from ARE the level-1 extrema (rm_bounds_l1.h) -- a kept pair cannot stop at level 1,
    // the running minimum of tile_eval_below() and its wave reduction (~70 instructions a visit) are left out.
    // (ONE loop body -- the frame in `cur`, the next two in n1 / n2, moved up by register copies: unrolling the body per prefetch slot
    //  doubled the evaluator's code and cost a wave per SIMD)
    double cur[F::PF], n1[F::PF], n2[F::PF];
#pragma unroll
    for (int p = 0; p < F::PF; ++p) { cur[p] = 0.0; n1[p] = 0.0; n2[p] = 0.0; }
    auto fetch = [&](double (&dst)[F::PF], int i) __attribute__((always_inline)) {
        if (i >= nlist) return;   // (uniform)
        const int t = uniform((int)s_list[i]);
        if (l1_stop) { const int u = sym_frame(t, T); if (!((s_mask[u >> 6] >> (u & 63)) & 1ull)) return; }   // (uniform) its first visit stopped at level 1: nothing to fetch
        const double *src = cS + (size_t)sym_frame(t, T) * fs;
#pragma unroll
        for (int p = 0; p < F::PF; ++p) dst[p] = src[ts.off_g[p]];
    };
    fetch(cur, 0); fetch(n1, 1);
    int nkept = 0, gap = 0, t_done = t_first;
#pragma nounroll
    for (int i = 0; i < nlist; ++i) {
        fetch(n2, i + 2);
        const int t = uniform((int)s_list[i]);
        const int u = sym_frame(t, T);
        gap += t - t_done;              // the frames in front of this one that were not kept
        t_done = t + 1;
        const bool kept = !l1_stop || ((s_mask[u >> 6] >> (u & 63)) & 1ull) != 0;   // (uniform; a first visit that stopped at level 1 cleared the bit)
        bool below = false;             // (uniform) evaluated down to level 0: the tile may hold a value below top
        double v[16];
        if (kept) {
            wave_sync();   // the previous frame's reads of the slice are behind us
#pragma unroll
            for (int p = 0; p < F::PF; ++p) if (lane + 64 * p < F::NST) lds[F::off(S) + lane + 64 * p] = cur[p];
            wave_sync();
            if (S >= 2 && l1_stop) {
                if constexpr (S >= 2) below = tile_eval_below<S, false>(ts, lds, lane, top_m, v);
            } else { tile_eval<S, false>(ts, lds, lane, v); below = true; }
            if (!below) {
                if (lane == 0) s_mask[u >> 6] &= ~(1ull << (u & 63));
                wave_sync();
            }
        }
#pragma unroll
        for (int p = 0; p < F::PF; ++p) { cur[p] = n1[p]; n1[p] = n2[p]; }
        if (!below) { ++gap; continue; }
#pragma nounroll
        for (; gap > 0; --gap) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = acc[j] + min_val;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc[j] + ((v[j] >= top) ? min_val : v[j]);
        ++nkept;
    }
    gap += t_end - t_done;
#pragma nounroll
    for (; gap > 0; --gap) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc[j] + min_val;
    }
    // base.py:562: np.average = sum / T when the whole buffer was summed here; the heatmap's extrema for base.py:563
    const double cnt = (double)avg_T;
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = ts.Y0 + r;
        if (y < H0) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (ts.X + o < W0) {
                    const double a = acc[8 * o + r];
                    const double q = avg_T > 0 ? a / cnt : a;
                    heat_sum[(size_t)y * W0 + ts.X + o] = q;
                    hmn = (q < hmn) ? q : hmn; hmx = (q > hmx) ? q : hmx;
                }
            }
        }
    }
    if (tile_nkept && lane == 0) tile_nkept[tile] = nkept;   // 0: every pixel of the tile is the same constant (sparse heatmap exchange)
    if (avg_T > 0) {
        hmn = wave_min(hmn); hmx = wave_max(hmx);
        if (lane == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp_ = blockIdx.x & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
        }
    }
}

// ---- masked time sum, tile by tile ----------------------------------------------------------------------------------------------------
// Work item i = (heavy tile, half) [HALF] or one heavy tile; a workgroup of TS_NW waves takes the items i = blockIdx.x, + nworkers, ...
// The workgroups left without an item fill the tiles without kept pairs with their constant (as k_masked_sum_tiles did).
// LDS: the exchange [TS_NW frames][NV values][64 lanes] -- a wave's footprint slice overlays ITS frame's part of it (the slice is dead
// once the wave holds its level-0 values in registers) -- then the tile's kept frames s_kt[T].
constexpr int TS_NW = 16;

template <int S, bool HALF> __host__ __device__ constexpr int tile_sum_exchange_doubles()
{
    return TS_NW * (TileFoot<S, HALF>::NV * 64 > TileFoot<S, HALF>::TOTAL ? TileFoot<S, HALF>::NV * 64 : TileFoot<S, HALF>::TOTAL);
}

template <int S, bool HALF>
__device__ __forceinline__ void tile_sum_body(const double *cS, const ChainGeom &g, int t_first, int t_end, int T, int ntiles, const int *slot_of,
                                              CollapseState *st, double threshold, double *heat_sum, int avg_T, int *tile_nkept,
                                              const int *sel_cnt, const unsigned int *heavy, int nworkers, double *lds, int *s_wcnt, int tile0, int slot0,
                                              int nheavy)
{
    using F = TileFoot<S, HALF>;
    constexpr int NV = F::NV, NW = TS_NW;
    constexpr int WSTRIDE = tile_sum_exchange_doubles<S, HALF>() / NW;   // doubles per wave of the exchange (>= NV * 64 and >= the slice)
    constexpr int NSUB = HALF ? 2 : 1;
    int *s_kt = reinterpret_cast<int *>(lds + tile_sum_exchange_doubles<S, false>());   // behind the (full-tile sized) exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int H0 = g.h[0], W0 = g.w[0];
    const int nitems = nheavy * NSUB;
    // transforms.py:184-189: min, max, top = max - (max - min) * threshold
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;
    if (blockIdx.x == 0 && tid == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    const double cnt = (double)avg_T;
    const size_t fs = (size_t)g.h[S] * g.w[S];
    constexpr int NBUF = HALF ? 2 : 1;               // exchange buffers (both granularities fill the same LDS)
    static_assert(NV <= NW, "one running sum per thread: value v of lane l is owned by thread (wave v, lane l)");
    RM_TRACE_MARK(6, 0);
    for (int item = (int)blockIdx.x; item < nitems; item += nworkers) {
        const bool first = item == (int)blockIdx.x;
        const int tile = first ? tile0 : (int)heavy[item / NSUB], hsel = HALF ? item % NSUB : 0;
        const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
        // the tile's kept frames in time order (ballot + prefix popcount, 64 * NW frames per round)
        int nkept = 0;
        for (int c0 = t_first; c0 < t_end; c0 += 64 * NW) {
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
            for (int w = 0; w < NW; ++w) { const int c = s_wcnt[w]; off += (w < wave) ? c : 0; tot += c; }
            if (kept) s_kt[off + __popcll(m & ((1ull << lane) - 1ull))] = t;
            nkept += tot;
            __syncthreads();
        }
        if (tid == 0 && hsel == 0 && tile_nkept) tile_nkept[tile] = nkept;
        RM_TRACE_MARK(6, 1);
        TileSetup<S, HALF> ts;
        tile_setup<S, HALF>(g, tx, ty, hsel, lane, ts);
        RM_TRACE_MARK(6, 2);
        double acc = 0.0;                             // this thread's running sum: value `wave` of lane `lane` (waves >= NV own none)
        int t_done = t_first;
        // this wave's frame of the first round is requested now; inside the loop the next round's travels while this one is evaluated
        double stg[F::PF];
        auto fetch = [&](int fi) __attribute__((always_inline)) {
            const int t = s_kt[min(fi, nkept - 1)];
            const double *src = cS + (size_t)sym_frame(t, T) * fs;
#pragma unroll
            for (int p = 0; p < F::PF; ++p) stg[p] = src[ts.off_g[p]];
        };
        if (nkept > 0) fetch(wave);
        int round = 0;
        for (int r0 = 0; r0 < nkept; r0 += NW, ++round) {
            // half tiles: TWO exchange buffers, so that the additions of round r overlap the evaluation of round r + 1 and one
            // barrier per round is enough (a buffer is rewritten two rounds later, behind the next round's barrier)
            double *exb = lds + (size_t)(NBUF == 2 ? (round & 1) : 0) * NW * WSTRIDE;
            double *sl = exb + (size_t)wave * WSTRIDE;   // this wave's footprint slice == its part of the exchange
#pragma unroll
            for (int p = 0; p < F::PF; ++p) if (lane + 64 * p < F::NST) sl[F::off(S) + lane + 64 * p] = stg[p];
            fetch(r0 + NW + wave);
            wave_sync();
            double v[NV];
            tile_eval<S, HALF>(ts, sl, lane, v);
            wave_sync();   // every lane has its values: the slice may be overwritten
#pragma unroll
            for (int j = 0; j < NV; ++j) sl[j * 64 + lane] = (v[j] >= top) ? min_val : v[j];
            __syncthreads();
            const int nf = min(NW, nkept - r0);
            if (wave < NV) {
                // the round's values of this thread's pixel and the frames they belong to, requested together; then the chain of additions
                double ev[NW];
                int tk[NW];
#pragma unroll
                for (int f = 0; f < NW; ++f) { tk[f] = s_kt[min(r0 + f, nkept - 1)]; ev[f] = exb[(size_t)f * WSTRIDE + wave * 64 + lane]; }
#pragma unroll
                for (int f = 0; f < NW; ++f) {
                    if (f < nf) {   // (uniform)
                        const int t_stop = uniform(tk[f]);              // frames [t_done, t_stop) are pruned
                        for (int t = t_done; t < t_stop; ++t) acc = acc + min_val;
                        acc = acc + ev[f];
                        t_done = t_stop + 1;
                    }
                }
            }
            if (NBUF == 1) __syncthreads();   // (one buffer: the exchange and the slices are rewritten next round)
            RM_TRACE_MARK(6, 3 + (round < 9 ? round : 9));
        }
        for (int t = t_done; t < t_end; ++t) acc = acc + min_val;
        RM_TRACE_MARK(6, 13);
        // base.py:562: np.average = sum / T when the whole buffer was summed here; the heatmap's extrema for base.py:563
        double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
        if (wave < NV) {
            const int o = wave / (NV / 2), r = wave - o * (NV / 2);
            const int y = ts.Y0 + r, x = ts.X + o;
            if (y < H0 && x < W0) {
                const double a = avg_T > 0 ? acc / cnt : acc;
                heat_sum[(size_t)y * W0 + x] = a;
                hmn = a; hmx = a;
            }
        }
        if (avg_T > 0) {
            hmn = wave_min(hmn); hmx = wave_max(hmx);
            if (lane == 0 && wave < NV) {
                const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
                const int sp_ = (blockIdx.x * NW + wave) & (NSTRIPE - 1);
                striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
            }
        }
        RM_TRACE_MARK(6, 14);
        __syncthreads();   // s_kt is rewritten by the next item
    }
    // FILL: by the workgroups without items when there are any, by every workgroup otherwise
    const int idle = nworkers - min(nitems, nworkers);
    const int nfill = idle > 0 ? idle : nworkers;
    const int fid = idle > 0 ? (int)blockIdx.x - nitems : (int)blockIdx.x;
    if (fid < 0) return;
    double lead = 0.0;
    for (int t = t_first; t < t_end; ++t) lead = lead + min_val;
    const double fv = avg_T > 0 ? lead / cnt : lead;
    bool any = false;
    const int tiles_x = g.tiles_x;
    // a tile = 16 rows x 64 columns = 1024 values: one per thread (wave = row)
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
            const int x = tx * CT_W + lane, y = ty * CT_H + wave;
            if (x < W0 && y < H0) heat_sum[(size_t)y * W0 + x] = fv;
            if (tid == 0 && tile_nkept) tile_nkept[tile] = 0;     // 0: every pixel of the tile is the same constant
        }
    }
    if (any && tid == 0 && avg_T > 0) {
        const unsigned long long kv = f64_key(fv);
        const int sp_ = blockIdx.x & (NSTRIPE - 1);
        striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kv, kv);
    }
}

// one launch serves both granularities: half tiles when that still leaves every work item a workgroup of its own (the sparse case: few
// heavy tiles, the longest chain of rounds decides), whole tiles otherwise (no part of the chain is evaluated twice)
template <int S>
__global__ __launch_bounds__(64 * TS_NW) void k_tile_sum(const double *cS, ChainGeom g, int t_first, int t_end, int T, int ntiles, const int *slot_of,
                                                          CollapseState *st, double threshold, double *heat_sum, int avg_T, int *tile_nkept,
                                                          const int *sel_cnt, const unsigned int *heavy, int nworkers, int force_half, SumPlan sp,
                                                          int only_if_dense)
{
    RM_TRACE_SCOPE(6);
    HIP_DYNAMIC_SHARED(double, lds)
    __shared__ int s_wcnt[TS_NW];
    const int tid = threadIdx.x;
    // enqueued behind the sparse sum kernel as its stand-in for a selection the value store cannot hold: that kernel took the sum
    if (only_if_dense && !sum_is_dense(st, sp)) return;   // (uniform over the grid)
    // requested before the state: the tile of this workgroup's first item under either granularity (heavy[] is valid memory whatever
    // n_heavy turns out to be) and its first slot_of column
    const int tile_a = (int)(heavy[blockIdx.x] % (unsigned)ntiles), tile_b = (int)(heavy[blockIdx.x >> 1] % (unsigned)ntiles);
    int slot_a = SLOT_PRUNED, slot_b = SLOT_PRUNED;
    if (t_first + tid < t_end) {
        const int u0 = sym_frame(t_first + tid, T), Th_ = sym_frames(T);
        slot_a = slot_of[slot_index(u0, tile_a, Th_)]; slot_b = slot_of[slot_index(u0, tile_b, Th_)];
    }
    const int nheavy = (int)st->n_heavy;
    const bool half = force_half >= 0 ? force_half != 0 : 2 * nheavy <= nworkers;   // (uniform over the grid; force_half: test hook)
    if (half) tile_sum_body<S, true>(cS, g, t_first, t_end, T, ntiles, slot_of, st, threshold, heat_sum, avg_T, tile_nkept, sel_cnt, heavy, nworkers, lds, s_wcnt, tile_b, slot_b, nheavy);
    else tile_sum_body<S, false>(cS, g, t_first, t_end, T, ntiles, slot_of, st, threshold, heat_sum, avg_T, tile_nkept, sel_cnt, heavy, nworkers, lds, s_wcnt, tile_a, slot_a, nheavy);
}

}  // namespace rm
