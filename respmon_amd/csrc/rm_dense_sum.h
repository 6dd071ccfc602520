// respmon_amd/csrc/rm_dense_sum.h -- the masked time sum of a DENSE stream (round 2)
//
//   heat_sum[y, x] = sum_t (raw[t, y, x] >= top ? min : raw[t, y, x])        transforms.py:184-192 + base.py:562
//
// The sparse path (k_select_pairs -> k_eval_pairs -> k_masked_sum_tiles, rm_kernels.h) evaluates the few (tile, frame) pairs
// that can hold a value below `top`, parks their values in the value store and sums them afterwards.  When most pairs are
// such pairs -- skip_levels_at_top = 2 on a noisy video: every 64x16 tile of every frame -- that is one 8 KB round trip
// through memory per pair on top of single-wave workgroups whose pyrUp steps use a third of their lanes: 12.4 + 5.2 ms at
// 4K x 512, 0.33 + 0.22 ms at 720p x 128.  Here the exact extrema come first, from the pairs that can hold them alone (C pairs,
// k_eval_pairs without a store), and then ONE kernel recomputes the full-resolution values frame after frame and adds them up in
// registers: a workgroup owns a 64 x (16 NW) super-tile of the heatmap for all frames, its NW waves run the pyrUp chain of the
// super-tile's footprint together (same per-pixel arithmetic as chain_step / level0_rows: bit-identical values), wave w keeps
// rows 16 w .. 16 w + 15, lane = column, 16 running sums per lane.  Nothing but C_S is read, nothing but the heatmap is written.
// A pruned pair needs no special case: all of its values are >= top, so every pixel adds `min`, exactly what the sparse
// path adds for it.
#pragma once

namespace rm {

// Which path takes the sum is decided on the device from this call's own selection: sum_is_dense() in rm_kernels.h.  Both sum
// kernels are enqueued; the one whose turn it is not returns at once.

struct DenseGeom {
    int rows;                    // super-tile rows (waves per workgroup x rows per wave)
    int nsx, nsy;                // super-tiles across / down
    int lds_off[MAX_CHAIN];      // level k's footprint buffer, k = 1 .. S
    int lds_hb[MAX_CHAIN];       // scratch of the horizontal pass of step k -> k-1
    int lds_total;               // doubles
};

// footprint of super-tile (sxi, syi) at level k (0 = the super-tile itself): the recurrence of tile_region
__host__ __device__ __forceinline__ Region super_region(const ChainGeom &g, int rows, int sxi, int syi, int k)
{
    Region R;
    R.y0 = syi * rows; R.y1 = min(R.y0 + rows, g.h[0]) - 1;
    R.x0 = sxi * CT_W; R.x1 = min(R.x0 + CT_W, g.w[0]) - 1;
    for (int i = 1; i <= k; ++i) {
        R.y0 = max(0, floordiv2(R.y0) - 1);
        R.y1 = min(g.h[i] - 1, floordiv2(R.y1) + 1);
        R.x0 = max(0, floordiv2(R.x0) - 1);
        R.x1 = min(g.w[i] - 1, floordiv2(R.x1) + 1);
    }
    return R;
}

// one pyrUp step of the footprint inside LDS, level k (Rk) -> level k-1 (Rd): chain_step's arithmetic, regions handed in.
// Rows go to the waves in batches of DS_U whose LDS reads are issued together: one read-to-write round trip per batch instead
// of one per row (the compiler cannot reorder the loads of a row past the previous row's store).
constexpr int DS_U = 4;
__device__ __forceinline__ void dense_step(const ChainGeom &g, const double *src, double *hb, double *dst, int k, const Region &Rk,
                                           const Region &Rd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    const int sp = Rk.x1 - Rk.x0 + 1, srows = Rk.y1 - Rk.y0 + 1;
    const int dw = Rd.x1 - Rd.x0 + 1, drows = Rd.y1 - Rd.y0 + 1;
    const int sh = g.h[k], sw = g.w[k];
    for (int c = lane; c < dw; c += 64) {
        const HTap t = make_htap(Rd.x0 + c, sw);
        const int oa = t.ia - Rk.x0, ob = t.ib - Rk.x0, oc = t.ic - Rk.x0;
        for (int rb = wave; rb < srows; rb += nwaves * DS_U) {
            double a[DS_U], b[DS_U], cc[DS_U];
#pragma unroll
            for (int u = 0; u < DS_U; ++u) {
                const int r = rb + u * nwaves;
                const double *row = src + (r < srows ? r : rb) * sp;
                a[u] = row[oa]; b[u] = row[ob]; cc[u] = row[oc];
            }
#pragma unroll
            for (int u = 0; u < DS_U; ++u) {
                const int r = rb + u * nwaves;
                if (r < srows) hb[r * dw + c] = (a[u] * t.wa + b[u] * t.wb) + cc[u] * t.wc;
            }
        }
    }
    __syncthreads();
    for (int c = lane; c < dw; c += 64) {
        for (int rb = wave; rb < drows; rb += nwaves * DS_U) {
            double h0[DS_U], h1[DS_U], h2[DS_U];
#pragma unroll
            for (int u = 0; u < DS_U; ++u) {
                const int r = min(rb + u * nwaves, drows - 1);
                const int y = Rd.y0 + r, i = y >> 1;                         // uniform per wave
                const int r2 = ((i == sh - 1) ? i : i + 1) - Rk.y0, r1 = i - Rk.y0;
                const int r0 = (y & 1) ? r1 : ((i == 0) ? (sh > 1 ? 1 : 0) : i - 1) - Rk.y0;   // (odd rows: unused)
                h0[u] = hb[r0 * dw + c]; h1[u] = hb[r1 * dw + c]; h2[u] = hb[r2 * dw + c];
            }
#pragma unroll
            for (int u = 0; u < DS_U; ++u) {
                const int r = rb + u * nwaves;
                if (r < drows) {
                    const int y = Rd.y0 + r;
                    dst[r * dw + c] = (y & 1) ? ((h1[u] + h2[u]) * 4) * (1.0 / 64) : (h0[u] + h1[u] * 6 + h2[u]) * (1.0 / 64);
                }
            }
        }
    }
    __syncthreads();
}

// level 1 (LDS, footprint R1) -> level 0 rows y_first .. y_first + NR - 1 of column x (y_first, NR even): level0_rows' arithmetic
template <int NR>
__device__ __forceinline__ void dense_level0(const ChainGeom &g, const double *src, const Region &R1, int x, int y_first, double (&out)[NR])
{
    const int sp = R1.x1 - R1.x0 + 1;
    const int sh = g.h[1], sw = g.w[1];
    const HTap t = make_htap(x, sw);
    const int oa = t.ia - R1.x0, ob = t.ib - R1.x0, oc = t.ic - R1.x0;
    const int i0 = y_first >> 1;
    double hv[NR / 2 + 2];
#pragma unroll
    for (int k = 0; k < NR / 2 + 2; ++k) {
        int i = i0 - 1 + k;
        int r = (i < 0) ? (sh > 1 ? 1 : 0) : (i > sh - 1 ? sh - 1 : i);
        r = min(max(r, R1.y0), R1.y1);   // rows past the image end are computed and dropped: keep their reads inside the footprint
        const double *row = src + (r - R1.y0) * sp;
        hv[k] = (row[oa] * t.wa + row[ob] * t.wb) + row[oc] * t.wc;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        int k = (j >> 1) + 1;
        if (j & 1) out[j] = ((hv[k] + hv[k + 1]) * 4) * (1.0 / 64);
        else out[j] = (hv[k - 1] + hv[k] * 6 + hv[k + 1]) * (1.0 / 64);
    }
}

// NW waves per workgroup, RPW rows of the heatmap per wave: a 64 x (NW RPW) super-tile.  <4,16> and <2,16> for images with
// super-tiles to spare; <4,4> gives one 64x16 tile to four waves when tiles are few and the per-frame latency is what counts.
template <int NW, int RPW>
__global__ __launch_bounds__(64 * NW) void k_dense_sum(const double *cS, ChainGeom g, DenseGeom dg, int t_first, int t_end, int T, CollapseState *st,
                                                        double threshold, double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp)
{
    HIP_DYNAMIC_SHARED(double, lds)
    if (!sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = g.S;
    const int syi = (int)blockIdx.x / dg.nsx, sxi = (int)blockIdx.x - syi * dg.nsx;
    constexpr int ROWS = RPW * NW;
    // transforms.py:184-189: min, max, top = max - (max - min) * threshold
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;
    if (blockIdx.x == 0 && tid == 0) {
        st->min_val = min_val; st->max_val = max_val; st->top = top;
    }
    const Region R0 = super_region(g, ROWS, sxi, syi, 0), RS = super_region(g, ROWS, sxi, syi, S);
    const int nwS = RS.x1 - RS.x0 + 1, nS = (RS.y1 - RS.y0 + 1) * nwS, wS = g.w[S];
    const float inv_nwS = 1.0f / (float)nwS;
    const size_t fs = (size_t)g.h[S] * wS;
    double *dS = lds + dg.lds_off[S];
    const int x = R0.x0 + lane;
    const int y_first = R0.y0 + wave * RPW;
    const bool col_ok = x <= R0.x1 && y_first <= R0.y1;
    const int rows = min(R0.y1 - y_first + 1, RPW);   // <= 0 for a wave below the image
    double acc[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) acc[j] = 0.0;
    // the footprint's own elements of C_S: position fixed for all frames
    constexpr int PF = 4;   // staged elements per thread ...
    constexpr int PD = 3;   // ... held in registers PD frames ahead: a footprint is a few short rows gathered from L2 / HBM, and
                            // one frame of arithmetic (~1 us) does not cover that latency
    int off_g[PF], off_l[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int i = tid + p * 64 * NW;
        int r, c;
        split_rc(i < nS ? i : 0, nwS, inv_nwS, r, c);
        off_g[p] = (RS.y0 + r) * wS + RS.x0 + c;
        off_l[p] = i < nS ? i : -1;
    }
    const bool small = nS <= PF * 64 * NW;
    double nxt[PD][PF];
    auto compute = [&]() __attribute__((always_inline)) {
        __syncthreads();
        Region Rk = RS;   // (the footprints are recomputed per step on the scalar unit: keeping all of them live costs registers)
        for (int k = S; k >= 2; --k) {
            const Region Rd = super_region(g, ROWS, sxi, syi, k - 1);
            dense_step(g, lds + dg.lds_off[k], lds + dg.lds_hb[k], lds + dg.lds_off[k - 1], k, Rk, Rd);
            Rk = Rd;
        }
        if (col_ok) {
            double v[RPW];
            dense_level0<RPW>(g, lds + dg.lds_off[1], Rk, x, y_first, v);
#pragma unroll
            for (int j = 0; j < RPW; ++j) acc[j] = acc[j] + ((v[j] >= top) ? min_val : v[j]);
        }
        if (S == 1) __syncthreads();   // (S >= 2: the barriers of the next frame's first step stand between these reads and its writes)
    };
    if (small) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int t = t_first + d;
            const double *src = cS + (size_t)sym_frame(t < t_end ? t : t_first, T) * fs;
#pragma unroll
            for (int p = 0; p < PF; ++p) nxt[d][p] = (off_l[p] >= 0 && t < t_end) ? src[off_g[p]] : 0.0;
        }
        for (int tb = t_first; tb < t_end; tb += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int t = tb + d;
                if (t < t_end) {   // (uniform)
#pragma unroll
                    for (int p = 0; p < PF; ++p) if (off_l[p] >= 0) dS[off_l[p]] = nxt[d][p];
                    const int tn = t + PD;
                    const double *src = cS + (size_t)sym_frame(tn < t_end ? tn : t_first, T) * fs;
#pragma unroll
                    for (int p = 0; p < PF; ++p) nxt[d][p] = (off_l[p] >= 0 && tn < t_end) ? src[off_g[p]] : 0.0;
                    compute();
                }
            }
        }
    } else {
        for (int t = t_first; t < t_end; ++t) {
            const double *src = cS + (size_t)sym_frame(t, T) * fs;
            for (int i = tid; i < nS; i += 64 * NW) {
                int r, c;
                split_rc(i, nwS, inv_nwS, r, c);
                dS[i] = src[(size_t)(RS.y0 + r) * wS + RS.x0 + c];
            }
            compute();
        }
    }
    // base.py:562: np.average = sum / T when the whole buffer was summed here; the heatmap's extrema for base.py:563
    const double cnt = (double)avg_T;
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
    if (col_ok) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if (j < rows) {
                const double v = avg_T > 0 ? acc[j] / cnt : acc[j];
                heat_sum[(size_t)(y_first + j) * g.w[0] + x] = v;
                hmn = (v < hmn) ? v : hmn; hmx = (v > hmx) ? v : hmx;
            }
        }
    }
    if (tile_nkept && lane == 0 && y_first <= R0.y1 && y_first % CT_H == 0)   // (the sparse heatmap exchange: no tile is known to be the constant)
        tile_nkept[(y_first / CT_H) * g.tiles_x + sxi] = t_end - t_first;
    if (avg_T > 0) {
        block_minmax(hmn, hmx);
        if (tid == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp = blockIdx.x & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp, kmn, kmx);
        }
    }
}

// ---- the same kernel for skip_levels_at_top <= 2 (the reference's module defaults L=4, S=2; 4K L=6, S=2), table driven -----------
// k_dense_sum above is bound by instruction issue: ~600 VALU + ~400 scalar instructions per wave and frame, nearly all of them
// index arithmetic, border rules and tap weights that do not depend on the frame.  With at most ONE pyrUp step between the staged
// level and level 1, everything frame-invariant fits in registers: every thread owns a fixed list of outputs per stage (flat
// index over the stage's rows x columns, so all lanes work, not just the 35 of 64 that own a column) and keeps, per output, the
// three LDS offsets it reads, the offset it writes and its three tap weights.  A frame is then: store the prefetched footprint,
// barrier, NH x (3 LDS reads, 5 flops, 1 write), barrier, NV x (3 reads, 6 flops, 1 write), barrier, the level-0 rows from
// precomputed row offsets, 2 flops + a select per running sum.  Same expressions, same operand order: bit-identical values.
template <int NW, int RPW> struct DenseS2 {
    static constexpr int ROWS = NW * RPW, NT = 64 * NW;
    static constexpr int R2 = ROWS == 64 ? 21 : (ROWS == 32 ? 13 : 9);    // chain_extent(ROWS, 2) + 1
    static constexpr int R1 = ROWS == 64 ? 35 : (ROWS == 32 ? 19 : 11);   // chain_extent(ROWS, 1) + 1
    static constexpr int C1 = 35;                                          // chain_extent(CT_W, 1) + 1
    static constexpr int NH = (R2 * C1 + NT - 1) / NT;                     // horizontal-pass outputs per thread
    static constexpr int NV = (R1 * C1 + NT - 1) / NT;                     // vertical-pass outputs (= level-1 elements) per thread
};

template <int NW, int RPW>
__global__ __launch_bounds__(64 * NW) void k_dense_sum_s2(const double *cS, ChainGeom g, DenseGeom dg, int t_first, int t_end, int T, CollapseState *st,
                                                           double threshold, double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp)
{
    using G = DenseS2<NW, RPW>;
    constexpr int NT = G::NT, NH = G::NH, NV = G::NV, PF = G::NV, ROWS = G::ROWS;
    HIP_DYNAMIC_SHARED(double, lds)
    if (!sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = g.S;   // 1 or 2
    const int syi = (int)blockIdx.x / dg.nsx, sxi = (int)blockIdx.x - syi * dg.nsx;
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && tid == 0) {
        st->min_val = min_val; st->max_val = max_val; st->top = top;
    }
    const Region R0 = super_region(g, ROWS, sxi, syi, 0), R1 = super_region(g, ROWS, sxi, syi, 1), R2 = super_region(g, ROWS, sxi, syi, 2 <= S ? 2 : 1);
    const Region RS = S == 2 ? R2 : R1;
    const int nwS = RS.x1 - RS.x0 + 1, nS = (RS.y1 - RS.y0 + 1) * nwS, wS = g.w[S];
    const size_t fs = (size_t)g.h[S] * wS;
    double *dS = lds + dg.lds_off[S];
    const double *l1 = lds + dg.lds_off[1];
    // staged elements of this thread (position fixed for all frames)
    int off_g[PF], off_l[PF];
    {
        const float inv = 1.0f / (float)nwS;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = tid + p * NT;
            int r, c;
            split_rc(i < nS ? i : 0, nwS, inv, r, c);
            off_g[p] = (RS.y0 + r) * wS + RS.x0 + c;
            off_l[p] = i < nS ? i : -1;
        }
    }
    // horizontal pass of step 2 -> 1: output i = r * dw + c of the scratch buffer, r = level-2 row, c = level-1 column
    const int sp2 = R2.x1 - R2.x0 + 1, srows = R2.y1 - R2.y0 + 1;
    const int dw = R1.x1 - R1.x0 + 1, drows = R1.y1 - R1.y0 + 1;
    int h_src[NH][3], h_dst[NH];
    float h_w[NH][3];
    int v_src[NV][3], v_dst[NV];   // vertical pass: output i = r * dw + c of level 1; v_src[n][0] < 0: odd row
    if (S == 2) {
        const float inv = 1.0f / (float)dw;
        const int sh = g.h[2], sw = g.w[2];
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int i = tid + n * NT;
            const bool ok = i < srows * dw;
            int r, c;
            split_rc(ok ? i : 0, dw, inv, r, c);
            const HTap t = make_htap(R1.x0 + c, sw);
            h_src[n][0] = r * sp2 + t.ia - R2.x0; h_src[n][1] = r * sp2 + t.ib - R2.x0; h_src[n][2] = r * sp2 + t.ic - R2.x0;
            h_w[n][0] = (float)t.wa; h_w[n][1] = (float)t.wb; h_w[n][2] = (float)t.wc;   // 0, 1, 2, 4, 6, 7: exact
            h_dst[n] = ok ? i : -1;
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int i = tid + n * NT;
            const bool ok = i < drows * dw;
            int r, c;
            split_rc(ok ? i : 0, dw, inv, r, c);
            const int y = R1.y0 + r, ii = y >> 1;
            const int r1 = ii - R2.y0, r2 = ((ii == sh - 1) ? ii : ii + 1) - R2.y0;
            const int r0 = ((ii == 0) ? (sh > 1 ? 1 : 0) : ii - 1) - R2.y0;
            v_src[n][0] = (y & 1) ? -1 : r0 * dw + c; v_src[n][1] = r1 * dw + c; v_src[n][2] = r2 * dw + c;
            v_dst[n] = ok ? i : -1;
        }
    }
    // level 1 -> level 0: rows y_first .. y_first + RPW - 1 of column x (dense_level0's arithmetic with the row offsets fixed)
    const int x = R0.x0 + lane;
    const int y_first = R0.y0 + wave * RPW;
    const bool col_ok = x <= R0.x1 && y_first <= R0.y1;
    const int rows = min(R0.y1 - y_first + 1, RPW);
    int l0_row[RPW / 2 + 2], l0_b, l0_c;
    double l0_wa, l0_wb, l0_wc;
    {
        const int sp = R1.x1 - R1.x0 + 1, sh = g.h[1];
        const HTap t = make_htap(col_ok ? x : R0.x0, g.w[1]);
        const int oa = t.ia - R1.x0;
        l0_b = t.ib - t.ia; l0_c = t.ic - t.ia;
        l0_wa = t.wa; l0_wb = t.wb; l0_wc = t.wc;
        const int i0 = y_first >> 1;
#pragma unroll
        for (int k = 0; k < RPW / 2 + 2; ++k) {
            const int i = i0 - 1 + k;
            int r = (i < 0) ? (sh > 1 ? 1 : 0) : (i > sh - 1 ? sh - 1 : i);
            r = min(max(r, R1.y0), R1.y1);
            l0_row[k] = (r - R1.y0) * sp + oa;
        }
    }
    double acc[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) acc[j] = 0.0;
    const double *hb = lds + dg.lds_hb[2];
    double *hbw = lds + dg.lds_hb[2], *l1w = lds + dg.lds_off[1];
    const double *l2 = lds + dg.lds_off[2];
    const bool small = nS <= PF * NT;
    double nxt[PF];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const double *src = cS + (size_t)sym_frame(t < t_end ? t : t_first, T) * fs;
#pragma unroll
        for (int p = 0; p < PF; ++p) nxt[p] = (off_l[p] >= 0 && t < t_end) ? src[off_g[p]] : 0.0;
    };
    if (small) fetch(t_first);
    for (int t = t_first; t < t_end; ++t) {
        if (small) {
#pragma unroll
            for (int p = 0; p < PF; ++p) if (off_l[p] >= 0) dS[off_l[p]] = nxt[p];
            fetch(t + 1);
        } else {
            const double *src = cS + (size_t)sym_frame(t, T) * fs;
            const float inv = 1.0f / (float)nwS;
            for (int i = tid; i < nS; i += NT) {
                int r, c;
                split_rc(i, nwS, inv, r, c);
                dS[i] = src[(size_t)(RS.y0 + r) * wS + RS.x0 + c];
            }
        }
        __syncthreads();
        if (S == 2) {
            double a[NH], b[NH], c[NH];
#pragma unroll
            for (int n = 0; n < NH; ++n) { a[n] = l2[h_src[n][0]]; b[n] = l2[h_src[n][1]]; c[n] = l2[h_src[n][2]]; }
#pragma unroll
            for (int n = 0; n < NH; ++n)
                if (h_dst[n] >= 0) hbw[h_dst[n]] = (a[n] * (double)h_w[n][0] + b[n] * (double)h_w[n][1]) + c[n] * (double)h_w[n][2];
            __syncthreads();
            double h0[NV], h1[NV], h2[NV];
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int o0 = v_src[n][0] < 0 ? v_src[n][1] : v_src[n][0];
                h0[n] = hb[o0]; h1[n] = hb[v_src[n][1]]; h2[n] = hb[v_src[n][2]];
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if (v_dst[n] >= 0) {
                    const double odd = (h1[n] + h2[n]) * 4, even = h0[n] + h1[n] * 6 + h2[n];
                    l1w[v_dst[n]] = ((v_src[n][0] < 0) ? odd : even) * (1.0 / 64);
                }
            }
            __syncthreads();
        }
        if (col_ok) {
            double hv[RPW / 2 + 2];
#pragma unroll
            for (int k = 0; k < RPW / 2 + 2; ++k) {
                const double *row = l1 + l0_row[k];
                hv[k] = (row[0] * l0_wa + row[l0_b] * l0_wb) + row[l0_c] * l0_wc;
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int k = (j >> 1) + 1;
                const double v = (j & 1) ? ((hv[k] + hv[k + 1]) * 4) * (1.0 / 64) : (hv[k - 1] + hv[k] * 6 + hv[k + 1]) * (1.0 / 64);
                acc[j] = acc[j] + ((v >= top) ? min_val : v);
            }
        }
        if (S == 1) __syncthreads();
    }
    const double cnt = (double)avg_T;   // base.py:562-563
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
    if (col_ok) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if (j < rows) {
                const double v = avg_T > 0 ? acc[j] / cnt : acc[j];
                heat_sum[(size_t)(y_first + j) * g.w[0] + x] = v;
                hmn = (v < hmn) ? v : hmn; hmx = (v > hmx) ? v : hmx;
            }
        }
    }
    if (tile_nkept && lane == 0 && y_first <= R0.y1 && y_first % CT_H == 0)
        tile_nkept[(y_first / CT_H) * g.tiles_x + sxi] = t_end - t_first;
    if (avg_T > 0) {
        block_minmax(hmn, hmx);
        if (tid == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp = blockIdx.x & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp, kmn, kmx);
        }
    }
}

// ---- wave-private tiles (round 3): the dense sum for skip_levels_at_top <= 2 -------------------------------------------------------
// k_dense_sum_s2 above holds four waves on three barriers per frame at 184 VGPRs (two waves per SIMD): at 4K x 512 the chip's
// fp64 units are 37 % busy, the rest is LDS latency nobody covers.  Here ONE wave owns a 64 x 16 tile of the heatmap for all
// frames, with a private 4 KB slice of LDS: no s_barrier anywhere (a wave's DS instructions execute in order: wave_sync() only
// fences the compiler), four to five independent waves per SIMD, and every index, border rule and tap weight is settled before
// the frame loop:
//   * VIRTUAL footprints.  The tile's level-1 footprint is always the 10 rows 8 ty - 1 .. 8 ty + 8 by 34 columns 32 tx - 1 ..
//     32 tx + 32, its level-2 footprint the 7 rows 4 ty - 1 .. 4 ty + 5 by 20 columns 16 tx - 2 .. 16 tx + 17.  Rows outside
//     the image are MATERIALISED as the rows OpenCV's border rules substitute for them (row -1 := row 1, row h := row h - 1:
//     up_at()'s r0 / r2), by address at staging time for the staged level and by a copy for the level computed here; columns
//     outside the image hold finite values that the column weights multiply by zero.  After that every read is base +
//     immediate offset and the row structure (which rows are even, which three values meet) is compile-time.
//   * column taps are always the three neighbours (j - 1, j, j + 1) with per-LANE weights: interior 1,6,1 / 0,4,4, left edge
//     0,6,2, right edge 1,7,0 / 0,8,0 -- make_htap()'s five shapes with the same operand order, bit for bit.
//   * level 2 -> 1: lane = level-1 column; the horizontal values of the 7 staged rows stay in REGISTERS and the 10 level-1 rows
//     are formed from them (odd row (a + b) / 16, even row (a + 6 b + c) / 64: the exact power-of-two scalings merged).
//   * level 1 -> 0: lane = (column pair, row half): 2 columns x 8 rows, so the three level-1 taps of a source row serve both
//     columns and six source rows serve eight output rows.
// Same expressions per value as chain_step / level0_rows (additions of the same operands, commuted at most): bit-identical to the
// sparse path.  Needs >= 2 rows and columns at levels 1 .. S; smaller images take k_dense_sum_s2.
template <int S> struct DenseW {
    static constexpr int R2 = 7, P2 = 20;            // staged level-2 footprint (virtual rows x virtual columns)
    static constexpr int R1 = 10, P1 = 34;           // level-1 footprint
    static constexpr int L1_OFF = S == 2 ? R2 * P2 : 0;
    static constexpr int TOTAL = L1_OFF + R1 * P1;   // doubles of LDS per wave
    static constexpr int NST = S == 2 ? R2 * P2 : R1 * P1;   // staged elements
    static constexpr int PF = (NST + 63) / 64;
    static constexpr int PD = 2;                     // frames the staged elements are requested ahead
};

// the row OpenCV substitutes for virtual row y of an image with h >= 2 rows (pyrUp: top reflect-101, bottom replicate)
__host__ __device__ __forceinline__ int up_virtual_row(int y, int h) { return y < 0 ? 1 : (y > h - 1 ? h - 1 : y); }

// (a wa + b wb) + c wc with wa in {0, 1} and wc in {0, 1, 2}: those two products are EXACT, so each addition of one of them
// may be written as a fused multiply-add without changing a bit (round(t + a wa) either way) -- three instructions instead of
// five on the fp64 units this kernel is bound by.  (b wb is inexact for wb = 6, 7: it stays a separate, rounded product.)
__device__ __forceinline__ double dw_tap3(double a, double b, double c, double wa, double wb, double wc)
{
    return __builtin_fma(c, wc, __builtin_fma(a, wa, b * wb));
}

__host__ __device__ __forceinline__ int dense_tile_of_block(int block, int ntiles) { return (block & 7) * ((ntiles + 7) >> 3) + (block >> 3); }
inline unsigned dense_tile_grid(int ntiles) { return (unsigned)(8 * ((ntiles + 7) >> 3)); }

inline bool dense_wave_ok(const ChainGeom &g)
{
    if (g.S < 1 || g.S > 2) return false;
    for (int k = 1; k <= g.S; ++k) if (g.h[k] < 2 || g.w[k] < 2) return false;
    return true;
}

// FR frames per trip of the frame loop, each with its own LDS slice: their stages interleave (stage both, first pyrUp step of
// both, ...), which gives a LONE wave two independent dependency chains -- for images with fewer tiles than the chip has SIMDs
// to fill (720p: 900 tiles on 1 024 SIMDs) the kernel is bound by the latency of one wave's chain, not by issue.  The running
// sums still take frame t before frame t + 1.
template <int S, int FR>
__global__ __launch_bounds__(64) void k_dense_sum_w(const double *cS, ChainGeom g, int t_first, int t_end, int T, CollapseState *st, double threshold,
                                                    double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp)
{
    using G = DenseW<S>;
    constexpr int R1 = G::R1, P1 = G::P1, R2 = G::R2, P2 = G::P2, PF = G::PF, PD = G::PD;
    HIP_DYNAMIC_SHARED(double, lds)
    if (!sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    const int lane = threadIdx.x;
    // workgroups are dealt to the 8 XCDs round robin (each with its own L2): XCD x takes the x-th EIGHTH of the tiles, so that the
    // tiles a CU's neighbours work on -- whose footprints overlap this one's -- are cached in the same L2
    const int ntiles_ = g.tiles_x * g.tiles_y;
    const int tile = dense_tile_of_block((int)blockIdx.x, ntiles_);
    if (tile >= ntiles_) return;   // (the grid is rounded up to a multiple of 8)
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && lane == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    const int H0 = g.h[0], W0 = g.w[0], sh1 = g.h[1], sw1 = g.w[1];
    const int yv1 = 8 * ty - 1, xv1 = 32 * tx - 1;   // first virtual row / column of the level-1 footprint
    // ---- staged elements of this lane (position fixed for all frames): virtual rows / columns resolved to addresses here
    const int hS = g.h[S], wS = g.w[S];
    const size_t fs = (size_t)hS * wS;
    int off_g[PF], off_l[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int i = lane + 64 * p;
        const int pitch = S == 2 ? P2 : P1;
        const int r = i / pitch, c = i - r * pitch;
        const int yv = (S == 2 ? 4 * ty - 1 : yv1) + r, xv = (S == 2 ? 16 * tx - 2 : xv1) + c;
        const int ya = up_virtual_row(yv, hS), xa = min(max(xv, 0), wS - 1);
        off_g[p] = ya * wS + xa;
        off_l[p] = i < G::NST ? i : -1;
    }
    // ---- level 2 -> 1: lane c < 34 owns level-1 column xv1 + c
    double hw_a = 0.0, hw_b = 0.0, hw_c = 0.0;
    int h_base = 0;
    if (S == 2) {
        const int sw2 = g.w[2];
        const int xv = xv1 + (lane < P1 ? lane : 0);
        if (lane < P1 && xv >= 0 && xv < sw1) {
            const int j = xv >> 1;
            const bool left = j == 0, right = j == sw2 - 1;
            if (xv & 1) { hw_b = right ? 8.0 : 4.0; hw_c = right ? 0.0 : 4.0; }
            else { hw_a = left ? 0.0 : 1.0; hw_b = right ? 7.0 : 6.0; hw_c = left ? 2.0 : (right ? 0.0 : 1.0); }
        }
        // taps j - 1, j, j + 1 of staged row q sit at h_base + q P2 + {0, 1, 2}: virtual column (xv >> 1) - 1 - (16 tx - 2)
        h_base = ((xv >> 1) - 1) - (16 * tx - 2);
        h_base = min(max(h_base, 0), P2 - 3);   // (lanes >= 34 and out-of-image columns: any valid address, weights are zero)
    }
    // ---- level 1 -> 0: lane = (column pair cp, row half rh): columns X, X + 1, rows Y0 .. Y0 + 7
    const int cp = lane & 31, rh = lane >> 5;
    const int X = 64 * tx + 2 * cp, Y0 = 16 * ty + 8 * rh;
    double we_a, we_b, we_c, wo_b, wo_c;
    {
        const int j = X >> 1;                       // = 32 tx + cp
        const bool left = j == 0, right = j >= sw1 - 1;
        we_a = left ? 0.0 : 1.0; we_b = right ? 7.0 : 6.0; we_c = left ? 2.0 : (right ? 0.0 : 1.0);
        wo_b = right ? 8.0 : 4.0; wo_c = right ? 0.0 : 4.0;
    }
    const int l0off = G::L1_OFF + (4 * rh) * P1 + cp;   // taps of source row k: slice[l0off + k P1 + {0, 1, 2}]
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    constexpr int NB = PD * FR;   // frames in flight (registers): trip i uses buffers 0 .. NB - 1 in frame order
    double nxt[NB][PF];
    auto fetch = [&](int d, int t) __attribute__((always_inline)) {
        const double *src = cS + (size_t)sym_frame(t < t_end ? t : t_first, T) * fs;
#pragma unroll
        for (int p = 0; p < PF; ++p) nxt[d][p] = src[off_g[p]];
    };
#pragma unroll
    for (int d = 0; d < NB; ++d) fetch(d, t_first + d);
    for (int tb = t_first; tb < t_end; tb += NB) {
#pragma unroll
        for (int d0 = 0; d0 < NB; d0 += FR) {
            if (tb + d0 >= t_end) break;   // (uniform)
            wave_sync();                   // the previous trip's reads of these slices are behind us
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                double *sl = lds + f * G::TOTAL;
#pragma unroll
                for (int p = 0; p < PF; ++p) if (off_l[p] >= 0) sl[off_l[p]] = nxt[d0 + f][p];
                fetch(d0 + f, tb + d0 + f + NB);
            }
            wave_sync();
            if (S == 2) {
#pragma unroll
                for (int f = 0; f < FR; ++f) {
                    double *sl = lds + f * G::TOTAL, *l1 = sl + G::L1_OFF;
                    // horizontal values of the 7 staged rows at this lane's level-1 column, in registers
                    double hq[R2];
#pragma unroll
                    for (int q = 0; q < R2; ++q) {
                        const double *row = sl + h_base + q * P2;
                        hq[q] = dw_tap3(row[0], row[1], row[2], hw_a, hw_b, hw_c);
                    }
                    // level-1 rows p = 0 .. 9 <-> virtual rows 8 ty - 1 + p: p even is an odd row (values of level-2 rows 4 ty - 1 + p / 2
                    // and the next one), p odd an even row (the three rows around 4 ty + (p - 1) / 2); hq[q] <-> level-2 row 4 ty - 1 + q
                    double prev = 0.0;
#pragma unroll
                    for (int p = 0; p < R1; ++p) {
                        const int q = p >> 1;
                        double v = (p & 1) ? (hq[q] + hq[q + 1] * 6 + hq[q + 2]) * (1.0 / 64) : (hq[q] + hq[q + 1]) * (1.0 / 16);
                        if (yv1 + p > sh1 - 1) v = prev;   // (uniform) virtual row past the bottom: the last row again (up_at()'s r2)
                        prev = v;
                        if (lane < P1) l1[p * P1 + lane] = v;
                    }
                }
                wave_sync();
            }
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                if (tb + d0 + f >= t_end) break;   // (uniform: a trip past the last frame computed on a repeated frame, nothing is added)
                const double *l0src = lds + f * G::TOTAL + l0off;
                double hve[6], hvo[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double *row = l0src + k * P1;
                    const double a = row[0], b = row[1], c = row[2];
                    hve[k] = dw_tap3(a, b, c, we_a, we_b, we_c);
                    hvo[k] = __builtin_fma(c, wo_c, b * wo_b);   // b * 4 + c * 4 (or b * 8 + c * 0): both products exact
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double e0 = (hve[m] + hve[m + 1] * 6 + hve[m + 2]) * (1.0 / 64), e1 = (hve[m + 1] + hve[m + 2]) * (1.0 / 16);
                    const double o0 = (hvo[m] + hvo[m + 1] * 6 + hvo[m + 2]) * (1.0 / 64), o1 = (hvo[m + 1] + hvo[m + 2]) * (1.0 / 16);
                    acc[2 * m] = acc[2 * m] + ((e0 >= top) ? min_val : e0);
                    acc[2 * m + 1] = acc[2 * m + 1] + ((e1 >= top) ? min_val : e1);
                    acc[8 + 2 * m] = acc[8 + 2 * m] + ((o0 >= top) ? min_val : o0);
                    acc[8 + 2 * m + 1] = acc[8 + 2 * m + 1] + ((o1 >= top) ? min_val : o1);
                }
            }
        }
    }
    // base.py:562: np.average = sum / T when the whole buffer was summed here; the heatmap's extrema for base.py:563
    const double cnt = (double)avg_T;
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = Y0 + r;
        if (y < H0) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (X + o < W0) {
                    const double a = acc[8 * o + r];
                    const double v = avg_T > 0 ? a / cnt : a;
                    heat_sum[(size_t)y * W0 + X + o] = v;
                    hmn = (v < hmn) ? v : hmn; hmx = (v > hmx) ? v : hmx;
                }
            }
        }
    }
    if (tile_nkept && lane == 0) tile_nkept[tile] = t_end - t_first;   // (the sparse heatmap exchange: no tile is known to be the constant)
    if (avg_T > 0) {
        hmn = wave_min(hmn); hmx = wave_max(hmx);
        if (lane == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp_ = blockIdx.x & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
        }
    }
}

// The same sum with NW waves per tile when an image has fewer tiles than the chip has SIMDs (720p: 900 tiles on 1 024 SIMDs -- a
// lone wave per tile is bound by the latency of its own dependency chain, and most of the VALU slots of its SIMD stay empty).
// The time sum is a chain in t, but only its ADDITIONS are: round r takes frames t0 + r NW .. + NW - 1, wave w evaluates frame
// t0 + r NW + w (the whole pyrUp chain of the tile, as in k_dense_sum_w) and parks its 16 masked values per lane in LDS; after a
// barrier wave w adds the NW frames, in frame order, to ITS 16 / NW running sums per lane.  Same values, same order of additions:
// bit-identical to k_dense_sum_w.
// Round 6: the rounds walk the tile's KEPT frames only.  With the level-1 bounds of rm_bounds_l1.h the second look with the exact
// `top` (slot_of / lo given) drops a third of the pairs of the 720p stream -- and a round of NW kept frames is as balanced as a round of
// NW consecutive ones (round 5 tried the same with the level-2 bounds, which drop nothing: 115 against 107 us).  The frames in between
// add `min`, in order, in front of the next kept frame.  slot_of == nullptr: every frame is kept (the exhaustive form).
__device__ __forceinline__ double masked_gap_small(double acc, int n, double min_val)   // n sequential additions of `min`
{
    for (; n > 0; --n) acc = acc + min_val;
    return acc;
}

template <int S, int NW>
__global__ __launch_bounds__(64 * NW) void k_dense_sum_wf(const double *cS, ChainGeom g, int t_first, int t_end, int T, CollapseState *st, double threshold,
                                                          double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp, const int *slot_of, const double *lo)
{
    using G = DenseW<S>;
    constexpr int R1 = G::R1, P1 = G::P1, R2 = G::R2, P2 = G::P2, PF = G::PF, PD = G::PD;
    constexpr int QA = 16 / NW;   // running sums per lane and wave
    HIP_DYNAMIC_SHARED(double, lds)
    if (!sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exchange: [NW frames][16 values][64 lanes]; a wave's footprint slice overlays ITS frame's part of the exchange (the slice is
    // dead once the wave has its level-1 taps in registers, and DS operations of one wave execute in order): 8 KB per wave
    static_assert(G::TOTAL <= 16 * 64, "the footprint slice must fit the wave's part of the exchange");
    double *ex = lds;
    double *sl = lds + (size_t)wave * 16 * 64;
    unsigned short *s_list = reinterpret_cast<unsigned short *>(lds + (size_t)NW * 16 * 64);   // the tile's kept frames of [t_first, t_end) in time order
    __shared__ int s_wcnt[NW];
    // workgroups are dealt to the 8 XCDs round robin (each with its own L2): XCD x takes the x-th EIGHTH of the tiles, so that the
    // tiles a CU's neighbours work on -- whose footprints overlap this one's -- are cached in the same L2
    const int ntiles_ = g.tiles_x * g.tiles_y;
    const int tile = dense_tile_of_block((int)blockIdx.x, ntiles_);
    if (tile >= ntiles_) return;   // (the grid is rounded up to a multiple of 8)
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    const int H0 = g.h[0], W0 = g.w[0], sh1 = g.h[1], sw1 = g.w[1];
    const int yv1 = 8 * ty - 1, xv1 = 32 * tx - 1;
    const int hS = g.h[S], wS = g.w[S];
    const size_t fs = (size_t)hS * wS;
    // the kept frames: what the selection kept and the exact top does not clear (a pair whose lower bound clears it adds `min` to
    // every pixel, evaluated or not)
    int nlist = 0;
    {
        const int Th = sym_frames(T);
        const double margin = st->margin;
        for (int c0 = t_first; c0 < t_end; c0 += 64 * NW) {
            const int t = c0 + (int)threadIdx.x;
            bool kept = t < t_end;
            if (kept && slot_of) {
                const int u = sym_frame(t, T);
                kept = slot_of[slot_index(u, tile, Th)] != SLOT_PRUNED;
                if (kept && lo) kept = !(lo[(size_t)u * ntiles_ + tile] - margin >= top);   // (a NaN bound keeps the pair: NaN must reach the sum)
            }
            const unsigned long long mk = __ballot(kept);
            if (lane == 0) s_wcnt[wave] = __popcll(mk);
            __syncthreads();
            int off = nlist, tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const int c = s_wcnt[w]; off += (w < wave) ? c : 0; tot += c; }
            if (kept) s_list[off + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)t;
            nlist += tot;
            __syncthreads();
        }
    }
    int off_g[PF], off_l[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int i = lane + 64 * p;
        const int pitch = S == 2 ? P2 : P1;
        const int r = i / pitch, c = i - r * pitch;
        const int yv = (S == 2 ? 4 * ty - 1 : yv1) + r, xv = (S == 2 ? 16 * tx - 2 : xv1) + c;
        const int ya = up_virtual_row(yv, hS), xa = min(max(xv, 0), wS - 1);
        off_g[p] = ya * wS + xa;
        off_l[p] = i < G::NST ? i : -1;
    }
    double hw_a = 0.0, hw_b = 0.0, hw_c = 0.0;
    int h_base = 0;
    if (S == 2) {
        const int sw2 = g.w[2];
        const int xv = xv1 + (lane < P1 ? lane : 0);
        if (lane < P1 && xv >= 0 && xv < sw1) {
            const int j = xv >> 1;
            const bool left = j == 0, right = j == sw2 - 1;
            if (xv & 1) { hw_b = right ? 8.0 : 4.0; hw_c = right ? 0.0 : 4.0; }
            else { hw_a = left ? 0.0 : 1.0; hw_b = right ? 7.0 : 6.0; hw_c = left ? 2.0 : (right ? 0.0 : 1.0); }
        }
        h_base = ((xv >> 1) - 1) - (16 * tx - 2);
        h_base = min(max(h_base, 0), P2 - 3);
    }
    const int cp = lane & 31, rh = lane >> 5;
    const int X = 64 * tx + 2 * cp, Y0 = 16 * ty + 8 * rh;
    double we_a, we_b, we_c, wo_b, wo_c;
    {
        const int j = X >> 1;
        const bool left = j == 0, right = j >= sw1 - 1;
        we_a = left ? 0.0 : 1.0; we_b = right ? 7.0 : 6.0; we_c = left ? 2.0 : (right ? 0.0 : 1.0);
        wo_b = right ? 8.0 : 4.0; wo_c = right ? 0.0 : 4.0;
    }
    const int l0off = G::L1_OFF + (4 * rh) * P1 + cp;
    double acc[QA];
#pragma unroll
    for (int j = 0; j < QA; ++j) acc[j] = 0.0;
    double nxt[PD][PF];   // this wave's frames of the next PD rounds
    auto fetch = [&](int d, int i) __attribute__((always_inline)) {
        const int t = (int)s_list[min(i, max(nlist - 1, 0))];   // (an entry past the end: a repeated frame nobody adds)
        const double *src = cS + (size_t)sym_frame(nlist > 0 ? t : t_first, T) * fs;
#pragma unroll
        for (int p = 0; p < PF; ++p) nxt[d][p] = src[off_g[p]];
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch(d, d * NW + wave);
    int t_done = t_first;
    for (int ib = 0; ib < nlist; ib += NW * PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int i0 = ib + d * NW;      // first list entry of this round
            if (i0 >= nlist) break;          // (uniform over the workgroup)
#pragma unroll
            for (int p = 0; p < PF; ++p) if (off_l[p] >= 0) sl[off_l[p]] = nxt[d][p];
            fetch(d, i0 + wave + NW * PD);
            wave_sync();
            if (S == 2) {
                double *l1 = sl + G::L1_OFF;
                double hq[R2];
#pragma unroll
                for (int q = 0; q < R2; ++q) {
                    const double *row = sl + h_base + q * P2;
                    hq[q] = dw_tap3(row[0], row[1], row[2], hw_a, hw_b, hw_c);
                }
                double prev = 0.0;
#pragma unroll
                for (int p = 0; p < R1; ++p) {
                    const int q = p >> 1;
                    double v = (p & 1) ? (hq[q] + hq[q + 1] * 6 + hq[q + 2]) * (1.0 / 64) : (hq[q] + hq[q + 1]) * (1.0 / 16);
                    if (yv1 + p > sh1 - 1) v = prev;
                    prev = v;
                    if (lane < P1) l1[p * P1 + lane] = v;
                }
                wave_sync();
            }
            {
                const double *l0src = sl + l0off;
                double hve[6], hvo[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double *row = l0src + k * P1;
                    const double a = row[0], b = row[1], c = row[2];
                    hve[k] = dw_tap3(a, b, c, we_a, we_b, we_c);
                    hvo[k] = __builtin_fma(c, wo_c, b * wo_b);
                }
                wave_sync();   // every lane has its taps: the slice may be overwritten (a no-op for the lockstep hardware wave)
                double *exw = ex + (size_t)wave * 16 * 64 + lane;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double e0 = (hve[m] + hve[m + 1] * 6 + hve[m + 2]) * (1.0 / 64), e1 = (hve[m + 1] + hve[m + 2]) * (1.0 / 16);
                    const double o0 = (hvo[m] + hvo[m + 1] * 6 + hvo[m + 2]) * (1.0 / 64), o1 = (hvo[m + 1] + hvo[m + 2]) * (1.0 / 16);
                    exw[(2 * m) * 64] = (e0 >= top) ? min_val : e0;
                    exw[(2 * m + 1) * 64] = (e1 >= top) ? min_val : e1;
                    exw[(8 + 2 * m) * 64] = (o0 >= top) ? min_val : o0;
                    exw[(8 + 2 * m + 1) * 64] = (o1 >= top) ? min_val : o1;
                }
            }
            __syncthreads();
            const int nf = min(NW, nlist - i0);   // frames of this round that exist
#pragma unroll
            for (int f = 0; f < NW; ++f) {
                if (f < nf) {
                    const int t = uniform((int)s_list[i0 + f]);
                    const int ngap = t - t_done;       // the frames in front of this one that are not kept: `min` each
                    t_done = t + 1;
                    const double *exf = ex + (size_t)f * 16 * 64 + (size_t)(wave * QA) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < QA; ++q) acc[q] = masked_gap_small(acc[q], ngap, min_val) + exf[q * 64];
                }
            }
            __syncthreads();   // (the exchange and the slices are rewritten next round)
        }
    }
#pragma unroll
    for (int q = 0; q < QA; ++q) acc[q] = masked_gap_small(acc[q], t_end - t_done, min_val);
    const double cnt = (double)avg_T;
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int j = wave * QA + q, o = j >> 3, r = j & 7;
        const int y = Y0 + r;
        if (y < H0 && X + o < W0) {
            const double v = avg_T > 0 ? acc[q] / cnt : acc[q];
            heat_sum[(size_t)y * W0 + X + o] = v;
            hmn = (v < hmn) ? v : hmn; hmx = (v > hmx) ? v : hmx;
        }
    }
    if (tile_nkept && threadIdx.x == 0) tile_nkept[tile] = nlist;   // 0: every pixel of the tile is the same constant (sparse heatmap exchange, ROI stage)
    if (avg_T > 0) {
        hmn = wave_min(hmn); hmx = wave_max(hmx);
        if (lane == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp_ = (blockIdx.x * NW + wave) & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
        }
    }
}

}  // namespace rm
