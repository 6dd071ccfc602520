// respmon_amd/csrc/rm_xstore.h -- the masked time sum of a DENSE selection through an exception store (round 6)
//
//   heat = (1 / T) sum_t (raw[t] >= top ? min : raw[t]),  t = 0 .. T - 1 in order           (transforms.py:184-192, base.py:562)
//
// The store-less sum kernels (rm_dense_sum.h k_dense_sum_w / wf, rm_tile_eval.h k_dense_sum_t) own a tile for ALL frames and evaluate
// its pyrUp chain frame after frame, in time order -- every unique frame TWICE (the band-passed signal is even in time: frame T - u is
// frame u again), one wave per tile however few of its frames matter, and at 4K x 512 1.4 - 1.8 ms of fp64 issue.  What the sum needs
// of a (tile, frame) pair is very little: on the streams measured 0.6 % of the full-resolution values lie below `top` (720p x 128:
// 45 % of the pairs hold one at all, 14 of 1024 values on average) -- everything else adds `min`.  So the work is split:
//
//   k_xs_eval<S>   a FLAT pass over the pairs the selection kept (k_select_pairs' lists: perfectly balanced, every SIMD busy whatever tile
//                  the pairs belong to), each pair ONCE: exact `top` first (a pair whose level-1 lower bound clears it adds `min`
//                  everywhere: no evaluation), TileEval of the others (rm_tile_eval.h: the same values bit for bit), and the EXCEPTIONS
//                  -- the values that are not >= top -- leave as a record: 16 lane masks (value j of lane l is an exception), 16 running
//                  counts, the values themselves compacted in (j, lane) order.  Records are carved from chunks a wave reserves with
//                  one atomic; a table entry per pair says where its record is and which of its 16 masks are not empty.
//   k_xs_sum<NW>   the time-ordered additions: NW waves per tile, wave w owns the running sums j = w 16 / NW .. of every lane, walks
//                  the tile's records up (t = u) and down (t = T - u) -- the second visit of a pair costs a 512-byte load --, adds
//                  `min` lazily (one counter per running sum: the additions happen, in order, in front of the next exception) and the
//                  exception values where a lane's bit is set.  Same additions in the same order as every other sum kernel: bit-identical.
//
// 4K x 512 (fp16 buffer, skip 2): k_dense_sum_t 1.41 ms -> eval + sum.  A selection whose exceptions overflow the store
// (capacity: rm_collapse_sum.hip) raises a flag in the state and the store-less kernel enqueued behind k_xs_sum takes the sum.
#pragma once

namespace rm {

constexpr unsigned XS_NONE = 0xffffffffu;
struct alignas(8) XsEntry { unsigned off; unsigned nz; };   // off: the pair's record, in 8-byte words from the start of the store (XS_NONE: no record -- every value of the pair adds `min`); nz: bit j = mask j of the record is not empty
constexpr int XS_HDR = 24;             // words of a record in front of its values: 16 masks, then 16 counts (exceptions in masks 0 .. j - 1) as 32-bit halves of 8 words
constexpr int XS_HEAD_VALUES = 64 - XS_HDR;   // values that travel with the header in the one 512-byte load of a visit
constexpr unsigned XS_CHUNK = 2048;    // words a wave reserves at a time (16 KB)

struct XsPlan {
    unsigned long long *store;         // records
    XsEntry *tab;                      // [tile][unique frame] (slot_index): written XS_NONE by k_select_pairs, then by k_xs_eval
    unsigned long long cap_words;      // capacity of the store
};

// 64-bit readlane (lane index wave-uniform)
__device__ __forceinline__ unsigned long long xs_readlane64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// ---- evaluation: one wave per listed pair, exceptions into the store -----------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(64) void k_xs_eval(const double *cS, ChainGeom g, int ntiles, const unsigned int *list_a, const unsigned int *list_b,
                                                const int *slot_of, const double *lo, CollapseState *st, double threshold, SumPlan sp, int Th,
                                                XsPlan xp, int only_if_dense)
{
    using F = TileFoot<S, false>;
    HIP_DYNAMIC_SHARED(double, lds)
    const int lane = threadIdx.x;
    if (only_if_dense && !sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    const unsigned nA = st->n_list_a, nB = st->n_list_b, n = nA + nB;
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    const double margin = st->margin;
    const size_t fs = (size_t)g.h[S] * g.w[S];
    const int H0 = g.h[0], W0 = g.w[0];
    // (uniform) this wave's chunk of the store: words [cur, end).  The first chunk of every wave is its own by position -- no atomic: the
    // 7 000 waves of a 720p launch all asking ONE counter for their first chunk took 100 us (same-address atomics serialise) --, the
    // counter hands out what lies behind those
    const unsigned long long chunks0 = (unsigned long long)gridDim.x * XS_CHUNK;
    unsigned long long cur = (unsigned long long)blockIdx.x * XS_CHUNK, end = cur + XS_CHUNK;
    if (end > xp.cap_words) { cur = 0; end = 0; }
    auto stage_offsets = [&](unsigned idx, int (&off)[F::PF]) __attribute__((always_inline)) {   // tile_setup()'s staging part alone
        const int u = (int)(idx / (unsigned)ntiles), tile = (int)(idx - (unsigned)u * (unsigned)ntiles);
        const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
        const int hS = g.h[S], wS = g.w[S];
        const int fy = ((16 * ty) >> S) - 1, fx = ((64 * tx) >> S) - 1;
#pragma unroll
        for (int p = 0; p < F::PF; ++p) {
            const int i = min(lane + 64 * p, F::NST - 1);
            const int r = i / F::nc(S), c = i - r * F::nc(S);
            const int yv = fy + r, xv = fx + c;
            const int ya = yv < 0 ? 1 : (yv > hS - 1 ? hS - 1 : yv), xa = min(max(xv, 0), wS - 1);
            off[p] = ya * wS + xa;
        }
        return cS + (size_t)u * fs;
    };
    for (unsigned c0 = blockIdx.x; c0 < n; c0 += gridDim.x * 64u) {
        // 64 listed pairs at a time, gridDim.x apart (every wave a sample of the whole list: kept and pruned pairs, light and heavy
        // tiles alike): which of them can hold a value below the exact top at all?
        const unsigned c = c0 + gridDim.x * (unsigned)lane;
        const unsigned my_idx = c < nA ? list_a[c] : (c < n ? list_b[c - nA] : 0u);
        bool kept = c < n;
        if (kept) {
            const int u = (int)(my_idx / (unsigned)ntiles), tile = (int)(my_idx - (unsigned)u * (unsigned)ntiles);
            kept = slot_of[slot_index(u, tile, Th)] != SLOT_PRUNED;               // (a C pair that is not a D pair: evaluated for the extrema only)
            if (kept && lo) kept = !(lo[my_idx] - margin >= top);                 // (a NaN bound keeps the pair: NaN must reach the sum)
        }
        unsigned long long todo = __ballot(kept);
        // the footprint of the NEXT kept pair travels while this one is evaluated
        double nxt[F::PF];
#pragma unroll
        for (int p = 0; p < F::PF; ++p) nxt[p] = 0.0;
        if (todo) {
            int off[F::PF];
            const double *src = stage_offsets((unsigned)__builtin_amdgcn_readlane((int)my_idx, (int)__builtin_ctzll(todo)), off);
#pragma unroll
            for (int p = 0; p < F::PF; ++p) nxt[p] = src[off[p]];
        }
        while (todo) {   // (uniform)
            const int b = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const unsigned idx = (unsigned)__builtin_amdgcn_readlane((int)my_idx, b);
            const int u = (int)(idx / (unsigned)ntiles), tile = (int)(idx - (unsigned)u * (unsigned)ntiles);
            const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
            TileSetup<S, false> ts;
            tile_setup<S, false>(g, tx, ty, 0, lane, ts);
            wave_sync();   // the previous pair's reads of the slice are behind us
#pragma unroll
            for (int p = 0; p < F::PF; ++p) if (lane + 64 * p < F::NST) lds[F::off(S) + lane + 64 * p] = nxt[p];
            if (todo) {
                int off[F::PF];
                const double *src = stage_offsets((unsigned)__builtin_amdgcn_readlane((int)my_idx, (int)__builtin_ctzll(todo)), off);
#pragma unroll
                for (int p = 0; p < F::PF; ++p) nxt[p] = src[off[p]];
            }
            wave_sync();
            double v[16];
            tile_eval<S, false>(ts, lds, lane, v);
            // exceptions: the values that do NOT mask (raw >= top is false: below top, or NaN), inside the image
            const bool ragged = 16 * ty + CT_H > H0 || 64 * tx + CT_W > W0;   // (uniform: the tile sticks out of the image)
            unsigned long long m[16];
            unsigned nz = 0, total = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                bool ex = !(v[j] >= top);
                if (ragged) ex = ex && ts.Y0 + (j & 7) < H0 && ts.X + (j >> 3) < W0;
                m[j] = __ballot(ex);
                nz |= (m[j] != 0ull ? 1u : 0u) << j;
                total += (unsigned)__popcll(m[j]);
            }
            if (total == 0) continue;   // (the table says XS_NONE already: k_select_pairs)
            const unsigned need = (unsigned)XS_HDR + total;
            unsigned long long at;
            if (need > XS_CHUNK / 4) {   // a large record: its own reservation (the chunk keeps serving the small ones: at most a quarter of a chunk is ever left unused)
                unsigned long long got = 0;
                if (lane == 0) got = atomicAdd(&st->xs_next, (unsigned long long)need);
                at = chunks0 + xs_readlane64(got, 0);
                if (at + need > xp.cap_words) { if (lane == 0) st->xs_overflow = 1u; return; }
            } else {
                if (cur + need > end) {
                    unsigned long long got = 0;
                    if (lane == 0) got = atomicAdd(&st->xs_next, (unsigned long long)XS_CHUNK);
                    cur = chunks0 + xs_readlane64(got, 0); end = cur + XS_CHUNK;
                    if (end > xp.cap_words) {   // (uniform) the store is full: the stand-in behind k_xs_sum takes the sum
                        if (lane == 0) st->xs_overflow = 1u;
                        return;
                    }
                }
                at = cur; cur += need;
            }
            unsigned long long *rec = xp.store + at;
            // header: lanes 0 .. 15 hold the masks, lanes 16 .. 23 the running counts (two 32-bit halves each)
            unsigned long long hv = 0;
            unsigned base = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (lane == j) hv = m[j];
                if (lane == 16 + (j >> 1)) hv |= (unsigned long long)base << (32 * (j & 1));
                base += (unsigned)__popcll(m[j]);
            }
            if (lane < XS_HDR) rec[lane] = hv;
            // values, compacted in (j, lane) order
            const unsigned long long below = (1ull << lane) - 1ull;
            base = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (m[j] == 0ull) continue;   // (uniform)
                if ((m[j] >> lane) & 1ull) rec[XS_HDR + base + (unsigned)__popcll(m[j] & below)] = __double_as_longlong(v[j]);
                base += (unsigned)__popcll(m[j]);
            }
            if (lane == 0) xp.tab[slot_index(u, tile, Th)] = XsEntry{(unsigned)at, nz};
        }
    }
}

// ---- the time-ordered additions ----------------------------------------------------------------------------------------------------------
// Dynamic LDS: the tile's records in frame order -- s_u[Th], s_off[Th], s_nz[Th].
template <int NW>
__global__ __launch_bounds__(64 * NW) RM_WAVES_PER_EU(8) void k_xs_sum(ChainGeom g, int t_first, int t_end, int T, int ntiles, CollapseState *st, double threshold,
                                                    double *heat_sum, int avg_T, int *tile_nkept, SumPlan sp, XsPlan xp, int only_if_dense, int *ran_host)
{
    constexpr int QA = 16 / NW;   // running sums per lane and wave
    HIP_DYNAMIC_SHARED(int, s_u)
    __shared__ int s_wcnt[NW];
    const int Th = sym_frames(T);
    unsigned *s_off = reinterpret_cast<unsigned *>(s_u + Th), *s_nz = s_off + Th;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    if (only_if_dense && !sum_is_dense(st, sp)) return;   // (uniform over the grid: the sparse path took the sum)
    if (st->xs_overflow) return;                          // (uniform over the grid: the exception store was too small -- the stand-in takes the sum)
    if (ran_host && blockIdx.x == 0 && tid == 0) *ran_host = 2;   // (pinned: tells rm_locate that the stand-in it enqueued on a hint was needed)
    const int tile = dense_tile_of_block((int)blockIdx.x, ntiles);   // XCD x takes the x-th eighth of the tiles (rm_dense_sum.h)
    if (tile >= ntiles) return;
    const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
    const double min_val = f64_unkey(fold_min_keys(st->min_keys, st->min_key)), max_val = f64_unkey(fold_max_keys(st->max_keys, st->max_key));
    const double top = max_val - (max_val - min_val) * threshold;   // transforms.py:184-189
    if (blockIdx.x == 0 && tid == 0) { st->min_val = min_val; st->max_val = max_val; st->top = top; }
    // the tile's records, ascending in u (ballot + prefix popcount, 64 NW unique frames per round)
    int m = 0;
    for (int c0 = 0; c0 < Th; c0 += 64 * NW) {
        const int u = c0 + tid;
        XsEntry e{XS_NONE, 0u};
        if (u < Th && sym_in_range(u, T, t_first, t_end)) e = xp.tab[slot_index(u, tile, Th)];   // (a frame shard: only its own pairs were selected and written)
        const bool have = e.off != XS_NONE;
        const unsigned long long mk = __ballot(have);
        int off = m, tot;
        if (NW > 1) {
            if (lane == 0) s_wcnt[wave] = __popcll(mk);
            __syncthreads();
            tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const int c = s_wcnt[w]; off += (w < wave) ? c : 0; tot += c; }
        } else tot = __popcll(mk);
        if (have) { const int pos = off + __popcll(mk & ((1ull << lane) - 1ull)); s_u[pos] = u; s_off[pos] = e.off; s_nz[pos] = e.nz; }
        m += tot;
        if (NW > 1) __syncthreads(); else wave_sync();
    }
    // lane = (column pair cp, row group rg): value j = 8 o + r of a record is pixel (Y0 + r, X + o)   (rm_tile_eval.h tile_setup)
    const int X = 64 * tx + 2 * (lane & 31), Y0 = 16 * ty + 8 * (lane >> 5);
    const int H0 = g.h[0], W0 = g.w[0];
    const int j0 = wave * QA;
    const unsigned my_nz = ((1u << QA) - 1u) << j0;
    const unsigned long long my_bit = 1ull << lane, below = my_bit - 1ull;
    double acc[QA];
    int ap[QA];       // (uniform) frames of [t_first, t_end) whose contribution acc[q] already holds: [t_first, ap[q])
#pragma unroll
    for (int q = 0; q < QA; ++q) { acc[q] = 0.0; ap[q] = t_first; }
    int nvis = 0;     // visits of this tile's records inside [t_first, t_end): the kept frames in time order
    const int u_down = (T + 1) / 2 - 1;   // the way down starts at t = T / 2 + 1, i.e. u = T - t = u_down, and ends at u = 1
    // A visit is ONE 512-byte load (the record's header and its first 40 values) and a few additions: what it costs is the load's
    // latency.  XS_PF loads are in flight per wave -- the records of the next XS_PF visits, in visit order (the list is known up
    // front); the ring is indexed statically (the visit loops are unrolled XS_PF times).
    auto head = [&](int i) __attribute__((always_inline)) -> unsigned long long {
        if (i < 0 || i >= m) return 0ull;   // (uniform)
        const unsigned off = (unsigned)uniform((int)s_off[i]);
        return xp.store[(size_t)off + lane];   // (the store is allocated 64 words longer than its capacity: a short record at its very end)
    };
    auto visit = [&](int i, int t, unsigned long long data) __attribute__((always_inline)) {
        const unsigned nz = (unsigned)uniform((int)s_nz[i]) & my_nz;
        if (nz == 0u) return;   // (uniform) every value of this wave's running sums adds `min`: later, lazily
        const unsigned off = (unsigned)uniform((int)s_off[i]);
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int j = j0 + q;
            if (!((nz >> j) & 1u)) continue;   // (uniform)
            const unsigned long long mj = xs_readlane64(data, j);
            const unsigned long long cw = xs_readlane64(data, 16 + (j >> 1));
            const unsigned base = (unsigned)(cw >> (32 * (j & 1)));
            const unsigned pos = base + (unsigned)__popcll(mj & below);
            const bool mine = (mj & my_bit) != 0ull;
            unsigned long long bits;
            if (base + (unsigned)__popcll(mj) <= (unsigned)XS_HEAD_VALUES) {   // (uniform) the values arrived with the header
                const int src = (int)(XS_HDR + pos) & 63;
                const unsigned blo = (unsigned)__shfl((int)(unsigned)data, src), bhi = (unsigned)__shfl((int)(unsigned)(data >> 32), src);
                bits = ((unsigned long long)bhi << 32) | blo;
            } else {
                bits = mine ? xp.store[(size_t)off + XS_HDR + pos] : 0ull;
            }
            const double val = mine ? __longlong_as_double((long long)bits) : min_val;
            acc[q] = masked_gap(acc[q], t - ap[q], min_val);   // frames [ap, t) added `min` to this running sum
            acc[q] = acc[q] + ((val >= top) ? min_val : val);   // (an exception is not >= top; the select keeps the expression every sum kernel uses)
            ap[q] = t + 1;
        }
    };
    constexpr int XS_PF = NW == 1 ? 5 : 8;   // (one wave per tile: 64 registers keep eight waves per SIMD -- every tile of a 4K frame resident at once)
    unsigned long long ring[XS_PF];
    // the way up: t = u
#pragma unroll
    for (int k = 0; k < XS_PF; ++k) ring[k] = head(k);
    for (int i0 = 0; i0 < m; i0 += XS_PF) {
#pragma unroll
        for (int k = 0; k < XS_PF; ++k) {
            const int i = i0 + k;
            if (i >= m) break;   // (uniform)
            const unsigned long long data = ring[k];
            ring[k] = head(i + XS_PF);
            const int t = uniform(s_u[i]);
            if (t >= t_first && t < t_end) { ++nvis; visit(i, t, data); }
        }
    }
    // the way down: t = T - u for u in [1, u_down], largest u first
#pragma unroll
    for (int k = 0; k < XS_PF; ++k) ring[k] = head(m - 1 - k);
    for (int i0 = m - 1; i0 >= 0; i0 -= XS_PF) {
#pragma unroll
        for (int k = 0; k < XS_PF; ++k) {
            const int i = i0 - k;
            if (i < 0) break;   // (uniform)
            const unsigned long long data = ring[k];
            ring[k] = head(i - XS_PF);
            const int u = uniform(s_u[i]), t = T - u;
            if (u >= 1 && u <= u_down && t >= t_first && t < t_end) { ++nvis; visit(i, t, data); }
        }
    }
    // base.py:562: np.average = sum / T when the whole buffer was summed here; the heatmap's extrema for base.py:563
    const double cnt = (double)avg_T;
    double hmn = __builtin_huge_val(), hmx = -__builtin_huge_val();
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int j = j0 + q, o = j >> 3, r = j & 7;
        const double a = masked_gap(acc[q], t_end - ap[q], min_val);
        const int y = Y0 + r, x = X + o;
        if (y < H0 && x < W0) {
            const double v = avg_T > 0 ? a / cnt : a;
            heat_sum[(size_t)y * W0 + x] = v;
            hmn = (v < hmn) ? v : hmn; hmx = (v > hmx) ? v : hmx;
        }
    }
    if (tile_nkept && tid == 0) tile_nkept[tile] = nvis;   // 0: every pixel of the tile is the same constant (sparse heatmap exchange, ROI stage)
    if (avg_T > 0) {
        hmn = wave_min(hmn); hmx = wave_max(hmx);
        if (lane == 0) {
            const unsigned long long kmn = f64_key(hmn), kmx = f64_key(hmx);
            const int sp_ = ((int)blockIdx.x * NW + wave) & (NSTRIPE - 1);
            striped_min_max(st->heat_min_keys, st->heat_max_keys, sp_, kmn, kmx);
        }
    }
}

}  // namespace rm
