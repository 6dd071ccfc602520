// respmon_amd/csrc/rm_front.hip -- front half of the calibration: frames -> collapsed band-passed level C_S
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// ------------------------------------------------------------------------------------------
// front half of calibration in two steps:
//   front_pyramid: frames[T,H,W] -> Laplacian levels S..L-2 side by side, lap[T,NP]     (per frame)
//   front_filter : lap[T,NP]     -> collapsed band-passed level S, C_S[T,hS,wS]         (needs every frame)
// rm_calibrate runs them back to back; the frame-sharded path (rm_shard_*) all-gathers lap in between.
// ------------------------------------------------------------------------------------------

void pyr_geom(int H, int W, int levels, int skip, unsigned flags, PyrGeom &pg)
{
    level_sizes(H, W, levels, pg.h, pg.w);
    const int L = levels, S = skip;
    pg.L = L; pg.S = S;
    pg.all_zero = skip >= L - 1;
    if (pg.all_zero) { pg.S = 0; return; }
    pg.chain = S >= 1 && S <= 5 && !(flags & RM_FLAG_UNFUSED_DOWN);
    pg.off.assign(L, 0);
    pg.NP = 0;
    for (int l = S; l <= L - 2; ++l) { pg.off[l] = pg.NP; pg.NP += (size_t)pg.h[l] * pg.w[l]; }
    pg.lds_levels = 0;
    for (int l = S; l < L; ++l) pg.lds_levels += (size_t)pg.h[l] * pg.w[l];
    const size_t LDS_LIMIT = 150 * 1024;
    pg.fuse_small = pg.chain && !(flags & RM_FLAG_UNFUSED_SMALL) && L <= SMALL_MAX_LEVELS &&
                    pg.lds_levels * sizeof(double) <= LDS_LIMIT && pg.NP * sizeof(double) <= LDS_LIMIT;
    if (pg.fuse_small) {
        SmallGeom &sg = pg.sg;
        sg.S = S; sg.L = L; sg.NP = (int)pg.NP;
        int o = 0;
        for (int l = 0; l < L; ++l) {
            sg.h[l] = pg.h[l]; sg.w[l] = pg.w[l];
            sg.g_off[l] = 0; sg.np_off[l] = (int)pg.off[l];
            if (l >= S) { sg.g_off[l] = o; o += pg.h[l] * pg.w[l]; }
        }
        // filter-first form (rm_kernels.h k_small_filter_first): the Gaussian levels and the row-extrema table of the tile
        // bounds must fit LDS together
        if (!(flags & RM_FLAG_FILTER_LAPLACIANS) && S >= 1 && S < MAX_CHAIN) {
            const size_t nS = (size_t)pg.h[S] * pg.w[S];
            const size_t tiles_x = (size_t)(W + CT_W - 1) / CT_W;
            const size_t need = sizeof(double) * (pg.lds_levels + 2 * (size_t)pg.h[S] * tiles_x);
            if (need <= LDS_LIMIT && !(flags & RM_FLAG_FF_PER_LEVEL)) { pg.filter_first = true; pg.NP = nS; }   // (flag: test hook)
        }
    }
    // the same form with one launch per pyramid level when the levels are too large for LDS (4K, skip 2): the temporal filter
    // runs over G_S only and the Laplacian levels are never materialised
    if (pg.chain && !pg.filter_first && (!pg.fuse_small || (flags & RM_FLAG_FF_PER_LEVEL)) &&
        !(flags & (RM_FLAG_FILTER_LAPLACIANS | RM_FLAG_UNFUSED_SMALL)) && S >= 1 && S < MAX_CHAIN) {
        pg.ff_levels = true; pg.fuse_small = false; pg.NP = (size_t)pg.h[S] * pg.w[S];
    }
}

int front_pyramid(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, const PyrGeom &pg, unsigned flags,
                         double *lap, hipStream_t s)
{
    const std::vector<int> &h = pg.h, &w = pg.w;
    const int L = pg.L, S = pg.S;
    const size_t NP = pg.NP;
    // Gaussian chain (pyramid.py:9-17).  Levels < S are stepping stones (ping-pong scratch);
    // levels S..L-1 are kept for the Laplacians.
    std::vector<double *> g(L, nullptr);
    if (dtype == RM_BGR8 && !pg.chain) {
        // [T,H,W,3] uint8 on the paths that do not start with the fused chain (skip 0, unfused test flags): cvtColor of the whole buffer first
        RM_TRY(bgr_buffer_to_gray(ctx, frames, (size_t)T * H * W, &frames, s));
        dtype = RM_U8;
    }
    const void *cur = frames; int cur_dtype = dtype;
    int first = 1;
    if (pg.filter_first || pg.ff_levels) {
        // filter-first form: the array the stages exchange is G_S itself; the small pyramid is built after the temporal filter
        PhaseTimer pt(ctx, 0, s);
        RM_TRY(launch_down_chain(ctx, frames, dtype, T, h, w, S, lap, s, (flags & RM_FLAG_TINY_STRIPS) != 0));
        host_mark(ctx, 1);
        ctx->state_fresh = false;
        return RM_OK;
    }
    if (pg.chain) {
        // one launch reads the frame buffer once and writes only G_S
        double *dst = nullptr;
        RM_TRY(ws(ctx, "g" + std::to_string(S), (size_t)T * h[S] * w[S], &dst));
        {
            PhaseTimer pt(ctx, 0, s);
            RM_TRY(launch_down_chain(ctx, frames, dtype, T, h, w, S, dst, s, (flags & RM_FLAG_TINY_STRIPS) != 0));
        }
        g[S] = dst; cur = dst; cur_dtype = RM_F64;
        first = S + 1;
    }
    if (!pg.fuse_small) {
        for (int l = first; l < L; ++l) {
            double *dst = nullptr;
            if (l < S) RM_TRY(ws(ctx, (l & 1) ? "g_ping" : "g_pong", (size_t)T * h[l] * w[l], &dst));
            else RM_TRY(ws(ctx, "g" + std::to_string(l), (size_t)T * h[l] * w[l], &dst));
            {
                PhaseTimer pt(ctx, (l == 1) ? 0 : 1, s);
                RM_TRY(launch_pyr_down(cur, cur_dtype, T, h[l - 1], w[l - 1], dst, s));
            }
            g[l] = dst; cur = dst; cur_dtype = RM_F64;
        }
    }
    if (S == 0) {
        double *g0 = nullptr;
        RM_TRY(ws(ctx, "g0", (size_t)T * H * W, &g0));
        RM_TRY(launch_to_f64(frames, dtype, (size_t)T * H * W, g0, s));
        g[0] = g0;
    }
    PhaseTimer pt_small(ctx, 1, s);
    // The filtered levels S .. L-2 live side by side in [T, NP] buffers (level S first), so the temporal
    // filter is two launches for the whole small pyramid.
    if (pg.fuse_small) {
        // Gaussian levels S+1..L-1 and all Laplacians in one launch, one workgroup per frame, in LDS
        const size_t shmem = pg.lds_levels * sizeof(double);
        if (shmem > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_pyramid<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k_small_pyramid<>, dim3(T), dim3(SMALL_NT), shmem, s, (const double *)g[S], pg.sg, lap, ctx->d_state);
        LAUNCH_CHECK();
        ctx->state_fresh = true;
    } else {
        // Laplacian levels (pyramid.py:23-26): L_l = G_l - pyrUp(G_{l+1})
        for (int l = L - 2; l >= S; --l)
            RM_TRY(launch_pyr_up(g[l + 1], T, h[l + 1], w[l + 1], lap + pg.off[l], h[l], w[l], 1, g[l], s, 0, NP, 0));
    }
    return RM_OK;
}

int front_filter(rm_ctx *ctx, const double *lap, int T, const PyrGeom &pg, double fps, double fmin, double fmax, double amp,
                        SmallLevels &out, hipStream_t s)
{
    const std::vector<int> &h = pg.h, &w = pg.w;
    const int L = pg.L, S = pg.S;
    const size_t NP = pg.NP;
    const int Th = sym_frames(T);   // the band-passed signal is even in time: everything below handles the unique frames only
    out.h = pg.h; out.w = pg.w; out.S = S; out.all_zero = false;
    // consumed here, on every path: whatever follows reduces into d_state, so the reset by front_pyramid's last kernel
    // vouches for this call only (a later rm_shard_collapse with a foreign lap buffer must reset the state itself)
    const bool state_fresh = ctx->state_fresh;
    ctx->state_fresh = false;
    TemporalOp op;
    RM_TRY(get_operator(ctx, T, fps, fmin, fmax, &op, s));
    PhaseTimer pt_small(ctx, 1, s);
    double *bp = nullptr;
    RM_TRY(ws(ctx, "bp_all", (size_t)Th * NP, &bp));
    if (pg.filter_first) {
        // X = B(G_S) for the unique frames (its workgroup 0 resets the reduction state), then ONE per-frame kernel: Gaussian levels
        // of X, Laplacians, collapse to C_S, tile bounds and lattice samples
        RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s, ctx->d_state));
        ChainGeom cg;
        SmallLevels probe; probe.h = pg.h; probe.w = pg.w; probe.S = S;
        RM_TRY(make_geom(probe, cg));
        const long long npairs = (long long)cg.tiles_x * cg.tiles_y * Th;
        if (npairs >= (1ll << 31)) return fail(RM_E_UNSUPPORTED, "calibration: %lld (tile, frame) pairs exceed 2^31", npairs);
        double *dst = nullptr, *lo = nullptr, *hi = nullptr;
        int *sel_cnt = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * NP, &dst));
        RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &lo));
        RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &hi));
        RM_TRY(ws(ctx, "sel_cnt", (size_t)cg.tiles_x * cg.tiles_y, &sel_cnt));
        const size_t sh = sizeof(double) * (pg.lds_levels + 2 * (size_t)h[S] * cg.tiles_x);
        if (sh > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_filter_first<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        // two workgroups per frame when one per frame leaves CUs idle and the frame has tile rows to share
        int cus_ff = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus_ff, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        int parts = (2 * Th <= cus_ff && cg.tiles_y >= 8) ? 2 : 1;   // (120 KB of LDS: one workgroup per CU, so 2 Th must fit the chip in ONE round -- 258 workgroups at T = 256 took 45 us instead of 27)
        if (ctx->dbg.ff_parts > 0) parts = std::min(ctx->dbg.ff_parts, std::max(1, cg.tiles_y / 2));
        hipLaunchKernelGGL(k_small_filter_first<>, dim3(Th * parts), dim3(SMALL_NT), sh, s, (const double *)bp, pg.sg, (int)pg.lds_levels, dst, ctx->d_state, cg,
                           cg.tiles_x * cg.tiles_y, lo, hi, sel_cnt, parts);
        LAUNCH_CHECK();
        out.state_ready = true; out.bounds_ready = true;
        out.cS = dst;
        return RM_OK;
    }
    if (pg.ff_levels) {
        // X_S = B(G_S); X_{L-1} = pyrDown^(L-1-S)(X_S); U_{L-1} = X_{L-1}, U_l = pyrUp(U_{l+1}); C_S = X_S - pyrUp(U_{S+1})
        // (rm_kernels.h k_small_filter_first: the telescoped collapse, here with one launch per step): only the COARSEST level of
        // the filtered pyramid is needed, so the way down is the fused pyrDown chain (rm_down_chain.h) on the float64 level X_S
        RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s, ctx->d_state));   // (its workgroup 0 resets the reduction state: no k_state_init launch)
        out.state_ready = true;
        std::vector<double *> x(L, nullptr);
        x[S] = bp;
        const int depth = L - 1 - S;
        const bool fused_down = depth >= 1 && depth <= 5;
        auto level_buf = [&](int l) { return x[l] ? RM_OK : ws(ctx, "g" + std::to_string(l), (size_t)Th * h[l] * w[l], &x[l]); };
        RM_TRY(level_buf(L - 1));
        if (fused_down) {
            std::vector<int> hh(h.begin() + S, h.end()), ww(w.begin() + S, w.end());
            RM_TRY(launch_down_chain(ctx, bp, RM_F64, Th, hh, ww, depth, x[L - 1], s, false));
        } else {
            for (int l = S + 1; l < L; ++l) { RM_TRY(level_buf(l)); RM_TRY(launch_pyr_down(x[l - 1], RM_F64, Th, h[l - 1], w[l - 1], x[l], s)); }
        }
        double *dst = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * NP, &dst));
        // the way back up and the subtraction in one launch (k_ff_collapse): levels S .. L-1 as a pyrUp chain of `depth` steps
        SmallLevels up; up.S = depth;
        up.h.assign(h.begin() + S, h.end()); up.w.assign(w.begin() + S, w.end());
        ChainGeom ug;
        if (depth >= 1 && depth < MAX_CHAIN && make_geom(up, ug) == RM_OK && (long long)ug.tiles_x * ug.tiles_y * Th < (1ll << 31)) {
            const int utiles = ug.tiles_x * ug.tiles_y, nitems = utiles * Th;
            const size_t sh = sizeof(double) * (size_t)ug.lds_total;
            if (sh > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_ff_collapse<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL(k_ff_collapse<>, dim3((unsigned)std::min(nitems, 256 * 64)), dim3(64), sh, s, (const double *)bp, (const double *)x[L - 1],
                               ug, utiles, nitems, dst);
            LAUNCH_CHECK();
        } else {
            for (int l = L - 2; l > S; --l) {
                RM_TRY(level_buf(l));
                RM_TRY(launch_pyr_up(x[l + 1], Th, h[l + 1], w[l + 1], x[l], h[l], w[l], 0, nullptr, s));
            }
            RM_TRY(launch_pyr_up(x[S + 1], Th, h[S + 1], w[S + 1], dst, h[S], w[S], 1, bp, s));
        }
        out.cS = dst;
        return RM_OK;
    }
    // temporal band-pass of every level at once (transforms.py:162,169)
    RM_TRY(launch_temporal(ctx, lap, T, NP, op, amp, bp, s));
    // collapse of the band-passed levels L-2 .. S (pyramid.py:51-57; the coarsest level is zeros: 0 + x == x);
    // the result is a contiguous [Th,h_S,w_S] array for the full-resolution passes
    const double *c = bp + pg.off[L - 2];
    if (L - 2 == S) {
        // single filtered level: NP == h_S*w_S, bp_all is already C_S
    } else if (pg.fuse_small) {
        double *dst = nullptr;
        RM_TRY(ws(ctx, "cS", (size_t)Th * h[S] * w[S], &dst));
        const size_t shmem = NP * sizeof(double);
        if (shmem > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_small_collapse<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        // when the row-extrema table of a frame fits beside its small pyramid, the tile bounds of the collapse passes
        // are taken here, from the LDS copy of C_S (k_small_collapse_bounds)
        ChainGeom cg;
        SmallLevels probe; probe.h = pg.h; probe.w = pg.w; probe.S = S;
        const bool geom_ok = S >= 1 && S < MAX_CHAIN && make_geom(probe, cg) == RM_OK;
        const size_t tbl = geom_ok ? 2 * sizeof(double) * (size_t)h[S] * cg.tiles_x : 0;
        const long long npairs = geom_ok ? (long long)cg.tiles_x * cg.tiles_y * Th : 0;
        if (geom_ok && shmem + tbl <= 150 * 1024 && npairs < (1ll << 31) && !ctx->dbg.no_fused_bounds) {
            double *lo = nullptr, *hi = nullptr;
            int *sel_cnt = nullptr;
            RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &lo));
            RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &hi));
            RM_TRY(ws(ctx, "sel_cnt", (size_t)cg.tiles_x * cg.tiles_y, &sel_cnt));
            if (!state_fresh) {   // the lap buffer did not come from front_pyramid on this context just now
                hipLaunchKernelGGL(k_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, ctx->d_state);
                LAUNCH_CHECK();
            }
            const size_t sh2 = shmem + tbl;
            if (sh2 > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)k_small_collapse_bounds<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
            hipLaunchKernelGGL(k_small_collapse_bounds<>, dim3(Th), dim3(SMALL_NT), sh2, s, (const double *)bp, pg.sg, dst, ctx->d_state, cg,
                               cg.tiles_x * cg.tiles_y, lo, hi, sel_cnt);
            out.state_ready = true; out.bounds_ready = true;
        } else {
            hipLaunchKernelGGL(k_small_collapse<>, dim3(Th), dim3(SMALL_NT), shmem, s, (const double *)bp, pg.sg, dst, ctx->d_state);
            out.state_ready = true;
        }
        LAUNCH_CHECK();
        c = dst;
    } else {
        size_t c_fs = NP;
        for (int l = L - 3; l >= S; --l) {
            double *dst = bp + pg.off[l];
            size_t dst_fs = NP;
            if (l == S) {
                RM_TRY(ws(ctx, "cS", (size_t)Th * h[S] * w[S], &dst));
                dst_fs = (size_t)h[S] * w[S];
            }
            RM_TRY(launch_pyr_up(c, Th, h[l + 1], w[l + 1], dst, h[l], w[l], 2, bp + pg.off[l], s, c_fs, dst_fs, NP));
            c = dst; c_fs = dst_fs;
        }
    }
    out.cS = c;
    return RM_OK;
}

int front_half(rm_ctx *ctx, const void *frames, int dtype, int T, int H, int W, double fps, double fmin, double fmax,
                      double amp, int levels, int skip, unsigned flags, SmallLevels &out, hipStream_t s)
{
    PyrGeom pg;
    pyr_geom(H, W, levels, skip, flags, pg);
    out.h = pg.h; out.w = pg.w;
    if (pg.all_zero) { out.all_zero = true; out.S = 0; return RM_OK; }
    double *lap = nullptr;
    RM_TRY(ws(ctx, "lap_all", (size_t)T * pg.NP, &lap));
    RM_TRY(front_pyramid(ctx, frames, dtype, T, H, W, pg, flags, lap, s));
    return front_filter(ctx, lap, T, pg, fps, fmin, fmax, amp, out, s);
}

int make_geom(const SmallLevels &sl, ChainGeom &g)
{
    const int S = sl.S;
    if (S < 1 || S >= MAX_CHAIN) return fail(RM_E_UNSUPPORTED, "fused collapse supports 1 <= skip_levels_at_top <= %d", MAX_CHAIN - 1);
    g.S = S;
    for (int k = 0; k <= S; ++k) { g.h[k] = sl.h[k]; g.w[k] = sl.w[k]; }
    // LDS layout of one evaluation workgroup (doubles).  Step k -> k-1 of the chain needs level k, level k-1 and the scratch of
    // its horizontal pass (every source row of level k at the destination columns of level k-1); level k is dead afterwards:
    //   [ level 1 ][ level 2 ][ B ]   B = levels 3 .. S and the scratch of steps S .. 3 behind them, reused as the (largest)
    //                                     scratch of step 2 -> 1 once those levels are dead
    // 889 doubles at S = 4 instead of 1034 side by side: 22 single-wave workgroups per CU instead of 18.
    auto lvl = [](int k) { return (chain_extent(CT_H, k) + 1) * (chain_extent(CT_W, k) + 1); };
    auto scratch = [](int k) { return (chain_extent(CT_H, k) + 1) * (chain_extent(CT_W, k - 1) + 1); };
    for (int k = 0; k < MAX_CHAIN; ++k) { g.lds_off[k] = 0; g.lds_hb[k] = 0; }
    int off = lvl(1);
    if (S >= 2) { g.lds_off[2] = off; off += lvl(2); }
    const int B = off;
    int small = 0, hb_small = 0;
    for (int k = 3; k <= S; ++k) { g.lds_off[k] = B + small; small += lvl(k); hb_small = std::max(hb_small, scratch(k)); }
    for (int k = 3; k <= S; ++k) g.lds_hb[k] = B + small;
    if (S >= 2) g.lds_hb[2] = B;
    g.lds_total = B + (S >= 2 ? std::max(scratch(2), S >= 3 ? small + hb_small : 0) : 0);
    g.tiles_x = (sl.w[0] + CT_W - 1) / CT_W;
    g.tiles_y = (sl.h[0] + CT_H - 1) / CT_H;
    // weights of the lattice samples (rm_kernels.h lattice_sample): a unit impulse pushed through S interior 1-D pyrUp
    // steps (even: (s[j-1] + 6 s[j] + s[j+1]) / 8, odd: (s[j] + s[j+1]) / 2), read at position 1 << S of a 3-pixel line
    {
        double w[3];
        for (int k = 0; k < 3; ++k) {
            std::vector<double> v(5, 0.0);
            v[1 + k] = 1.0;                      // pixels y-1, y, y+1 sit at 1, 2, 3; 0 and 4 are never reached from the lattice point
            int centre = 2;
            for (int i = 0; i < S; ++i) {
                std::vector<double> u(2 * v.size(), 0.0);
                for (size_t j = 0; j < v.size(); ++j) {
                    const double a = j > 0 ? v[j - 1] : 0.0, c = j + 1 < v.size() ? v[j + 1] : 0.0;
                    u[2 * j] = (a + 6 * v[j] + c) / 8;
                    u[2 * j + 1] = (v[j] + c) / 2;
                }
                v.swap(u);
                centre *= 2;
            }
            w[k] = v[centre];
        }
        g.lat_a = w[0]; g.lat_b = w[1];   // w[2] == w[0]
    }
    return RM_OK;
}

