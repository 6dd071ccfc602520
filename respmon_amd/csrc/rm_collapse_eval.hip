// respmon_amd/csrc/rm_collapse_eval.hip -- collapse, first pass: tile bounds, pruning, exact extrema
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// the flat evaluation pass over the listed pairs (exact extrema; values of the kept pairs into the value store)
int launch_eval_pairs(rm_ctx *ctx, const CollapsePlan &cp, hipStream_t s)
{
    CollapseState *st = ctx->d_state;
    const ChainGeom &g = cp.g;
    const int ntiles = cp.ntiles, npairs = cp.npairs, Th = sym_frames(cp.T);
    const SumPlan &sp = cp.sp;
    struct { const double *cS; int S; } sl{cp.cS, cp.S};
    // one resident round of single-wave workgroups that loop over the lists: their lengths live on the device, and
    // dispatching thousands of workgroups that find nothing to do costs more than the loop.  "Resident" is what the
    // kernel's registers and this geometry's LDS footprint allow per CU (asked of the runtime once per footprint).
    unsigned egrid = 64;   // (host emulation: a fiber per lane -- few, looping workgroups compute the same thing)
#ifndef RM_HIPEMU
    {
        size_t &cached_shmem = ctx->eval_shmem;
        int &cached_per_cu = ctx->eval_per_cu, &cached_cus = ctx->eval_cus;
        if (cached_shmem != cp.shmem) {
            int per_cu = 0, cus = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_eval_pairs<>, 64, cp.shmem));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
            per_cu -= 1;   // (measured: the runtime's figure ignores the LDS allocation granule -- its last workgroup queues)
#ifdef RM_EVAL_PER_CU
            per_cu = RM_EVAL_PER_CU;
#endif
            cached_per_cu = per_cu < 1 ? 1 : per_cu; cached_cus = cus < 1 ? 1 : cus; cached_shmem = cp.shmem;
        }
        const long long capw = (long long)cached_per_cu * cached_cus;
        egrid = (unsigned)(npairs < capw ? npairs : capw);
    }
#else
    if ((long long)egrid > npairs) egrid = (unsigned)npairs;
#endif
    if (tile_eval_ok(g) && ctx->dbg.eval_fast) {
        // the wave-private evaluator (rm_tile_eval.h): ~70 VGPRs and < 5 KB of LDS per single-wave workgroup -- 24 per CU stay resident
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        const unsigned fgrid = (unsigned)std::min<long long>(npairs, 12ll * cus);   // (launching more single-wave workgroups than pairs costs ~0.5 us of ramp per thousand)
#else
        const unsigned fgrid = egrid;
#endif
#define RM_EVAL_FAST(SS)                                                                                                              \
        do {                                                                                                                          \
            using FootE = TileFoot<SS, false>;                                                                                        \
            hipLaunchKernelGGL((k_eval_pairs_fast<SS>), dim3(fgrid), dim3(64), sizeof(double) * FootE::TOTAL, s, sl.cS, g, ntiles, cp.list_a, cp.list_b, \
                               cp.slot_of, st, cp.store, sp, Th);                                                                     \
        } while (0)
        switch (sl.S) { case 1: RM_EVAL_FAST(1); break; case 2: RM_EVAL_FAST(2); break; case 3: RM_EVAL_FAST(3); break; default: RM_EVAL_FAST(4); break; }
#undef RM_EVAL_FAST
        (void)cus;
    } else {
        hipLaunchKernelGGL(k_eval_pairs<>, dim3(egrid), dim3(64), cp.shmem, s, sl.cS, g, ntiles, cp.list_a, cp.list_b, cp.slot_of, st, cp.store, sp, Th);
    }
    LAUNCH_CHECK();
    return RM_OK;
}

// ------------------------------------------------------------------------------------------
// back half: C_S -> exact raw.min()/raw.max() -> masked time sum, for the frames [t0, t1) of the buffer.
// The bounds and the pruning decisions always cover all T frames (they are cheap and every rank of a
// frame-sharded run must agree on them); full-resolution evaluation and the sum touch only [t0, t1).
// ------------------------------------------------------------------------------------------
int collapse_eval(rm_ctx *ctx, const SmallLevels &sl, int T, int t0, int t1, double thr, unsigned flags, CollapsePlan &cp,
                         hipStream_t s)
{
    CollapseState *st = ctx->d_state;
    cp.valid = false;
    cp.cS = sl.cS; cp.T = T; cp.t0 = t0; cp.t1 = t1; cp.H = sl.h[0]; cp.W = sl.w[0]; cp.S = sl.S;
    const size_t npix = (size_t)cp.H * cp.W;
    const int Th = sym_frames(T);   // C_S, the bounds and the pairs exist for the unique frames only (rm_kernels.h sym_frame)
    if (!sl.state_ready) {
        hipLaunchKernelGGL(k_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, st);
        LAUNCH_CHECK();
    }
    const int no_prune = (flags & RM_FLAG_NO_PRUNE) ? 1 : 0;
    if (sl.S == 0) {
        if (t0 != 0 || t1 != T) return fail(RM_E_UNSUPPORTED, "frame-sharded calibration needs skip_levels_at_top >= 1");
        size_t n = (size_t)Th * npix;
        hipLaunchKernelGGL(k_minmax_plain<>, dim3(nblk(n, 256, 1024)), dim3(256), 0, s, sl.cS, n, st);
        LAUNCH_CHECK();
        cp.valid = true;
        return RM_OK;
    }
    ChainGeom &g = cp.g;
    RM_TRY(make_geom(sl, g));
    const int ntiles = g.tiles_x * g.tiles_y;
    const long long npairs_ll = (long long)ntiles * Th;
    if (npairs_ll >= (1ll << 31)) return fail(RM_E_UNSUPPORTED, "calibration: %lld (tile, frame) pairs exceed 2^31", npairs_ll);
    const int npairs = (int)npairs_ll;
    cp.ntiles = ntiles; cp.npairs = npairs;
    int mine_frames = 0;   // unique frames this rank's frame range [t0, t1) holds
    for (int u = 0; u < Th; ++u) mine_frames += sym_in_range(u, T, t0, t1) ? 1 : 0;
    const long long npairs_mine = (long long)ntiles * mine_frames;
    RM_TRY(ws(ctx, "tile_lo", (size_t)npairs, &cp.lo));
    RM_TRY(ws(ctx, "tile_hi", (size_t)npairs, &cp.hi));
    RM_TRY(ws(ctx, "pair_list_a", (size_t)npairs, &cp.list_a));
    RM_TRY(ws(ctx, "pair_list_b", (size_t)npairs, &cp.list_b));
    RM_TRY(ws(ctx, "pair_slot", (size_t)npairs, &cp.slot_of));
    // The value store: 8 KB slots for the pairs the selection keeps, handed out tile by tile (rm_kernels.h k_select_pairs).
    // Capped at STORE_BUDGET_SLOTS (1 GiB): a selection that keeps more takes the dense sum kernel, which needs no store --
    // decided on the device, in this same call (sum_is_dense).  The exhaustive-evaluation baseline (RM_FLAG_NO_PRUNE) and a
    // forced sparse path park every pair they are told to, so they get a slot per pair.
    // (round 4) The store starts at STORE_DEFAULT_SLOTS (128 MiB) and GROWS when a selection of this context has overflowed it
    // (rm_locate reads the kept count after its host synchronisation and runs the evaluation and the sum again with a store that
    // holds it: memory is committed for the streams that need it, up to STORE_MAX_SLOTS = 4 GiB; ctx->store_hint_slots).
    const long long STORE_DEFAULT_SLOTS = ctx->dbg.store_default_slots > 0 ? ctx->dbg.store_default_slots : 16384;   // (knob: test hook)
    const long long STORE_BUDGET_SLOTS = std::min(STORE_MAX_SLOTS, std::max(STORE_DEFAULT_SLOTS, ctx->store_hint_slots));
    cp.no_prune = no_prune != 0;
    SumPlan &sp = cp.sp;
    sp.mode = (flags & RM_FLAG_DENSE_SUM) ? 1 : ((flags & RM_FLAG_SPARSE_SUM) || no_prune) ? 2 : 0;
    sp.auto_dense_ok = sl.S <= 2 ? 1 : 0;
    sp.npairs_mine = (unsigned)npairs_mine;
    long long cap = sp.mode == 2 ? npairs_mine : std::min(npairs_mine, STORE_BUDGET_SLOTS);
    if (flags & RM_FLAG_TINY_STORE) cap = std::min(cap, (long long)8);   // test hook: nearly every selection overflows
    if (ctx->dbg.store_slots > 0) cap = std::min(npairs_mine, ctx->dbg.store_slots);
    if (sp.mode == 1) cap = 0;
    sp.cap_slots = (unsigned)cap;
    cp.store = nullptr;
    // skip 3 / 4 (locate()'s default): the kept pairs are evaluated where they are summed, tile by tile, and nothing is stored
    // (rm_tile_eval.h); the flags that name a sum kernel of the store-based path keep that path (tests compare the two bit for bit)
    cp.fused = tile_eval_ok(g) && !(flags & (RM_FLAG_DENSE_SUM | RM_FLAG_SPARSE_SUM | RM_FLAG_TINY_STORE)) && ctx->dbg.store_slots <= 0 &&
               ctx->dbg.collapse_fused > 0;
    if (cp.fused) { cap = 0; sp.cap_slots = 0; sp.mode = 0; }
    if (cap > 0) RM_TRY(ws(ctx, "value_store", (size_t)cap * CT_H * CT_W, &cp.store));
    ctx->dbg_pairs = npairs; ctx->dbg_cap = cap; ctx->dbg_mine = npairs_mine; ctx->dbg_mode = sp.mode; ctx->dbg_auto_dense = sp.auto_dense_ok;
    RM_TRY(ws(ctx, "sel_cnt", (size_t)ntiles, &cp.sel_cnt));
    RM_TRY(ws(ctx, "heavy_tiles", (size_t)ntiles, &cp.heavy));
    cp.xs_tab = nullptr;
    if (ctx->dbg.xs && tile_eval_ok(g) && !cp.fused) RM_TRY(ws(ctx, "xs_tab", (size_t)npairs, &cp.xs_tab));   // exception store (rm_xstore.h): one entry per pair
    if (!sl.bounds_ready) {
        // per-frame separable form, in bands of tile rows whose row-extrema table fits 64 KB of LDS; the per-pair kernel
        // remains for geometries where even one tile row does not fit
        const size_t row_bytes = 2 * sizeof(double) * (size_t)g.tiles_x;
        int band = g.tiles_y;
        auto tbl_rows_of = [&](int b) {   // most level-S rows any band of b tile rows touches (exact: the device's own footprint rule)
            int most = 0;
            for (int ty0 = 0; ty0 < g.tiles_y; ty0 += b) {
                const int ty1 = std::min(ty0 + b, g.tiles_y) - 1;
                most = std::max(most, tile_region(g, ty1 * g.tiles_x, g.S).y1 - tile_region(g, ty0 * g.tiles_x, g.S).y0 + 1);
            }
            return most;
        };
        size_t tbl_max = 64 * 1024;
        // wide levels (>= 256 columns, at most 64 tile columns): a wave per FB_TR tile rows, streaming (k_frame_bounds_rows)
        const int wS_b = g.w[g.S];
        const size_t rowbufs = 4 * sizeof(double) * (size_t)fb_row_pitch(wS_b);
        // (small images -- 720p: 45 tile rows x 65 frames -- have too few waves of 8 tile rows to fill the chip: the table form stays,
        //  measured 31.7 us against 45 us with 2 tile rows per wave)
        const bool by_rows = (ctx->dbg.bounds_scalar == 2 || (!ctx->dbg.bounds_scalar && (long long)Th * ((g.tiles_y + 7) / 8) >= 2048)) && wS_b >= 256 &&
                             wS_b <= 64 * FB_MAXNL && g.tiles_x <= 64 && ntiles < (1 << 24);
        if (ctx->dbg.bounds_table_bytes > 0) tbl_max = (size_t)ctx->dbg.bounds_table_bytes;   // test hook: force small bands
        while (band > 1 && (size_t)tbl_rows_of(band) * row_bytes > tbl_max) band = (band + 1) / 2;
        // ... and enough workgroups to fill the chip: one workgroup per frame leaves half of it idle at T = 128
        while (band > 4 && (long long)Th * ((g.tiles_y + band - 1) / band) < 1024) band = (band + 1) / 2;
        const int tbl_rows = tbl_rows_of(band);
        const size_t tbl = (size_t)tbl_rows * row_bytes;
        cp.l1_bounds = false;
        if (ctx->dbg.bounds_l1 && bounds_l1_ok(g) && ntiles < (1 << 24)) {
            cp.l1_bounds = true;
            // skip 2: the extrema of the LEVEL-1 footprints, streaming (rm_bounds_l1.h) -- a wave per three tile columns and band of tile rows
            const int nchunks = (g.tiles_x + BL1_TILES - 1) / BL1_TILES;
            int trb = 32;   // (4K x 512: 543 / 375 / 340 us with 8 / 16 / 32 tile rows per wave -- three halo rows per band, and fewer, longer waves)
            while (trb > 8 && (long long)Th * nchunks * ((g.tiles_y + trb - 1) / trb) < 1024) trb >>= 1;   // (small frames: 720p x 128 65 / 52 / 38 / 33 us with 2 / 4 / 8 / 16)
            if (ctx->dbg.bounds_l1_rows > 0) trb = ctx->dbg.bounds_l1_rows;
            const int nbands = (g.tiles_y + trb - 1) / trb;
            // (bounds_l1 2, the default: the level-1 values in packed float32, the bounds widened by what float32 can have lost; 1: in the
            //  chain's own float64 operations -- the exact extrema)
            if (ctx->dbg.bounds_l1 >= 2)
                hipLaunchKernelGGL(k_frame_bounds_l1f<>, dim3(Th, (unsigned)((nchunks * nbands + 3) / 4)), dim3(256), 0, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, cp.sel_cnt,
                                   nchunks, nbands, trb);
            else
                hipLaunchKernelGGL(k_frame_bounds_l1<>, dim3(Th, (unsigned)((nchunks * nbands + 3) / 4)), dim3(256), 0, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, cp.sel_cnt,
                                   nchunks, nbands, trb);
        } else if (by_rows) {
            hipLaunchKernelGGL(k_frame_bounds_rows<8>, dim3(Th, (unsigned)((g.tiles_y + 31) / 32)), dim3(256), rowbufs, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, cp.sel_cnt);
        } else if (tbl <= std::max(tbl_max, (size_t)64 * 1024) && ntiles < (1 << 24)) {
            const unsigned nbands = (unsigned)((g.tiles_y + band - 1) / band);
            hipLaunchKernelGGL(k_frame_bounds<>, dim3(Th, nbands), dim3(256), tbl, s, sl.cS, g, ntiles, cp.lo, cp.hi, st, band, tbl_rows, cp.sel_cnt);
        } else {
            hipLaunchKernelGGL(k_tile_bounds<>, dim3((npairs + 255) / 256), dim3(256), 0, s, sl.cS, g, Th, ntiles, cp.lo, cp.hi, st, cp.sel_cnt);
        }
        LAUNCH_CHECK();
    }
    // skip 3 / 4 behind a call that kept many pairs (a stream of noise): the bounds once more, from the level-(S - 1) footprint
    if ((sl.S == 3 || sl.S == 4) && tile_eval_ok(g) && !no_prune && (ctx->dbg.bounds_up1 == 1 || (ctx->dbg.bounds_up1 < 0 && ctx->refine_hint))) {
        const dim3 rgrid((unsigned)((ntiles + 255) / 256), (unsigned)Th);
        if (sl.S == 4) hipLaunchKernelGGL(k_bounds_up1<4>, rgrid, dim3(256), 0, s, sl.cS, g, ntiles, cp.lo, cp.hi);
        else hipLaunchKernelGGL(k_bounds_up1<3>, rgrid, dim3(256), 0, s, sl.cS, g, ntiles, cp.lo, cp.hi);
        LAUNCH_CHECK();
    }
    const int prune_ok = (!no_prune && thr >= 0.0 && thr <= 1.0) ? 1 : 0;
    hipLaunchKernelGGL(k_select_pairs<>, dim3((ntiles + SEL_TILES - 1) / SEL_TILES, (Th + SEL_PH * SEL_U - 1) / (SEL_PH * SEL_U)), dim3(256), 0, s,
                       cp.lo, cp.hi, ntiles, Th, T, t0, t1, st, cp.list_a, cp.list_b, cp.slot_of, prune_ok ? 0 : 1, thr, cp.sel_cnt, cp.heavy, cp.xs_tab);
    LAUNCH_CHECK();
    if (cp.fused) {
        // exact extrema from the C pairs: one wave per pair, a grid that covers the few pairs of a pruned selection at once and loops
        // over an exhaustive one
        const unsigned cgrid = (unsigned)std::min<long long>(npairs, 8192);
        ctx->dbg_fused = 1;
#define RM_EVAL_C(SS)                                                                                            \
        do {                                                                                                     \
            using FootC = TileFoot<SS, false>;                                                                   \
            hipLaunchKernelGGL((k_eval_c<SS>), dim3(cgrid), dim3(64), sizeof(double) * FootC::TOTAL, s, sl.cS, g, ntiles, cp.list_a, st); \
        } while (0)
        switch (sl.S) { case 1: RM_EVAL_C(1); break; case 2: RM_EVAL_C(2); break; case 3: RM_EVAL_C(3); break; default: RM_EVAL_C(4); break; }
#undef RM_EVAL_C
        LAUNCH_CHECK();
        cp.valid = true;
        return RM_OK;
    }
    ctx->dbg_fused = 0;
    cp.shmem = sizeof(double) * (size_t)g.lds_total;
    RM_TRY(launch_eval_pairs(ctx, cp, s));
    cp.valid = true;
    return RM_OK;
}

