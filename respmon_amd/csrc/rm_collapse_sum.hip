// respmon_amd/csrc/rm_collapse_sum.hip -- collapse, second pass: the masked time sum (sparse / dense / store-less forms)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// heat_sum[H*W] = sum over t in [t0, t1) of (raw >= top ? min : raw), with min/max as they stand in the state
// avg_T > 0: the sum covers the whole buffer, write heat = sum / avg_T and leave the heatmap's min / max in the state
// host_rescue: the caller synchronises the stream soon and looks at the slot's h_unserved (rm_locate): a dense kernel that could only
// be chosen because the value store overflowed is then not enqueued here
int collapse_sum(rm_ctx *ctx, const CollapsePlan &cp, double thr, double *heat_sum, hipStream_t s, int avg_T, bool host_rescue)
{
    CollapseState *st = ctx->d_state;
    const size_t npix = (size_t)cp.H * cp.W;
    if (cp.S == 0) {
        hipLaunchKernelGGL(k_finish_minmax<>, dim3(1), dim3(NSTRIPE), 0, s, st, thr);
        LAUNCH_CHECK();
        double *sum = heat_sum;
        if (avg_T > 0) RM_TRY(ws(ctx, "heat_sum", npix, &sum));
        hipLaunchKernelGGL(k_masked_sum_plain<>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, cp.cS, cp.T, npix, st, sum);
        LAUNCH_CHECK();
        if (avg_T > 0) {
            hipLaunchKernelGGL(k_heat_avg_minmax<>, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, sum, npix, avg_T, heat_sum, st);
            LAUNCH_CHECK();
        }
        return RM_OK;
    }
    int *tile_nkept = nullptr;
    RM_TRY(ws(ctx, "tile_nkept", (size_t)cp.ntiles, &tile_nkept));
    int *unserved_dev = nullptr;
    auto launch_tile_sum = [&](int only_if_dense) -> int {
        // one workgroup of TS_NW waves per CU (the exchange takes most of a CU's LDS): the heavy tiles' items first, the workgroups
        // left without one fill the constant tiles
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#else
        cus = 6;   // (host emulation: few, looping workgroups compute the same thing)
#endif
        const int nworkers = std::max(1, std::min(2 * cp.ntiles, cus));
#define RM_TILE_SUM(SS)                                                                                                                  \
        do {                                                                                                                             \
            constexpr int exd = tile_sum_exchange_doubles<SS, false>();                                                                  \
            const size_t shb = sizeof(double) * (size_t)exd + sizeof(int) * (size_t)cp.T;                                                \
            if (shb > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_tile_sum<SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb)); \
            hipLaunchKernelGGL((k_tile_sum<SS>), dim3(nworkers), dim3(64 * TS_NW), shb, s, cp.cS, cp.g, cp.t0, cp.t1, cp.T, cp.ntiles, cp.slot_of, st, thr,  \
                               heat_sum, avg_T, tile_nkept, cp.sel_cnt, cp.heavy, nworkers, ctx->dbg.tile_sum_half, cp.sp, only_if_dense);          \
        } while (0)
        switch (cp.S) { case 1: RM_TILE_SUM(1); break; case 2: RM_TILE_SUM(2); break; case 3: RM_TILE_SUM(3); break; default: RM_TILE_SUM(4); break; }
#undef RM_TILE_SUM
        LAUNCH_CHECK();
        return RM_OK;
    };
    auto launch_dense_t = [&](int only_if_dense, int xs_standin = 0) -> int {
        // one wave per tile, frame after frame (rm_tile_eval.h k_dense_sum_t)
#define RM_DENSE_T(SS)                                                                                                                   \
        do {                                                                                                                             \
            using FootD = TileFoot<SS, false>;                                                                                           \
            hipLaunchKernelGGL((k_dense_sum_t<SS>), dim3(dense_tile_grid(cp.ntiles)), dim3(64), sizeof(double) * (FootD::TOTAL + DST_MAXW) + sizeof(unsigned short) * (size_t)((cp.T + 3) & ~3), s, cp.cS, cp.g, cp.t0, \
                               cp.t1, cp.T, cp.ntiles, cp.slot_of, st, thr, heat_sum, avg_T, tile_nkept, cp.sp, only_if_dense, unserved_dev, (ctx->dbg.dense_exact_top && !cp.no_prune) ? cp.lo : nullptr, xs_standin, (cp.l1_bounds && ctx->dbg.dense_exact_top && !cp.no_prune) ? 0 : 1); \
        } while (0)
        switch (cp.S) { case 1: RM_DENSE_T(1); break; case 2: RM_DENSE_T(2); break; case 3: RM_DENSE_T(3); break; default: RM_DENSE_T(4); break; }
#undef RM_DENSE_T
        LAUNCH_CHECK();
        return RM_OK;
    };
    // The exception store (rm_xstore.h): every kept pair evaluated ONCE by a flat pass, its values below `top` parked in a compact
    // record, the time-ordered additions by a kernel that reads records only.  Takes the place of the store-less kernels wherever
    // TileEval applies; k_dense_sum_t stays behind it as the stand-in for a selection whose exceptions do not fit the store.
    auto launch_xs = [&](int only_if_dense) -> int {
        const int Th = sym_frames(cp.T);
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        // evaluation grid: single-wave workgroups, each a strided sample of the lists (at least ~8 listed pairs per wave)
#ifndef RM_HIPEMU
        const unsigned egrid = (unsigned)std::max<long long>(1, std::min<long long>((cp.npairs + 7) / 8, 32ll * cus));
#else
        const unsigned egrid = (unsigned)std::max(1, std::min(cp.npairs, 24));   // (host emulation: few, looping workgroups compute the same thing)
#endif
        // capacity: the worst case of the geometry (every value of every pair an exception) when that is small, 1 GiB otherwise; plus
        // the chunk every wave may leave unfinished
        const unsigned long long worst = (unsigned long long)cp.sp.npairs_mine * (XS_HDR + CT_H * CT_W);
        unsigned long long budget = ctx->dbg.xs_budget_words > 0 ? (unsigned long long)ctx->dbg.xs_budget_words : (1ull << 27);
        const bool never_overflows = ctx->dbg.xs_budget_words <= 0 && worst <= budget;
        XsPlan xp;
        // (+ a third: a chunk is abandoned with at most a quarter of it unused; + every wave's first chunk)
        xp.cap_words = std::min(worst, budget) + std::min(worst, budget) / 3 + (unsigned long long)(egrid + 1) * XS_CHUNK;
        if (ctx->dbg.xs_budget_words > 0) xp.cap_words = budget;
        if (xp.cap_words > 0xfffffff0ull) xp.cap_words = 0xfffffff0ull;   // (record offsets are 32-bit words)
        RM_TRY(ws(ctx, "xs_store", (size_t)xp.cap_words + 64, &xp.store));   // (+ 64: k_xs_sum reads 64 words from the start of a record)
        xp.tab = reinterpret_cast<XsEntry *>(cp.xs_tab);
        const double *lo = (ctx->dbg.dense_exact_top && !cp.no_prune) ? cp.lo : nullptr;
#define RM_XS_EVAL(SS)                                                                                                                   \
        do {                                                                                                                             \
            using FootX = TileFoot<SS, false>;                                                                                           \
            hipLaunchKernelGGL((k_xs_eval<SS>), dim3(egrid), dim3(64), sizeof(double) * FootX::TOTAL, s, cp.cS, cp.g, cp.ntiles, cp.list_a, cp.list_b, \
                               cp.slot_of, lo, st, thr, cp.sp, Th, xp, only_if_dense);                                                  \
        } while (0)
        switch (cp.S) { case 1: RM_XS_EVAL(1); break; case 2: RM_XS_EVAL(2); break; case 3: RM_XS_EVAL(3); break; default: RM_XS_EVAL(4); break; }
#undef RM_XS_EVAL
        LAUNCH_CHECK();
        // the additions: one wave per tile where tiles are plenty, 2 / 4 waves per tile (each its share of the running sums) where a
        // lone wave per SIMD would be bound by the latency of its own chain
        int nw = cp.ntiles >= 4 * cus ? 1 : (cp.ntiles >= 2 * cus ? 2 : 4);
        if (ctx->dbg.xs_waves == 1 || ctx->dbg.xs_waves == 2 || ctx->dbg.xs_waves == 4) nw = ctx->dbg.xs_waves;
        const size_t shx = 3 * sizeof(int) * (size_t)Th;
#define RM_XS_SUM(NN)                                                                                                                    \
        hipLaunchKernelGGL((k_xs_sum<NN>), dim3(dense_tile_grid(cp.ntiles)), dim3(64 * NN), shx, s, cp.g, cp.t0, cp.t1, cp.T, cp.ntiles, st, thr, heat_sum, \
                           avg_T, tile_nkept, cp.sp, xp, only_if_dense, unserved_dev)
        if (nw == 1) RM_XS_SUM(1); else if (nw == 2) RM_XS_SUM(2); else RM_XS_SUM(4);
#undef RM_XS_SUM
        LAUNCH_CHECK();
        if (!never_overflows) RM_TRY(launch_dense_t(only_if_dense, 1));
        return RM_OK;
    };
    const bool xs_ok = ctx->dbg.xs != 0 && cp.xs_tab != nullptr && tile_eval_ok(cp.g) && !ctx->dbg.dense_rows && !ctx->dbg.dense_general;
    if (cp.fused) {
        if (ctx->dbg.dense_tiles) RM_TRY(launch_dense_t(0)); else
        RM_TRY(launch_tile_sum(0));
        ctx->nkept_H = cp.H; ctx->nkept_W = cp.W;
        return RM_OK;
    }
    const SumPlan &sp = cp.sp;
    // Both sum kernels are enqueued and the one whose turn it is not returns at once (sum_is_dense, decided from this call's own
    // selection); a launch that can never be chosen is left out: the sparse one when the dense kernel is forced, the dense one when
    // the store has a slot for every pair and the automatic rule cannot pick it.
    const bool may_sparse = sp.mode != 1;
    const bool auto_dense = sp.mode == 0 && sp.auto_dense_ok;
    const bool overflow_only = !auto_dense && sp.mode != 1 && sp.cap_slots < sp.npairs_mine;
    // (a context whose last selection was dense enqueues the stand-in behind the sparse kernel instead of waiting for the host to
    //  find the store overflowed: ctx->dense_hint, set and cleared by rm_locate)
    const bool may_dense = sp.mode == 1 || auto_dense || (overflow_only && (!host_rescue || ctx->dense_hint));
    if (overflow_only && host_rescue) {
        RoiSlot &rs = ctx->slots[ctx->cur_slot];
        if (!rs.h_unserved) HIP_TRY(hipHostMalloc((void **)&rs.h_unserved, 2 * sizeof(int), hipHostMallocDefault));   // [0]: the word above; [1]: pairs the selection kept
        rs.h_unserved[0] = 0; rs.h_unserved[1] = 0;
        HIP_TRY(hipHostGetDevicePointer((void **)&unserved_dev, rs.h_unserved, 0));
    }
    if (may_sparse) {
        // worker items for the tiles with kept pairs (MS_Q each); the workgroups left without an item fill the other tiles
#ifdef RM_HIPEMU
        const int nworkers = std::min(cp.ntiles * MS_Q, 24);    // (host emulation: fewer, looping workgroups compute the same thing)
#else
        const int nworkers = std::min(cp.ntiles * MS_Q, MS_B > 16 ? 512 : 768);   // 2-3 workgroups per CU (registers): one resident round
#endif
        if (cp.t0 == 0 && cp.t1 == cp.T && avg_T == cp.T && ctx->dbg.sum_rows) {
            // the whole buffer: one wave per (heavy tile, row), the kept unique frames' values staged by LDS-DMA (rm_tile_eval.h)
            const size_t shr = sizeof(double) * MSR_CHUNK * 64 + 2 * sizeof(int) * (size_t)sym_frames(cp.T);
#ifdef RM_HIPEMU
            const int nw3 = std::min(cp.ntiles * CT_H, 40);
#else
            // one resident round: what the LDS footprint lets a CU hold (a queued wave starts its chain of round trips late)
            int cus3 = 256;
            HIP_TRY(hipDeviceGetAttribute(&cus3, hipDeviceAttributeMultiprocessorCount, ctx->device));
            const int per_cu3 = (int)std::max<size_t>(1, std::min<size_t>(32, ((size_t)160 * 1024) / (shr + 512)));
            const int nw3 = std::min(cp.ntiles * CT_H, per_cu3 * cus3);
#endif
            hipLaunchKernelGGL(k_masked_sum_rows<>, dim3(nw3), dim3(64), shr, s, cp.T, cp.ntiles, cp.W, cp.H, cp.slot_of, cp.store, st, thr, heat_sum,
                               tile_nkept, cp.sel_cnt, cp.heavy, nw3, sp, unserved_dev);
        } else if (cp.t0 == 0 && cp.t1 == cp.T && avg_T == cp.T && ctx->dbg.sum_sym) {
            // the whole buffer: every unique frame loaded once and added on the way up and on the way down (rm_tile_eval.h)
            const int nw2 = std::min(nworkers, 512);   // 220 VGPRs: two workgroups per CU stay resident
            hipLaunchKernelGGL(k_masked_sum_sym<>, dim3(nw2), dim3(64 * MS_RQ), 2 * sizeof(int) * (size_t)sym_frames(cp.T), s, cp.T, cp.ntiles, cp.W, cp.H,
                               cp.slot_of, cp.store, st, thr, heat_sum, tile_nkept, cp.sel_cnt, cp.heavy, nw2, sp, unserved_dev);
        } else {
            hipLaunchKernelGGL(k_masked_sum_tiles<>, dim3(nworkers), dim3(64 * MS_RQ), 2 * sizeof(int) * (size_t)cp.T, s, cp.t0, cp.t1, cp.T, cp.ntiles,
                               cp.W, cp.H, cp.slot_of, cp.store, st, thr, heat_sum, avg_T, tile_nkept, cp.sel_cnt, cp.heavy, nworkers, sp, unserved_dev);
        }
        LAUNCH_CHECK();
    }
    // skip <= 2 on large frames (four waves' worth of tiles per SIMD): the TileEval kernel of the deeper chains is the faster one-wave-per-
    // tile form there too (4K x 512 skip 2: 2.26 -> 2.18 ms); smaller frames keep the several-waves-per-tile forms below
    const bool t_low = ctx->dbg.dense_t_low >= 0 ? ctx->dbg.dense_t_low != 0 : (cp.S == 2 && cp.ntiles >= 4096 && tile_eval_ok(cp.g));
    if (may_dense && xs_ok) {
        RM_TRY(launch_xs(sp.mode == 1 ? 0 : 1));
    } else if (may_dense && cp.S <= 2 && ctx->dbg.dense_wave && !t_low && !ctx->dbg.dense_rows && !ctx->dbg.dense_general && dense_wave_ok(cp.g)) {
        // one wave per 64 x 16 tile, no barriers (rm_dense_sum.h k_dense_sum_w)
        const ChainGeom &g = cp.g;
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        // fewer than two waves per SIMD: two frames per trip, interleaved (the lone wave's dependency chain is what takes the time)
        int fr = cp.ntiles < 8 * cus ? 2 : 1;
        if (ctx->dbg.dense_frames == 1 || ctx->dbg.dense_frames == 2) fr = ctx->dbg.dense_frames;
        // fewer tiles than SIMDs: NW waves per tile, each evaluating every NW-th frame (k_dense_sum_wf)
        int split = cp.ntiles < 4 * cus ? 4 : 1;
        if (ctx->dbg.dense_split == 1 || ctx->dbg.dense_split == 2 || ctx->dbg.dense_split == 4) split = ctx->dbg.dense_split;
        if (split > 1) {
            // (the kept frames only: what the selection kept and the exact top does not clear -- dense_wf_list 0: every frame, as before round 6)
            const int *wf_slot = (ctx->dbg.dense_wf_list && !cp.no_prune) ? cp.slot_of : nullptr;
            const double *wf_lo = (wf_slot && ctx->dbg.dense_exact_top) ? cp.lo : nullptr;
#define RM_DENSE_WF(SS, NN)                                                                                                               \
            do {                                                                                                                          \
                const size_t shf = sizeof(double) * (size_t)NN * 16 * 64 + sizeof(unsigned short) * (size_t)((cp.T + 3) & ~3);           \
                hipLaunchKernelGGL((k_dense_sum_wf<SS, NN>), dim3(dense_tile_grid(cp.ntiles)), dim3(64 * NN), shf, s, cp.cS, g, cp.t0, cp.t1, cp.T, st, thr,  \
                                   heat_sum, avg_T, tile_nkept, sp, wf_slot, wf_lo);                                                      \
            } while (0)
            if (cp.S == 2) { if (split == 4) RM_DENSE_WF(2, 4); else RM_DENSE_WF(2, 2); }
            else { if (split == 4) RM_DENSE_WF(1, 4); else RM_DENSE_WF(1, 2); }
#undef RM_DENSE_WF
            LAUNCH_CHECK();
        } else {
#define RM_DENSE_W(SS, FF)                                                                                                                \
        hipLaunchKernelGGL((k_dense_sum_w<SS, FF>), dim3(dense_tile_grid(cp.ntiles)), dim3(64), sizeof(double) * DenseW<SS>::TOTAL * FF, s, cp.cS, g, cp.t0, \
                           cp.t1, cp.T, st, thr, heat_sum, avg_T, tile_nkept, sp)
        if (cp.S == 2) { if (fr == 2) RM_DENSE_W(2, 2); else RM_DENSE_W(2, 1); }
        else { if (fr == 2) RM_DENSE_W(1, 2); else RM_DENSE_W(1, 1); }
#undef RM_DENSE_W
        LAUNCH_CHECK();
        }
    } else if (may_dense && (cp.S >= 3 || t_low) && tile_eval_ok(cp.g) && !ctx->dbg.dense_rows && !ctx->dbg.dense_general) {
        // deeper chains: every kept pair evaluated where it is summed, tile by tile (rm_tile_eval.h k_tile_sum); it looks at the
        // selection itself when the sparse kernel was enqueued in front of it
        if (ctx->dbg.dense_tiles) RM_TRY(launch_dense_t(sp.mode == 1 ? 0 : 1));
        else RM_TRY(launch_tile_sum(sp.mode == 1 ? 0 : 1));
    } else if (may_dense) {
        // super-tiles of 64 x 64 pixels (four waves, 16 rows each) when that still gives every CU two workgroups, 64 x 32 (two
        // waves) next; with fewer tiles than that, one 64 x 16 tile per workgroup and four rows per wave: the per-frame latency counts
        int cus = 256;
#ifndef RM_HIPEMU
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
#endif
        const ChainGeom &g = cp.g;
        int rows = 64;
        while (rows > 16 && (long long)g.tiles_x * ((cp.H + rows - 1) / rows) < 2ll * cus) rows >>= 1;
        { const int v = ctx->dbg.dense_rows; if (v == 16 || v == 32 || v == 64) rows = v; }   // test hook
        DenseGeom dg;
        dg.rows = rows; dg.nsx = g.tiles_x; dg.nsy = (cp.H + rows - 1) / rows;
        const int S = cp.S;
        auto lvl = [&](int k) { return (chain_extent(rows, k) + 1) * (chain_extent(CT_W, k) + 1); };
        auto scratch = [&](int k) { return (chain_extent(rows, k) + 1) * (chain_extent(CT_W, k - 1) + 1); };
        for (int k = 0; k < MAX_CHAIN; ++k) { dg.lds_off[k] = 0; dg.lds_hb[k] = 0; }
        int off = lvl(1);                                   // [ level 1 ][ level 2 ][ B ], as make_geom lays out k_eval_pairs
        if (S >= 2) { dg.lds_off[2] = off; off += lvl(2); }
        const int B = off;
        int small = 0, hb_small = 0;
        for (int k = 3; k <= S; ++k) { dg.lds_off[k] = B + small; small += lvl(k); hb_small = std::max(hb_small, scratch(k)); }
        for (int k = 3; k <= S; ++k) dg.lds_hb[k] = B + small;
        if (S >= 2) dg.lds_hb[2] = B;
        dg.lds_total = B + (S >= 2 ? std::max(scratch(2), S >= 3 ? small + hb_small : 0) : 0);
        const size_t sh = sizeof(double) * (size_t)dg.lds_total;
        const unsigned grid = (unsigned)(dg.nsx * dg.nsy);
#define RM_DENSE_LAUNCH(KERNEL, NW, RPW)                                                                                                 \
        do {                                                                                                                             \
            if (sh > 64 * 1024)                                                                                                          \
                HIP_TRY(hipFuncSetAttribute((const void *)KERNEL<NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));       \
            hipLaunchKernelGGL((KERNEL<NW, RPW>), dim3(grid), dim3(64 * NW), sh, s, cp.cS, g, dg, cp.t0, cp.t1, cp.T, st, thr, heat_sum, \
                               avg_T, tile_nkept, sp);                                                                                   \
        } while (0)
        if (S <= 2 && !ctx->dbg.dense_general) {   // table-driven form (knob: test hook for the general kernel)
            if (rows == 64) RM_DENSE_LAUNCH(k_dense_sum_s2, 4, 16); else if (rows == 32) RM_DENSE_LAUNCH(k_dense_sum_s2, 2, 16); else RM_DENSE_LAUNCH(k_dense_sum_s2, 4, 4);
        } else {
            if (rows == 64) RM_DENSE_LAUNCH(k_dense_sum, 4, 16); else if (rows == 32) RM_DENSE_LAUNCH(k_dense_sum, 2, 16); else RM_DENSE_LAUNCH(k_dense_sum, 4, 4);
        }
#undef RM_DENSE_LAUNCH
        LAUNCH_CHECK();
    }
    ctx->nkept_H = cp.H; ctx->nkept_W = cp.W;   // the constant tiles of this heatmap (or partial heat sum of a frame shard) are known
    return RM_OK;
}

