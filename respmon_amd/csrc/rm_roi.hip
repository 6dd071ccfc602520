// respmon_amd/csrc/rm_roi.hip -- heatmap -> ROI (base.py:563-575) and the sparse heatmap packets
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

constexpr int CCL_TABLE_MAX_COMPONENTS = 2048;   // more components than this last time: k_ccl_bbox without its LDS table
constexpr int LABEL_REPROBE = 64;         // labelled stages in a row before the host-only stage is timed again
constexpr int LABEL_MIN_CONTOURS = 512;   // ~0.13 us per followed border on the host against ~40 us of labelling kernels
constexpr long long LABEL_MIN_STEPS = 24000;   // ... or this many border steps (5-10 ns each, cache misses included) against ~100 us of labelling kernels
static_assert(sizeof(CclComp) == sizeof(LabelComp), "record layout shared by rm_ccl.h and rm_contour.h");

// The ROI stage in two halves: roi_launch enqueues the device work (threshold -> packed image in the pinned memory of slot
// ctx->cur_slot, component labelling when the rule asks for it), roi_finish -- once the stream (or the event recorded behind the
// launches) has been waited for -- runs the host contour stage on that slot.  heatmap_to_roi_impl is the two with stream_wait
// between them; rm_locate_submit / rm_locate_result put the next call's frame-buffer kernel there instead.
int roi_launch(rm_ctx *ctx, const double *heat, int H, int W, int threshold, uint8_t *avg_u8, uint8_t *binary, void *stream,
                      bool have_minmax, RoiPending &pd, bool xywh_given)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_heatmap_to_roi: bad argument");
    // the one-call clip request of rm_locate (RM_FLAG_CONTOUR_CLIP_FRAME) is consumed here, whatever happens below
    const bool clip_once = ctx->clip_frame_once, tiles_once = ctx->tiles_const_once;
    ctx->clip_frame_once = false;
    ctx->tiles_const_once = false;   // (consumed here too: a failure below must not leave it set for a foreign heatmap of this geometry)
    if (!heat || !xywh_given || H < 1 || W < 1) return fail(RM_E_BADARG, "rm_heatmap_to_roi: bad argument");
    RoiSlot &rs = ctx->slots[ctx->cur_slot];
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const size_t npix = (size_t)H * W;
    CollapseState *st = ctx->d_state;
    ctx->state_fresh = false;   // the heatmap extrema are (or have been) reduced into the state
    // the thresholded image goes to the host bit-packed (npix / 8 bytes) plus one "row holds foreground" byte per row: the
    // kernel stores both straight into pinned, device-mapped host memory (no copy-engine hop)
    // ... and, for images whose rows are whole words, one 8-byte record per row (k_heat_rows_u8) behind the row flags
    const size_t nwords = (npix + 63) / 64;
    const size_t rec_off = (nwords * 8 + (size_t)H + 7) & ~(size_t)7;
    const size_t need = rec_off + (size_t)H * 8;
    if (rs.h_bin_cap < need) {
        if (rs.h_bin) { HIP_TRY(stream_wait(s)); HIP_TRY(hipHostFree(rs.h_bin)); }
        rs.h_bin = nullptr; rs.h_bin_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&rs.h_bin, need, hipHostMallocDefault));
        rs.h_bin_cap = need;
        rs.h_rows_dirty = nullptr;
    }
    uint8_t *h_rows = rs.h_bin + nwords * 8;
    // invariant between calls: image words, row flags and row records are all zero.  The flag path zeroes what it read in roi_finish;
    // the record path leaves that to the NEXT launch on the slot -- here, while the GPU is busy with the frame-buffer kernel, instead
    // of on the host path between two calibrations
    if (rs.h_rows_dirty != h_rows || rs.dirty_geom != need) { std::memset(rs.h_bin, 0, rs.h_bin_cap); rs.h_rows_dirty = h_rows; rs.dirty_geom = need; rs.dirty_w1 = 0; rs.dirty_w0 = 1; }
    else if (rs.dirty_w1 >= rs.dirty_w0) {
        std::memset(rs.h_bin + rs.dirty_w0 * 8, 0, (rs.dirty_w1 - rs.dirty_w0 + 1) * 8);
        std::memset(rs.h_bin + rec_off + (size_t)rs.dirty_r0 * 8, 0, (size_t)(rs.dirty_r1 - rs.dirty_r0 + 1) * 8);
        rs.dirty_w1 = 0; rs.dirty_w0 = 1;
    }
    uint8_t *dev_bin = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dev_bin, rs.h_bin, 0));
    PhaseTimer *pt_roi = new PhaseTimer(ctx, 3, s);
    struct Guard { PhaseTimer *&p; ~Guard() { delete p; p = nullptr; } } guard{pt_roi};
    if (!have_minmax) {  // rm_calibrate has just left the heatmap's min / max in the state
        hipLaunchKernelGGL(k_heat_state_init<>, dim3(1), dim3(NSTRIPE), 0, s, st);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_heat_minmax<>, dim3(nblk(npix, 256, 256)), dim3(256), 0, s, heat, npix, st);
        LAUNCH_CHECK();
    }
    // noisy images: label the components on the device so that the host follows only borders that can win (rm_ccl.h)
    const bool clip = ctx->clip_frame || clip_once;
    const bool same_geom = ctx->label_H == H && ctx->label_W == W;
    if (!same_geom) { ctx->label_unl_steps = -1; ctx->label_streak = 0; ctx->label_lazy = false; }
    const bool many = ctx->label_last_n > LABEL_MIN_CONTOURS;
    // (decided from what the previous extraction of this geometry COUNTED -- contours met, border steps walked --, never from how long
    //  it took: the path a stream takes, and with it its step time, is the same in every run)
    bool slow_host = ctx->label_unl_steps > (ctx->dbg.label_host_steps > 0 ? ctx->dbg.label_host_steps : LABEL_MIN_STEPS);
    if (slow_host && !many && ctx->label_mode < 0 && ctx->label_streak >= LABEL_REPROBE) { slow_host = false; ctx->label_streak = 0; }
    const bool label = !clip && npix < (size_t)0x7fffffff &&
                       (ctx->label_mode == 1 || (ctx->label_mode < 0 && same_geom && (many || slow_host)));
    // rm_locate's own heatmap: the sum kernel that wrote it knows which 64 x 16 tiles are one constant (tile_nkept == 0)
    const int *tile_const = nullptr;
    if (tiles_once && ctx->nkept_H == H && ctx->nkept_W == W && W % CT_W == 0 && ctx->dbg.heat_const_tiles) {
        int *tk = nullptr;
        RM_TRY(ws(ctx, "tile_nkept", (size_t)((W + CT_W - 1) / CT_W) * ((H + CT_H - 1) / CT_H), &tk));
        tile_const = tk;
    }
    unsigned long long *d_bits = nullptr;
    const size_t comps_cap = std::min<size_t>(npix / 4 + 2, (size_t)1 << 18);
    bool rows = false;
    // Lazy: the last labelled extraction of this geometry needed nothing but the 2 KB of summary records (the area bound of the top
    // component beat every rival's box: rm_ccl.h) -- this one keeps the packed image and the full list in device buffers of the slot and
    // sends the summaries only.  At 4K x 512 that is 1.7 MB less over PCIe behind the last kernel and a 1 MB memset less on the host
    // between two calibrations; roi_finish fetches both if the summaries do not settle the winner after all.
    // (not in the two-call form: the fetch roi_finish may have to make would queue behind the NEXT submission's calibration on the
    //  tickets' one stream -- rm_locate_result(A) would wait for buffer B's kernels, the overlap the two calls exist for)
    const bool lazy = label && ctx->label_lazy && ctx->dbg.label_lazy != 0 && ctx->cur_slot == 0;
    CclComp *d_complist = nullptr;
    if (label) {
        int *d_label = nullptr; CclBox *d_box = nullptr; unsigned int *d_cnt = nullptr; int *d_list = nullptr;
        RM_TRY(ws(ctx, "ccl_list", comps_cap, &d_list));
        RM_TRY(ws(ctx, "ccl_bits_slot" + std::to_string(ctx->cur_slot), nwords, &d_bits));   // (per slot: a lazy fetch may follow the NEXT submission's kernels)
        if (lazy) RM_TRY(ws(ctx, "ccl_comps_slot" + std::to_string(ctx->cur_slot), comps_cap * 2, (double **)&d_complist));
        RM_TRY(ws(ctx, "ccl_label", npix, &d_label));
        RM_TRY(ws(ctx, "ccl_box", npix, &d_box));
        RM_TRY(ws(ctx, "ccl_counters", (size_t)2, &d_cnt));
        if (rs.h_comps_cap < comps_cap + 1 + 2 * CCL_PUB_BLOCKS) {
            if (rs.h_comps) { HIP_TRY(stream_wait(s)); HIP_TRY(hipHostFree(rs.h_comps)); }
            rs.h_comps = nullptr; rs.h_comps_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&rs.h_comps, (comps_cap + 1 + 2 * CCL_PUB_BLOCKS) * sizeof(CclComp), hipHostMallocDefault));
            rs.h_comps_cap = comps_cap + 1 + 2 * CCL_PUB_BLOCKS;
        }
        CclComp *dev_comps = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dev_comps, rs.h_comps, 0));
        // rows of whole words: the components of every 64 x 32 tile in LDS first, then the seams (rm_ccl.h k_ccl_tile / k_ccl_seam / k_ccl_fold)
        const bool tiles = (W & 63) == 0 && ctx->dbg.ccl_tiles != 0;
        hipLaunchKernelGGL(k_heat_to_u8<>, dim3(nblk(npix, 256, 2048)), dim3(256), 0, s, heat, npix, W, st, threshold, avg_u8, binary,
                           lazy ? (unsigned long long *)nullptr : (unsigned long long *)dev_bin, lazy ? (uint8_t *)nullptr : dev_bin + nwords * 8,
                           d_bits, tiles ? (int *)nullptr : d_label, d_box, d_cnt, tile_const);
        LAUNCH_CHECK();
        if (tiles) {
            const int ntx = W >> 6, nty = (H + CCL_TILE_ROWS - 1) / CCL_TILE_ROWS, ntiles = ntx * nty;
            int *d_troots = nullptr, *d_tile_n = nullptr;
            RM_TRY(ws(ctx, "ccl_tile_roots", (size_t)ntiles * CCL_TILE_CAP, &d_troots));
            RM_TRY(ws(ctx, "ccl_tile_n", (size_t)ntiles, &d_tile_n));
            // waves per tile workgroup: 16 (two rows each) while one round of workgroups covers the image -- the kernel's time is then one
            // workgroup's life (1080p noise 1.0475 / 1.0532 ms with 16 / 8, 720p 0.4634 / 0.4644) --, 8 (four 25 KB workgroups per CU instead
            // of two) where the tiles come in several rounds (4K x 512: 5.119 / 5.097 ms)
            const int tw = ctx->dbg.ccl_tile_waves > 0 ? ctx->dbg.ccl_tile_waves : (ntiles >= 2048 ? 8 : 16);
            if (tw == 4) hipLaunchKernelGGL(k_ccl_tile<4>, dim3((unsigned)ntx, (unsigned)nty), dim3(256), 0, s, d_bits, H, W, d_label, d_box, d_troots, d_tile_n);
            else if (tw == 8) hipLaunchKernelGGL(k_ccl_tile<8>, dim3((unsigned)ntx, (unsigned)nty), dim3(512), 0, s, d_bits, H, W, d_label, d_box, d_troots, d_tile_n);
            else hipLaunchKernelGGL(k_ccl_tile<16>, dim3((unsigned)ntx, (unsigned)nty), dim3(1024), 0, s, d_bits, H, W, d_label, d_box, d_troots, d_tile_n);
            LAUNCH_CHECK();
            const size_t nseam = (size_t)((H - 1) / CCL_TILE_ROWS) * W + (size_t)H * ntx;
            hipLaunchKernelGGL(k_ccl_seam<>, dim3((unsigned)((nseam + 255) / 256)), dim3(256), 0, s, d_bits, H, W, d_label);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_ccl_fold<>, dim3((unsigned)((ntiles + CCL_FOLD_TILES - 1) / CCL_FOLD_TILES)), dim3(256), 0, s, W, d_label, d_box, d_troots,
                               d_tile_n, ntiles, d_cnt, d_list, (unsigned int)comps_cap);
            LAUNCH_CHECK();
        } else {
        // (a thread per 64-bit word walking its set bits through the same rule was measured: 97 / 95 us instead of 21 / 19 at 720p --
        //  ten dependent find / atomic round trips per thread cost more than launching 84 % idle threads)
        const dim3 grid((unsigned)((npix + 255) / 256));
        hipLaunchKernelGGL(k_ccl_union<>, grid, dim3(256), 0, s, d_bits, npix, H, W, d_label);
        LAUNCH_CHECK();
        // per-tile LDS table of boxes unless the last extraction of this geometry met thousands of components (specks: see k_ccl_bbox)
        const bool table = ctx->dbg.ccl_table >= 0 ? ctx->dbg.ccl_table != 0 : !(same_geom && ctx->label_last_n > CCL_TABLE_MAX_COMPONENTS);
        const dim3 bgrid((unsigned)((W + 63) / 64), (unsigned)((H + CCL_BOX_ROWS - 1) / CCL_BOX_ROWS));
        if (table) hipLaunchKernelGGL(k_ccl_bbox<true>, bgrid, dim3(64 * CCL_BOX_ROWS), 0, s, d_bits, npix, H, W, d_label, d_box, d_cnt, d_list, (unsigned int)comps_cap);
        else hipLaunchKernelGGL(k_ccl_bbox<false>, bgrid, dim3(64 * CCL_BOX_ROWS), 0, s, d_bits, npix, H, W, d_label, d_box, d_cnt, d_list, (unsigned int)comps_cap);
        LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_ccl_publish<>, dim3(CCL_PUB_BLOCKS), dim3(256), 0, s, d_list, d_box, W, d_cnt, (unsigned int)comps_cap, dev_comps,
                           lazy ? d_complist : dev_comps + 1 + 2 * CCL_PUB_BLOCKS);
        LAUNCH_CHECK();
    } else if ((W & 63) == 0 && (W >> 6) <= HR_MAXW && ctx->dbg.heat_rows) {
        // rows of whole words: a workgroup per row, one record per row with foreground beside the packed image (k_heat_rows_u8)
        rows = true;
        hipLaunchKernelGGL(k_heat_rows_u8<>, dim3((unsigned)H), dim3(256), 0, s, heat, H, W, st, threshold, avg_u8, binary, (unsigned long long *)dev_bin,
                           (unsigned long long *)(dev_bin + rec_off), tile_const);
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_heat_to_u8<>, dim3(nblk(npix, 256, 2048)), dim3(256), 0, s, heat, npix, W, st, threshold, avg_u8, binary,
                           (unsigned long long *)dev_bin, dev_bin + nwords * 8, (unsigned long long *)nullptr, (int *)nullptr,
                           (CclBox *)nullptr, (unsigned int *)nullptr, tile_const);
        LAUNCH_CHECK();
    }
    delete pt_roi; pt_roi = nullptr;
    // From here on the pinned area of the slot is NOT known to be clean: a kernel that writes it is enqueued.  roi_finish puts the
    // marker back together with the exact range it found written; if it never runs for this launch (a failed wait, an abandoned
    // ticket, an error return above) the next roi_launch on the slot clears everything (stale foreground bits and records
    // would be OR-ed into the next image -- the kernels store only non-zero words).
    rs.h_rows_dirty = nullptr;
    pd.H = H; pd.W = W; pd.slot = ctx->cur_slot; pd.nwords = nwords; pd.comps_cap = comps_cap; pd.label = label; pd.clip = clip;
    pd.rows = rows; pd.rec_off = rec_off;
    pd.lazy = lazy; pd.stream = s; pd.d_bits = d_bits; pd.d_list = d_complist;
    return RM_OK;
}

int roi_finish(rm_ctx *ctx, const RoiPending &pd, int32_t *xywh)
{
    RoiSlot &rs = ctx->slots[pd.slot];
    const int H = pd.H, W = pd.W;
    const size_t nwords = pd.nwords, comps_cap = pd.comps_cap;
    const bool label = pd.label, clip = pd.clip;
    uint8_t *h_rows = rs.h_bin + nwords * 8;
    RoiResult r{};
    {
        const auto t0 = std::chrono::steady_clock::now();
        int y0 = H, y1 = -1;   // rows that hold foreground
        bool settled = false;
        if (pd.rows) {
            // one record per row with foreground: the one-blob rule needs nothing else (the image is read only when it fails)
            settled = simple_shape_row_records((const uint64_t *)(rs.h_bin + pd.rec_off), H, W, &y0, &y1, &r) && ctx->dbg.host_simple_shape && !clip;
            rs.dirty_w1 = 0; rs.dirty_w0 = 1;
            if (y1 >= y0) {   // what the next launch on this slot zeroes (roi_launch): the words of rows y0 .. y1 and their records
                rs.dirty_w0 = ((size_t)y0 * W) >> 6; rs.dirty_w1 = (((size_t)(y1 + 1) * W) - 1) >> 6;
                rs.dirty_r0 = y0; rs.dirty_r1 = y1;
            }
        } else {
            for (int y = 0; y < H; ++y)
                if (h_rows[y]) { if (y < y0) y0 = y; y1 = y; h_rows[y] = 0; }
        }
        if (clip && y1 >= y0) {
            // OpenCV <= 3.1: the 1-pixel image frame is zeroed before tracing (the host copy is ours to change)
            uint64_t *hb = (uint64_t *)rs.h_bin;
            auto clear_bit = [&](size_t p) { hb[p >> 6] &= ~(1ull << (p & 63)); };
            for (int y = y0; y <= y1; ++y) {
                const size_t r0 = (size_t)y * W;
                if (y == 0 || y == H - 1) { for (int x = 0; x < W; ++x) clear_bit(r0 + x); }
                else { clear_bit(r0); clear_bit(r0 + W - 1); }
            }
        }
        const size_t ncomp = label ? (size_t)(unsigned int)rs.h_comps[0].root : 0;
        ctx->label_used = label && ncomp <= comps_cap;
        ctx->roi_path = RM_ROI_PATH_ONE_BLOB;
        if (pd.lazy) {
            // the summaries alone first; image and list only when they leave the winner open (or the list overflowed: full scan)
            if (ctx->label_used && ctx->dbg.host_area_bound &&
                labelled_tops_settled((const LabelComp *)(rs.h_comps + 1), CCL_PUB_BLOCKS, W, ncomp, &r)) {
                settled = true;
                ctx->roi_path = RM_ROI_PATH_AREA_BOUND;
            } else {
                HIP_TRY(hipMemcpyAsync(rs.h_bin, pd.d_bits, nwords * 8, hipMemcpyDeviceToHost, pd.stream));
                if (ctx->label_used && ncomp > 0)
                    HIP_TRY(hipMemcpyAsync(rs.h_comps + 1 + 2 * CCL_PUB_BLOCKS, pd.d_list, ncomp * sizeof(CclComp), hipMemcpyDeviceToHost, pd.stream));
                HIP_TRY(stream_wait(pd.stream));
                y0 = 0; y1 = H - 1;   // (the whole image arrived: all of it is put back to zero below)
            }
        }
        if (settled) {
        } else if (ctx->label_used) {  // (an overflowing record list falls through to the full scan: the image is here either way)
            largest_external_contour_labelled_tops((const uint64_t *)rs.h_bin, H, W, (const LabelComp *)(rs.h_comps + 1), CCL_PUB_BLOCKS,
                                                   (const LabelComp *)(rs.h_comps + 1 + 2 * CCL_PUB_BLOCKS), ncomp, &r, ctx->dbg.host_area_bound != 0);
            ctx->roi_path = (r.found && r.area < 0.0) ? RM_ROI_PATH_AREA_BOUND : RM_ROI_PATH_LABELLED;
        } else if (!(ctx->dbg.host_simple_shape && y1 >= y0 && simple_shape_bits_rows((const uint64_t *)rs.h_bin, H, W, y0, y1, &r))) {
            largest_external_contour_bits_rows((const uint64_t *)rs.h_bin, H, W, y0, y1, &r);
            ctx->roi_path = RM_ROI_PATH_SCAN;
        }
        ctx->label_H = H; ctx->label_W = W; ctx->label_last_n = r.n_contours;
        ctx->label_lazy = label && ctx->roi_path == RM_ROI_PATH_AREA_BOUND;
        if (y1 >= y0 && !pd.rows) {   // restore the all-zero image: the words that cover rows y0 .. y1
            const size_t w0 = ((size_t)y0 * W) >> 6, w1 = (((size_t)(y1 + 1) * W) - 1) >> 6;
            std::memset(rs.h_bin + w0 * 8, 0, (w1 - w0 + 1) * 8);
        }
        rs.h_rows_dirty = h_rows;   // the slot is clean again but for the range recorded above (roi_launch took the marker away)
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->label_used) {
            ++ctx->label_streak;
            if ((long long)ncomp * 2 < (long long)ctx->label_unl_n) ctx->label_unl_steps = -1;   // a different kind of image: count the host stage's steps afresh
        } else if (!clip) {
            ctx->label_unl_steps = r.steps; ctx->label_unl_n = r.n_contours; ctx->label_streak = 0;
        }
        if (ctx->prof_on) ctx->prof_host_ms[3] += host_us * 1e-3;
    }
    if (!r.found) { xywh[0] = xywh[1] = xywh[2] = xywh[3] = 0; return RM_NO_CONTOUR; }
    xywh[0] = r.x; xywh[1] = r.y; xywh[2] = r.w; xywh[3] = r.h;
    return RM_OK;
}

int heatmap_to_roi_impl(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                               uint8_t *binary, void *stream, bool have_minmax)
{
    RoiPending pd;
    RM_TRY(roi_launch(ctx, heat, H, W, threshold, avg_u8, binary, stream, have_minmax, pd, xywh != nullptr));
    host_mark(ctx, 2);
    HIP_TRY(stream_wait((hipStream_t)stream));
    host_mark(ctx, 3);
    const int rc = roi_finish(ctx, pd, xywh);
    host_mark(ctx, 4);
    return rc;
}

// ------------------------------------------------------------------------------------------
// sparse heatmap exchange (kernels: k_sparse_*)
// ------------------------------------------------------------------------------------------
extern "C" size_t rm_heat_sparse_packet_doubles(int cap_tiles)
{
    return cap_tiles < 1 ? 0 : (size_t)SP_HDR + (size_t)cap_tiles + (size_t)cap_tiles * CT_H * CT_W;
}

extern "C" int rm_heat_sparse_pack(rm_ctx *ctx, const double *heat, int H, int W, int cap_tiles, double *packet, void *stream)
{
    if (!ctx || !heat || !packet || H < 1 || W < 1 || cap_tiles < 1) return fail(RM_E_BADARG, "rm_heat_sparse_pack: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, ntiles = tiles_x * tiles_y;
    if (ctx->nkept_H != H || ctx->nkept_W != W) {
        // no pruning bookkeeping for this heatmap (skip 0, zero result, foreign heatmap): report overflow -> dense exchange
        HIP_TRY(hipMemsetAsync(packet, 0, sizeof(double) * SP_HDR, s));
        HIP_TRY(hipMemsetAsync(packet, 0xff, sizeof(unsigned int), s));   // SP_DENSE_ONLY
        return RM_OK;
    }
    int *tile_nkept = nullptr;
    RM_TRY(ws(ctx, "tile_nkept", (size_t)ntiles, &tile_nkept));
    hipLaunchKernelGGL(k_sparse_background<>, dim3(1), dim3(64), 0, s, heat, W, tiles_x, ntiles, tile_nkept, cap_tiles, packet);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sparse_pack<>, dim3(ntiles), dim3(256), 0, s, heat, H, W, tiles_x, tile_nkept, cap_tiles, packet);
    LAUNCH_CHECK();
    return RM_OK;
}

extern "C" int rm_heat_sparse_merge_roi(rm_ctx *ctx, const double *packets, int world, int H, int W, int cap_tiles, int threshold,
                                        int avg_T, double *fused, int32_t *xywh, void *stream)
{
    if (!ctx || !packets || !fused || !xywh || world < 1 || H < 1 || W < 1 || cap_tiles < 1)
        return fail(RM_E_BADARG, "rm_heat_sparse_merge_roi: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ctx->device));
    RM_TRY(ctx_stream_ok(ctx, stream, __func__));
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, ntiles = tiles_x * tiles_y;
    const size_t pd = rm_heat_sparse_packet_doubles(cap_tiles);
    int *map = nullptr, *any = nullptr, *flag_dev = nullptr;
    RM_TRY(ws(ctx, "sparse_map", (size_t)world * ntiles, &map));
    RM_TRY(ws(ctx, "sparse_any", (size_t)ntiles, &any));
    if (!ctx->h_flag) {
        HIP_TRY(hipHostMalloc((void **)&ctx->h_flag, 2 * sizeof(int), hipHostMallocDefault));
        ctx->h_flag[0] = ctx->h_flag[1] = 0;
    }
    HIP_TRY(hipHostGetDevicePointer((void **)&flag_dev, ctx->h_flag, 0));
    hipLaunchKernelGGL(k_sparse_index<>, dim3(1), dim3(256), 0, s, packets, pd, world, cap_tiles, ntiles, map, any, flag_dev,
                       ctx->d_state, avg_T);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sparse_merge<>, dim3(ntiles), dim3(256), 0, s, packets, pd, world, cap_tiles, H, W, tiles_x, ntiles, map, any,
                       fused, ctx->d_state, avg_T);
    LAUNCH_CHECK();
    // the ROI stage synchronises the stream; the overflow flag is in pinned memory by then
    const int rc = heatmap_to_roi_impl(ctx, fused, H, W, threshold, xywh, nullptr, nullptr, stream, true);
    if (rc < 0) return rc;
    if (ctx->h_flag[0]) return RM_SPARSE_FALLBACK;
    return rc;
}

extern "C" int rm_heat_sparse_tiles_needed(rm_ctx *ctx, int *tiles)
{
    if (!ctx || !tiles) return fail(RM_E_BADARG, "rm_heat_sparse_tiles_needed: bad argument");
    *tiles = ctx->h_flag ? ctx->h_flag[1] : 0;
    return RM_OK;
}

extern "C" int rm_set_contour_labelling(rm_ctx *ctx, int mode)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_set_contour_labelling: ctx is NULL");
    ctx->label_mode = mode < 0 ? -1 : mode > 0 ? 1 : 0;
    return RM_OK;
}

extern "C" int rm_get_contour_labelling(rm_ctx *ctx, int *mode)
{
    if (!ctx || !mode) return fail(RM_E_BADARG, "rm_get_contour_labelling: bad argument");
    *mode = ctx->label_mode;
    return RM_OK;
}

extern "C" int rm_get_contour_clip_frame(rm_ctx *ctx, int *on)
{
    if (!ctx || !on) return fail(RM_E_BADARG, "rm_get_contour_clip_frame: bad argument");
    *on = ctx->clip_frame ? 1 : 0;
    return RM_OK;
}

extern "C" int rm_contour_stats(rm_ctx *ctx, int *n_components, int *labelled)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_contour_stats: ctx is NULL");
    if (n_components) *n_components = ctx->label_last_n;
    if (labelled) *labelled = ctx->label_used;
    return RM_OK;
}

extern "C" int rm_set_contour_clip_frame(rm_ctx *ctx, int on)
{
    if (!ctx) return fail(RM_E_BADARG, "rm_set_contour_clip_frame: ctx is NULL");
    ctx->clip_frame = on != 0;
    return RM_OK;
}

extern "C" int rm_heatmap_to_roi(rm_ctx *ctx, const double *heat, int H, int W, int threshold, int32_t *xywh, uint8_t *avg_u8,
                                 uint8_t *binary, void *stream)
{
    return heatmap_to_roi_impl(ctx, heat, H, W, threshold, xywh, avg_u8, binary, stream, false);
}

