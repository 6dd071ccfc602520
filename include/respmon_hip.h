/*
 * include/respmon_hip.h -- C-ABI of librespmon_hip.so, the MI355X (gfx950) implementation of
 * respmon's Eulerian-magnification calibration and ROI motion-extraction hot path.
 *
 * The reference (kevroy314/respmon) is pure Python and has no FFI layer; the "interface each
 * entry point replaces" is therefore the Python function (and the cv2 / scipy / numpy calls
 * under it) cited next to each declaration, paths relative to the reference root.
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer named *_dev is a DEVICE pointer (HBM) unless the
 *    comment says host; images are row-major, videos are [T,H,W] with frame stride H*W.
 *  - every function returns int: RM_OK (0) or a negative RM_E_* code; rm_last_error_string()
 *    gives the text for the calling thread's last failure.  RM_NO_CONTOUR (+1) is the
 *    reference's `return None` (base.py:569-570).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work on
 *    it; functions that return host results synchronise that stream before returning.
 *  - a context is not thread-safe; use one per stream / GPU.
 *  - developer switches, hooks for the tests and diagnostics counters are NOT part of this interface: include/respmon_hip_debug.h.
 */
#ifndef RESPMON_HIP_H
#define RESPMON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_OK 0
#define RM_NO_CONTOUR 1
#define RM_SPARSE_FALLBACK 2 /* rm_heat_sparse_merge_roi: a packet overflowed, use the dense all-reduce instead */
#define RM_E_BADARG (-1)
#define RM_E_HIP (-2)
#define RM_E_NOMEM (-3)
#define RM_E_UNSUPPORTED (-4)
#define RM_E_INTERNAL (-5)
#define RM_E_COMM (-6)     /* RCCL: library missing, communicator or collective failed */
#define RM_E_BUSY (-7)     /* rm_locate_submit: every ticket of the context is waiting for its rm_locate_result */

/* element type of a frame buffer handed to the library */
#define RM_U8 0  /* gray uint8; the kernels apply uint8_to_float's  k * (1./255)  (transforms.py:20-23) */
#define RM_F16 1 /* IEEE half, widened exactly */
#define RM_F32 2 /* float, widened exactly */
#define RM_F64 3 /* double: the reference's calibration_buffer dtype (base.py:119-120) */
/* Frame BUFFERS only -- the `frames` argument of rm_calibrate, rm_locate, rm_locate_submit, rm_locate_streams, rm_locate_sharded,
 * rm_shard_pyramid and rm_eulerian_magnification_bandpass: [T,H,W,3] uint8 in cv2.VideoCapture's channel order.  The kernels apply
 * base.py:230-231 (cv2.cvtColor(BGR2GRAY), then uint8_to_float) as they read the buffer: results are bit-identical to the same call on
 * the RM_U8 buffer that rm_bgr_to_gray makes of it.  Every other entry point takes gray frames and answers RM_E_BADARG. */
#define RM_BGR8 4

/* rm_calibrate flags */
#define RM_FLAG_NO_PRUNE 1u      /* evaluate every (tile, frame) of the collapse passes (A/B + verification) */
#define RM_FLAG_UNFUSED_DOWN 2u  /* build the Gaussian levels one pyrDown launch per level */
#define RM_FLAG_UNFUSED_SMALL 16u /* build / collapse the small pyramid with one launch per level (A/B + fallback path) */
#define RM_FLAG_CONTOUR_CLIP_FRAME 32u /* rm_locate: cv2.findContours as OpenCV <= 3.1 did it (see rm_set_contour_clip_frame) */
#define RM_FLAG_FILTER_LAPLACIANS 64u /* build the small pyramid in the reference's order -- Laplacians first, filter them, collapse (bit for bit equal to the per-level
                                         path) -- instead of the filter-first form (filter G_S, then Laplacians + collapse in one kernel; equal to ~1e-15) */
#define RM_FLAG_DENSE_SUM 128u    /* take the masked time sum with the dense kernel (recomputes every value, no value store) ... */
#define RM_FLAG_SPARSE_SUM 256u   /* ... or with the sparse path (evaluate + store the pairs that can fall below `top`).  Neither: decided on
                                     the device from what THIS call's selection kept (nothing is remembered between calls): dense when the
                                     kept pairs outnumber the value store's slots, or at skip <= 2 when more than half are kept.
                                     Bit-identical results. */

typedef struct rm_ctx rm_ctx;

/* ---- context -------------------------------------------------------------------------- */
int rm_ctx_create(int device, rm_ctx **out);
int rm_ctx_destroy(rm_ctx *ctx);
const char *rm_last_error_string(void);
int rm_abi_version(void);
/* bytes of device workspace currently held by the context */
size_t rm_ctx_workspace_bytes(const rm_ctx *ctx);

/* ---- measurement hook for bench.py: mode 1 brackets only the frame-buffer kernel with hipEvents on the
 *      caller's stream (cheap enough for the timed region), mode 2 brackets every phase of rm_calibrate /
 *      rm_heatmap_to_roi (each bracket costs ~10 us of stream idle time), mode 0 turns it off.  rm_profile_read waits for them and returns the summed
 *      milliseconds since the last read: ms_host[0] = the kernel that reads the [T,H,W] frame buffer
 *      (the roofline kernel), [1] = remaining pyramid + temporal kernels, [2] = collapse passes,
 *      [3] = heatmap -> ROI (device part + host contour stage); *n_host = rm_calibrate calls the sums cover
 *      (mode 2: every call; mode 1: every 8th call is bracketed, the others run without the two event records). */
#define RM_PROFILE_PHASES 4
int rm_profile_enable(rm_ctx *ctx, int mode);
int rm_profile_read(rm_ctx *ctx, double *ms_host, int *n_host);

/* ---- dtype helpers: transforms.py:20-23 uint8_to_float, transforms.py:26-29 float_to_uint8 */
int rm_uint8_to_float(rm_ctx *ctx, const uint8_t *src_dev, double *dst_dev, size_t n, void *stream);
int rm_float_to_uint8(rm_ctx *ctx, const double *src_dev, uint8_t *dst_dev, size_t n, void *stream);

/* ---- pyramid.py building blocks (materialising forms, API parity + tests) ------------- */
/* cv2.pyrDown on every frame of src[T,h,w] -> dst[T,(h+1)/2,(w+1)/2] float64.  pyramid.py:14 */
int rm_pyr_down(rm_ctx *ctx, const void *src_dev, int src_dtype, int T, int h, int w,
                double *dst_dev, void *stream);
/* cv2.pyrUp(src, dstsize=(dw,dh)) on every frame, fused with the reference's add/subtract:
 *   mode 0: dst = up(src)                      pyramid.py:55 with a zero level
 *   mode 1: dst = other - up(src)              pyramid.py:24-26  (Laplacian level)
 *   mode 2: dst = up(src) + other              pyramid.py:55     (collapse step)
 * `other_dev` is [T,dh,dw] (ignored for mode 0); dst may alias other. */
int rm_pyr_up(rm_ctx *ctx, const double *src_dev, int T, int sh, int sw, double *dst_dev, int dh, int dw,
              int mode, const double *other_dev, void *stream);
/* pyramid.py:31-48 create_laplacian_video_pyramid: level_ptrs_host[l] -> device [T,h_l,w_l] float64,
 * h_0=H, h_{l+1}=(h_l+1)/2.  All `levels` arrays are written. */
int rm_create_laplacian_video_pyramid(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W,
                                      int levels, double *const *level_ptrs_host, void *stream);
/* pyramid.py:60-69 collapse_laplacian_video_pyramid: result into out_dev[T,H,W] (may be level 0). */
int rm_collapse_laplacian_video_pyramid(rm_ctx *ctx, const double *const *level_ptrs_host, int T, int H, int W,
                                        int levels, double *out_dev, void *stream);

/* ---- transforms.py:82-102 temporal_bandpass_filter_fft -------------------------------- */
/* out[s,p] = amplification * sum_t M[s,t] data[t,p], M = the reference's packed-rfft / mask /
 * Re(ifft) operator (quirk kept).  data/out: [T, npix] float64. */
int rm_temporal_bandpass_filter_fft(rm_ctx *ctx, const double *data_dev, int T, size_t npix, double fps,
                                    double freq_min, double freq_max, double amplification,
                                    double *out_dev, void *stream);
/* the operator itself (host, row-major [T,T], without the amplification) and the band bounds */
int rm_temporal_operator(int T, double fps, double freq_min, double freq_max, double *M_host,
                         int *bound_low, int *bound_high);

/* ---- np.average(video, axis=0) (base.py:562, 579, 587, 589: heatmap and the panels of the calibration image):
 *      out_dev[npix] float64 = (sum over t, in t order, of data[t, :]) / T for a [T, npix] array of any frame dtype. */
int rm_time_average(rm_ctx *ctx, const void *data_dev, int dtype, int T, size_t npix, double *out_dev, void *stream);

/* ---- transforms.py:72-79 temporal_bandpass_filter, transforms.py:53-55 butter_bandpass_filter_fast:
 *      out = scipy.signal.lfilter(b, a, data, axis=0) * scale on data[T, npix] float64 (transposed direct form II,
 *      the operation order of scipy's C loop).  b_host / a_host: ncoef coefficients each (ncoef <= 16; the
 *      reference's order-6 Butterworth band-pass has 13), designed on the host with scipy.signal.butter exactly
 *      as transforms.py:38-44 does.  SURVEY 8f row f4. */
int rm_lfilter(rm_ctx *ctx, const double *data_dev, int T, size_t npix, const double *b_host, const double *a_host,
               int ncoef, double scale, double *out_dev, void *stream);

/* ---- transforms.py:184-192 on a materialised array: minmax_host = {raw.min(), raw.max()} (may be NULL);
 *      masked_dev (may be NULL) = raw with every value >= max - (max - min) * threshold replaced by min.
 *      Lets eulerian_magnification_bandpass run with ANY temporal_filter_function (transforms.py:146). */
int rm_threshold_mask(rm_ctx *ctx, const double *raw_dev, size_t n, double threshold, double *masked_dev,
                      double *minmax_host, void *stream);

/* ---- transforms.py:144-198 eulerian_magnification_bandpass (materialised outputs) ------ */
/* masked_dev / raw_dev: [T,H,W] float64 (either may be NULL); minmax_host[2] = {min, max} of raw. */
int rm_eulerian_magnification_bandpass(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W,
                                       double fps, double freq_min, double freq_max, double amplification,
                                       int pyramid_levels, int skip_levels_at_top, double threshold,
                                       double *masked_dev, double *raw_dev, double *minmax_host, void *stream);

/* ---- the fused calibration path: base.py:555-562 (eulerian ... np.average(op, axis=0)) -- */
/* Reads the frame buffer once; never materialises a [T,H,W] intermediate.  heatmap_dev[H*W]
 * receives np.average(masked, axis=0) (sequential-in-t float64 sum / T).  minmax_host (may be
 * NULL) receives {raw.min(), raw.max()}.  Asynchronous unless minmax_host != NULL. */
int rm_calibrate(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W,
                 double fps, double freq_min, double freq_max, double amplification,
                 int pyramid_levels, int skip_levels_at_top, double temporal_threshold,
                 unsigned flags, double *heatmap_dev, double *minmax_host, void *stream);

/* ---- base.py:563-575: normalise, float_to_uint8, threshold, findContours(EXTERNAL,SIMPLE),
 *      max contourArea, boundingRect.  avg_u8_dev / binary_dev (device, H*W, may be NULL) receive
 *      the uint8 heatmap and the thresholded image.  Returns RM_OK + xywh_host[4], or RM_NO_CONTOUR. */
int rm_heatmap_to_roi(rm_ctx *ctx, const double *heatmap_dev, int H, int W, int threshold,
                      int32_t *xywh_host, uint8_t *avg_u8_dev, uint8_t *binary_dev, void *stream);

/* ---- sparse exchange of per-stream heatmaps between GPUs (one stream per GPU, respmon_amd/dist.py::locate_streams).
 *      Not a reference function: it replaces the dense all-reduce(sum) of the [H,W] float64 heatmaps (16.6 MB per GPU
 *      at 1080p) by ONE all-gather of small packets.  The heatmap rm_calibrate leaves behind is a single constant (the
 *      time average of `min`, transforms.py:190-192 + base.py:562) in every 64x16 tile whose frames were all masked,
 *      so a packet holds that constant plus the values of the tiles that differ from it (cap_tiles of them at most).
 *        rm_heat_sparse_packet_doubles(cap)  packet length in doubles
 *        rm_heat_sparse_pack      packet_dev <- heatmap of the LAST rm_calibrate on this context
 *          -- all-gather the packets: packets_dev[world][packet] --
 *        rm_heat_sparse_merge_roi fused_dev[H*W] = sum over ranks, in rank order, of the per-rank heatmaps (avg_T = 0),
 *                                 or of the partial heat sums of a frame-sharded buffer divided by avg_T (rm_shard_heat
 *                                 outputs: the sparse form of the heat-sum all-reduce + rm_shard_finish), then
 *                                 base.py:563-575 on it -> xywh_host.  Returns RM_OK / RM_NO_CONTOUR, or
 *                                 RM_SPARSE_FALLBACK when some rank needed more than cap_tiles tiles (every rank sees
 *                                 the same packets, so every rank falls back to the dense all-reduce together).
 *        rm_heat_sparse_tiles_needed  the largest tile count any rank needed in the LAST rm_heat_sparse_merge_roi on this
 *                                 context (also when it overflowed): identical on every rank, so the ranks can size the
 *                                 next exchange's packets from it without talking to each other. */
size_t rm_heat_sparse_packet_doubles(int cap_tiles);
int rm_heat_sparse_pack(rm_ctx *ctx, const double *heat_dev, int H, int W, int cap_tiles, double *packet_dev, void *stream);
int rm_heat_sparse_merge_roi(rm_ctx *ctx, const double *packets_dev, int world, int H, int W, int cap_tiles, int threshold,
                             int avg_T, double *fused_dev, int32_t *xywh_host, void *stream);
int rm_heat_sparse_tiles_needed(rm_ctx *ctx, int *tiles_host);

/* ---- base.py:547-601 RespiratoryMonitor.locate = rm_calibrate + rm_heatmap_to_roi ------- */
int rm_locate(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W, double fps,
              double freq_min, double freq_max, double amplification, int pyramid_levels,
              int skip_levels_at_top, double temporal_threshold, int threshold, unsigned flags,
              int32_t *xywh_host, void *stream);

/* ---- the same in two calls, for back-to-back calibration buffers (base.py:547-601 called once per buffer: the streams of
 *      BASELINE config 4, the state machine's recalibrations).  rm_locate_submit enqueues the device work of rm_locate on `stream`
 *      and returns a ticket without waiting; rm_locate_result(ticket) waits for it, runs the host contour stage (base.py:568-575)
 *      and fills xywh_host -- same return values and bit-identical ROI as rm_locate.  A context holds up to RM_LOCATE_TICKETS
 *      submissions (RM_E_BUSY beyond), all on ONE stream; results may be fetched in any order.  Submitting buffer k+1 before
 *      fetching buffer k puts the frame-buffer kernel of k+1 where the synchronous call leaves the GPU idle (host wait + contour
 *      stage + launch latency of the next call).  frames_dev must stay valid and unchanged until the ticket's rm_locate_result
 *      (a selection that overflows the value store is taken again synchronously there); between a submit and its result the
 *      context takes no other calls than rm_locate_submit / rm_locate_result / rm_locate on the same stream. */
#define RM_LOCATE_TICKETS 2
int rm_locate_submit(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W, double fps,
                     double freq_min, double freq_max, double amplification, int pyramid_levels,
                     int skip_levels_at_top, double temporal_threshold, int threshold, unsigned flags,
                     void *stream, int *ticket_host);
int rm_locate_result(rm_ctx *ctx, int ticket, int32_t *xywh_host);

/* ---- frame-sharded calibration (one [T,H,W] buffer split by frame index over the GPUs of a node; SURVEY 8e
 *      "Mode A", BASELINE north_star).  Same arithmetic as rm_calibrate (transforms.py:144-198, base.py:562),
 *      cut at the three points where frames meet; the caller runs a collective at each cut
 *      (respmon_amd/dist.py::locate_sharded uses torch.distributed / RCCL):
 *        rm_shard_pyramid   frames[t0:t1] -> lap_local[(t1-t0), NP]   the Gaussian level `skip` of every frame (filter-first form, the
 *                           default where it fits LDS) or the Laplacian levels skip..L-2 (pyramid.py:20-48): see rm_shard_layout_flags
 *          -- all-gather lap_local -> lap_all[T, NP] --
 *        rm_shard_collapse  temporal band-pass + collapse of every frame (cheap, identical on every rank),
 *                           full-resolution evaluation of this rank's frames [t0,t1) only;
 *                           negmin_max_dev[2] (DEVICE) = { -min, max } of raw over those frames
 *          -- all-reduce(MAX) of negmin_max_dev --
 *        rm_shard_heat      heat_sum_dev[H*W] = sum_{t in [t0,t1)} (raw >= top ? min : raw)   (transforms.py:184-192)
 *          -- all-reduce(SUM) of heat_sum_dev: the single [H,W] heatmap collective --
 *        rm_shard_finish    heatmap = heat_sum / T (base.py:562), then base.py:563-575 -> xywh_host (may be NULL).
 *      With one rank (t0 = 0, t1 = T) the result equals rm_calibrate / rm_locate bit for bit; with several
 *      ranks the time sum is associated per shard (<= 1e-15 relative on the heatmap).
 *      rm_shard_layout returns NP (0 when no level is filtered).  Requires skip_levels_at_top >= 1. */
int rm_shard_layout(int H, int W, int pyramid_levels, int skip_levels_at_top, size_t *np_out);
/* ... for the `flags` the other stages will be given (RM_FLAG_FILTER_LAPLACIANS / RM_FLAG_UNFUSED_SMALL change what travels between the
 * stages: the Laplacian levels S .. L-2, NP = their pixel count, instead of the Gaussian level S, NP = h_S * w_S); rm_shard_layout = flags 0 */
int rm_shard_layout_flags(int H, int W, int pyramid_levels, int skip_levels_at_top, unsigned flags, size_t *np_out);
int rm_shard_pyramid(rm_ctx *ctx, const void *frames_local_dev, int dtype, int T_local, int H, int W, int pyramid_levels,
                     int skip_levels_at_top, unsigned flags, double *lap_local_dev, void *stream);
int rm_shard_collapse(rm_ctx *ctx, const double *lap_all_dev, int T, int t0, int t1, int H, int W, double fps,
                      double freq_min, double freq_max, double amplification, int pyramid_levels, int skip_levels_at_top,
                      double temporal_threshold, unsigned flags, double *negmin_max_dev, void *stream);
int rm_shard_heat(rm_ctx *ctx, const double *negmin_max_dev, double temporal_threshold, double *heat_sum_dev, void *stream);
int rm_shard_finish(rm_ctx *ctx, const double *heat_sum_dev, int T, int H, int W, int threshold, double *heatmap_dev,
                    int32_t *xywh_host, void *stream);

/* ---- base.py:355-358 + 471: extract_motion('average') = np.average(frame[y:y+h, x:x+w]) -- */
int rm_roi_mean(rm_ctx *ctx, const void *frame_dev, int dtype, int H, int W, int x, int y, int w, int h,
                double *out_host, void *stream);
/* ROI crop + float_to_uint8 (base.py:364, 371, 381): dst_dev[h*w] uint8 */
int rm_roi_to_uint8(rm_ctx *ctx, const void *frame_dev, int dtype, int H, int W, int x, int y, int w, int h,
                    uint8_t *dst_dev, void *stream);

/* ---- base.py:365-366 cv2.goodFeaturesToTrack(img_u8, mask=None, maxCorners, qualityLevel,
 *      minDistance, blockSize).  pts_host: float32 (x,y) pairs, capacity max_corners; *n_host = count
 *      (0 <=> cv2 returns None). */
int rm_good_features_to_track(rm_ctx *ctx, const uint8_t *img_dev, int h, int w, int max_corners,
                              double quality_level, double min_distance, int block_size,
                              float *pts_host, int *n_host, void *stream);

/* ---- base.py:371-372 cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize, maxLevel,
 *      criteria=(EPS|COUNT, max_count, epsilon)).  pts in/out: host float32 (x,y) pairs; status host u8. */
int rm_calc_optical_flow_pyr_lk(rm_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int h, int w,
                                const float *pts_in_host, int npts, int win_w, int win_h, int max_level,
                                int max_count, double epsilon, float *pts_out_host, uint8_t *status_host,
                                void *stream);

/* ---- base.py:388: np.mean(good_old - good_new, axis=0) (float32, sequential over points) and
 *      base.py:396-405: np.cov -> np.linalg.eig -> argsort desc -> ROW unpack -> projection[-1].
 *      motion_host: float32 [n,2] (the motion_data deque); *out_host = the value extract_motion returns. */
int rm_mean_flow(rm_ctx *ctx, const float *old_host, const float *new_host, const uint8_t *status_host,
                 int npts, float *mean_xy_host, int *n_good_host, void *stream);
int rm_pca_reduce(rm_ctx *ctx, const float *motion_host, int n, double *out_host, void *stream);

/* ---- base.py:363-388 as ONE call per frame (SURVEY 8b "rm_flow_step"): the ROI crop, its pyramids, the tracked points and
 *      their status never leave the device between two frames.
 *      rm_flow_begin = base.py:364-366: crop + float_to_uint8 of the ROI, cv2.goodFeaturesToTrack on it; the corners are
 *        returned (pts_host, capacity max(max_corners, 1) float32 pairs; *n_host = 0 <=> cv2 returns None) and kept on the device.
 *      rm_flow_step  = base.py:371-388: crop of the new frame, cv2.calcOpticalFlowPyrLK from the previous crop and the kept
 *        points, np.mean(good_old - good_new, axis=0) over status == 1 (float32, point order) -> mean_xy_host[2], *n_good_host;
 *        the new crop and p1[st == 1] become the state the next call starts from (base.py:381-382).  With no points left the
 *        call only advances the previous image (mean 0, n_good 0: the caller returns NaN, base.py:385-386).
 *      rm_flow_points: the points the next step will track (what the reference holds in self.motion_key_points).
 *      Bit-identical to rm_roi_to_uint8 + rm_calc_optical_flow_pyr_lk + rm_mean_flow called in turn. */
/*      The session's state -- previous crop, its LK pyramid and derivatives, the tracked points -- lives in an rm_flow_state the
 *      caller owns (one per RespiratoryMonitor: self.previous_cropped_image / self.motion_key_points of base.py:111-112 are
 *      per object), so several monitors share a GPU and an rm_ctx without seeing each other's tracking.  What a step builds for
 *      the new crop is kept for the next step, which tracks FROM it: one pyramid and one set of derivatives per frame. */
typedef struct rm_flow_state rm_flow_state;
int rm_flow_state_create(rm_ctx *ctx, rm_flow_state **out);
int rm_flow_state_destroy(rm_flow_state *state);
int rm_flow_begin(rm_ctx *ctx, rm_flow_state *state, const void *frame_dev, int dtype, int H, int W, int x, int y, int w, int h, int max_corners,
                  double quality_level, double min_distance, int block_size, float *pts_host, int *n_host, void *stream);
int rm_flow_step(rm_ctx *ctx, rm_flow_state *state, const void *frame_dev, int dtype, int H, int W, int x, int y, int w, int h, int win_w, int win_h,
                 int max_level, int max_count, double epsilon, float *mean_xy_host, int *n_good_host, void *stream);
int rm_flow_points(rm_ctx *ctx, rm_flow_state *state, float *pts_host, int cap, int *n_host, void *stream);

/* ---- multi-GPU steps with RCCL behind the C-ABI (the call site they replace is base.py:444, run once per GPU).
 *      One process per GPU, one context per process.  librccl is opened at run time (dlopen), so single-GPU users never need it.
 *        rm_comm_unique_id   rank 0 makes the id (RM_COMM_ID_BYTES bytes = ncclUniqueId) and hands it to the other ranks by any
 *                            out-of-band way (a file, MPI, a torch.distributed broadcast: INTEGRATION.md)
 *        rm_comm_init        ncclCommInitRank on the context's device.  world == 1 with unique_id == NULL: no library involved,
 *                            the collectives are the identity (rm_locate_streams then equals rm_locate)
 *        rm_comm_info        rank / world of the context and ncclCommCount of its communicator (0: none)
 *        rm_shard_frames     Mode A frame shard [t0, t1) of `rank`: contiguous, sizes differ by at most one
 *        rm_locate_streams   Mode B (BASELINE config 4): every rank calibrates ITS OWN [T,H,W] buffer; the float64 heatmaps are
 *                            summed over the ranks in rank order -- one ncclAllGather of sparse packets (a stream's heatmap is one
 *                            constant outside the tiles that survive the pruning), dense ncclAllReduce(sum) when a packet
 *                            overflows (the cap then grows / the exchange stays dense for a while, identically on every rank) --
 *                            and every rank extracts the same ROI.  fused_heat_dev (nullable): the summed heatmap.
 *        rm_locate_sharded   Mode A: ONE [T,H,W] buffer whose frames rm_shard_frames(T, rank, world) are frames_local_dev on this
 *                            rank: pyramid of the local frames -> ncclAllGather -> temporal filter + collapse + pruning for all
 *                            frames -> ncclAllReduce(max) of {-min, max} -> masked time sum of the local frames -> sparse packets
 *                            / dense all-reduce of the partial sums -> heatmap = sum / T -> ROI (every rank the same).
 *      Both enqueue on `stream` and synchronise the host once (twice after a packet overflow).  *exchange_out (nullable):
 *      RM_EXCHANGE_SPARSE / RM_EXCHANGE_DENSE.  Return RM_OK / RM_NO_CONTOUR like rm_locate; RM_E_COMM on an RCCL failure. */
#define RM_COMM_ID_BYTES 128
#define RM_EXCHANGE_SPARSE 1
#define RM_EXCHANGE_DENSE 2
#define RM_SPARSE_CAP_TILES 128   /* tiles (64x16 px) a packet carries at first: 1 MB per rank */
#define RM_SPARSE_MAX_TILES 512   /* ... and at most: 4 MB */
#define RM_DENSE_HOLD 64          /* steps a stream that does not fit stays on the dense all-reduce before the sparse form is tried again */
int rm_comm_unique_id(void *id_out);
int rm_comm_init(rm_ctx *ctx, int rank, int world, const void *unique_id);
int rm_comm_destroy(rm_ctx *ctx);
int rm_comm_info(rm_ctx *ctx, int *rank, int *world, int *rccl_ranks);
int rm_shard_frames(int T, int rank, int world, int *t0, int *t1);
int rm_locate_streams(rm_ctx *ctx, const void *frames_dev, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp,
                      int levels, int skip, double temporal_thr, int threshold, unsigned flags, double *fused_heat_dev, int32_t *xywh_host,
                      int *exchange_out, void *stream);
int rm_locate_sharded(rm_ctx *ctx, const void *frames_local_dev, int dtype, int T, int H, int W, double fps, double fmin, double fmax, double amp,
                      int levels, int skip, double temporal_thr, int threshold, unsigned flags, double *heatmap_dev, int32_t *xywh_host,
                      int *exchange_out, void *stream);

/* ---- base.py:230-231: cv2.cvtColor(BGR2GRAY) then uint8_to_float, on device ("next" row f3) */
int rm_bgr_to_gray(rm_ctx *ctx, const uint8_t *bgr_dev, size_t npix, uint8_t *gray_dev, void *stream);

/* ---- base.py:567-568: which cv2.findContours the ROI stage reproduces.  The reference pins no OpenCV version (README.md:12
 *      installs "opencv3" from conda channel menpo; base.py:567 keeps a `thresh_copy` because the version its author ran MUTATED
 *      the input).  OpenCV <= 3.1 zeroes the 1-pixel image frame in place before tracing, so components are clipped to
 *      [1, W-2] x [1, H-2]; OpenCV >= 3.2 traces on a zero-padded copy and frame pixels count.
 *      on = 0 (default): the >= 3.2 rule.  on = 1: the <= 3.1 rule, for every later ROI extraction of this context
 *      (rm_heatmap_to_roi, rm_locate, rm_shard_finish, rm_heat_sparse_merge_roi); rm_locate also takes it per call as
 *      RM_FLAG_CONTOUR_CLIP_FRAME. */
int rm_set_contour_clip_frame(rm_ctx *ctx, int on);
int rm_get_contour_clip_frame(rm_ctx *ctx, int *on);   /* so that a per-call override can put back what the context had */

/* ---- base.py:568-575 on noisy thresholded images.  locate() keeps ONE contour (max contourArea -> boundingRect); when the
 *      image holds thousands of specks (BASELINE configs 2 / 5) the device labels the 8-connected components, reduces their
 *      bounding boxes and the host follows only the borders whose (w-1)*(h-1) bound can reach the best area -- same contour,
 *      same tie rule, no per-speck border following (csrc/rm_ccl.h).
 *      mode = -1 (default): taken when the previous ROI extraction of this geometry met more than 512 components, or when its
 *      host stage walked more than 24 000 border steps (few components with long borders, e.g. a frame of noise blobs; the
 *      host-only stage runs again every 64th call to refresh the count).  Counts only, no clock: a stream takes the same path in
 *      every run;
 *      0: never (every border is followed on the host); 1: always.  The ROI does not depend on the mode.
 *      rm_contour_stats: components / contours met by the last ROI extraction and whether it ran labelled (diagnostics). */
int rm_set_contour_labelling(rm_ctx *ctx, int mode);
int rm_get_contour_labelling(rm_ctx *ctx, int *mode);
int rm_contour_stats(rm_ctx *ctx, int *n_components, int *labelled);

#ifdef __cplusplus
}
#endif
#endif /* RESPMON_HIP_H */
