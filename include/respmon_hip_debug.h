/*
 * include/respmon_hip_debug.h -- developer switches, test hooks and diagnostics of librespmon_hip.so.
 *
 * NOT part of the drop-in interface (include/respmon_hip.h): nothing a caller of the reference's path needs is declared here,
 * and a reference-side binding never includes this file.  tests/, bench.py's diagnostics legs and tools/ do.
 * Every switch selects between implementations with identical results, or shrinks a tuning constant so that a test reaches a
 * rare path; the library never reads the process environment.
 */
#ifndef RESPMON_HIP_DEBUG_H
#define RESPMON_HIP_DEBUG_H

#include "respmon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rm_calibrate / rm_locate flags that exist for tests only */
#define RM_FLAG_TINY_STORE 4u     /* a value store of 8 slots, so that nearly every selection overflows into the store-less sum kernel */
#define RM_FLAG_TINY_STRIPS 8u    /* 3-column strips / 2-row segments in the fused pyrDown chain */
#define RM_FLAG_FF_PER_LEVEL 512u /* the filter-first small pyramid with one launch per level although it would fit LDS */

/* per-context switches; unknown key -> RM_E_BADARG.  The keys and their meaning: struct DebugKnobs in respmon_amd/csrc/rm_internal.h
 * ("temporal_valu" 0|1, "dc_segs" n, "dense_split" 0|1|2|4, "store_slots" n, "exchange_dense" 0|1, ...).
 * "exchange_dense" changes which collective a step issues: set it identically on EVERY rank of a communicator. */
int rm_debug_set(rm_ctx *ctx, const char *key, long long value);

/* counters of the last rm_calibrate on this context: out_host[0] = (frame, tile) pairs, [1] = pairs evaluated at full resolution
 * by the selection's evaluation pass, [2] = pairs the selection kept for the masked sum, [3] = capacity of the value store (pairs);
 * 0 = the store-less sum kernel took the sum (it recomputes every pair itself, [1] then counts only the pairs evaluated for the
 * exact raw.min() / raw.max()) */
int rm_debug_counters(rm_ctx *ctx, long long *out_host, void *stream);

/* first 16 hex digits of the sha256 over the sources of the frame-buffer kernels this library was built from (csrc/Makefile
 * STAMP_SRCS, concatenated in that order).  The committed PMC figures (profiles/hbm_traffic.json, profiles/valu_issue.json) carry the
 * stamp of the library they were measured on; bench.py reports them only while the stamps agree. */
const char *rm_debug_kernel_source_stamp(void);

/* a copy of the first `bytes` bytes of the context's workspace buffer `name` as the last call left it ("tile_lo", "tile_hi": the
 * [unique frame][tile] bounds of the selection; "cS": the collapsed band-passed level) -- tests check the bounds themselves with it */
int rm_debug_workspace(rm_ctx *ctx, const char *name, void *out_host, size_t bytes, void *stream);

/* host timeline of the last rm_locate on this context, microseconds on the steady clock relative to the entry of that call:
 * out_host[0] = entry of the call minus the return of the PREVIOUS rm_locate (what the caller spent between two calls),
 * [1] = first kernel launch issued, [2] = every launch issued, [3] = device work seen complete, [4] = host contour stage done
 * (the call returns right after).  bench.py reports the medians as `host_timeline_us`. */
#define RM_HOST_MARKS 5
int rm_debug_host_timeline(rm_ctx *ctx, double *out_host);

/* how the host contour stage of the last ROI extraction on this context found its contour (base.py:568-575) */
#define RM_ROI_PATH_NONE 0          /* no extraction yet, or settled without the host stage */
#define RM_ROI_PATH_ONE_BLOB 1      /* the one-blob rule on the row records / packed rows */
#define RM_ROI_PATH_SCAN 2          /* every border followed on the host */
#define RM_ROI_PATH_LABELLED 3      /* device labelling: the borders that can win followed */
#define RM_ROI_PATH_AREA_BOUND 4    /* device labelling: the top component's area bound beat every rival's box, no border followed */
int rm_debug_roi_path(rm_ctx *ctx, int *path_out);

#ifdef __cplusplus
}
#endif
#endif /* RESPMON_HIP_DEBUG_H */
