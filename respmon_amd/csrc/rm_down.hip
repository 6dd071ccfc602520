// respmon_amd/csrc/rm_down.hip -- which fused pyrDown chain reads the frame buffer (rm_down_chain.h, rm_down_chain_u8.h)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_internal.h"

using namespace rm;

// the three units that hold the kernels (each its own translation unit: they are the bulk of the library's compile time)
int launch_down_chain_f64(rm_ctx *ctx, const void *frames, int T, const std::vector<int> &h, const std::vector<int> &w, int S, int vec_ok, double *out,
                          hipStream_t s, bool tiny);                                  // rm_down_f64.hip
int launch_down_chain_generic(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S, int vec_ok,
                              double *out, hipStream_t s, bool tiny);                 // rm_down_generic.hip
int launch_down_chain_narrow(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S, double *out,
                             hipStream_t s, bool tiny);                               // rm_down_narrow.hip; 1 = not applicable
int launch_down_chain_bgr(rm_ctx *ctx, const void *frames, int T, const std::vector<int> &h, const std::vector<int> &w, int S, double *out,
                          hipStream_t s, bool tiny);                                  // rm_down_bgr.hip; 1 = not applicable

// frames[T,H,W] -> G_S[T,h_S,w_S] in one launch
int launch_down_chain(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S,
                             double *out, hipStream_t s, bool tiny)
{
    const int V = dtype_vec(dtype);
    const size_t esz = dtype_size(dtype);
    const bool vec_ok = (w[0] % V == 0) && (((size_t)h[0] * w[0] * esz) % 16 == 0) && (((uintptr_t)frames) % 16 == 0);
    if (S < 1 || S > 5) return fail(RM_E_UNSUPPORTED, "fused pyrDown chain supports 1..5 levels, got %d", S);
    const int vo = vec_ok ? 1 : 0;
    if (dtype == RM_BGR8) {
        // [T,H,W,3] uint8: the register chain converts while it unpacks (rm_down_chain_u8.h bgr8_t); shapes it does not take are
        // converted as a whole first and continue as a gray uint8 buffer
        if (vec_ok && !ctx->dbg.dc_lds_front_end && !ctx->dbg.bgr_unfused) {
            const int rc = launch_down_chain_bgr(ctx, frames, T, h, w, S, out, s, tiny);
            if (rc != 1) return rc;
        }
        const void *gray = nullptr;
        RM_TRY(bgr_buffer_to_gray(ctx, frames, (size_t)T * h[0] * w[0], &gray, s));
        return launch_down_chain(ctx, gray, RM_U8, T, h, w, S, out, s, tiny);
    }
    if ((dtype == RM_U8 || dtype == RM_F16 || dtype == RM_F32) && vec_ok && !ctx->dbg.dc_lds_front_end) {
        // all-register variant for narrow frame buffers (rm_down_chain_u8.h): a lane owns 16 adjacent pixels
        const int rc = launch_down_chain_narrow(ctx, frames, dtype, T, h, w, S, out, s, tiny);
        if (rc != 1) return rc;   // 1: no narrow geometry for this shape -- the generic chain below takes it
    }
    if (dtype == RM_F64) return launch_down_chain_f64(ctx, frames, T, h, w, S, vo, out, s, tiny);
    return launch_down_chain_generic(ctx, frames, dtype, T, h, w, S, vo, out, s, tiny);
}
