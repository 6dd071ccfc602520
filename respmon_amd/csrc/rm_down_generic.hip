// respmon_amd/csrc/rm_down_generic.hip -- k_down_chain<uint8 / half / float, S>: narrow frame buffers whose shape the register kernels do not take
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_down_launch.h"

using namespace rm;

int launch_down_chain_generic(rm_ctx *ctx, const void *frames, int dtype, int T, const std::vector<int> &h, const std::vector<int> &w, int S, int vec_ok,
                              double *out, hipStream_t s, bool tiny)
{
    switch (dtype) {
    case RM_U8: return launch_down_chain_t<uint8_t>(ctx, frames, T, h, w, S, vec_ok, out, s, tiny);
    case RM_F16: return launch_down_chain_t<__half>(ctx, frames, T, h, w, S, vec_ok, out, s, tiny);
    case RM_F32: return launch_down_chain_t<float>(ctx, frames, T, h, w, S, vec_ok, out, s, tiny);
    }
    return fail(RM_E_BADARG, "unknown dtype %d", dtype);
}
