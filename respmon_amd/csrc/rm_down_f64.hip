// respmon_amd/csrc/rm_down_f64.hip -- k_down_chain<double, S>: the frame-buffer kernel of the reference's float64 calibration_buffer (base.py:119)
// (one translation unit of librespmon_hip.so; shared host-side declarations: rm_internal.h)
#include "rm_down_launch.h"

using namespace rm;

int launch_down_chain_f64(rm_ctx *ctx, const void *frames, int T, const std::vector<int> &h, const std::vector<int> &w, int S, int vec_ok, double *out,
                          hipStream_t s, bool tiny)
{
    return launch_down_chain_t<double>(ctx, frames, T, h, w, S, vec_ok, out, s, tiny);
}
