// respmon_amd/csrc/rm_flow.h -- optical-flow motion extraction (base.py:360-407): placeholder
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/respmon_hip.h"
namespace rm {
struct FlowWorkspace {};
inline int flow_good_features(FlowWorkspace &, const uint8_t *, int, int, int, double, double, int, float *, int *, hipStream_t, std::string &e) { e = "not built yet"; return RM_E_UNSUPPORTED; }
inline int flow_pyr_lk(FlowWorkspace &, const uint8_t *, const uint8_t *, int, int, const float *, int, int, int, int, int, double, float *, uint8_t *, hipStream_t, std::string &e) { e = "not built yet"; return RM_E_UNSUPPORTED; }
inline int flow_mean(FlowWorkspace &, const float *, const float *, const uint8_t *, int, float *, int *, hipStream_t, std::string &e) { e = "not built yet"; return RM_E_UNSUPPORTED; }
inline int flow_pca(FlowWorkspace &, const float *, int, double *, hipStream_t, std::string &e) { e = "not built yet"; return RM_E_UNSUPPORTED; }
}
