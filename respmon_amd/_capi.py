"""
ctypes declarations for the C-ABI in include/respmon_hip.h and the loader of the HIP library.

The product path is librespmon_hip.so built by hipcc for gfx950 -- there is NO CPU fallback:
`load()` raises if the library is missing or cannot be loaded.
"""
import ctypes
import os

RM_OK = 0
RM_NO_CONTOUR = 1
RM_SPARSE_FALLBACK = 2
RM_COMM_ID_BYTES = 128
RM_EXCHANGE_SPARSE, RM_EXCHANGE_DENSE = 1, 2
RM_E_BADARG = -1
RM_E_COMM = -6
RM_E_BUSY = -7
RM_LOCATE_TICKETS = 2
RM_U8, RM_F16, RM_F32, RM_F64 = 0, 1, 2, 3
RM_BGR8 = 4   # frame buffers only: [T,H,W,3] uint8 in cv2.VideoCapture's channel order (include/respmon_hip.h)
RM_FLAG_NO_PRUNE = 1
RM_FLAG_UNFUSED_DOWN = 2
RM_FLAG_TINY_STORE = 4
RM_FLAG_TINY_STRIPS = 8
RM_FLAG_UNFUSED_SMALL = 16
RM_FLAG_CONTOUR_CLIP_FRAME = 32
RM_FLAG_FILTER_LAPLACIANS = 64
RM_FLAG_DENSE_SUM = 128
RM_FLAG_SPARSE_SUM = 256
RM_FLAG_FF_PER_LEVEL = 512

_c = ctypes
_vp, _i, _d, _sz, _u = _c.c_void_p, _c.c_int, _c.c_double, _c.c_size_t, _c.c_uint

# name -> (restype, argtypes); every symbol include/respmon_hip.h and include/respmon_hip_debug.h declare
SIGNATURES = {
    "rm_ctx_create": (_i, [_i, _c.POINTER(_vp)]),
    "rm_ctx_destroy": (_i, [_vp]),
    "rm_last_error_string": (_c.c_char_p, []),
    "rm_abi_version": (_i, []),
    "rm_ctx_workspace_bytes": (_sz, [_vp]),
    "rm_debug_set": (_i, [_vp, _c.c_char_p, _c.c_longlong]),
    "rm_profile_enable": (_i, [_vp, _i]),
    "rm_set_contour_clip_frame": (_i, [_vp, _i]),
    "rm_set_contour_labelling": (_i, [_vp, _i]),
    "rm_get_contour_clip_frame": (_i, [_vp, _c.POINTER(_i)]),
    "rm_get_contour_labelling": (_i, [_vp, _c.POINTER(_i)]),
    "rm_contour_stats": (_i, [_vp, _vp, _vp]),
    "rm_profile_read": (_i, [_vp, _vp, _c.POINTER(_i)]),
    "rm_debug_counters": (_i, [_vp, _vp, _vp]),
    "rm_debug_workspace": (_i, [_vp, _c.c_char_p, _vp, _c.c_size_t, _vp]),
    "rm_debug_kernel_source_stamp": (_c.c_char_p, []),
    "rm_debug_host_timeline": (_i, [_vp, _vp]),
    "rm_debug_roi_path": (_i, [_vp, _vp]),
    "rm_uint8_to_float": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "rm_float_to_uint8": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "rm_pyr_down": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rm_pyr_up": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "rm_create_laplacian_video_pyramid": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _c.POINTER(_vp), _vp]),
    "rm_collapse_laplacian_video_pyramid": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _vp, _vp]),
    "rm_temporal_bandpass_filter_fft": (_i, [_vp, _vp, _i, _sz, _d, _d, _d, _d, _vp, _vp]),
    "rm_temporal_operator": (_i, [_i, _d, _d, _d, _vp, _c.POINTER(_i), _c.POINTER(_i)]),
    "rm_time_average": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp]),
    "rm_lfilter": (_i, [_vp, _vp, _i, _sz, _vp, _vp, _i, _d, _vp, _vp]),
    "rm_threshold_mask": (_i, [_vp, _vp, _sz, _d, _vp, _vp, _vp]),
    "rm_eulerian_magnification_bandpass": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _vp, _vp, _vp, _vp]),
    "rm_calibrate": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _u, _vp, _vp, _vp]),
    "rm_heatmap_to_roi": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rm_heat_sparse_packet_doubles": (_sz, [_i]),
    "rm_heat_sparse_pack": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "rm_heat_sparse_merge_roi": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rm_heat_sparse_tiles_needed": (_i, [_vp, _vp]),
    "rm_locate": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _i, _u, _vp, _vp]),
    "rm_locate_submit": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _i, _u, _vp, _vp]),
    "rm_locate_result": (_i, [_vp, _i, _vp]),
    "rm_shard_layout": (_i, [_i, _i, _i, _i, _c.POINTER(_sz)]),
    "rm_shard_layout_flags": (_i, [_i, _i, _i, _i, _c.c_uint, _c.POINTER(_sz)]),
    "rm_shard_pyramid": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _u, _vp, _vp]),
    "rm_shard_collapse": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _u, _vp, _vp]),
    "rm_shard_heat": (_i, [_vp, _vp, _d, _vp, _vp]),
    "rm_shard_finish": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rm_roi_mean": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rm_roi_to_uint8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rm_good_features_to_track": (_i, [_vp, _vp, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp]),
    "rm_calc_optical_flow_pyr_lk": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    "rm_flow_state_create": (_i, [_vp, _vp]),
    "rm_flow_state_destroy": (_i, [_vp]),
    "rm_flow_begin": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp]),
    "rm_flow_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    "rm_flow_points": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "rm_mean_flow": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "rm_pca_reduce": (_i, [_vp, _vp, _i, _vp, _vp]),
    "rm_bgr_to_gray": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "rm_comm_unique_id": (_i, [_vp]),
    "rm_comm_init": (_i, [_vp, _i, _i, _vp]),
    "rm_comm_destroy": (_i, [_vp]),
    "rm_comm_info": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "rm_shard_frames": (_i, [_i, _i, _i, _c.POINTER(_i), _c.POINTER(_i)]),
    "rm_locate_streams": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _i, _u, _vp, _vp, _c.POINTER(_i), _vp]),
    "rm_locate_sharded": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _i, _d, _i, _u, _vp, _vp, _c.POINTER(_i), _vp]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librespmon_hip.so")


class RespmonError(RuntimeError):
    pass


def bind(cdll):
    """Attach restype/argtypes for every declared symbol; raises AttributeError if one is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_LIB = None


def load():
    """Load the gfx950 HIP library.  Fails loudly: there is no other implementation to fall back to."""
    global _LIB
    if _LIB is None:
        # PyTorch-ROCm ships its own HIP / HSA runtime libraries (torch/lib) under the same SONAMEs as /opt/rocm's.
        # They must be in the process BEFORE this library is opened, so that its libamdhip64.so.7 dependency resolves
        # to the runtime that owns torch's device memory and streams; loaded the other way round the process ends
        # up with two runtimes and the second one finds "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RespmonError("HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` or `make -C respmon_amd/csrc`)" % LIB_PATH)
        try:
            _LIB = bind(ctypes.CDLL(LIB_PATH))
        except OSError as e:
            raise RespmonError("cannot load HIP extension %s: %s" % (LIB_PATH, e))
    return _LIB


def check(lib, rc, what):
    if rc < 0:
        raise RespmonError("%s failed (%d): %s" % (what, rc, lib.rm_last_error_string().decode("utf-8", "replace")))
    return rc
