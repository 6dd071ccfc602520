"""Device plumbing: PyTorch-ROCm tensors are only the device-memory container; every computation
goes through the C-ABI of librespmon_hip.so (respmon_amd/_capi.py)."""
import ctypes

import numpy as np

from . import _capi

_CTX = {}

_NP_CODES = {np.dtype(np.uint8): _capi.RM_U8, np.dtype(np.float16): _capi.RM_F16,
             np.dtype(np.float32): _capi.RM_F32, np.dtype(np.float64): _capi.RM_F64}


def torch():
    import torch as _t
    return _t


def require_gpu():
    t = torch()
    if not t.cuda.is_available():
        raise _capi.RespmonError("respmon_amd needs an MI355X (HIP) device: torch.cuda.is_available() is False "
                                 "and there is no CPU implementation of the hot path")
    return t


def ctx(device_index=None):
    """One library context per GPU (lazily created)."""
    t = require_gpu()
    if device_index is None:
        device_index = t.cuda.current_device()
    if device_index not in _CTX:
        lib = _capi.load()
        h = ctypes.c_void_p()
        _capi.check(lib, lib.rm_ctx_create(int(device_index), ctypes.byref(h)), "rm_ctx_create")
        _CTX[device_index] = h
    return _CTX[device_index]


def debug_set(key, value, device_index=None):
    """Developer / test switch of this GPU's library context (include/respmon_hip.h rm_debug_set)."""
    lib = _capi.load()
    _capi.check(lib, lib.rm_debug_set(ctx(device_index), key.encode(), int(value)), "rm_debug_set")


def debug_workspace(name, shape, dtype=np.float64, device_index=None):
    """Developer / test hook: a host copy of this GPU's workspace buffer `name` as the last call left it (rm_debug_workspace)."""
    lib = _capi.load()
    out = np.empty(shape, dtype=dtype)
    _capi.check(lib, lib.rm_debug_workspace(ctx(device_index), name.encode(), ctypes.c_void_p(out.ctypes.data), out.nbytes, stream_ptr()),
                "rm_debug_workspace")
    return out


def debug_counters(device_index=None):
    """(pairs, evaluated, kept, store capacity) of the last calibration on this GPU's context (rm_debug_counters)."""
    lib = _capi.load()
    out = (ctypes.c_longlong * 4)()
    _capi.check(lib, lib.rm_debug_counters(ctx(device_index), out, stream_ptr()), "rm_debug_counters")
    return [int(v) for v in out]


def raw_stream(device_index):
    """hipStream_t of torch's current stream on `device_index` as an int (the raw getter costs ~0.3 us; the Stream object of
    torch.cuda.current_stream() ~3 us, which sits on the host path between two calibrations)."""
    t = torch()
    get = getattr(t._C, "_cuda_getCurrentRawStream", None)
    if get is not None:
        return int(get(device_index))
    return int(t.cuda.current_stream(device_index).cuda_stream)


def stream_ptr():
    t = torch()
    return ctypes.c_void_p(raw_stream(t.cuda.current_device()))


def dtype_code(tensor):
    t = torch()
    table = {t.uint8: _capi.RM_U8, t.float16: _capi.RM_F16, t.float32: _capi.RM_F32, t.float64: _capi.RM_F64}
    if tensor.dtype not in table:
        raise TypeError("unsupported frame dtype %s (uint8, float16, float32, float64)" % tensor.dtype)
    return table[tensor.dtype]


def is_bgr_buffer(tensor):
    """[T,H,W,3] uint8: frames as cv2.VideoCapture.read() delivers them (base.py:229), stored unconverted (RM_BGR8)."""
    return tensor.dim() == 4 and tensor.shape[-1] == 3 and tensor.dtype == torch().uint8


def buffer_dtype_code(buf):
    """dtype code of a calibration frame BUFFER: dtype_code of a [T,H,W] tensor, RM_BGR8 of a [T,H,W,3] uint8 one."""
    if buf.dim() == 4:
        if not is_bgr_buffer(buf):
            raise TypeError("a 4-D frame buffer must be [T,H,W,3] uint8 (BGR), got %s %s" % (tuple(buf.shape), buf.dtype))
        return _capi.RM_BGR8
    return dtype_code(buf)


def buffer_shape(buf):
    """(T, H, W) of a [T,H,W] or [T,H,W,3] frame buffer."""
    if buf.dim() not in (3, 4):
        raise ValueError("frame buffer must be [T,H,W] or [T,H,W,3], got %s" % (tuple(buf.shape),))
    return int(buf.shape[0]), int(buf.shape[1]), int(buf.shape[2])


def to_device(a, dtype=None):
    """numpy array or torch tensor -> contiguous CUDA(HIP) tensor (no copy if already there)."""
    t = require_gpu()
    if isinstance(a, t.Tensor):
        x = a
    else:
        x = t.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    if not x.is_cuda:
        x = x.cuda()
    return x.contiguous()


def ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else None


def like_input(result_tensor, original):
    """Return numpy when the caller passed numpy (the reference's convention), else the tensor."""
    t = torch()
    if isinstance(original, t.Tensor):
        return result_tensor
    return result_tensor.cpu().numpy()
