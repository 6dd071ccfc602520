"""TEST INFRASTRUCTURE: drives the C-ABI of the host-emulated build (tests/emu) with numpy arrays
standing in for device memory.  See tests/emu/build.py."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from respmon_amd import _capi  # noqa: E402

DT = {np.dtype(np.uint8): _capi.RM_U8, np.dtype(np.float16): _capi.RM_F16, np.dtype(np.float32): _capi.RM_F32,
      np.dtype(np.float64): _capi.RM_F64}


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class Emu:
    def __init__(self):
        from tests.emu import build as emu_build
        self.lib = _capi.bind(ctypes.CDLL(emu_build.build()))
        h = ctypes.c_void_p()
        self.ck(self.lib.rm_ctx_create(0, ctypes.byref(h)), "ctx_create")
        self.ctx = h

    def ck(self, rc, what):
        return _capi.check(self.lib, rc, what)

    def pyr_down(self, src):
        src = np.ascontiguousarray(src)
        T, h, w = src.shape
        dst = np.empty((T, (h + 1) // 2, (w + 1) // 2))
        self.ck(self.lib.rm_pyr_down(self.ctx, ptr(src), DT[src.dtype], T, h, w, ptr(dst), None), "pyr_down")
        return dst

    def pyr_up(self, src, dh, dw, mode=0, other=None):
        src = np.ascontiguousarray(src, dtype=np.float64)
        T, sh, sw = src.shape
        dst = np.empty((T, dh, dw))
        self.ck(self.lib.rm_pyr_up(self.ctx, ptr(src), T, sh, sw, ptr(dst), dh, dw, mode, ptr(other), None), "pyr_up")
        return dst

    def temporal(self, data, fps, fmin, fmax, amp):
        data = np.ascontiguousarray(data, dtype=np.float64)
        T = data.shape[0]
        npix = data[0].size
        out = np.empty_like(data)
        self.ck(self.lib.rm_temporal_bandpass_filter_fft(self.ctx, ptr(data), T, npix, fps, fmin, fmax, amp, ptr(out), None), "temporal")
        return out

    def operator(self, T, fps, fmin, fmax):
        M = np.empty((T, T))
        lo, hi = ctypes.c_int(), ctypes.c_int()
        self.ck(self.lib.rm_temporal_operator(T, fps, fmin, fmax, ptr(M), ctypes.byref(lo), ctypes.byref(hi)), "operator")
        return M, lo.value, hi.value

    def eulerian(self, frames, fps, fmin, fmax, amp, levels, skip, thr=0.7):
        frames = np.ascontiguousarray(frames)
        T, H, W = frames.shape
        masked = np.empty((T, H, W)); raw = np.empty((T, H, W)); mm = np.empty(2)
        self.ck(self.lib.rm_eulerian_magnification_bandpass(self.ctx, ptr(frames), DT[frames.dtype], T, H, W, fps, fmin, fmax,
                                                            amp, levels, skip, thr, ptr(masked), ptr(raw), ptr(mm), None), "eulerian")
        return masked, raw, mm

    def calibrate(self, frames, fps, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, flags=0):
        frames = np.ascontiguousarray(frames)
        T, H, W = frames.shape
        heat = np.empty((H, W)); mm = np.empty(2)
        self.ck(self.lib.rm_calibrate(self.ctx, ptr(frames), DT[frames.dtype], T, H, W, fps, fmin, fmax, amp, levels, skip, thr,
                                      flags, ptr(heat), ptr(mm), None), "calibrate")
        return heat, mm

    def heatmap_to_roi(self, heat, threshold=20):
        heat = np.ascontiguousarray(heat, dtype=np.float64)
        H, W = heat.shape
        xywh = np.zeros(4, np.int32); u8 = np.empty((H, W), np.uint8); b = np.empty((H, W), np.uint8)
        rc = self.ck(self.lib.rm_heatmap_to_roi(self.ctx, ptr(heat), H, W, threshold, ptr(xywh), ptr(u8), ptr(b), None), "roi")
        return (None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)), u8, b

    def locate(self, frames, fps, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, threshold=20, flags=0):
        frames = np.ascontiguousarray(frames)
        T, H, W = frames.shape
        xywh = np.zeros(4, np.int32)
        rc = self.ck(self.lib.rm_locate(self.ctx, ptr(frames), DT[frames.dtype], T, H, W, fps, fmin, fmax, amp, levels, skip, thr,
                                        threshold, flags, ptr(xywh), None), "locate")
        return None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)
