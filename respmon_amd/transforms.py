"""Device-backed counterparts of the reference's transforms.py hot-path functions (same names,
argument meaning and return conventions).  numpy in -> numpy out; CUDA tensor in -> tensor out."""
import ctypes
import logging

import numpy as np

from . import _capi, device


def uint8_to_float(img):
    """reference transforms.py:20-23: img * (1./255) as float64."""
    t = device.require_gpu()
    lib = _capi.load()
    src = device.to_device(img, t.uint8)
    dst = t.empty(src.shape, dtype=t.float64, device=src.device)
    _capi.check(lib, lib.rm_uint8_to_float(device.ctx(), device.ptr(src), device.ptr(dst), src.numel(), device.stream_ptr()),
                "rm_uint8_to_float")
    return device.like_input(dst, img)


def float_to_uint8(img):
    """reference transforms.py:26-29: img*255 stored into uint8 (C truncation)."""
    t = device.require_gpu()
    lib = _capi.load()
    src = device.to_device(img, t.float64)
    dst = t.empty(src.shape, dtype=t.uint8, device=src.device)
    _capi.check(lib, lib.rm_float_to_uint8(device.ctx(), device.ptr(src), device.ptr(dst), src.numel(), device.stream_ptr()),
                "rm_float_to_uint8")
    return device.like_input(dst, img)


def temporal_operator(n, fps, freq_min, freq_max):
    """The reference's rfft / mask / Re(ifft) filter (transforms.py:86-98) as its explicit n x n matrix
    (without the amplification) plus (bound_low, bound_high).  Host only; needs no GPU."""
    lib = _capi.load()
    M = np.empty((n, n), dtype=np.float64)
    lo, hi = ctypes.c_int(), ctypes.c_int()
    _capi.check(lib, lib.rm_temporal_operator(int(n), float(fps), float(freq_min), float(freq_max),
                                              ctypes.c_void_p(M.ctypes.data), ctypes.byref(lo), ctypes.byref(hi)),
                "rm_temporal_operator")
    return M, lo.value, hi.value


def temporal_bandpass_filter_fft(data, fps, freq_min=0.833, freq_max=1, axis=0,
                                 amplification_factor=50, verbose=False, debug=''):
    """reference transforms.py:82-102 (packed-rfft mask quirk kept).  `data` is [T,h,w]; like the
    reference the inverse transform always runs along axis 0, so only axis=0 is accepted."""
    if axis != 0:
        raise NotImplementedError("the reference's ifft is hard-wired to axis 0 (transforms.py:98); only axis=0 is meaningful")
    t = device.require_gpu()
    lib = _capi.load()
    x = device.to_device(data, t.float64)
    T = x.shape[0]
    npix = x[0].numel()
    out = t.empty(x.shape, dtype=t.float64, device=x.device)
    _capi.check(lib, lib.rm_temporal_bandpass_filter_fft(device.ctx(), device.ptr(x), T, npix, float(fps), float(freq_min),
                                                         float(freq_max), float(amplification_factor), device.ptr(out),
                                                         device.stream_ptr()), "rm_temporal_bandpass_filter_fft")
    if verbose:
        print('{0}{1},{2}'.format(debug, float(out.min()), float(out.max())))
    return device.like_input(out, data)


def eulerian_magnification_bandpass(vid_data, fps, freq_min, freq_max, amplification,
                                    pyramid_levels=4, skip_levels_at_top=2, verbose=False,
                                    temporal_filter_function=temporal_bandpass_filter_fft, threshold=0.7):
    """reference transforms.py:144-198 -> (bandpassed_data, raw_bandpassed_data), both [T,H,W] float64.
    This is the MATERIALISING form kept for API parity; calibration itself uses the fused path
    (RespiratoryMonitor.locate -> rm_calibrate) that never writes a [T,H,W] array."""
    if temporal_filter_function is not temporal_bandpass_filter_fft:
        raise NotImplementedError("only temporal_bandpass_filter_fft is on the accelerated path")
    t = device.require_gpu()
    lib = _capi.load()
    vid = device.to_device(vid_data)
    T, H, W = vid.shape
    masked = t.empty((T, H, W), dtype=t.float64, device=vid.device)
    raw = t.empty((T, H, W), dtype=t.float64, device=vid.device)
    mm = (ctypes.c_double * 2)()
    _capi.check(lib, lib.rm_eulerian_magnification_bandpass(device.ctx(), device.ptr(vid), device.dtype_code(vid), T, H, W,
                                                            float(fps), float(freq_min), float(freq_max), float(amplification),
                                                            int(pyramid_levels), int(skip_levels_at_top), float(threshold),
                                                            device.ptr(masked), device.ptr(raw), mm, device.stream_ptr()),
                "rm_eulerian_magnification_bandpass")
    if verbose:
        logging.info("eulerian_magnification_bandpass: min=%r max=%r", mm[0], mm[1])
    return device.like_input(masked, vid_data), device.like_input(raw, vid_data)


def butter_lowpass_filter(data, cutoff, fs, order=5):
    """reference transforms.py:58-69 (used by measure(), base.py:342): 128-sample 1-D signal, host scipy."""
    from scipy.signal import butter, filtfilt
    b, a = butter(order, cutoff / (0.5 * fs), btype='low', analog=False)
    return filtfilt(b, a, data)
