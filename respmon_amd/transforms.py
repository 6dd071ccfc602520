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


def bgr_buffer_to_gray(buf):
    """[T,H,W,3] uint8 -> [T,H,W] uint8: cv2.cvtColor(BGR2GRAY) of every frame (base.py:230) on the device (rm_bgr_to_gray) -- what the
    calibration kernels do on the fly with an RM_BGR8 buffer, for the entry points that take gray frames."""
    t = device.require_gpu()
    lib = _capi.load()
    src = device.to_device(buf)
    if not device.is_bgr_buffer(src):
        raise TypeError("expected a [T,H,W,3] uint8 buffer")
    gray = t.empty(tuple(src.shape[:3]), dtype=t.uint8, device=src.device)
    _capi.check(lib, lib.rm_bgr_to_gray(device.ctx(), device.ptr(src), gray.numel(), device.ptr(gray), device.stream_ptr()), "rm_bgr_to_gray")
    return gray


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


def butter_bandpass(_lowcut, _highcut, _fs, order=5):
    """reference transforms.py:38-44: Butterworth band-pass design (host, scipy like the reference)."""
    from scipy.signal import butter
    _nyq = 0.5 * _fs
    _b, _a = butter(order, [_lowcut / _nyq, _highcut / _nyq], btype='band', output='ba')
    return _b, _a


def butter_bandpass_filter(_data, _lowcut, _highcut, _fs, order=5):
    """reference transforms.py:47-50: lfilter along the LAST axis (1-D signals in the reference; host scipy)."""
    from scipy.signal import lfilter
    _b, _a = butter_bandpass(_lowcut, _highcut, _fs, order=order)
    return lfilter(_b, _a, _data)


def butter_bandpass_filter_fast(_data, _b, _a, axis=0):
    """reference transforms.py:53-55: scipy.signal.lfilter(b, a, data, axis).  Videos ([T,...], axis 0) run on the
    device (rm_lfilter: one lane per pixel, the recurrence in scipy's operation order); anything else is a small
    host-side signal and goes to scipy like the reference."""
    x_in = _data
    is_tensor = hasattr(_data, "is_cuda")
    if axis != 0 or (not is_tensor and np.ndim(_data) < 2):
        from scipy.signal import lfilter
        return lfilter(_b, _a, _data.cpu().numpy() if is_tensor else _data, axis=axis)
    t = device.require_gpu()
    lib = _capi.load()
    x = device.to_device(x_in, t.float64)
    T = x.shape[0]
    npix = x[0].numel()
    out = t.empty(x.shape, dtype=t.float64, device=x.device)
    b = np.ascontiguousarray(_b, dtype=np.float64)
    a = np.ascontiguousarray(_a, dtype=np.float64)
    n = max(len(b), len(a))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    _capi.check(lib, lib.rm_lfilter(device.ctx(), device.ptr(x), T, npix, ctypes.c_void_p(b.ctypes.data),
                                    ctypes.c_void_p(a.ctypes.data), n, 1.0, device.ptr(out), device.stream_ptr()), "rm_lfilter")
    return device.like_input(out, x_in)


def butter_lowpass(cutoff, fs, order=5):
    """reference transforms.py:58-63."""
    from scipy.signal import butter
    return butter(order, cutoff / (0.5 * fs), btype='low', analog=False)


def temporal_bandpass_filter(data, fps, freq_min=0.833, freq_max=1, axis=0,
                             amplification_factor=50, verbose=False, debug=''):
    """reference transforms.py:72-79: order-6 Butterworth band-pass (lfilter) along `axis`, times the
    amplification -- the IIR alternative to temporal_bandpass_filter_fft."""
    b, a = butter_bandpass(freq_min, freq_max, fps, order=6)
    result = butter_bandpass_filter_fast(data, b, a, axis=axis)
    result *= amplification_factor
    if verbose:
        print('{0}{1},{2}'.format(debug, float(result.min()), float(result.max())))
    return result


def eulerian_magnification_bandpass(vid_data, fps, freq_min, freq_max, amplification,
                                    pyramid_levels=4, skip_levels_at_top=2, verbose=False,
                                    temporal_filter_function=None, threshold=0.7):
    """reference transforms.py:144-198 -> (bandpassed_data, raw_bandpassed_data), both [T,H,W] float64.
    This is the MATERIALISING form kept for API parity; calibration itself uses the fused path
    (RespiratoryMonitor.locate -> rm_calibrate) that never writes a [T,H,W] array.
    `temporal_filter_function` defaults to temporal_bandpass_filter_fft (one fused C-ABI call); any other
    callable with the reference's filter signature (e.g. temporal_bandpass_filter, or a plain scipy filter) runs the
    reference's own sequence -- pyramid, filter per level, collapse, mask; the callable is handed numpy levels when
    `vid_data` is numpy and device tensors when it is a device tensor."""
    t = device.require_gpu()
    lib = _capi.load()
    vid = device.to_device(vid_data)
    T, H, W = device.buffer_shape(vid)
    if vid.dim() == 4 and not (temporal_filter_function is None or temporal_filter_function is temporal_bandpass_filter_fft):
        vid = bgr_buffer_to_gray(vid)
    masked = t.empty((T, H, W), dtype=t.float64, device=vid.device)
    mm = (ctypes.c_double * 2)()
    if temporal_filter_function is None or temporal_filter_function is temporal_bandpass_filter_fft:
        raw = t.empty((T, H, W), dtype=t.float64, device=vid.device)
        _capi.check(lib, lib.rm_eulerian_magnification_bandpass(device.ctx(), device.ptr(vid), device.buffer_dtype_code(vid), T, H, W,
                                                                float(fps), float(freq_min), float(freq_max), float(amplification),
                                                                int(pyramid_levels), int(skip_levels_at_top), float(threshold),
                                                                device.ptr(masked), device.ptr(raw), mm, device.stream_ptr()),
                    "rm_eulerian_magnification_bandpass")
    else:
        from . import pyramid
        vid_pyramid = pyramid.create_laplacian_video_pyramid(vid, pyramid_levels)                    # transforms.py:148
        bandpassed = [t.zeros_like(lv) for lv in vid_pyramid]                                       # :150-152
        for i, lv in enumerate(vid_pyramid):                                                         # :156-170
            if i < skip_levels_at_top or i >= len(vid_pyramid) - 1:
                continue
            # the callable sees what the caller works in: numpy levels for numpy input (a scipy / numpy filter with the
            # reference's signature must run unchanged), device tensors for device input
            bandpassed[i] = device.to_device(temporal_filter_function(device.like_input(lv, vid_data), fps, freq_min=freq_min,
                                                                      freq_max=freq_max, amplification_factor=amplification,
                                                                      debug='{0},{1}:'.format('n/a', i), verbose=verbose), t.float64)
        raw = pyramid.collapse_laplacian_video_pyramid(bandpassed)                                   # :182
        _capi.check(lib, lib.rm_threshold_mask(device.ctx(), device.ptr(raw), raw.numel(), float(threshold), device.ptr(masked), mm,
                                               device.stream_ptr()), "rm_threshold_mask")           # :184-192
    if verbose:
        logging.info("eulerian_magnification_bandpass: min=%r max=%r", mm[0], mm[1])
    return device.like_input(masked, vid_data), device.like_input(raw, vid_data)


def butter_lowpass_filter(data, cutoff, fs, order=5):
    """reference transforms.py:58-69 (used by measure(), base.py:342): 128-sample 1-D signal, host scipy."""
    from scipy.signal import butter, filtfilt
    b, a = butter(order, cutoff / (0.5 * fs), btype='low', analog=False)
    return filtfilt(b, a, data)
