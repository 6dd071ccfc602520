"""
oracle/respmon_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + scipy + oracle/libcvref.so) of the reference's
Eulerian-magnification calibration and ROI motion-extraction hot path.  It is
the parity checker for the HIP implementation and the `cpu_baseline` leg of
bench.py.  Nothing under respmon_amd/ may import, load or call it.

Every function follows the reference's MATERIALISING algorithm step for step and
cites the reference lines it restates (paths relative to /root/reference).

Pinning status
--------------
* Glue (dtype helpers, FFT mask quirk, level selection, collapse order, min/max
  mask, time average, normalisation, uint8 truncation, PCA row-unpack quirk,
  reduce_bounding_box): PINNED against outputs of the reference itself, run in
  the build container through oracle/ref_loader.py; fixtures are committed in
  tests/golden/ together with the generating script oracle/make_golden.py.
* OpenCV arithmetic (pyrDown/pyrUp/threshold/findContours/contourArea/
  boundingRect/goodFeaturesToTrack/calcOpticalFlowPyrLK/cvtColor), restated in
  oracle/cvref*.c: PARITY UNPINNED -- cv2 is an un-vendored, un-pinned
  dependency (README.md:12 "opencv3") that is absent from /root/reference and
  not installable here; anchored by analytic known-answer tests only.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.fftpack
import scipy.signal

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

THRESH_BINARY = 0
RETR_EXTERNAL = 0
CHAIN_APPROX_NONE = 1
CHAIN_APPROX_SIMPLE = 2


def build(force=False):
    """Compile oracle/libcvref.so (gcc).  Called by __graft_entry__.build() and tests."""
    so = os.path.join(_HERE, "libcvref.so")
    srcs = [os.path.join(_HERE, f) for f in ("cvref.c", "cvref_flow.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcvref.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libcvref.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        L.rmo_pyr_down.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
        L.rmo_pyr_up.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int, ctypes.c_int]
        L.rmo_contour_area.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------- #
# cv2 call-site shims (semantics: SURVEY.md Appendix B; code: oracle/cvref*.c)
# --------------------------------------------------------------------------- #
def pyrDown(img):
    """cv2.pyrDown on float64, default dst size -- pyramid.py:14."""
    src = np.ascontiguousarray(img, dtype=np.float64)
    h, w = src.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), dtype=np.float64)
    rc = _lib().rmo_pyr_down(_dptr(src), h, w, _dptr(dst))
    assert rc == 0
    return dst


def pyrUp(img, dstsize):
    """cv2.pyrUp(img, dstsize=(w, h)) on float64 -- pyramid.py:25-26, 55."""
    src = np.ascontiguousarray(img, dtype=np.float64)
    sh, sw = src.shape
    dw, dh = int(dstsize[0]), int(dstsize[1])
    dst = np.empty((dh, dw), dtype=np.float64)
    rc = _lib().rmo_pyr_up(_dptr(src), sh, sw, _dptr(dst), dh, dw)
    if rc != 0:
        raise ValueError("pyrUp: dstsize %r incompatible with source %r" % ((dw, dh), (sw, sh)))
    return dst


def threshold(img_u8, thresh, maxval, threshold_type=THRESH_BINARY):
    """cv2.threshold(..., THRESH_BINARY) -- base.py:566."""
    assert threshold_type == THRESH_BINARY
    src = np.ascontiguousarray(img_u8, dtype=np.uint8)
    dst = np.empty_like(src)
    _lib().rmo_threshold_binary(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(src.size),
                                ctypes.c_int(int(thresh)), ctypes.c_int(int(maxval)),
                                dst.ctypes.data_as(ctypes.c_void_p))
    return float(thresh), dst


def findContours(img_u8, mode=RETR_EXTERNAL, method=CHAIN_APPROX_SIMPLE, clip_frame=False):
    """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_*) -- base.py:568.

    Returns the contour list in cv2's order (reverse raster discovery), each an
    int32 array of shape [n, 1, 2] holding (x, y).
    clip_frame: the reference pins no OpenCV version.  False = OpenCV >= 3.2 (tracing on a zero-padded copy: pixels on the
    image frame count); True = OpenCV <= 3.1, which zeroes the 1-pixel image frame (in place -- hence base.py:567's
    thresh_copy) before tracing, so components are clipped to [1, W-2] x [1, H-2].
    """
    assert mode == RETR_EXTERNAL
    src = np.ascontiguousarray(img_u8, dtype=np.uint8)
    if clip_frame:
        src = src.copy()
        src[0, :] = 0; src[-1, :] = 0; src[:, 0] = 0; src[:, -1] = 0
    h, w = src.shape
    cap_pts = 2 * src.size + 16
    cap_off = src.size + 2
    pts = np.empty((cap_pts, 2), dtype=np.int32)
    offs = np.empty(cap_off, dtype=np.int32)
    n = _lib().rmo_find_contours_external(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(h), ctypes.c_int(w),
                                          ctypes.c_int(1 if method == CHAIN_APPROX_SIMPLE else 0),
                                          pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cap_pts),
                                          offs.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cap_off))
    assert n >= 0
    out = [pts[offs[i]:offs[i + 1]].reshape(-1, 1, 2).copy() for i in range(n)]
    out.reverse()
    return out


def contourArea(contour):
    """cv2.contourArea -- base.py:571-572."""
    c = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    return float(_lib().rmo_contour_area(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(c))))


def boundingRect(contour):
    """cv2.boundingRect -- base.py:575."""
    c = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    r = np.empty(4, dtype=np.int32)
    _lib().rmo_bounding_rect(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(c)),
                             r.ctypes.data_as(ctypes.c_void_p))
    return int(r[0]), int(r[1]), int(r[2]), int(r[3])


def cvtColor_bgr2gray(frame_u8):
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) -- base.py:230."""
    src = np.ascontiguousarray(frame_u8, dtype=np.uint8)
    h, w, c = src.shape
    assert c == 3
    dst = np.empty((h, w), dtype=np.uint8)
    _lib().rmo_bgr2gray(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(h * w),
                        dst.ctypes.data_as(ctypes.c_void_p))
    return dst


def goodFeaturesToTrack(img_u8, maxCorners, qualityLevel, minDistance, blockSize=3, mask=None):
    """cv2.goodFeaturesToTrack -- base.py:365-366.  Returns float32 [N,1,2] or None."""
    assert mask is None
    src = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = src.shape
    cap = max(int(maxCorners), 1) if maxCorners > 0 else src.size
    out = np.empty((cap, 2), dtype=np.float32)
    n = _lib().rmo_good_features_to_track(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(h), ctypes.c_int(w),
                                          ctypes.c_int(int(maxCorners)), ctypes.c_double(qualityLevel),
                                          ctypes.c_double(minDistance), ctypes.c_int(int(blockSize)),
                                          out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cap))
    assert n >= 0
    if n == 0:
        return None
    return out[:n].reshape(-1, 1, 2).copy()


def cornerMinEigenVal(img_u8, blockSize):
    src = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = src.shape
    out = np.empty((h, w), dtype=np.float32)
    rc = _lib().rmo_corner_min_eigen_val(src.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(h), ctypes.c_int(w),
                                         ctypes.c_int(int(blockSize)), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def calcOpticalFlowPyrLK(prev_u8, next_u8, prev_pts, next_pts=None, winSize=(21, 21), maxLevel=3,
                         criteria=(3, 30, 0.01)):
    """cv2.calcOpticalFlowPyrLK -- base.py:371-372.  Returns (p1 [N,1,2] f32, st [N,1] u8, err None)."""
    assert next_pts is None
    a = np.ascontiguousarray(prev_u8, dtype=np.uint8)
    b = np.ascontiguousarray(next_u8, dtype=np.uint8)
    assert a.shape == b.shape
    h, w = a.shape
    p0 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    n = len(p0)
    p1 = np.empty((n, 2), dtype=np.float32)
    st = np.empty(n, dtype=np.uint8)
    ctype, max_count, eps = criteria
    if not (ctype & 1):
        max_count = 30
    if not (ctype & 2):
        eps = 0.01
    rc = _lib().rmo_calc_optical_flow_pyr_lk(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_int(h), ctypes.c_int(w),
                                             p0.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                                             ctypes.c_int(int(winSize[0])), ctypes.c_int(int(winSize[1])),
                                             ctypes.c_int(int(maxLevel)), ctypes.c_int(int(max_count)),
                                             ctypes.c_double(float(eps)),
                                             p1.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
    assert rc >= 0
    return p1.reshape(-1, 1, 2), st.reshape(-1, 1), None


# --------------------------------------------------------------------------- #
# transforms.py dtype helpers
# --------------------------------------------------------------------------- #
def uint8_to_float(img):
    """transforms.py:20-23 -- `img * (1./255)` as float64."""
    return np.asarray(img) * (1. / 255)


def float_to_uint8(img):
    """transforms.py:26-29 -- `img*255` stored into uint8 (C cast: truncation toward 0)."""
    v = np.asarray(img, dtype=np.float64) * 255
    out = np.empty(v.shape, dtype=np.uint8)
    with np.errstate(invalid="ignore"):
        out[...] = v
    return out


# --------------------------------------------------------------------------- #
# pyramid.py
# --------------------------------------------------------------------------- #
def create_gaussian_image_pyramid(image, pyramid_levels):
    """pyramid.py:9-17 -- float64 copy, then (levels-1) x pyrDown."""
    g = np.array(image, dtype=np.float64)
    out = [g]
    for _ in range(1, pyramid_levels):
        g = pyrDown(g)
        out.append(g)
    return out


def create_laplacian_image_pyramid(image, pyramid_levels):
    """pyramid.py:20-28 -- L_i = G_i - pyrUp(G_{i+1}, size of G_i); last = coarsest Gaussian."""
    g = create_gaussian_image_pyramid(image, pyramid_levels)
    lap = []
    for i in range(pyramid_levels - 1):
        lap.append(g[i] - pyrUp(g[i + 1], dstsize=(g[i].shape[1], g[i].shape[0])))
    lap.append(g[-1])
    return lap


def create_laplacian_video_pyramid(video, pyramid_levels):
    """pyramid.py:31-48 -- list of per-level [T, h_l, w_l] float64 arrays."""
    T = video.shape[0]
    levels = None
    for t in range(T):
        lap = create_laplacian_image_pyramid(video[t], pyramid_levels)
        if levels is None:
            levels = [np.zeros((T,) + l.shape, dtype=np.float64) for l in lap]
        for k, l in enumerate(lap):
            levels[k][t] = l
    return levels


def collapse_laplacian_pyramid(image_pyramid):
    """pyramid.py:51-57 -- coarsest to finest: img = pyrUp(img, size of next) + next."""
    levels = list(image_pyramid)
    img = levels.pop()
    while levels:
        nxt = levels.pop()
        img = pyrUp(img, dstsize=(nxt.shape[1], nxt.shape[0])) + (nxt - 0)
    return img


def collapse_laplacian_video_pyramid(pyramid):
    """pyramid.py:60-69 -- per frame collapse, written in place into pyramid[0]."""
    T = pyramid[0].shape[0]
    for t in range(T):
        pyramid[0][t] = collapse_laplacian_pyramid([lvl[t] for lvl in pyramid])
    return pyramid[0]


# --------------------------------------------------------------------------- #
# transforms.py:82-102  temporal FFT band-pass (with the packed-rfft quirk)
# --------------------------------------------------------------------------- #
def band_bounds(n, fps, freq_min, freq_max):
    """transforms.py:88-90 -- argmin over the FULL fftfreq vector (incl. negative bins)."""
    freqs = scipy.fftpack.fftfreq(n, d=1.0 / fps)
    lo = int(np.abs(freqs - freq_min).argmin())
    hi = int(np.abs(freqs - freq_max).argmin())
    return lo, hi


def temporal_bandpass_filter_fft(data, fps, freq_min=0.833, freq_max=1, axis=0, amplification_factor=50):
    """transforms.py:82-102.  rfft output is the PACKED real layout; the mask indexes it as if
    it were a complex spectrum; the inverse is Re(ifft(packed)) along axis 0; then *amplification."""
    data = np.asarray(data, dtype=np.float64)
    n = data.shape[0]
    spec = scipy.fftpack.rfft(data, axis=axis)
    lo, hi = band_bounds(n, fps, freq_min, freq_max)
    spec[hi:-hi] = 0
    if lo != 0:
        spec[:lo] = 0
        spec[-lo:] = 0
    out = np.real(scipy.fftpack.ifft(spec, axis=0)).astype(np.float64)
    out *= amplification_factor
    return out


def temporal_bandpass_filter(data, fps, freq_min=0.833, freq_max=1, axis=0, amplification_factor=50):
    """transforms.py:72-79 (with 38-44, 53-55): order-6 Butterworth band-pass, lfilter along `axis`, *amplification."""
    import scipy.signal
    nyq = 0.5 * fps
    b, a = scipy.signal.butter(6, [freq_min / nyq, freq_max / nyq], btype='band', output='ba')
    result = scipy.signal.lfilter(b, a, np.asarray(data, dtype=np.float64), axis=axis)
    result *= amplification_factor
    return result


def temporal_operator(n, fps, freq_min, freq_max):
    """The filter above as an explicit real n x n matrix (WITHOUT the amplification):
    out = M @ x along time.  Built by applying transforms.py:86-98 literally to the identity."""
    eye = np.eye(n, dtype=np.float64).reshape(n, n, 1)
    return temporal_bandpass_filter_fft(eye, fps, freq_min, freq_max, amplification_factor=1.0).reshape(n, n)


# --------------------------------------------------------------------------- #
# transforms.py:144-198  eulerian_magnification_bandpass
# --------------------------------------------------------------------------- #
def eulerian_magnification_bandpass(vid_data, fps, freq_min, freq_max, amplification,
                                    pyramid_levels=4, skip_levels_at_top=2, threshold=0.7, temporal_filter_function=None):
    temporal_filter_function = temporal_filter_function or temporal_bandpass_filter_fft        # :146
    vid_pyramid = create_laplacian_video_pyramid(vid_data, pyramid_levels)          # :148
    bandpassed = [np.zeros(l.shape) for l in vid_pyramid]                            # :150-152
    for i, vid in enumerate(vid_pyramid):                                            # :156
        if i < skip_levels_at_top or i >= len(vid_pyramid) - 1:                      # :157
            continue
        bandpassed[i] += temporal_filter_function(vid, fps, freq_min=freq_min, freq_max=freq_max,
                                                  amplification_factor=amplification)       # :162,169
    raw = collapse_laplacian_video_pyramid(bandpassed)                               # :182
    min_val = raw.min()                                                              # :185
    max_val = raw.max()                                                              # :187
    top = max_val - (max_val - min_val) * threshold                                  # :188-189
    masked = raw.copy()                                                              # :191
    masked[raw >= top] = min_val                                                     # :190,192
    return masked, raw                                                               # :198


# --------------------------------------------------------------------------- #
# base.py:547-601  locate
# --------------------------------------------------------------------------- #
def heatmap_u8(masked):
    """base.py:562-564 -- time average, min-max normalise, float_to_uint8."""
    avg_frame = np.array(np.average(masked, axis=0))
    with np.errstate(invalid="ignore", divide="ignore"):
        avg_norm = (avg_frame - avg_frame.min()) / (avg_frame.max() - avg_frame.min())
    return avg_frame, float_to_uint8(avg_norm)


def roi_from_heatmap_u8(avg_u8, thresh_value, clip_frame=False):
    """base.py:566-575 -- threshold, external contours, largest contourArea, boundingRect."""
    _, binary = threshold(avg_u8, thresh_value, 255, THRESH_BINARY)
    contours = findContours(binary, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE, clip_frame=clip_frame)
    if len(contours) <= 0:
        return None
    c = max(contours, key=contourArea)
    return boundingRect(c)


def locate(calibration_video_data, fps, freq_min=0.1, freq_max=1.0, amplification=500,
           pyramid_levels=9, skip_levels_at_top=4, temporal_threshold=0.7, threshold=20,
           return_intermediates=False, contour_clip_frame=False):
    masked, raw = eulerian_magnification_bandpass(calibration_video_data, fps, freq_min, freq_max, amplification,
                                                  skip_levels_at_top=skip_levels_at_top,
                                                  pyramid_levels=pyramid_levels, threshold=temporal_threshold)
    avg_frame, avg = heatmap_u8(masked)
    roi = roi_from_heatmap_u8(avg, threshold, clip_frame=contour_clip_frame)
    if return_intermediates:
        return roi, dict(avg_frame=avg_frame, avg_u8=avg, min=float(raw.min()), max=float(raw.max()))
    return roi


def locate_parallel(calibration_video_data, fps, freq_min=0.1, freq_max=1.0, amplification=500,
                    pyramid_levels=9, skip_levels_at_top=4, temporal_threshold=0.7, threshold=20,
                    workers=8, return_intermediates=False):
    """locate() above with its independent loops spread over `workers` host threads -- frames for the pyramid build and
    the collapse (pyramid.py:31-69), pixel rows for the temporal filter and the time average, frame chunks for the
    whole-array passes of transforms.py:184-192.  Every element goes through the same operations in the same order, so
    the results equal locate()'s bit for bit (tests/test_oracle_golden.py).  The reference itself is single-threaded:
    this exists so that bench.py can quote an all-host-cores CPU figure beside the single-thread one."""
    from concurrent.futures import ThreadPoolExecutor
    vid = calibration_video_data
    T = vid.shape[0]
    L = pyramid_levels
    workers = max(1, int(workers))

    def chunks(n, parts):
        parts = max(1, min(parts, n))
        base, rem = divmod(n, parts)
        out, a = [], 0
        for k in range(parts):
            b = a + base + (1 if k < rem else 0)
            out.append((a, b))
            a = b
        return out

    with ThreadPoolExecutor(workers) as ex:
        # transforms.py:148 / pyramid.py:31-48
        first = create_laplacian_image_pyramid(vid[0], L)
        levels = [np.zeros((T,) + l.shape, dtype=np.float64) for l in first]

        def build(span):
            for t in range(*span):
                for k, l in enumerate(create_laplacian_image_pyramid(vid[t], L)):
                    levels[k][t] = l
        list(ex.map(build, chunks(T, 4 * workers)))
        # transforms.py:150-170
        bandpassed = [np.zeros(l.shape) for l in levels]
        for i, lv in enumerate(levels):
            if i < skip_levels_at_top or i >= len(levels) - 1:
                continue

            def filt(span, lv=lv, dst=bandpassed[i]):
                y0, y1 = span
                dst[:, y0:y1] += temporal_bandpass_filter_fft(lv[:, y0:y1], fps, freq_min=freq_min, freq_max=freq_max,
                                                              amplification_factor=amplification)
            list(ex.map(filt, chunks(lv.shape[1], 2 * workers)))
        del levels
        # transforms.py:182 / pyramid.py:60-69
        def collapse(span):
            for t in range(*span):
                bandpassed[0][t] = collapse_laplacian_pyramid([lvl[t] for lvl in bandpassed])
        list(ex.map(collapse, chunks(T, 4 * workers)))
        raw = bandpassed[0]
        spans = chunks(T, 2 * workers)
        mins = list(ex.map(lambda sp: raw[sp[0]:sp[1]].min(), spans))
        maxs = list(ex.map(lambda sp: raw[sp[0]:sp[1]].max(), spans))
        min_val, max_val = min(mins), max(maxs)                                           # :185,187
        top = max_val - (max_val - min_val) * temporal_threshold                         # :188-189
        masked = np.empty_like(raw)

        def mask(sp):
            m = raw[sp[0]:sp[1]].copy()                                                   # :191
            m[raw[sp[0]:sp[1]] >= top] = min_val                                          # :190,192
            masked[sp[0]:sp[1]] = m
        list(ex.map(mask, spans))
        # base.py:562: np.average over t, sequential per pixel
        H = raw.shape[1]
        avg_frame = np.empty(raw.shape[1:], dtype=np.float64)

        def avg(sp):
            avg_frame[sp[0]:sp[1]] = np.average(masked[:, sp[0]:sp[1]], axis=0)
        list(ex.map(avg, chunks(H, 2 * workers)))
    with np.errstate(invalid="ignore", divide="ignore"):
        avg_norm = (avg_frame - avg_frame.min()) / (avg_frame.max() - avg_frame.min())
    avg_u8 = float_to_uint8(avg_norm)
    roi = roi_from_heatmap_u8(avg_u8, threshold)
    if return_intermediates:
        return roi, dict(avg_frame=avg_frame, avg_u8=avg_u8, min=float(min_val), max=float(max_val))
    return roi


# --------------------------------------------------------------------------- #
# tools.py:48-57
# --------------------------------------------------------------------------- #
def reduce_bounding_box(x, y, w, h, maximum_area):
    start_area = w * h
    if start_area <= maximum_area:
        return x, y, w, h
    s = np.sqrt(float(maximum_area) / float(start_area))
    nw, nh = w * s, h * s
    nx, ny = x + (w - nw) / 2., y + (h - nh) / 2.
    return int(np.round(nx)), int(np.round(ny)), int(np.round(nw)), int(np.round(nh))


# --------------------------------------------------------------------------- #
# base.py:354-407  extract_motion
# --------------------------------------------------------------------------- #
def roi_average(frame, x, y, w, h):
    """base.py:471 + 355-358."""
    return np.average(frame[y:y + h, x:x + w])


def pca_first_component(motion_data):
    """base.py:396-407 -- np.cov (ddof=1) -> np.linalg.eig -> argsort desc -> ROW unpack -> last projection."""
    if len(motion_data) < 2:
        return 0.0
    xs, ys = np.transpose(motion_data)
    cov = np.cov(np.vstack([xs, ys]))
    vals, vecs = np.linalg.eig(cov)
    order = np.argsort(vals)[::-1]
    evec1, _evec2 = vecs[:, order]
    return np.array(motion_data).dot(evec1)[-1]


class FlowState:
    """State carried by extract_motion('flow') across frames (base.py:136,138,128)."""

    def __init__(self, feature_params=None, lk_params=None, buffer_len=128):
        self.prev = None
        self.points = None
        self.motion_data = []
        self.buffer_len = buffer_len
        self.feature_params = feature_params or dict(maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7)
        self.lk_params = lk_params or dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))


def extract_motion_flow(state, cropped_image):
    """base.py:360-407 (with the deque pop of base.py:473-475 applied to motion_data)."""
    if len(state.motion_data) >= state.buffer_len:
        state.motion_data.pop(0)
    if state.prev is None:
        state.prev = float_to_uint8(np.array(cropped_image))
        state.points = goodFeaturesToTrack(state.prev, mask=None, **state.feature_params)
        return 0.0
    cur = float_to_uint8(np.array(cropped_image))
    p1, st, _ = calcOpticalFlowPyrLK(state.prev, cur, state.points, None, **state.lk_params)
    good_new = p1[st == 1]
    good_old = state.points[st == 1]
    state.prev = cur
    state.points = good_new.reshape(-1, 1, 2)
    if len(good_new) == 0 or len(good_old) == 0:
        return np.nan
    state.motion_data.append(list(np.mean(good_old - good_new, axis=0)))
    return pca_first_component(state.motion_data)


# --------------------------------------------------------------------------- #
# transforms.py:58-69 (used by measure(), base.py:342) -- "next" row f2
# --------------------------------------------------------------------------- #
def butter_lowpass_filter(data, cutoff, fs, order=5):
    b, a = scipy.signal.butter(order, cutoff / (0.5 * fs), btype="low", analog=False)
    return scipy.signal.filtfilt(b, a, data)
