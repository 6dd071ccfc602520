"""
RespiratoryMonitor -- drop-in for the reference's base.py surface on the hot path
(run / locate / calibrate / measure / extract_motion / skip_calibration), backed by the
gfx950 HIP library.  Reference lines are cited per method (paths relative to the reference).

Differences from the reference that are deliberate and documented in DESIGN.md:
  * no pyqtgraph UI (`visualize` must be None or 'pyqtgraph'; the latter is accepted and ignored),
    no cv2.VideoWriter / np.save side outputs (out of scope: capture / codec I/O);
  * `capture_target` may be any object with the cv2.VideoCapture duck type
    (isOpened / read / get / release); a path or camera index needs cv2 at run time;
  * the calibration buffer lives in HBM (`calibration_buffer` is a CUDA tensor);
  * `run_on_init=True` keeps the reference's "constructor blocks in run()" behaviour
    (base.py:164); pass False to drive the object step by step;
  * `reset()` also forgets the flow state (`previous_cropped_image`, `motion_key_points`).  This DIFFERS from the reference:
    its reset() (base.py:515-533) never sets `previous_cropped_image` back to None, so after a reset and a new calibration it
    keeps tracking from the stale crop and the stale points (base.py:371) -- and `calcOpticalFlowPyrLK` fails outright if the
    new ROI has another size.  Here the first frame after a recalibration initialises the corners again (a bug fix, listed
    under "deliberate behavioural differences" in DESIGN.md);
  * after a calibration that finds no ROI the retry goes through `update_ui()` / `sync_to_fps()` like every other
    iteration; the reference `continue`s past them (base.py:451-454) -- no effect on the data path (the UI is a no-op here);
  * the default operation order of the calibration commutes two linear stages (`reference_operation_order`).
"""
import logging
import time
from collections import deque

import numpy as np

from . import _capi, device
from .tools import Benchmarker, reduce_bounding_box
from .transforms import butter_lowpass_filter

THRESH_BINARY = 0  # cv2.THRESH_BINARY


class FlowState:
    """Owner of one rm_flow_state (the device-resident tracking session of extract_motion('flow')); released with the object."""

    def __init__(self, lib):
        import ctypes
        self._lib = lib
        self.handle = ctypes.c_void_p()
        _capi.check(lib, lib.rm_flow_state_create(device.ctx(), ctypes.byref(self.handle)), "rm_flow_state_create")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self._lib.rm_flow_state_destroy(self.handle)
            self.handle.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


class _Backend:
    """The only door from the state machine to compute: every method is one C-ABI call.
    Tests of the host logic replace it with a recording double; there is no CPU implementation."""

    def __init__(self):
        import ctypes
        self.lib = _capi.load()
        self.t = device.require_gpu()
        self._xywh = (ctypes.c_int32 * 4)()

    # -- calibration ------------------------------------------------------------------
    def locate(self, buf, fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top,
               temporal_threshold, threshold, flags=0):
        import ctypes
        T, H, W = device.buffer_shape(buf)
        xywh = self._xywh
        # (this call sits on the host path between two calibrations: the context of the buffer's device is looked up without asking
        #  the runtime, the stream is the current one of that device)
        idx = buf.device.index
        ctx = device._CTX.get(idx) or device.ctx(idx)
        stream = ctypes.c_void_p(device.raw_stream(idx))
        rc = self.lib.rm_locate(ctx, ctypes.c_void_p(buf.data_ptr()), device.buffer_dtype_code(buf), T, H, W,
                                float(fps), float(freq_min), float(freq_max), float(amplification),
                                int(pyramid_levels), int(skip_levels_at_top), float(temporal_threshold),
                                int(threshold), int(flags), xywh, stream)
        if rc < 0:
            _capi.check(self.lib, rc, "rm_locate")
        if rc == _capi.RM_NO_CONTOUR:
            return None
        return xywh[0], xywh[1], xywh[2], xywh[3]

    def locate_submit(self, buf, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4,
                      temporal_threshold=0.7, threshold=20, flags=0):
        """locate() without the wait: the device work is enqueued and a ticket returned (rm_locate_submit).  Up to
        _capi.RM_LOCATE_TICKETS buffers per GPU may be in flight; `buf` must stay alive and unchanged until locate_result."""
        import ctypes
        T, H, W = device.buffer_shape(buf)
        idx = buf.device.index
        ctx = device._CTX.get(idx) or device.ctx(idx)
        stream = ctypes.c_void_p(device.raw_stream(idx))
        ticket = ctypes.c_int(-1)
        rc = self.lib.rm_locate_submit(ctx, ctypes.c_void_p(buf.data_ptr()), device.buffer_dtype_code(buf), T, H, W,
                                       float(fps), float(freq_min), float(freq_max), float(amplification),
                                       int(pyramid_levels), int(skip_levels_at_top), float(temporal_threshold),
                                       int(threshold), int(flags), stream, ctypes.byref(ticket))
        if rc < 0:
            _capi.check(self.lib, rc, "rm_locate_submit")
        return (idx, ticket.value, buf)     # (the buffer rides along: it must outlive the submission)

    def locate_result(self, ticket):
        """The ROI of a locate_submit: (x, y, w, h) or None, bit-identical to locate() on the same buffer (rm_locate_result)."""
        idx, tk, _buf = ticket
        ctx = device._CTX.get(idx) or device.ctx(idx)
        xywh = self._xywh
        rc = self.lib.rm_locate_result(ctx, tk, xywh)
        if rc < 0:
            _capi.check(self.lib, rc, "rm_locate_result")
        if rc == _capi.RM_NO_CONTOUR:
            return None
        return xywh[0], xywh[1], xywh[2], xywh[3]

    # -- ingest -----------------------------------------------------------------------
    def bgr_to_gray(self, bgr_u8_host):
        t = self.t
        src = t.from_numpy(np.ascontiguousarray(bgr_u8_host)).cuda()
        H, W = src.shape[0], src.shape[1]
        gray = t.empty((H, W), dtype=t.uint8, device=src.device)
        _capi.check(self.lib, self.lib.rm_bgr_to_gray(device.ctx(), device.ptr(src), H * W, device.ptr(gray),
                                                      device.stream_ptr()), "rm_bgr_to_gray")
        self.last_bgr = src     # the frame as captured, on the device: what a 'bgr8' calibration buffer stores (store_frame's `bgr`)
        return gray

    def store_frame(self, buf, idx, gray_u8, bgr=None):
        """calibration_buffer[idx][:] = uint8_to_float(gray) (base.py:231, 431).
        A 'bgr8' buffer ([T,H,W,3] uint8) stores the frame AS CAPTURED: pass the device BGR tensor the gray frame came from as
        `bgr` (RespiratoryMonitor.step does: the tensor next_frame() kept).  Without it -- a gray frame from another source:
        replayed, processed, another backend's ingest -- the gray value goes into all three planes, which base.py:230's
        conversion maps back to exactly that gray value (Y(x, x, x) == x), so the calibration reads the same frame either way."""
        t = self.t
        if buf.dim() == 4:
            # buffer_dtype 'bgr8': rm_locate applies base.py:230-231 when it reads the buffer (RM_BGR8)
            if bgr is not None:
                buf[idx].copy_(bgr)
            else:
                buf[idx].copy_(gray_u8.unsqueeze(-1).expand(-1, -1, 3))
        elif buf.dtype == t.uint8:
            buf[idx].copy_(gray_u8)
        else:
            dst64 = buf[idx] if buf.dtype == t.float64 else t.empty(gray_u8.shape, dtype=t.float64, device=gray_u8.device)
            _capi.check(self.lib, self.lib.rm_uint8_to_float(device.ctx(), device.ptr(gray_u8), device.ptr(dst64),
                                                             gray_u8.numel(), device.stream_ptr()), "rm_uint8_to_float")
            if buf.dtype != t.float64:   # narrower float buffer: round the float64 value to the storage dtype
                buf[idx].copy_(dst64)

    # -- measurement ------------------------------------------------------------------
    def roi_mean(self, gray_u8, x, y, w, h):
        import ctypes
        H, W = gray_u8.shape
        out = ctypes.c_double()
        _capi.check(self.lib, self.lib.rm_roi_mean(device.ctx(), device.ptr(gray_u8), device.dtype_code(gray_u8), H, W,
                                                   x, y, w, h, ctypes.byref(out), device.stream_ptr()), "rm_roi_mean")
        return out.value

    def roi_to_uint8(self, gray_u8, x, y, w, h):
        t = self.t
        H, W = gray_u8.shape
        dst = t.empty((h, w), dtype=t.uint8, device=gray_u8.device)
        _capi.check(self.lib, self.lib.rm_roi_to_uint8(device.ctx(), device.ptr(gray_u8), device.dtype_code(gray_u8), H, W,
                                                       x, y, w, h, device.ptr(dst), device.stream_ptr()), "rm_roi_to_uint8")
        return dst

    def good_features_to_track(self, img_u8, maxCorners, qualityLevel, minDistance, blockSize):
        import ctypes
        h, w = img_u8.shape
        cap = max(int(maxCorners), 1)
        pts = np.empty((cap, 2), dtype=np.float32)
        n = ctypes.c_int()
        _capi.check(self.lib, self.lib.rm_good_features_to_track(device.ctx(), device.ptr(img_u8), h, w, int(maxCorners),
                                                                 float(qualityLevel), float(minDistance), int(blockSize),
                                                                 ctypes.c_void_p(pts.ctypes.data), ctypes.byref(n),
                                                                 device.stream_ptr()), "rm_good_features_to_track")
        if n.value == 0:
            return None
        return pts[:n.value].reshape(-1, 1, 2).copy()

    def calc_optical_flow_pyr_lk(self, prev_u8, next_u8, pts, winSize, maxLevel, criteria):
        import ctypes
        h, w = prev_u8.shape
        p0 = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        n = len(p0)
        p1 = np.empty((n, 2), dtype=np.float32)
        st = np.empty(n, dtype=np.uint8)
        ctype, max_count, eps = criteria
        if not (ctype & 1):
            max_count = 30
        if not (ctype & 2):
            eps = 0.01
        _capi.check(self.lib, self.lib.rm_calc_optical_flow_pyr_lk(device.ctx(), device.ptr(prev_u8), device.ptr(next_u8), h, w,
                                                                   ctypes.c_void_p(p0.ctypes.data), n, int(winSize[0]),
                                                                   int(winSize[1]), int(maxLevel), int(max_count), float(eps),
                                                                   ctypes.c_void_p(p1.ctypes.data), ctypes.c_void_p(st.ctypes.data),
                                                                   device.stream_ptr()), "rm_calc_optical_flow_pyr_lk")
        return p1.reshape(-1, 1, 2), st.reshape(-1, 1)

    def mean_flow(self, old_pts, new_pts, status):
        import ctypes
        o = np.ascontiguousarray(old_pts, dtype=np.float32).reshape(-1, 2)
        nw = np.ascontiguousarray(new_pts, dtype=np.float32).reshape(-1, 2)
        st = np.ascontiguousarray(status, dtype=np.uint8).reshape(-1)
        mean = np.empty(2, dtype=np.float32)
        ng = ctypes.c_int()
        _capi.check(self.lib, self.lib.rm_mean_flow(device.ctx(), ctypes.c_void_p(o.ctypes.data), ctypes.c_void_p(nw.ctypes.data),
                                                    ctypes.c_void_p(st.ctypes.data), len(o), ctypes.c_void_p(mean.ctypes.data),
                                                    ctypes.byref(ng), device.stream_ptr()), "rm_mean_flow")
        return mean, ng.value

    # -- extract_motion('flow') with the crops and points resident on the device: one C-ABI call per frame --------------------
    # -- resident flow session (include/respmon_hip.h rm_flow_state): one handle per monitor -----------------------------
    def flow_state(self):
        """A new rm_flow_state on the current device: previous crop, its LK pyramid, the tracked points of ONE tracking session."""
        return FlowState(self.lib)

    def flow_begin(self, state, gray_u8, x, y, w, h, maxCorners, qualityLevel, minDistance, blockSize):
        import ctypes
        H, W = gray_u8.shape
        pts = np.empty((max(int(maxCorners), 1), 2), dtype=np.float32)
        n = ctypes.c_int()
        _capi.check(self.lib, self.lib.rm_flow_begin(device.ctx(), state.handle, device.ptr(gray_u8), device.dtype_code(gray_u8), H, W, x, y, w, h,
                                                     int(maxCorners), float(qualityLevel), float(minDistance), int(blockSize),
                                                     ctypes.c_void_p(pts.ctypes.data), ctypes.byref(n), device.stream_ptr()),
                    "rm_flow_begin")
        return None if n.value == 0 else pts[:n.value].reshape(-1, 1, 2).copy()

    def flow_step(self, state, gray_u8, x, y, w, h, winSize, maxLevel, criteria):
        import ctypes
        H, W = gray_u8.shape
        ctype, max_count, eps = criteria
        if not (ctype & 1):
            max_count = 30
        if not (ctype & 2):
            eps = 0.01
        mean = np.empty(2, dtype=np.float32)
        ng = ctypes.c_int()
        _capi.check(self.lib, self.lib.rm_flow_step(device.ctx(), state.handle, device.ptr(gray_u8), device.dtype_code(gray_u8), H, W, x, y, w, h,
                                                    int(winSize[0]), int(winSize[1]), int(maxLevel), int(max_count), float(eps),
                                                    ctypes.c_void_p(mean.ctypes.data), ctypes.byref(ng), device.stream_ptr()),
                    "rm_flow_step")
        return mean, ng.value

    def flow_points(self, state, cap):
        import ctypes
        pts = np.empty((max(int(cap), 1), 2), dtype=np.float32)
        n = ctypes.c_int()
        _capi.check(self.lib, self.lib.rm_flow_points(device.ctx(), state.handle, ctypes.c_void_p(pts.ctypes.data), len(pts), ctypes.byref(n),
                                                      device.stream_ptr()), "rm_flow_points")
        return pts[:min(n.value, len(pts))].reshape(-1, 1, 2).copy()

    def pca_reduce(self, motion_data):
        import ctypes
        m = np.ascontiguousarray(motion_data, dtype=np.float32).reshape(-1, 2)
        out = ctypes.c_double()
        _capi.check(self.lib, self.lib.rm_pca_reduce(device.ctx(), ctypes.c_void_p(m.ctypes.data), len(m), ctypes.byref(out),
                                                     device.stream_ptr()), "rm_pca_reduce")
        return out.value


class RespiratoryMonitor:
    CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS = 3, 4, 5
    TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS = 1, 2

    def __init__(self, capture_target=0, save_calibration_image=False, visualize='pyqtgraph', fig_size=None,
                 fps_limit=10, error_reset_delay=10.0, save_all_data=True,
                 motion_extraction_method='average', buffer_dtype='float64', run_on_init=True, backend=None):
        """Arguments as reference base.py:24-34, plus (not in the reference):
        buffer_dtype -- element type of the calibration buffer in HBM: 'float64' (the reference's, base.py:119), 'float32',
            'float16', 'uint8' (the gray frame as ingested; uint8_to_float happens where the buffer is read) or 'bgr8' (a
            [T,H,W,3] uint8 buffer of the frames AS CAPTURED: cvtColor + uint8_to_float happen where the buffer is read; next_frame()
            keeps the device copy of the captured frame for step(), a frame from anywhere else is stored as its gray value in
            three planes -- the same calibration either way);
        run_on_init -- False: construct without entering run(); backend -- a stand-in for the device backend (tests)."""
        # argument contract of reference base.py:24-34
        assert isinstance(fps_limit, (int, float)) and fps_limit > 0, "fps_limit must be a positive int or float"
        assert isinstance(save_calibration_image, bool), "save_calibration_image must be bool"
        assert visualize == 'pyqtgraph' or visualize is None, "visualize must be 'pyqtgraph' or None"
        assert fig_size is None or (isinstance(fig_size, (tuple, list)) and len(fig_size) == 2), \
            "fig_size should be None or length 2 tuple or list"
        assert isinstance(error_reset_delay, (int, float)) and error_reset_delay >= 0, \
            "error_reset_delay must be a positive int or float"
        assert isinstance(save_all_data, bool), "save_all_data should be bool"
        assert motion_extraction_method in ("average", "flow"), "motion_extraction_method must be 'average' or 'flow'"
        assert buffer_dtype in ("float64", "float32", "float16", "uint8", "bgr8")

        self.benchmarker = Benchmarker()
        self.error_reset_delay = error_reset_delay
        self.save_all_data = save_all_data
        self.fig_size = fig_size
        self.save_calibration_image = save_calibration_image
        self.capture_target = capture_target
        self.visualize = visualize
        self.motion_extraction_method = motion_extraction_method
        self.buffer_dtype = buffer_dtype
        self._backend = backend if backend is not None else _Backend()

        self.cap = self._open_capture(capture_target)                       # base.py:48
        self.fps = int(self.cap.get(self.CAP_PROP_FPS))                     # base.py:49
        self.width = int(self.cap.get(self.CAP_PROP_FRAME_WIDTH))
        self.height = int(self.cap.get(self.CAP_PROP_FRAME_HEIGHT))

        # hyperparameters, reference base.py:80-106
        self.maximum_bounding_box_area = np.inf
        self.calibration_buffer_target_length = 128
        self.freq_min = 0.1
        self.freq_max = 1.0
        self.temporal_threshold = 0.7
        self.threshold = 0.08
        self.measure_buffer_length = 128
        self.confidence_interval = 0.95
        self.feature_params = dict(maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7)
        self.lk_params = dict(winSize=(15, 15), maxLevel=2,
                              criteria=(self.TERM_CRITERIA_EPS | self.TERM_CRITERIA_COUNT, 10, 0.03))
        self.gaussian_cutoff = 10.0
        self.filter_order = 3
        self.peak_minimum_sample_distance = 0
        self.measure_initialization_length = 12
        if self.fps == 0:
            self.fps = np.nan                                               # base.py:109-110
        self.fps_limit = fps_limit

        self.x, self.y, self.w, self.h = None, None, None, None
        self.disable_error_detection = False
        self.calibration_buffer_idx = 0
        self.calibration_buffer = self._alloc_buffer()                      # base.py:119-120, in HBM
        self.all_data = []
        self.data = deque()
        self.t = deque()
        self.freq = deque()
        self.confidence = deque()
        self.num_peaks = deque()
        self.num_peaks_mean = deque()
        self.motion_data = deque()
        self.filtered_data = []
        self.peak_indices = []
        self.peak_times = []
        self._frame_u8 = None            # current gray frame, device uint8 [H,W]
        self._frame_bgr = None           # ... and, for buffer_dtype 'bgr8' only, the captured frame it came from, device uint8 [H,W,3]
        self.cropped_image = None        # (x, y, w, h) view descriptor of the current frame
        self.previous_cropped_image = None
        self.display_frame = None
        self.motion_key_points = None
        self.video_out = None
        self.error_message = None
        self.buffers = [self.data, self.confidence, self.t, self.freq, self.num_peaks, self.num_peaks_mean,
                        self.motion_data]
        self.state = 'initialize'
        logging.info("Capturing {0} calibration frames.".format(self.calibration_buffer_target_length))
        self.calibration_start_time = np.nan
        self.loop_start_time = np.nan
        self.reset_start_time = np.nan
        self.frames_consumed = 0
        if run_on_init:
            self.run()                                                      # base.py:164

    # ------------------------------------------------------------------ plumbing
    def _open_capture(self, target):
        if all(hasattr(target, a) for a in ("isOpened", "read", "get", "release")):
            return target
        try:
            import cv2  # not part of this build's dependencies; only needed for real cameras / files
        except ImportError:
            raise _capi.RespmonError("capture_target=%r needs OpenCV's VideoCapture, which is not installed; pass an "
                                     "object with isOpened/read/get/release (e.g. respmon_amd.synth.FakeCapture)" % (target,))
        return cv2.VideoCapture(target)

    def _alloc_buffer(self):
        if isinstance(self._backend, _Backend):
            t = device.require_gpu()
            if self.buffer_dtype == "bgr8":    # north_star's [T,H,W,C] frame buffer: frames as captured, 3 bytes per pixel
                return t.zeros((self.calibration_buffer_target_length, self.height, self.width, 3), dtype=t.uint8, device="cuda")
            dt = {"float64": t.float64, "float32": t.float32, "float16": t.float16, "uint8": t.uint8}[self.buffer_dtype]
            return t.zeros((self.calibration_buffer_target_length, self.height, self.width), dtype=dt, device="cuda")
        return self._backend.alloc_buffer(self.calibration_buffer_target_length, self.height, self.width, self.buffer_dtype)

    @property
    def current_frame(self):
        """float64 [H,W] host copy of the current frame (uint8_to_float(gray), base.py:231)."""
        if self._frame_u8 is None:
            return None
        g = self._frame_u8
        g = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
        return g * (1. / 255)

    # ------------------------------------------------------------------ reference methods
    def skip_calibration(self, x, y, w, h):
        """base.py:166-172."""
        self.x, self.y, self.w, self.h = x, y, w, h
        self.peak_minimum_sample_distance = int(np.floor(self.fps / self.freq_max))
        self.state = 'measure'

    def next_frame(self):
        """base.py:227-233: read -> BGR2GRAY -> (uint8_to_float is applied by the consumers); False at end."""
        ret, frame = self.cap.read()
        if ret:
            self._frame_u8 = self._backend.bgr_to_gray(frame)
            # a 'bgr8' calibration buffer stores the frame as captured: keep the device copy the backend made of it for step()
            # (only then: nothing else needs it, and it is released with the next frame)
            self._frame_bgr = getattr(self._backend, "last_bgr", None) if self.buffer_dtype == "bgr8" else None
            self.frames_consumed += 1
            return self._frame_u8
        return False

    def initialize(self):
        """base.py:299-301."""
        self.calibration_start_time = time.time()
        self.calibration_buffer_idx = 0

    def detect_fps(self):
        """base.py:303-310: measured fps if the capture reports none, then clamp to fps_limit."""
        if self.fps == 0 or self.fps is np.nan:
            self.fps = self.calibration_buffer_target_length / (time.time() - self.calibration_start_time)
            logging.info("Computer FPS as {0}.".format(self.fps))
        if self.fps > self.fps_limit:
            self.fps = self.fps_limit
            logging.info("FPS Limited to {0}.".format(self.fps))
        logging.info("Final FPS is {0}.".format(self.fps))

    def trigger_error(self, message):
        """base.py:249-253."""
        self.error_message = message
        self.reset_start_time = time.time()
        logging.warning(message)
        self.state = 'error'

    def detect_errors(self):
        """base.py:543-545 (identity test against np.nan, as the reference does)."""
        if self.data[-1] is np.nan:
            return True

    def reset(self):
        """base.py:515-533 without the UI calls."""
        self.state = 'initialize'
        for b in self.buffers:
            b.clear()
        self.filtered_data = []
        self.peak_indices = []
        self.peak_times = []
        self.calibration_buffer_idx = 0
        self.previous_cropped_image = None
        self.motion_key_points = None

    def sync_to_fps(self):
        """base.py:535-541."""
        fps_x = self.fps_limit if self.fps is np.nan else self.fps
        sleep_time = (1.0 / fps_x) - (time.time() - self.loop_start_time)
        if sleep_time > 0:
            time.sleep(sleep_time)

    def update_ui(self):
        """base.py:255-297: no UI in this build."""

    # ------------------------------------------------------------------ BPM estimation ("next" row f2)
    def find_peaks(self):
        """base.py:312-338 with own restatements of peakutils.indexes / gaussian_fit (peakutils is an
        un-pinned dependency that is not installable here: parity unpinned)."""
        from . import peaks
        width = self.peak_minimum_sample_distance
        idxs = peaks.indexes(np.asarray(self.filtered_data), min_dist=width)
        final, fits = [], []
        t_arr = np.array(self.t)
        f_arr = np.array(self.filtered_data)
        for idx in idxs:
            w = width
            if idx - width < 0:
                w = idx
            if idx + w > len(self.t):
                w = len(self.t) - idx
            ti, di = t_arr[idx - w:idx + w], f_arr[idx - w:idx + w]
            try:
                params = peaks.gaussian_fit(ti, di)
                fits.append(0.0)  # the reference's r2 is identically 0 (ssr == sst, base.py:330-332)
                if params[2] < self.gaussian_cutoff:
                    final.append(idx)
            except (RuntimeError, TypeError, ValueError):
                pass
        return final, fits

    def measure(self):
        """base.py:340-352."""
        self.filtered_data = np.array(butter_lowpass_filter(self.data, self.freq_max * 0.5, self.fps, self.filter_order))
        self.peak_indices, _fits = self.find_peaks()
        self.peak_times = np.take(self.t, self.peak_indices)
        diffs = [a - b for b, a in zip(self.peak_times, self.peak_times[1:])]
        if len(diffs) > 0:
            self.freq.append(60.0 / np.mean(diffs))

    # ------------------------------------------------------------------ hot path B
    def extract_motion(self):
        """base.py:354-407."""
        x, y, w, h = self.cropped_image
        if self.motion_extraction_method == "average":
            return self._backend.roi_mean(self._frame_u8, x, y, w, h)       # np.average(crop), base.py:357
        be = self._backend
        if self.fused_flow_step and hasattr(be, "flow_step"):
            return self._extract_motion_flow_resident(be, x, y, w, h)
        if self.previous_cropped_image is None:                            # base.py:363-369
            self.previous_cropped_image = be.roi_to_uint8(self._frame_u8, x, y, w, h)
            self.motion_key_points = be.good_features_to_track(self.previous_cropped_image, **self.feature_params)
            if self.motion_key_points is None or len(self.motion_key_points) < 1:
                self.trigger_error("No motion key points found.")
            return 0.0
        cur = be.roi_to_uint8(self._frame_u8, x, y, w, h)
        if self.motion_key_points is None or len(self.motion_key_points) == 0:
            return np.nan
        p1, st = be.calc_optical_flow_pyr_lk(self.previous_cropped_image, cur, self.motion_key_points, **self.lk_params)
        mean, n_good = be.mean_flow(self.motion_key_points, p1, st)         # base.py:377-378, 388
        self.previous_cropped_image = cur                                   # base.py:381
        self.motion_key_points = p1[st == 1].reshape(-1, 1, 2)              # base.py:382
        if n_good == 0:
            return np.nan                                                   # base.py:385-386
        self.motion_data.append([mean[0], mean[1]])                         # base.py:389
        if len(self.motion_data) >= 2:
            return be.pca_reduce(np.array(self.motion_data, dtype=np.float32))  # base.py:396-405
        return 0.0

    # One C-ABI call per frame (rm_flow_begin / rm_flow_step): the previous crop and the tracked points stay on the device, so
    # `previous_cropped_image` only marks that tracking has begun and `motion_key_points` is fetched when somebody reads it.
    # False: the reference's four steps as four calls, points and status through host memory (same numbers).
    fused_flow_step = True
    _RESIDENT = "on the device (rm_flow_step)"

    def _extract_motion_flow_resident(self, be, x, y, w, h):
        if self.previous_cropped_image is None:                            # base.py:363-369
            if getattr(self, "_flow_state", None) is None:
                self._flow_state = be.flow_state()                         # this monitor's own tracking session
            self.motion_key_points = be.flow_begin(self._flow_state, self._frame_u8, x, y, w, h, **self.feature_params)
            self.previous_cropped_image = self._RESIDENT
            self._flow_cap = max(int(self.feature_params["maxCorners"]), 1)
            self._flow_n = 0 if self.motion_key_points is None else len(self.motion_key_points)
            if self._flow_n < 1:
                self.trigger_error("No motion key points found.")
            return 0.0
        if self.previous_cropped_image is not self._RESIDENT:              # tracking was begun by the four-call path
            self.fused_flow_step = False
            return self.extract_motion()
        if self._flow_n == 0:
            be.flow_step(self._flow_state, self._frame_u8, x, y, w, h, **self.lk_params)     # (the previous image still advances, base.py:381)
            return np.nan
        mean, n_good = be.flow_step(self._flow_state, self._frame_u8, x, y, w, h, **self.lk_params)   # base.py:371-388
        self._flow_n = n_good
        self._points_stale = True
        if n_good == 0:
            return np.nan                                                   # base.py:385-386
        self.motion_data.append([mean[0], mean[1]])                         # base.py:389
        if len(self.motion_data) >= 2:
            return be.pca_reduce(np.array(self.motion_data, dtype=np.float32))  # base.py:396-405
        return 0.0

    @property
    def motion_key_points(self):
        if getattr(self, "_points_stale", False):
            self._motion_key_points = self._backend.flow_points(self._flow_state, self._flow_cap)
            self._points_stale = False
        return getattr(self, "_motion_key_points", None)

    @motion_key_points.setter
    def motion_key_points(self, value):
        self._motion_key_points = value
        self._points_stale = False

    # Which cv2.findContours the ROI stage reproduces (the reference pins no OpenCV version, README.md:12): False = OpenCV >= 3.2
    # (pixels on the image frame count), True = OpenCV <= 3.1 (the 1-pixel frame is zeroed before tracing; base.py:567's
    # `thresh_copy` exists because that version mutated its input).  include/respmon_hip.h rm_set_contour_clip_frame.
    opencv_contours_clip_frame = False
    # Operation order of the calibration.  False (default): the temporal filter runs on the Gaussian level G_S and the Laplacians are
    # taken of the filtered images -- the linear stages commuted, results within ~1e-15 of the band-passed magnitudes of the
    # reference's order.  True: the reference's own order (Laplacians first, transforms.py:148-170; RM_FLAG_FILTER_LAPLACIANS),
    # bit-identical to the per-level kernels, a few per cent slower.  locate() is a static method (as in the reference, base.py:547):
    # both switches are read from RespiratoryMonitor itself, not from a subclass.
    reference_operation_order = False

    # ------------------------------------------------------------------ hot path A
    @staticmethod
    def locate(calibration_video_data, fps,
               freq_min=0.1, freq_max=1.0, amplification=500,
               pyramid_levels=9, skip_levels_at_top=4, temporal_threshold=0.7,
               threshold=20, threshold_type=THRESH_BINARY,
               verbose=False, save_calibration_image=False):
        """base.py:547-601 -> (x, y, w, h) or None.  Fused device path: the frame buffer is read once
        and no [T,H,W] intermediate is written (rm_locate)."""
        if threshold_type != THRESH_BINARY:
            raise NotImplementedError("only cv2.THRESH_BINARY is used by the reference (base.py:448,551)")
        logging.info("Beginning processing calibration frames...")
        buf = device.to_device(calibration_video_data)
        roi = _Backend().locate(buf, fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top,
                                temporal_threshold, threshold,
                                flags=(_capi.RM_FLAG_CONTOUR_CLIP_FRAME if RespiratoryMonitor.opencv_contours_clip_frame else 0) |
                                      (_capi.RM_FLAG_FILTER_LAPLACIANS if RespiratoryMonitor.reference_operation_order else 0))
        if save_calibration_image and roi is not None:     # base.py:577-596 (the reference returns before it when no contour)
            from . import montage
            logging.info('Creating calibration image.')
            path, _ = montage.save_calibration_image(buf, fps, freq_min=freq_min, freq_max=freq_max, amplification=amplification,
                                                     pyramid_levels=pyramid_levels, skip_levels_at_top=skip_levels_at_top,
                                                     temporal_threshold=temporal_threshold, threshold=threshold)
            logging.info('Calibration image saved: %s', path)
        if verbose and roi is not None:
            print('x:{0}, y:{1}, w:{2}, h:{3}'.format(*roi))
        return roi

    calibrate = locate  # north_star names a calibrate(); the reference's calibration entry is locate()

    # ------------------------------------------------------------------ state machine
    def run(self):
        """base.py:409-513.  Frame accounting: frame 0 is consumed by 'initialize', the next
        T fill the buffer, one more triggers locate() and is dropped, measurement starts after that."""
        for tag in ('Measurement Loop', 'Frame Capture', 'Calibration Measurement'):
            if not self.benchmarker.has_tag(tag):
                self.benchmarker.add_tag(tag)
        while self.cap.isOpened():
            self.loop_start_time = time.time()
            self.benchmarker.tick_start('Frame Capture')
            frame = self.next_frame()
            if isinstance(frame, bool):
                break
            self.benchmarker.tick_end('Frame Capture')
            self.step(frame)
            self.update_ui()
            self.sync_to_fps()
        logging.info("Capture closed.")
        self.cap.release()

    def step(self, frame):
        """One iteration of the reference's state dispatch (base.py:423-500) on an already captured frame."""
        if self.state == 'initialize':
            self.initialize()
            self.state = 'calibration'
        elif self.state == 'calibration':
            if self.calibration_buffer_idx < self.calibration_buffer_target_length:
                if self.buffer_dtype == "bgr8":
                    bgr = self._frame_bgr if frame is getattr(self, "_frame_u8", None) else None   # (a frame step() was handed from elsewhere: its gray value, three planes)
                    self._backend.store_frame(self.calibration_buffer, self.calibration_buffer_idx, frame, bgr=bgr)
                else:
                    self._backend.store_frame(self.calibration_buffer, self.calibration_buffer_idx, frame)
                self.calibration_buffer_idx += 1
            else:
                logging.info("Finished capturing calibration frames. Beginning calibration...")
                self.detect_fps()
                self.peak_minimum_sample_distance = int(np.floor(self.fps / self.freq_max))
                self.benchmarker.tick_start('Calibration Measurement')
                location = self._locate_buffer()
                self.benchmarker.tick_end('Calibration Measurement')
                if location is None:
                    logging.info("Failed finding ROI during calibration. Retrying...")
                    self.calibration_buffer_idx = 0
                    return
                self.x, self.y, self.w, self.h = reduce_bounding_box(*location, self.maximum_bounding_box_area)
                logging.info("Finished calibration.")
                logging.info("Beginning measuring...")
                self.state = 'measure'
        elif self.state == 'measure':
            self.benchmarker.tick_start('Measurement Loop')
            self.cropped_image = (self.x, self.y, self.w, self.h)           # base.py:471 (a view descriptor)
            for b in self.buffers:                                          # base.py:473-475
                if len(b) >= self.measure_buffer_length:
                    b.popleft()
            value = self.extract_motion()
            self.data.append(value)
            if len(self.t) == 0:
                self.t.append(0.)
            else:
                self.t.append(self.t[-1] + (1. / self.fps))
            if self.save_all_data:
                self.all_data.append((self.t[-1], value))
            if len(self.data) > self.measure_initialization_length:
                self.measure()
                if not self.disable_error_detection and self.detect_errors():
                    self.trigger_error("error detection found poor signal")
            self.benchmarker.tick_end('Measurement Loop')
        elif self.state == 'error':
            if time.time() - self.reset_start_time >= self.error_reset_delay:
                logging.info('Benchmark Report...\r\n' + self.benchmarker.get_report())
                self.reset()
                self.state = 'calibration'

    def _locate_buffer(self):
        """the locate() call of run(), base.py:444-448: threshold = int(round(0.08*255)) = 20; pyramid_levels=9,
        skip_levels_at_top=4, amplification=500 are locate's defaults."""
        thr = int(np.round(self.threshold * 255))
        roi = self._backend.locate(self.calibration_buffer, self.fps, self.freq_min, self.freq_max, 500, 9, 4,
                                   self.temporal_threshold, thr)
        if self.save_calibration_image and roi is not None:     # base.py:448 passes the flag on; montage at base.py:577-596
            from . import montage
            path, _ = montage.save_calibration_image(self.calibration_buffer, self.fps, freq_min=self.freq_min, freq_max=self.freq_max,
                                                     temporal_threshold=self.temporal_threshold, threshold=thr)
            logging.info('Calibration image saved: %s', path)
        return roi
