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


def buf_args(frames):
    """(T, H, W, dtype code) of a frame buffer: [T,H,W] of a supported dtype, or [T,H,W,3] uint8 = RM_BGR8."""
    if frames.ndim == 4:
        assert frames.dtype == np.uint8 and frames.shape[3] == 3
        return frames.shape[0], frames.shape[1], frames.shape[2], _capi.RM_BGR8
    T, H, W = frames.shape
    return T, H, W, DT[frames.dtype]


class Emu:
    def __init__(self):
        from tests.emu import build as emu_build
        self.lib = _capi.bind(ctypes.CDLL(emu_build.build()))
        h = ctypes.c_void_p()
        self.ck(self.lib.rm_ctx_create(0, ctypes.byref(h)), "ctx_create")
        self.ctx = h

    def ck(self, rc, what):
        return _capi.check(self.lib, rc, what)

    def debug_set(self, key, value):
        self.ck(self.lib.rm_debug_set(self.ctx, key.encode(), int(value)), "debug_set")

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

    def lfilter(self, b, a, data, scale=1.0):
        data = np.ascontiguousarray(data, dtype=np.float64)
        T = data.shape[0]
        npix = data[0].size
        out = np.empty_like(data)
        b = np.ascontiguousarray(b, dtype=np.float64); a = np.ascontiguousarray(a, dtype=np.float64)
        n = max(len(b), len(a))
        b = np.concatenate([b, np.zeros(n - len(b))]); a = np.concatenate([a, np.zeros(n - len(a))])
        self.ck(self.lib.rm_lfilter(self.ctx, ptr(data), T, npix, ptr(b), ptr(a), n, float(scale), ptr(out), None), "lfilter")
        return out

    def threshold_mask(self, raw, thr=0.7):
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        masked = np.empty_like(raw); mm = np.empty(2)
        self.ck(self.lib.rm_threshold_mask(self.ctx, ptr(raw), raw.size, thr, ptr(masked), ptr(mm), None), "threshold_mask")
        return masked, mm

    def operator(self, T, fps, fmin, fmax):
        M = np.empty((T, T))
        lo, hi = ctypes.c_int(), ctypes.c_int()
        self.ck(self.lib.rm_temporal_operator(T, fps, fmin, fmax, ptr(M), ctypes.byref(lo), ctypes.byref(hi)), "operator")
        return M, lo.value, hi.value

    def eulerian(self, frames, fps, fmin, fmax, amp, levels, skip, thr=0.7):
        frames = np.ascontiguousarray(frames)
        T, H, W, code = buf_args(frames)
        masked = np.empty((T, H, W)); raw = np.empty((T, H, W)); mm = np.empty(2)
        self.ck(self.lib.rm_eulerian_magnification_bandpass(self.ctx, ptr(frames), code, T, H, W, fps, fmin, fmax,
                                                            amp, levels, skip, thr, ptr(masked), ptr(raw), ptr(mm), None), "eulerian")
        return masked, raw, mm

    def calibrate(self, frames, fps, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, flags=0):
        frames = np.ascontiguousarray(frames)
        T, H, W, code = buf_args(frames)
        heat = np.empty((H, W)); mm = np.empty(2)
        self.ck(self.lib.rm_calibrate(self.ctx, ptr(frames), code, T, H, W, fps, fmin, fmax, amp, levels, skip, thr,
                                      flags, ptr(heat), ptr(mm), None), "calibrate")
        return heat, mm

    def workspace(self, name, shape, dtype=np.float64):
        """A copy of the context's workspace buffer `name` (rm_debug_workspace) as an array of `shape`."""
        out = np.empty(shape, dtype=dtype)
        self.ck(self.lib.rm_debug_workspace(self.ctx, name.encode(), ptr(out), out.nbytes, None), "workspace")
        return out

    def counters(self):
        out = np.zeros(4, np.int64)
        self.ck(self.lib.rm_debug_counters(self.ctx, ptr(out), None), "counters")
        return [int(v) for v in out]

    def heatmap_to_roi(self, heat, threshold=20, clip_frame=False, labelling=-1):
        heat = np.ascontiguousarray(heat, dtype=np.float64)
        H, W = heat.shape
        xywh = np.zeros(4, np.int32); u8 = np.empty((H, W), np.uint8); b = np.empty((H, W), np.uint8)
        self.ck(self.lib.rm_set_contour_clip_frame(self.ctx, 1 if clip_frame else 0), "clip_frame")
        self.ck(self.lib.rm_set_contour_labelling(self.ctx, labelling), "labelling")
        rc = self.ck(self.lib.rm_heatmap_to_roi(self.ctx, ptr(heat), H, W, threshold, ptr(xywh), ptr(u8), ptr(b), None), "roi")
        self.lib.rm_set_contour_clip_frame(self.ctx, 0)
        self.lib.rm_set_contour_labelling(self.ctx, -1)
        return (None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)), u8, b

    def roi_path(self):
        """RM_ROI_PATH_* of the last host contour stage (include/respmon_hip_debug.h)."""
        v = ctypes.c_int(-1)
        self.ck(self.lib.rm_debug_roi_path(self.ctx, ctypes.byref(v)), "roi_path")
        return v.value

    def contour_stats(self):
        n, lab = ctypes.c_int(), ctypes.c_int()
        self.ck(self.lib.rm_contour_stats(self.ctx, ctypes.byref(n), ctypes.byref(lab)), "contour_stats")
        return n.value, lab.value

    def locate(self, frames, fps, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, threshold=20, flags=0):
        frames = np.ascontiguousarray(frames)
        T, H, W, code = buf_args(frames)
        xywh = np.zeros(4, np.int32)
        rc = self.ck(self.lib.rm_locate(self.ctx, ptr(frames), code, T, H, W, fps, fmin, fmax, amp, levels, skip, thr,
                                        threshold, flags, ptr(xywh), None), "locate")
        return None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)


    def locate_submit(self, frames, fps, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, threshold=20, flags=0):
        """rm_locate_submit -> (ticket, the frames kept alive) or the negative return code when the library refuses."""
        frames = np.ascontiguousarray(frames)
        T, H, W, code = buf_args(frames)
        tk = ctypes.c_int(-1)
        rc = self.lib.rm_locate_submit(self.ctx, ptr(frames), code, T, H, W, fps, fmin, fmax, amp, levels, skip, thr, threshold,
                                       flags, None, ctypes.byref(tk))
        if rc < 0:
            return rc
        return tk.value, frames

    def locate_result(self, ticket):
        xywh = np.zeros(4, np.int32)
        rc = self.ck(self.lib.rm_locate_result(self.ctx, ticket[0], ptr(xywh)), "locate_result")
        return None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)


def _shard_methods():
    def new_ctx(self):
        h = ctypes.c_void_p()
        self.ck(self.lib.rm_ctx_create(0, ctypes.byref(h)), "ctx_create")
        return h

    def locate_sharded(self, frames, world, fps=10.0, fmin=0.1, fmax=1.0, amp=500.0, levels=9, skip=4, thr=0.7, threshold=20, flags=0):
        """The rm_shard_* stages for `world` emulated ranks (a context each), collectives done with numpy."""
        frames = np.ascontiguousarray(frames)
        T, H, W, code = buf_args(frames)
        n = ctypes.c_size_t()
        self.ck(self.lib.rm_shard_layout_flags(H, W, levels, skip, flags, ctypes.byref(n)), "shard_layout")
        NP = int(n.value)
        spans = []
        base, rem = divmod(T, world)
        for r in range(world):
            t0 = r * base + min(r, rem)
            spans.append((t0, t0 + base + (1 if r < rem else 0)))
        ctxs = [self.new_ctx() for _ in range(world)]
        lap_all = np.zeros((T, max(NP, 1)))[:, :NP].copy()
        for c, (t0, t1) in zip(ctxs, spans):
            local = np.ascontiguousarray(frames[t0:t1])
            lap = np.empty((t1 - t0, NP))
            if NP:
                self.ck(self.lib.rm_shard_pyramid(c, ptr(local), code, t1 - t0, H, W, levels, skip, flags, ptr(lap), None),
                        "shard_pyramid")
            lap_all[t0:t1] = lap
        mms = []
        for c, (t0, t1) in zip(ctxs, spans):
            mm = np.empty(2)
            for _ in range(2):   # twice: the second call finds the reduction state used (not freshly reset by rm_shard_pyramid)
                self.ck(self.lib.rm_shard_collapse(c, ptr(lap_all) if NP else None, T, t0, t1, H, W, fps, fmin, fmax, amp, levels, skip,
                                                   thr, flags, ptr(mm), None), "shard_collapse")
            mms.append(mm)
        mm = np.max(np.stack(mms), axis=0)
        total = np.zeros((H, W))
        for c in ctxs:
            hs = np.empty((H, W))
            self.ck(self.lib.rm_shard_heat(c, ptr(mm), thr, ptr(hs), None), "shard_heat")
            total = total + hs
        heat = np.empty((H, W)); xywh = np.zeros(4, np.int32)
        rc = self.ck(self.lib.rm_shard_finish(ctxs[0], ptr(total), T, H, W, threshold, ptr(heat), ptr(xywh), None), "shard_finish")
        for c in ctxs:
            self.lib.rm_ctx_destroy(c)
        return (None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)), heat, (-mm[0], mm[1])

    def sparse_exchange(self, videos, cap, fps=10.0, levels=9, skip=4, threshold=20):
        """Mode B heatmap exchange for emulated ranks: calibrate + pack per rank (a context each), packets concatenated by
        hand (the all-gather), merge + ROI on rank 0.  Returns (rc, roi, fused, heats, counts)."""
        heats, packets = [], []
        pd = int(self.lib.rm_heat_sparse_packet_doubles(cap))
        ctxs = [self.new_ctx() for _ in videos]
        for c, v in zip(ctxs, videos):
            v = np.ascontiguousarray(v)
            T, H, W = v.shape
            heat = np.empty((H, W))
            self.ck(self.lib.rm_calibrate(c, ptr(v), DT[v.dtype], T, H, W, fps, 0.1, 1.0, 500.0, levels, skip, 0.7, 0, ptr(heat), None, None),
                    "calibrate")
            pk = np.zeros(pd)
            self.ck(self.lib.rm_heat_sparse_pack(c, ptr(heat), H, W, cap, ptr(pk), None), "sparse_pack")
            heats.append(heat); packets.append(pk)
        allp = np.concatenate(packets)
        fused = np.empty((H, W)); xywh = np.zeros(4, np.int32)
        rc = self.ck(self.lib.rm_heat_sparse_merge_roi(ctxs[0], ptr(allp), len(videos), H, W, cap, threshold, 0, ptr(fused), ptr(xywh), None),
                     "sparse_merge")
        for c in ctxs:
            self.lib.rm_ctx_destroy(c)
        counts = [int(pk[:1].view(np.uint32)[0]) for pk in packets]
        roi = tuple(int(v) for v in xywh) if rc == _capi.RM_OK else None
        return rc, roi, fused, heats, counts

    for f in (new_ctx, locate_sharded, sparse_exchange):
        setattr(Emu, f.__name__, f)


_shard_methods()


def _flow_methods():
    def good_features(self, img_u8, maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7):
        img = np.ascontiguousarray(img_u8, dtype=np.uint8)
        h, w = img.shape
        cap = max(int(maxCorners), 1) if maxCorners > 0 else img.size
        pts = np.empty((cap, 2), np.float32)
        n = ctypes.c_int()
        self.ck(self.lib.rm_good_features_to_track(self.ctx, ptr(img), h, w, int(maxCorners), float(qualityLevel), float(minDistance),
                                                   int(blockSize), ptr(pts), ctypes.byref(n), None), "gftt")
        return None if n.value == 0 else pts[:n.value].reshape(-1, 1, 2).copy()

    def pyr_lk(self, prev, nxt, pts, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03)):
        a = np.ascontiguousarray(prev, dtype=np.uint8); b = np.ascontiguousarray(nxt, dtype=np.uint8)
        h, w = a.shape
        p0 = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        n = len(p0)
        p1 = np.empty((n, 2), np.float32); st = np.empty(n, np.uint8)
        self.ck(self.lib.rm_calc_optical_flow_pyr_lk(self.ctx, ptr(a), ptr(b), h, w, ptr(p0), n, winSize[0], winSize[1], maxLevel,
                                                     criteria[1], float(criteria[2]), ptr(p1), ptr(st), None), "lk")
        return p1.reshape(-1, 1, 2), st.reshape(-1, 1)

    def mean_flow(self, old, new, st):
        o = np.ascontiguousarray(old, np.float32).reshape(-1, 2); nw = np.ascontiguousarray(new, np.float32).reshape(-1, 2)
        s = np.ascontiguousarray(st, np.uint8).reshape(-1)
        m = np.empty(2, np.float32); ng = ctypes.c_int()
        self.ck(self.lib.rm_mean_flow(self.ctx, ptr(o), ptr(nw), ptr(s), len(o), ptr(m), ctypes.byref(ng), None), "mean_flow")
        return m, ng.value

    def pca_reduce(self, motion):
        m = np.ascontiguousarray(motion, np.float32).reshape(-1, 2)
        out = ctypes.c_double()
        self.ck(self.lib.rm_pca_reduce(self.ctx, ptr(m), len(m), ctypes.byref(out), None), "pca")
        return out.value

    def roi_mean(self, frame, x, y, w, h):
        f = np.ascontiguousarray(frame)
        H, W = f.shape
        out = ctypes.c_double()
        self.ck(self.lib.rm_roi_mean(self.ctx, ptr(f), DT[f.dtype], H, W, x, y, w, h, ctypes.byref(out), None), "roi_mean")
        return out.value

    def flow_state(self):
        h = ctypes.c_void_p()
        self.ck(self.lib.rm_flow_state_create(self.ctx, ctypes.byref(h)), "flow_state_create")
        return h

    def flow_begin(self, frame_u8, x, y, w, h, maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7, state=None):
        f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
        H, W = f.shape
        pts = np.empty((max(int(maxCorners), 1), 2), np.float32)
        n = ctypes.c_int()
        if getattr(self, "_fstate", None) is None:
            self._fstate = self.flow_state()
        self.ck(self.lib.rm_flow_begin(self.ctx, (state or self._fstate), ptr(f), DT[f.dtype], H, W, x, y, w, h, int(maxCorners), float(qualityLevel),
                                       float(minDistance), int(blockSize), ptr(pts), ctypes.byref(n), None), "flow_begin")
        return None if n.value == 0 else pts[:n.value].reshape(-1, 1, 2).copy()

    def flow_step(self, frame_u8, x, y, w, h, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03), state=None):
        f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
        H, W = f.shape
        m = np.empty(2, np.float32); ng = ctypes.c_int()
        self.ck(self.lib.rm_flow_step(self.ctx, (state or self._fstate), ptr(f), DT[f.dtype], H, W, x, y, w, h, winSize[0], winSize[1], maxLevel, criteria[1],
                                      float(criteria[2]), ptr(m), ctypes.byref(ng), None), "flow_step")
        return m, ng.value

    def flow_points(self, cap=1000, state=None):
        pts = np.empty((cap, 2), np.float32); n = ctypes.c_int()
        self.ck(self.lib.rm_flow_points(self.ctx, (state or self._fstate), ptr(pts), cap, ctypes.byref(n), None), "flow_points")
        return pts[:min(n.value, cap)].reshape(-1, 1, 2).copy()

    for f in (good_features, pyr_lk, mean_flow, pca_reduce, roi_mean, flow_state, flow_begin, flow_step, flow_points):
        setattr(Emu, f.__name__, f)


_flow_methods()
