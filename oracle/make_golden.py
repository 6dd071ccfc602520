"""
oracle/make_golden.py -- generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF
(through oracle/ref_loader.py) in the build container.  Run once:

    python -m oracle.make_golden

Only input/output arrays are saved.  The reference's OpenCV calls are served by the build's
own restatement (oracle/cvref*.c), so fixtures pin the reference's numpy/scipy GLUE; the cv2
arithmetic itself stays "parity unpinned" (see oracle/respmon_oracle.py header).

Fixture ids follow SURVEY.md section 8c: G1 temporal FFT filter, G2 uint8<->float LUT,
G3 eulerian_magnification_bandpass, G4 locate, G5 extract_motion (scripted LK), G6 run()
frame accounting, G7 reduce_bounding_box / butter_lowpass_filter, G8 IIR temporal_bandpass_filter (row f4).
"""
import hashlib
import os
import sys
import types
import warnings
from collections import deque

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from respmon_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def g1(R):
    cases = [(128, 10), (256, 10), (512, 10), (128, 30), (64, 10), (128, 5.01), (128, 7.68), (32, 2.5)]
    out = {}
    for i, (n, fps) in enumerate(cases):
        rng = np.random.Generator(np.random.PCG64(100 + i))
        x = rng.standard_normal((n, 3, 5))
        y = R.transforms.temporal_bandpass_filter_fft(x.copy(), fps, freq_min=0.1, freq_max=1.0,
                                                      amplification_factor=500)
        out["x%d" % i] = x
        out["y%d" % i] = y
        out["meta%d" % i] = np.array([n, fps, 0.1, 1.0, 500.0])
        if n <= 128:
            eye = np.eye(n).reshape(n, n, 1)
            out["M%d" % i] = R.transforms.temporal_bandpass_filter_fft(
                eye, fps, freq_min=0.1, freq_max=1.0, amplification_factor=1.0).reshape(n, n)
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "g1_temporal_fft.npz"), **out)


def g2(R):
    k = np.arange(256, dtype=np.uint8)
    f = R.transforms.uint8_to_float(k)
    lut = R.transforms.float_to_uint8(f)
    edge_in = np.array([0.0, 1.0, 0.5, 0.999999, 1.0 / 255, 254.9999999 / 255, 20.0 / 255, 21.0 / 255, np.nan])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        edge_out = R.transforms.float_to_uint8(edge_in)
    np.savez_compressed(os.path.join(OUT, "g2_u8_float_lut.npz"), f=f, lut=lut, edge_in=edge_in, edge_out=edge_out)


def g3(R):
    out = {}
    cases = [(4, 2, 10.0), (6, 2, 2.5), (6, 4, 10.0)]
    vid_u8 = synth.synth_breathing(32, 36, 44, seed=77, amplitude=0.25)
    vid = R.transforms.uint8_to_float(vid_u8)
    out["vid_u8"] = vid_u8
    for i, (L, S, fps) in enumerate(cases):
        masked, raw = R.transforms.eulerian_magnification_bandpass(vid.copy(), fps, 0.1, 1.0, 500,
                                                                   pyramid_levels=L, skip_levels_at_top=S)
        out["raw%d" % i] = raw
        out["masked_sha%d" % i] = np.array(sha(masked))
        out["masked_stats%d" % i] = np.array([raw.min(), raw.max(), float((masked == raw.min()).sum())])
        out["avg%d" % i] = np.average(masked, axis=0)
        out["meta%d" % i] = np.array([L, S, fps])
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "g3_eulerian.npz"), **out)


def g4(R):
    out = {}
    cases = [
        dict(T=32, H=120, W=160, seed=5, L=9, S=4, fps=10),
        dict(T=32, H=96, W=128, seed=6, L=6, S=2, fps=10),
        dict(T=64, H=90, W=122, seed=7, L=7, S=3, fps=10),
    ]
    for i, c in enumerate(cases):
        vid_u8 = synth.synth_breathing(c["T"], c["H"], c["W"], seed=c["seed"])
        vid = R.transforms.uint8_to_float(vid_u8)
        captured = {}
        orig = R.transforms.float_to_uint8

        def spy(img, _orig=orig, _cap=captured):
            r = _orig(img)
            _cap.setdefault("avg_u8", r)
            return r

        R.base.float_to_uint8 = spy
        try:
            roi = R.base.RespiratoryMonitor.locate(vid, c["fps"], pyramid_levels=c["L"], skip_levels_at_top=c["S"])
        finally:
            R.base.float_to_uint8 = orig
        out["roi%d" % i] = np.array(roi if roi is not None else (-1, -1, -1, -1))
        out["avg_u8_%d" % i] = captured["avg_u8"]
        out["vid_sha%d" % i] = np.array(sha(vid_u8))
        out["meta%d" % i] = np.array([c["T"], c["H"], c["W"], c["seed"], c["L"], c["S"], c["fps"]])
    # no-contour case: a constant video -> flat heatmap -> None
    const = np.full((16, 40, 48), 0.5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        roi = R.base.RespiratoryMonitor.locate(const, 10, pyramid_levels=4, skip_levels_at_top=2)
    out["roi_const_is_none"] = np.array(roi is None)
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "g4_locate.npz"), **out)


def g5(R):
    """extract_motion('flow') with a scripted LK: pins mean-flow sign (old - new), float32 mean,
    row-unpack PCA and LAPACK eig signs (authentic numpy)."""
    rng = np.random.Generator(np.random.PCG64(55))
    npts, nframes = 7, 40
    pts0 = (rng.uniform(5, 60, (npts, 1, 2))).astype(np.float32)
    script_p1, script_st = [], []
    cur = pts0.copy()
    alive = np.ones(npts, bool)
    for t in range(nframes):
        d = np.array([0.8 * np.sin(0.5 * t), 0.3 * np.cos(0.5 * t) + 0.1], dtype=np.float32)
        n_alive = int(alive.sum())
        p1 = (cur[alive] + d + rng.normal(0, 0.02, (n_alive, 1, 2))).astype(np.float32)
        st = np.ones((n_alive, 1), np.uint8)
        if t in (9, 21) and n_alive > 3:
            st[rng.integers(0, n_alive)] = 0
        script_p1.append(p1)
        script_st.append(st)
        keep = st.ravel() == 1
        idx = np.flatnonzero(alive)
        alive[idx[~keep]] = False
        cur = cur.copy()
        cur[idx[keep]] = p1[keep]
    it = iter(zip(script_p1, script_st))
    R.cv2.goodFeaturesToTrack_saved = R.cv2.goodFeaturesToTrack
    R.cv2.calcOpticalFlowPyrLK_saved = R.cv2.calcOpticalFlowPyrLK
    R.cv2.goodFeaturesToTrack = lambda img, mask=None, **kw: pts0.copy()

    def scripted_lk(a, b, p, n, **kw):
        p1, st = next(it)
        assert len(p1) == len(p)
        return p1.copy(), st.copy(), None

    R.cv2.calcOpticalFlowPyrLK = scripted_lk
    self = types.SimpleNamespace(motion_extraction_method="flow", previous_cropped_image=None,
                                 cropped_image=np.full((64, 64), 0.5), motion_key_points=None,
                                 motion_data=deque(), feature_params={}, lk_params={},
                                 trigger_error=lambda msg: None)
    vals = []
    try:
        for t in range(nframes + 1):
            vals.append(R.base.RespiratoryMonitor.extract_motion(self))
    finally:
        R.cv2.goodFeaturesToTrack = R.cv2.goodFeaturesToTrack_saved
        R.cv2.calcOpticalFlowPyrLK = R.cv2.calcOpticalFlowPyrLK_saved
    out = dict(pts0=pts0, values=np.array(vals, dtype=np.float64),
               motion_data=np.array([[float(a), float(b)] for a, b in self.motion_data], dtype=np.float64),
               motion_data_f32=np.array(self.motion_data, dtype=np.float32), nframes=np.array(nframes))
    for t in range(nframes):
        out["p1_%d" % t] = script_p1[t]
        out["st_%d" % t] = script_st[t]
    np.savez_compressed(os.path.join(OUT, "g5_extract_motion.npz"), **out)


def _make_monitor(R, frames_bgr, fps, **kw):
    cls = R.base.RespiratoryMonitor
    real_run = cls.run
    R.cv2.VideoCapture = lambda target: synth.FakeCapture(frames_bgr, fps=fps)
    cls.run = lambda self: None
    try:
        mon = cls(capture_target="golden", visualize=None, **kw)
    finally:
        cls.run = real_run
    mon.sync_to_fps = lambda: None
    return mon


def g6(R):
    out = {}
    # (i) config 1: skip_calibration + 'average' on a 64x240x320 brightness video
    frames = synth.synth_brightness_video(64, 240, 320)
    mon = _make_monitor(R, frames, 10, save_all_data=False, motion_extraction_method="average")
    mon.skip_calibration(100, 80, 70, 51)
    mon.run()
    out["c1_data"] = np.array(mon.data, dtype=np.float64)
    out["c1_t"] = np.array(mon.t, dtype=np.float64)
    out["c1_fps"] = np.array(mon.fps)
    out["c1_peak_min_dist"] = np.array(mon.peak_minimum_sample_distance)
    out["c1_roi"] = np.array([mon.x, mon.y, mon.w, mon.h])
    # (ii) full state machine on a small video: 1 init frame + 128 buffer frames + 1 locate frame + measure
    vid = synth.synth_breathing(150, 48, 64, seed=11)
    mon = _make_monitor(R, vid, 30, save_all_data=False, motion_extraction_method="average")
    trace = []
    locate_kwargs = {}
    real_next = mon.next_frame

    def traced_next():
        trace.append((["initialize", "calibration", "measure", "error"].index(mon.state),
                      mon.calibration_buffer_idx))
        return real_next()

    mon.next_frame = traced_next
    real_locate = R.base.RespiratoryMonitor.locate

    def spy_locate(buf, fps, **kw):
        locate_kwargs.update(kw)
        locate_kwargs["fps"] = fps
        locate_kwargs["buf_sha"] = sha(buf)
        return real_locate(buf, fps, **kw)

    mon.locate = spy_locate
    mon.run()
    out["c2_trace"] = np.array(trace, dtype=np.int32)
    out["c2_roi"] = np.array([mon.x, mon.y, mon.w, mon.h])
    out["c2_fps"] = np.array(mon.fps)
    out["c2_locate_kw"] = np.array([locate_kwargs["fps"], locate_kwargs["freq_min"], locate_kwargs["freq_max"],
                                    locate_kwargs["temporal_threshold"], locate_kwargs["threshold"]], dtype=np.float64)
    out["c2_buf_sha"] = np.array(locate_kwargs["buf_sha"])
    out["c2_data"] = np.array(mon.data, dtype=np.float64)
    out["c2_t"] = np.array(mon.t, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g6_run_trace.npz"), **out)


def g7(R):
    cases = [(10, 20, 100, 50, np.inf), (10, 20, 100, 50, 1250.0), (538, 243, 70, 51, 1000.0), (0, 0, 3, 3, 4.0),
             (5, 7, 33, 91, 100.0)]
    res = [R.tools.reduce_bounding_box(*c) for c in cases]
    rng = np.random.Generator(np.random.PCG64(7))
    sig = np.sin(2 * np.pi * 0.3 * np.arange(128) / 10.0) + 0.2 * rng.standard_normal(128)
    filt = R.transforms.butter_lowpass_filter(sig, 0.5, 10, 3)
    np.savez_compressed(os.path.join(OUT, "g7_misc.npz"), rbb_in=np.array(cases, dtype=np.float64),
                        rbb_out=np.array(res, dtype=np.int64), sig=sig, filt=filt)


def g8(R):
    """the IIR alternative temporal_bandpass_filter (transforms.py:72-79) -- authentic scipy --
    and eulerian_magnification_bandpass(temporal_filter_function=temporal_bandpass_filter)."""
    out = {}
    cases = [(128, 10.0, 0.1, 1.0, 500.0), (64, 30.0, 0.833, 1.0, 50.0), (200, 10.0, 0.4, 2.0, 20.0)]
    for i, (n, fps, fmin, fmax, amp) in enumerate(cases):
        rng = np.random.Generator(np.random.PCG64(300 + i))
        x = rng.standard_normal((n, 4, 6))
        out["x%d" % i] = x
        out["y%d" % i] = R.transforms.temporal_bandpass_filter(x.copy(), fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
        out["meta%d" % i] = np.array([n, fps, fmin, fmax, amp])
    out["ncases"] = np.array(len(cases))
    vid_u8 = synth.synth_breathing(48, 36, 44, seed=78, amplitude=0.25)
    vid = R.transforms.uint8_to_float(vid_u8)
    masked, raw = R.transforms.eulerian_magnification_bandpass(vid.copy(), 10.0, 0.1, 1.0, 500, pyramid_levels=5, skip_levels_at_top=2,
                                                               temporal_filter_function=R.transforms.temporal_bandpass_filter)
    out["e_vid_u8"] = vid_u8
    out["e_raw"] = raw
    out["e_avg"] = np.average(masked, axis=0)
    out["e_meta"] = np.array([5, 2, 10.0])
    np.savez_compressed(os.path.join(OUT, "g8_iir.npz"), **out)


def g9(R):
    """the reference's own measure() / find_peaks() (base.py:312-352), run unbound on a plain namespace.
    scipy's filtfilt is authentic; peakutils (un-pinned pip dependency, not installable here) is served by the build's
    restatement respmon_amd/peaks.py, so the fixture pins the reference's GLUE (windowing of the fits, the gaussian_cutoff
    test, peak_times -> BPM), not peakutils itself."""
    import types
    from respmon_amd import peaks
    R.peakutils.indexes = peaks.indexes
    R.peakutils.gaussian = peaks.gaussian
    R.peakutils.gaussian_fit = lambda x, y, center_only=True: (peaks.gaussian_fit(x, y)[1] if center_only else peaks.gaussian_fit(x, y))
    g6v = np.load(os.path.join(OUT, "g6_run_trace.npz"))
    rng = np.random.Generator(np.random.PCG64(99))
    n = 128
    cases = [
        (np.array(g6v["c1_data"], dtype=np.float64), np.array(g6v["c1_t"], dtype=np.float64), 10.0),        # config 1: 0.4 Hz brightness, 64 frames
        (np.sin(2 * np.pi * 0.3 * np.arange(n) / 10.0) + 0.15 * rng.standard_normal(n), np.arange(n) / 10.0, 10.0),
        (0.5 + 0.2 * np.sin(2 * np.pi * 0.25 * np.arange(n) / 7.5 + 0.7) + 0.02 * rng.standard_normal(n), np.arange(n) / 7.5, 7.5),
    ]
    out = {"ncases": np.array(len(cases))}
    for i, (data, t, fps) in enumerate(cases):
        self = types.SimpleNamespace(data=list(data), t=list(t), fps=fps, freq_max=1.0, filter_order=3, gaussian_cutoff=10.0,
                                     peak_minimum_sample_distance=int(np.floor(fps / 1.0)), freq=[])       # base.py:83,100,101,171
        self.find_peaks = lambda self=self: R.base.RespiratoryMonitor.find_peaks(self)
        R.base.RespiratoryMonitor.measure(self)
        out["data%d" % i] = np.array(data)
        out["t%d" % i] = np.array(t)
        out["fps%d" % i] = np.array(fps)
        out["filtered%d" % i] = np.array(self.filtered_data)
        out["peaks%d" % i] = np.array(self.peak_indices, dtype=np.int64)
        out["freq%d" % i] = np.array(self.freq, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g9_measure.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = ref_loader.load()
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9):
        if only and fn.__name__ not in only:
            continue
        fn(R)
        print("wrote", fn.__name__)


if __name__ == "__main__":
    main()
