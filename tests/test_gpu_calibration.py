"""Parity tests proper: the gfx950 HIP path (through the C-ABI / the drop-in Python surface) against
the CPU oracle and the committed golden vectors.  Bit-exact for ROI coordinates and the uint8 heatmap;
float magnitudes within 1e-4 relative (north_star) -- in practice ~1e-15."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4  # north_star tolerance for FFT magnitudes


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from respmon_amd import _capi
    return _capi.load()  # raises if the HIP extension is missing: no fallback


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _close(a, b, tol=1e-12):
    """device tensors equal to rounding (two float64 operation orders of the same linear pipeline)"""
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-300)


def test_native_library_is_loaded(hip):
    import os
    from respmon_amd import _capi
    assert os.path.basename(_capi.LIB_PATH) == "librespmon_hip.so"
    with open("/proc/self/maps") as f:
        assert "librespmon_hip.so" in f.read()
    assert hip.rm_abi_version() == 1


def test_dtype_helpers_lut(hip, golden):
    from respmon_amd import transforms
    g = golden("g2_u8_float_lut.npz")
    k = np.arange(256, dtype=np.uint8)
    f = transforms.uint8_to_float(k)
    assert np.array_equal(f, g["f"])
    assert np.array_equal(transforms.float_to_uint8(f), g["lut"])
    assert np.array_equal(transforms.float_to_uint8(g["edge_in"]), g["edge_out"])


def test_pyramid_functions_bit_exact(hip, oracle):
    from respmon_amd import pyramid
    rng = np.random.default_rng(0)
    for shape, L in [((3, 21, 30), 3), ((2, 64, 96), 5), ((2, 9, 15), 4), ((1, 135, 240), 6), ((2, 1, 1), 2)]:
        vid = rng.random(shape)
        got = pyramid.create_laplacian_video_pyramid(vid, L)
        ref = oracle.create_laplacian_video_pyramid(vid, L)
        assert len(got) == L
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.array_equal(a, b)
        col = pyramid.collapse_laplacian_video_pyramid([x.copy() for x in got])
        col_ref = oracle.collapse_laplacian_video_pyramid([x.copy() for x in ref])
        assert np.array_equal(col, col_ref)
        assert _rel(col, vid) < 1e-12  # Laplacian pyramid round trip reconstructs the video


def test_temporal_filter_golden(hip, golden, oracle):
    from respmon_amd import transforms
    g = golden("g1_temporal_fft.npz")
    for i in range(int(g["ncases"])):
        n, fps, fmin, fmax, amp = g["meta%d" % i]
        y = transforms.temporal_bandpass_filter_fft(g["x%d" % i], fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
        assert _rel(y, g["y%d" % i]) < 1e-11
        M, lo, hi = transforms.temporal_operator(int(n), fps, fmin, fmax)
        if "M%d" % i in g.files:
            assert np.abs(M - g["M%d" % i]).max() < 1e-15
        # the three forms of the product -- K-split per 16 pixels (small levels), a wave per 16 pixel columns with the operator fragments
        # requested a group ahead (large levels: 4K x 512), the VALU twin (north_star: "no MFMA") -- forced on the same data
        from respmon_amd import device
        for knob in ("temporal_wide", "temporal_valu"):
            device.debug_set(knob, 1)
            try:
                y2 = transforms.temporal_bandpass_filter_fft(g["x%d" % i], fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
            finally:
                device.debug_set(knob, -1 if knob == "temporal_wide" else 0)
            assert _rel(y2, g["y%d" % i]) < 1e-11, (knob, int(n))
    rng = np.random.default_rng(12)
    for (T, fps, fmin, fmax, npx) in [(2, 10, 0.1, 6.0, 37), (3, 10, 0.1, 6.0, 37), (7, 10, 0.5, 3.0, 100), (31, 30, 0.5, 9.0, 1000), (62, 10, 0.1, 1.0, 5000),
                                      (66, 10, 0.3, 1.0, 4099), (97, 5.01, 0.1, 1.0, 777), (130, 10, 0.1, 2.5, 20000), (512, 10, 0.1, 1.0, 70000),
                                      (511, 10, 0.1, 1.0, 3000), (518, 30, 0.2, 3.0, 3000)]:
        x = rng.standard_normal((T, npx))
        want = oracle.temporal_bandpass_filter_fft(x, fps, freq_min=fmin, freq_max=fmax, amplification_factor=50)
        for knob in (None, "temporal_wide", "temporal_valu"):
            if knob:
                device.debug_set(knob, 1)
            try:
                got = transforms.temporal_bandpass_filter_fft(x, fps, freq_min=fmin, freq_max=fmax, amplification_factor=50)
            finally:
                if knob:
                    device.debug_set(knob, -1 if knob == "temporal_wide" else 0)
            assert np.abs(np.asarray(got) - want).max() <= 1e-11 * max(np.abs(want).max(), 1e-300), (T, fps, fmin, fmax, knob)


def test_eulerian_golden_and_fused_equals_materialised(hip, golden):
    import torch
    from respmon_amd import transforms, dist
    g = golden("g3_eulerian.npz")
    vid8 = g["vid_u8"]
    vid = vid8 * (1. / 255)
    for i in range(int(g["ncases"])):
        L, S, fps = g["meta%d" % i]
        masked, raw = transforms.eulerian_magnification_bandpass(vid, fps, 0.1, 1.0, 500, pyramid_levels=int(L),
                                                                 skip_levels_at_top=int(S))
        ref = g["raw%d" % i]
        assert _rel(raw, ref) < 1e-11 < REL
        mn, mx, cnt = g["masked_stats%d" % i]
        assert abs(raw.min() - mn) <= 1e-11 * abs(mn) and abs(raw.max() - mx) <= 1e-11 * abs(mx)
        # `masked == raw.min()` counts the voxels the mask (transforms.py:188-192) replaced plus those that ARE the minimum.  Against
        # the reference's count it may differ only by voxels of the reference's raw within the temporal filter's rounding distance
        # of `top` (they can fall on the other side) or of the minimum (the band-pass output attains its extrema more than
        # once by symmetry: scipy's FFT gives those twins bit-identically, the explicit operator to 1 ulp)
        tol = 1e-11 * np.abs(ref).max()
        top_ref = mx - (mx - mn) * 0.7
        near_top = int((np.abs(ref - top_ref) <= tol).sum())
        min_twins = int((np.abs(ref - mn) <= tol).sum()) - 1
        assert abs(int((masked == raw.min()).sum()) - int(cnt)) <= near_top + min_twins
        assert near_top + min_twins <= 2
        for dt in (torch.float64, torch.uint8):
            buf = torch.from_numpy(vid if dt == torch.float64 else vid8).cuda()
            heat = dist.hip_calibrate(buf, fps, pyramid_levels=int(L), skip_levels_at_top=int(S)).cpu().numpy()
            assert np.array_equal(heat, np.average(masked, axis=0))   # fused == materialised, bit for bit
            assert _rel(heat, g["avg%d" % i]) < 1e-11
            heat_np = dist.hip_calibrate(buf, fps, pyramid_levels=int(L), skip_levels_at_top=int(S), flags=1).cpu().numpy()
            assert np.array_equal(heat, heat_np)                      # pruning never changes a bit
            heat_ts = dist.hip_calibrate(buf, fps, pyramid_levels=int(L), skip_levels_at_top=int(S), flags=4).cpu().numpy()
            assert np.array_equal(heat, heat_ts)                      # value-store overflow path


def test_fused_down_chain_equals_per_level(hip):
    import torch
    from respmon_amd import dist
    rng = np.random.default_rng(3)
    for dt in (np.float64, np.uint8, np.float32, np.float16):
        for (T, H, W, L, S) in [(3, 64, 96, 4, 2), (2, 67, 131, 5, 3), (2, 135, 240, 6, 4), (9, 32, 48, 4, 2),
                                (1, 200, 320, 7, 5), (16, 270, 480, 9, 4), (8, 360, 640, 4, 2), (4, 540, 1936, 8, 4),
                                (3, 300, 2000, 6, 3)]:
            v = (rng.random((T, H, W)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((T, H, W)).astype(dt)
            buf = torch.from_numpy(v).cuda()
            kw = dict(pyramid_levels=L, skip_levels_at_top=S)
            # flags=2: one kernel per pyramid level, the reference's operation order (Laplacians first, then the temporal filter).
            # In that order (64) the fused pyrDown chain + LDS-resident small pyramid reproduce it bit for bit -- tiny strips (8)
            # and per-level small pyramid (16) included
            per_level = dist.hip_calibrate(buf, 10, flags=2, **kw)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=64, **kw), per_level), (dt, T, H, W, L, S)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=64 | 8, **kw), per_level), (dt, T, H, W, L, S)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=16, **kw), per_level), (dt, T, H, W, L, S)
            # the default filters G_S first and builds the Laplacians of the filtered images (the same linear map): equal to
            # rounding, and independent of the strip geometry bit for bit
            ff = dist.hip_calibrate(buf, 10, **kw)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=8, **kw), ff), (dt, T, H, W, L, S)
            assert _close(ff, per_level), (dt, T, H, W, L, S)


def test_locate_golden_roi_bit_exact(hip, golden):
    import torch
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    g = golden("g4_locate.npz")
    for i in range(int(g["ncases"])):
        T, H, W, seed, L, S, fps = (int(v) for v in g["meta%d" % i])
        v8 = synth.synth_breathing(T, H, W, seed=seed)
        frames = v8 * (1. / 255)
        roi = RespiratoryMonitor.locate(frames, fps, pyramid_levels=L, skip_levels_at_top=S)
        assert roi == tuple(int(v) for v in g["roi%d" % i])
        assert RespiratoryMonitor.locate(torch.from_numpy(v8).cuda(), fps, pyramid_levels=L, skip_levels_at_top=S) == roi
        # the uint8 heatmap itself
        from respmon_amd import _capi, device
        heat = dist.hip_calibrate(torch.from_numpy(frames).cuda(), fps, pyramid_levels=L, skip_levels_at_top=S)
        u8 = torch.empty((H, W), dtype=torch.uint8, device="cuda")
        xywh = (ctypes.c_int32 * 4)()
        rc = hip.rm_heatmap_to_roi(device.ctx(), device.ptr(heat), H, W, 20, xywh, device.ptr(u8), None, device.stream_ptr())
        assert rc == 0 and tuple(xywh) == roi
        assert np.array_equal(u8.cpu().numpy(), g["avg_u8_%d" % i])
    assert RespiratoryMonitor.locate(np.full((16, 40, 48), 0.5), 10, pyramid_levels=4, skip_levels_at_top=2) is None
    assert RespiratoryMonitor.calibrate is RespiratoryMonitor.locate     # north_star's calibrate() is the same static method


def test_locate_vs_oracle_midsize_and_ragged(hip, oracle):
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    cases = [(128, 90, 160, 9, 4, 21), (64, 135, 241, 8, 3, 22), (48, 77, 131, 6, 2, 23), (40, 64, 64, 5, 1, 24)]
    for (T, H, W, L, S, seed) in cases:
        v8 = synth.synth_breathing(T, H, W, seed=seed)
        frames = oracle.uint8_to_float(v8)
        ref, mid = oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
        assert RespiratoryMonitor.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S) == ref, (T, H, W, L, S)


_VIDEO_CACHE = {}


def _video_1080p():
    """synth_breathing(256, 1080, 1920, seed=1234): the north-star video, generated once per test session (0.5 GB)."""
    from respmon_amd import synth
    if "p" not in _VIDEO_CACHE:
        _VIDEO_CACHE["p"] = synth.synth_breathing(256, 1080, 1920, seed=1234)
    return _VIDEO_CACHE["p"]


def _host_can_hold(nbytes):
    import psutil
    return psutil.virtual_memory().available > 1.3 * nbytes


def test_full_size_1080p_vs_oracle(hip, oracle):
    """North-star size (BASELINE config 4 per GPU: 1080p x 256, L=9, S=4): the oracle's materialising run on all
    256 frames (~5 s, ~25 GB of host memory) against the HIP path -- ROI bit-exact, heatmap and raw extrema far
    inside north_star's 1e-4 -- plus the size-independent identities of the fused path."""
    import torch
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    T, H, W = 256, 1080, 1920
    v8 = _video_1080p()
    dev8 = torch.from_numpy(v8).cuda()
    heat_u8in = dist.hip_calibrate(dev8, 10)
    buf = torch.empty((T, H, W), dtype=torch.float64, device="cuda")
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = dev8[t0:t0 + 16].to(torch.float64) * (1.0 / 255)
    heat, (raw_min, raw_max) = dist.hip_calibrate(buf, 10, return_minmax=True)
    assert torch.equal(heat, heat_u8in)                       # uint8 ingest == float64 buffer, bit for bit
    heat_np = dist.hip_calibrate(buf, 10, flags=1)
    assert torch.equal(heat, heat_np)                         # pruned == exhaustive, bit for bit
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=64), dist.hip_calibrate(buf, 10, flags=2))   # fused chain == per-level kernels
    ref_order = dist.hip_calibrate(buf, 10, flags=16)          # Laplacians first, per-level launches (the reference's order)
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=64), ref_order)   # LDS-resident small pyramid, same order: bit for bit
    assert _close(heat, ref_order)                            # default: temporal filter on G_S first (linearity), equal to rounding
    assert dist.hip_heatmap_to_roi(ref_order, 20) == dist.hip_heatmap_to_roi(heat, 20)
    assert torch.equal(heat, dist.hip_calibrate(buf, 10))     # deterministic
    roi = dist.hip_heatmap_to_roi(heat, 20)
    assert RespiratoryMonitor.locate(buf, 10) == roi          # the one-call form (rm_locate)
    x, y, w, h = roi
    cx, cy = x + w / 2, y + h / 2
    assert abs(cx - 0.4 * W) < 0.08 * W and abs(cy - 0.6 * H) < 0.1 * H   # the ROI sits on the breathing blob
    avg_frame = heat.cpu().numpy()
    del buf, dev8, heat_np, heat_u8in
    torch.cuda.empty_cache()
    # the oracle on the full buffer: ~6 float64 [T,H,W] arrays on the host
    assert _host_can_hold(6.0 * T * H * W * 8), "host memory too small for the 1080p x 256 oracle run"
    frames = oracle.uint8_to_float(v8)
    ref, mid = oracle.locate(frames, 10, return_intermediates=True)
    del frames
    assert roi == ref                                                       # bit-exact ROI at the north-star size
    assert _rel(avg_frame, mid["avg_frame"]) <= 1e-12                       # heatmap magnitudes (gate: 1e-4)
    assert abs(raw_min - mid["min"]) <= 1e-9 * abs(mid["min"]) and abs(raw_max - mid["max"]) <= 1e-9 * abs(mid["max"])
    u8 = oracle.float_to_uint8((avg_frame - avg_frame.min()) / (avg_frame.max() - avg_frame.min()))
    assert np.array_equal(u8, mid["avg_u8"])                                # the uint8 heatmap, every pixel
    assert roi == oracle.roi_from_heatmap_u8(u8, 20)


def test_config4_roi_flow_on_the_located_roi(hip, oracle):
    """BASELINE config 4's "+ ROI flow" leg: Shi-Tomasi corners (the reference's feature_params, base.py:91-94) and five
    pyramidal-LK steps (lk_params, base.py:96-98) on the ROI the 1080p x 256 calibration finds, against the oracle --
    corner coordinates, tracked points, status and mean flow bit-exact."""
    import torch
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor, _Backend
    be = _Backend()
    T, H, W = 256, 1080, 1920
    v8 = _video_1080p()
    roi = RespiratoryMonitor.locate(torch.from_numpy(v8).cuda(), 10)
    assert roi is not None
    x, y, w, h = roi
    assert w > 200 and h > 150                                   # the ~351x235 box of the bench
    crops_dev = [be.roi_to_uint8(torch.from_numpy(v8[i]).cuda(), x, y, w, h) for i in range(6)]
    crops_ref = [oracle.float_to_uint8(oracle.uint8_to_float(v8[i])[y:y + h, x:x + w]) for i in range(6)]   # base.py:364,381
    for a, b in zip(crops_dev, crops_ref):
        assert np.array_equal(a.cpu().numpy(), b)
    pts = be.good_features_to_track(crops_dev[0], 100, 0.3, 7, 7)
    ref = oracle.goodFeaturesToTrack(crops_ref[0], 100, 0.3, 7, blockSize=7)
    assert pts is not None and ref is not None and np.array_equal(pts, ref)
    cur = ref
    for i in range(5):
        p1, st = be.calc_optical_flow_pyr_lk(crops_dev[i], crops_dev[i + 1], cur, (15, 15), 2, (3, 10, 0.03))
        r1, rs, _ = oracle.calcOpticalFlowPyrLK(crops_ref[i], crops_ref[i + 1], cur, None, winSize=(15, 15), maxLevel=2,
                                                criteria=(3, 10, 0.03))
        assert np.array_equal(st, rs) and np.array_equal(p1, r1), i
        mean, ng = be.mean_flow(cur, p1, st)
        assert ng == int((rs == 1).sum())
        if ng:
            assert np.array_equal(mean, np.mean(cur[rs == 1] - r1[rs == 1], axis=0))     # base.py:388 (old - new)
        cur = r1[rs == 1].reshape(-1, 1, 2)
        if len(cur) == 0:
            break


def test_roi_mean_and_state_machine_config1(hip, golden):
    """BASELINE config 1: 64x240x320 brightness video, skip_calibration, 'average' mode (G6)."""
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    g = golden("g6_run_trace.npz")
    frames = synth.synth_brightness_video(64, 240, 320)
    mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=10), visualize=None, save_all_data=False,
                             motion_extraction_method="average", run_on_init=False)
    mon.sync_to_fps = lambda: None
    mon.skip_calibration(100, 80, 70, 51)
    mon.run()
    assert len(mon.data) == 64 and mon.fps == 10 and mon.peak_minimum_sample_distance == int(g["c1_peak_min_dist"])
    assert np.allclose(np.array(mon.data), g["c1_data"], rtol=1e-13, atol=0)
    assert np.array_equal(np.array(mon.t), g["c1_t"])


def test_state_machine_full_calibration_trace(hip, golden):
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    g = golden("g6_run_trace.npz")
    vid = synth.synth_breathing(150, 48, 64, seed=11)
    for bdt in ("float64", "uint8", "bgr8"):    # 'bgr8': the frames stored as captured ([T,H,W,3]); rm_locate converts as it reads
        mon = RespiratoryMonitor(capture_target=synth.FakeCapture(vid, fps=30), visualize=None, save_all_data=False,
                                 motion_extraction_method="average", run_on_init=False, buffer_dtype=bdt)
        mon.sync_to_fps = lambda: None
        trace = []
        real_next = mon.next_frame

        def traced():
            trace.append((["initialize", "calibration", "measure", "error"].index(mon.state), mon.calibration_buffer_idx))
            return real_next()
        mon.next_frame = traced
        mon.run()
        assert np.array_equal(np.array(trace, dtype=np.int32), g["c2_trace"])
        assert [mon.x, mon.y, mon.w, mon.h] == [int(v) for v in g["c2_roi"]]
        assert mon.fps == int(g["c2_fps"])
        assert np.allclose(np.array(mon.data), g["c2_data"], rtol=1e-13, atol=0)
        assert np.array_equal(np.array(mon.t), g["c2_t"])


def test_reduce_bounding_box_on_a_located_roi(hip, oracle, golden):
    """a12 (reference tools.py:48-57, base.py:457): the G7 cases (outputs of the real reference function) and a run() with a finite
    maximum_bounding_box_area -- the ROI the state machine keeps is the reference's shrink of what locate() found on the device."""
    from respmon_amd import synth, tools
    from respmon_amd.base import RespiratoryMonitor
    g = golden("g7_misc.npz")
    for c, r in zip(g["rbb_in"], g["rbb_out"]):
        x, y, w, h, a = c
        assert tuple(tools.reduce_bounding_box(int(x), int(y), int(w), int(h), a)) == tuple(int(v) for v in r)
    vid = synth.synth_breathing(150, 48, 64, seed=11)
    frames = oracle.uint8_to_float(vid)
    located = RespiratoryMonitor.locate(frames[1:129].copy(), 10)      # frames 1..T fill the buffer
    assert located is not None and located == oracle.locate(frames[1:129].copy(), 10)
    full_area = located[2] * located[3]
    for area in (np.inf, float(full_area), full_area / 2.0, 12.0):
        mon = RespiratoryMonitor(capture_target=synth.FakeCapture(vid, fps=30), visualize=None, save_all_data=False,
                                 motion_extraction_method="average", run_on_init=False)
        mon.sync_to_fps = lambda: None
        mon.maximum_bounding_box_area = area
        mon.run()
        want = oracle.reduce_bounding_box(*located, area)
        assert (mon.x, mon.y, mon.w, mon.h) == tuple(int(v) for v in want)
        if area < full_area:
            assert mon.w * mon.h < full_area
        else:
            assert (mon.x, mon.y, mon.w, mon.h) == located
        # the measured signal is the mean of the REDUCED crop of every later frame (base.py:471, 355-358)
        if mon.w >= 1 and mon.h >= 1:
            crop = frames[130:, mon.y:mon.y + mon.h, mon.x:mon.x + mon.w]
            assert np.allclose(np.array(mon.data), crop.reshape(crop.shape[0], -1).mean(1), rtol=1e-13, atol=0)


def test_config_q_720p_full_size_vs_oracle(hip, oracle):
    """BASELINE config 2 at full size: 128 x 720p, 4-level pyramid, skip 2 (the module defaults of
    eulerian_magnification_bandpass, transforms.py:145) -- bit-exact ROI against the oracle's materialising run."""
    import torch
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    T, H, W, L, S = 128, 720, 1280, 4, 2
    v8 = synth.synth_breathing(T, H, W, seed=1234)
    frames = oracle.uint8_to_float(v8)
    ref, mid = oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
    buf = torch.from_numpy(frames).cuda()
    assert RespiratoryMonitor.locate(buf, 10, pyramid_levels=L, skip_levels_at_top=S) == ref
    n0, labelled0 = dist.contour_stats()
    # this heatmap thresholds to thousands of specks: the second locate() of the geometry labels the components on the device and
    # follows only the borders that can win (rm_ccl.h) -- same ROI
    assert n0 > 512 and not labelled0
    assert RespiratoryMonitor.locate(buf, 10, pyramid_levels=L, skip_levels_at_top=S) == ref
    assert dist.contour_stats()[1]
    heat = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
    assert _rel(heat.cpu().numpy(), mid["avg_frame"]) <= 1e-12          # north_star gate is 1e-4
    assert torch.equal(heat, dist.hip_calibrate(torch.from_numpy(v8).cuda(), 10, pyramid_levels=L, skip_levels_at_top=S))
    assert torch.equal(heat, dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S, flags=1))   # pruned == exhaustive


def test_config_r_fp16_buffer(hip, oracle):
    """BASELINE config 5 (6-level pyramid, skip 2, float16 frame buffer): the library widens the stored halves
    exactly, so against the oracle fed the SAME rounded values the ROI is bit-exact; against the float64 video
    the float16 storage error is reported (north_star: 'f16 config reports its own error and ROI match')."""
    import torch
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    L, S = 6, 2
    v8 = synth.synth_breathing(64, 270, 480, seed=55)
    f64 = oracle.uint8_to_float(v8)
    f16 = f64.astype(np.float16)
    ref16, mid16 = oracle.locate(f16.astype(np.float64), 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
    ref64, mid64 = oracle.locate(f64, 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
    buf16 = torch.from_numpy(f16).cuda()
    assert RespiratoryMonitor.locate(buf16, 10, pyramid_levels=L, skip_levels_at_top=S) == ref16
    heat = dist.hip_calibrate(buf16, 10, pyramid_levels=L, skip_levels_at_top=S).cpu().numpy()
    assert _rel(heat, mid16["avg_frame"]) <= 1e-12
    storage_err = _rel(mid16["avg_frame"], mid64["avg_frame"])
    print("config R float16 storage: heatmap rel. error vs float64 video %.3e, ROI f16 %s / f64 %s" % (storage_err, ref16, ref64))
    assert storage_err < 0.2     # the half-precision frame buffer perturbs, but does not destroy, the heatmap


def _note_branch(record_property, key, text):
    """Make a data- or host-dependent branch of a test visible: a junit property, a warning in the pytest summary (which -q keeps)
    and a line in gpurun_out/test_branches.log (copied to profiles/ with the round's test log)."""
    import warnings
    record_property(key, text)
    warnings.warn("%s: %s" % (key, text))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "test_branches.log"), "a") as f:
            f.write("%s: %s\n" % (key, text))


def test_config_r_4k_512_fp16_full_size(hip, oracle, record_property):
    """BASELINE config 5 at its stated size: 4K x 512 frames, 6-level pyramid, skip 2, float16 frame buffer (8.5 GB; the
    filtered levels are 2.8 GB per array, n = 512 keeps 92 rfft rows = 6 MFMA tiles, tile bounds in bands).  The oracle
    needs ~6 float64 [T,H,W] arrays on the host (T = 512: 204 GB): where the host has that much, the full buffer is compared with
    a complete oracle run; everywhere, it is checked through size-independent identities and the oracle's ROI stage on the GPU
    heatmap, and its first 64 frames (the largest T a 32 GB host budget holds at 4K) against a complete oracle run fed the SAME
    float16-rounded values."""
    import torch
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    L, S = 6, 2
    T, H, W = 512, 2160, 3840
    v8 = synth.synth_breathing_blocks(T, H, W, seed=1234)
    b16 = torch.empty((T, H, W), dtype=torch.float16, device="cuda")
    for t0 in range(0, T, 8):
        b16[t0:t0 + 8] = (torch.from_numpy(v8[t0:t0 + 8]).cuda().to(torch.float64) * (1.0 / 255)).to(torch.float16)
    kw = dict(pyramid_levels=L, skip_levels_at_top=S)
    h0 = dist.hip_calibrate(b16, 10, **kw)
    assert torch.equal(h0, dist.hip_calibrate(b16, 10, flags=1, **kw))   # pruned == exhaustive
    per_level = dist.hip_calibrate(b16, 10, flags=2, **kw)               # one kernel per level, the reference's operation order
    assert torch.equal(dist.hip_calibrate(b16, 10, flags=64, **kw), per_level)   # fused chain, same order: bit for bit
    assert _close(h0, per_level)                                         # default (temporal filter on G_S first): to rounding
    assert dist.hip_heatmap_to_roi(per_level, 20) == dist.hip_heatmap_to_roi(h0, 20)
    del per_level
    assert torch.equal(h0, dist.hip_calibrate(b16, 10, **kw))            # deterministic
    b64 = b16.to(torch.float64)
    assert torch.equal(h0, dist.hip_calibrate(b64, 10, **kw))            # exact widening of the stored halves
    del b64
    torch.cuda.empty_cache()
    roi = dist.hip_heatmap_to_roi(h0, 20)
    assert roi is not None and RespiratoryMonitor.locate(b16, 10, **kw) == roi
    a = h0.cpu().numpy()
    assert roi == oracle.roi_from_heatmap_u8(oracle.float_to_uint8((a - a.min()) / (a.max() - a.min())), 20)
    x, y, w, h = roi
    assert abs(x + w / 2 - 0.4 * W) < 0.1 * W and abs(y + h / 2 - 0.6 * H) < 0.12 * H   # on the breathing blob
    # a host with > 270 GB of free memory (the MI355X boxes of this pool have 3 TB) takes the oracle on the FULL buffer, all host
    # threads (oracle.locate_parallel: bit-identical to oracle.locate, tests/test_oracle_golden.py): ROI bit-exact, heatmap 1e-12
    if _host_can_hold(6.5 * T * H * W * 8):
        full_in = b16.cpu().numpy().astype(np.float64)
        ref_full, mid_full = oracle.locate_parallel(full_in, 10, pyramid_levels=L, skip_levels_at_top=S, workers=min(64, os.cpu_count() or 1),
                                                    return_intermediates=True)
        del full_in
        assert roi == ref_full
        assert _rel(a, mid_full["avg_frame"]) <= 1e-12
        _note_branch(record_property, "config_R_full_oracle", "ran: oracle on all 512 4K frames, ROI %s == GPU ROI" % (ref_full,))
    else:
        _note_branch(record_property, "config_R_full_oracle", "SKIPPED: host has < 270 GB free; only the first-64-frames oracle run below")
    # complete oracle run on the first 64 frames of the same float16 buffer
    Ts = 64
    assert _host_can_hold(6.0 * Ts * H * W * 8), "host memory too small for the 4K x 64 oracle run"
    sub16 = b16[:Ts].contiguous()
    ref_in = sub16.cpu().numpy().astype(np.float64)
    ref, mid = oracle.locate(ref_in, 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
    del ref_in
    hs, (mn, mx) = dist.hip_calibrate(sub16, 10, return_minmax=True, **kw)
    assert dist.hip_heatmap_to_roi(hs, 20) == ref
    assert _rel(hs.cpu().numpy(), mid["avg_frame"]) <= 1e-12
    assert abs(mn - mid["min"]) <= 1e-9 * abs(mid["min"]) and abs(mx - mid["max"]) <= 1e-9 * abs(mid["max"])


def test_image_pyramid_functions(hip, oracle):
    """pyramid.py:9-28, 51-57: the per-image forms, bit-exact against the oracle."""
    from respmon_amd import pyramid
    rng = np.random.default_rng(12)
    for shape, L in [((64, 96), 5), ((33, 47), 4), ((135, 240), 6), ((5, 8), 3)]:
        img = rng.random(shape)
        for a, b in zip(pyramid.create_gaussian_image_pyramid(img, L), oracle.create_gaussian_image_pyramid(img, L)):
            assert a.shape == b.shape and np.array_equal(a, b)
        lap = pyramid.create_laplacian_image_pyramid(img, L)
        lap_ref = oracle.create_laplacian_image_pyramid(img, L)
        for a, b in zip(lap, lap_ref):
            assert a.shape == b.shape and np.array_equal(a, b)
        assert np.array_equal(pyramid.collapse_laplacian_pyramid(lap), oracle.collapse_laplacian_pyramid(lap_ref))
    u8 = (rng.random((21, 30)) * 255).astype(np.uint8)       # the reference copies any dtype into float64 (pyramid.py:10-11)
    assert np.array_equal(pyramid.create_gaussian_image_pyramid(u8, 3)[2], oracle.create_gaussian_image_pyramid(u8, 3)[2])


def test_iir_temporal_filter_golden(hip, oracle, golden):
    """temporal_bandpass_filter (order-6 Butterworth lfilter along T, transforms.py:72-79) and
    eulerian_magnification_bandpass(temporal_filter_function=temporal_bandpass_filter) against the reference's outputs."""
    import torch
    from respmon_amd import transforms
    g = golden("g8_iir.npz")
    for i in range(int(g["ncases"])):
        n, fps, fmin, fmax, amp = g["meta%d" % i]
        y = transforms.temporal_bandpass_filter(g["x%d" % i].copy(), fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
        assert _rel(y, g["y%d" % i]) <= 1e-9        # north_star gate: 1e-4 relative
    L, S, fps = g["e_meta"]
    vid = transforms.uint8_to_float(g["e_vid_u8"])
    masked, raw = transforms.eulerian_magnification_bandpass(vid, fps, 0.1, 1.0, 500, pyramid_levels=int(L), skip_levels_at_top=int(S),
                                                             temporal_filter_function=transforms.temporal_bandpass_filter)
    assert _rel(raw, g["e_raw"]) <= 1e-9
    assert _rel(np.average(masked, axis=0), g["e_avg"]) <= 1e-9
    # any callable with the reference's filter signature works (transforms.py:146): here the FFT filter passed explicitly
    # through the general path (Laplacian levels filtered one by one, the reference's order) equals the fused default (temporal
    # filter applied to G_S once, then the Laplacians of the filtered images -- the same linear map) to rounding
    def fft_again(vid, fps, freq_min, freq_max, amplification_factor, **_):
        return transforms.temporal_bandpass_filter_fft(vid, fps, freq_min=freq_min, freq_max=freq_max, amplification_factor=amplification_factor)
    m1, r1 = transforms.eulerian_magnification_bandpass(vid, fps, 0.1, 1.0, 500, pyramid_levels=int(L), skip_levels_at_top=int(S),
                                                        temporal_filter_function=fft_again)
    m0, r0 = transforms.eulerian_magnification_bandpass(vid, fps, 0.1, 1.0, 500, pyramid_levels=int(L), skip_levels_at_top=int(S))
    assert _rel(r1, r0) <= 1e-12 and _rel(m1, m0) <= 1e-12
    # ... including a plain numpy / scipy filter (numpy video in -> the callable gets numpy levels): here scipy's lfilter with
    # the reference's Butterworth coefficients, which must reproduce the reference's own IIR outputs (G8)
    import scipy.signal

    def scipy_iir(data, fps, freq_min, freq_max, amplification_factor, **_):
        assert isinstance(data, np.ndarray)
        b, a = scipy.signal.butter(6, [freq_min / (0.5 * fps), freq_max / (0.5 * fps)], btype='band')
        return scipy.signal.lfilter(b, a, data, axis=0) * amplification_factor
    m2, r2 = transforms.eulerian_magnification_bandpass(vid, fps, 0.1, 1.0, 500, pyramid_levels=int(L), skip_levels_at_top=int(S),
                                                        temporal_filter_function=scipy_iir)
    assert isinstance(r2, np.ndarray) and _rel(r2, g["e_raw"]) <= 1e-9
    # device tensors stay on the device
    xt = torch.from_numpy(g["x0"]).cuda()
    yt = transforms.temporal_bandpass_filter(xt, 10.0, freq_min=0.1, freq_max=1.0, amplification_factor=500.0)
    assert yt.is_cuda and _rel(yt.cpu().numpy(), g["y0"]) <= 1e-9
    # 1-D host signals go to scipy like the reference (transforms.py:47-50)
    sig = np.sin(np.arange(200) * 0.3)
    import scipy.signal
    b, a = transforms.butter_bandpass(0.5, 2.0, 10.0, order=3)
    assert np.array_equal(transforms.butter_bandpass_filter(sig, 0.5, 2.0, 10.0, order=3), scipy.signal.lfilter(b, a, sig))


def test_calibration_image_panels(hip, oracle, tmp_path):
    """SURVEY 8f row f1, base.py:577-596: the data of the six panels (time averages, normalisation, float_to_uint8,
    threshold) bit-exact against the oracle; the montage is written as calibration<i>.png like the reference does."""
    import os
    from respmon_amd import montage, synth
    from respmon_amd.base import RespiratoryMonitor
    v8 = synth.synth_breathing(48, 96, 128, seed=5)
    frames = oracle.uint8_to_float(v8)
    L, S = 6, 2
    panels, roi = montage.calibration_panels(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
    masked, raw = oracle.eulerian_magnification_bandpass(frames, 10, 0.1, 1.0, 500, pyramid_levels=L, skip_levels_at_top=S)

    def norm_u8(a):
        return oracle.float_to_uint8((a - a.min()) / (a.max() - a.min()))
    avg = norm_u8(np.average(masked, axis=0))
    assert np.array_equal(panels["avg"], avg)
    assert np.array_equal(panels["avg_raw"], norm_u8(np.average(raw, axis=0)))
    assert np.array_equal(panels["avg_original"], oracle.float_to_uint8(np.average(frames, axis=0)))
    assert np.array_equal(panels["thresh"], np.where(avg > 20, 255, 0).astype(np.uint8))
    assert roi == oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S) and roi is not None
    x, y, w, h = roi
    total = oracle.float_to_uint8(np.average(frames, axis=0))
    assert np.all(panels["drawn"][y, x:x + w + 1] == 255)                       # the ROI rectangle is drawn on (mean + avg)
    inside = panels["drawn"][y + 3:y + h - 2, x + 3:x + w - 2]
    assert np.array_equal(inside, (total + avg).astype(np.uint8)[y + 3:y + h - 2, x + 3:x + w - 2])
    assert (panels["contour_img"] != total).any() and np.all(panels["contour_img"][panels["contour_img"] != total] == 0)
    # time average of other frame dtypes
    assert np.array_equal(montage.time_average(v8).cpu().numpy(), np.average(frames, axis=0))
    # locate(save_calibration_image=True) writes calibration0.png, then calibration1.png, ... into the working directory
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for i in range(2):
            assert RespiratoryMonitor.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S, save_calibration_image=True) == roi
            assert os.path.exists("calibration%d.png" % i)
        img = montage.read_png_gray("calibration1.png")
    finally:
        os.chdir(cwd)
    assert img.shape == (2 * 96, 3 * 128) and np.array_equal(img, montage.montage(panels))


def test_library_first_then_torch_in_a_fresh_process(hip):
    """`build()` opens the HIP library before anything touches the GPU; `smoke()` may follow in the same process.
    PyTorch bundles its own HIP runtime under the same SONAME, so the loader must see torch first (_capi.load)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from respmon_amd import _capi\n"
            "lib = _capi.load(); assert lib.rm_abi_version() == 1\n"
            "import __graft_entry__ as g\n"
            "g.build(); g.smoke(); print('ok')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_narrow_frames_found_by_the_fuzzer(hip, oracle):
    """16-pixel-wide frames with skip 4 reach a 2-column pyramid level, where BORDER_REFLECT_101 is not the -2 -> 2
    select of the register-resident chain (tools/fuzz_parity.py found it): those shapes must take the general kernel."""
    from respmon_amd import synth, dist
    for (T, H, W, L, S, dt) in [(40, 140, 16, 6, 4, np.float32), (96, 90, 16, 6, 4, np.uint8), (16, 175, 16, 7, 4, np.float32),
                                (40, 20, 16, 8, 4, np.float16), (24, 64, 32, 6, 4, np.uint8), (24, 50, 16, 5, 3, np.uint8)]:
        v8 = synth.synth_breathing(T, H, W, seed=H)
        frames = v8 if dt == np.uint8 else oracle.uint8_to_float(v8).astype(dt)
        ref_in = oracle.uint8_to_float(v8) if dt == np.uint8 else frames.astype(np.float64)
        with np.errstate(all="ignore"):
            ref, mid = oracle.locate(ref_in, 10, pyramid_levels=L, skip_levels_at_top=S, return_intermediates=True)
        import torch
        dev = torch.from_numpy(frames).cuda()
        heat = dist.hip_calibrate(dev, 10, pyramid_levels=L, skip_levels_at_top=S).cpu().numpy()
        assert _rel(heat, mid["avg_frame"]) <= 1e-12, (T, H, W, L, S, dt)
        assert dist.hip_heatmap_to_roi(torch.from_numpy(heat).cuda(), 20) == ref


def test_long_buffers(hip, oracle):
    """Long calibration buffers (T = 512 and 1024: 92 / 184 surviving rfft rows, several 256-frame rounds in the
    kept-frame compaction) against the oracle."""
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    for (T, H, W, L, S) in [(512, 48, 80, 5, 2), (1024, 40, 64, 5, 2)]:
        frames = oracle.uint8_to_float(synth.synth_breathing(T, H, W, seed=T))
        ref = oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
        assert RespiratoryMonitor.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S) == ref, (T, H, W)


def test_contour_version_switch(hip, oracle):
    """SURVEY App. B3 / VERDICT r1: the reference pins no OpenCV version.  Default = OpenCV >= 3.2 (frame pixels count);
    RM_FLAG_CONTOUR_CLIP_FRAME / rm_set_contour_clip_frame = OpenCV <= 3.1 (1-pixel frame zeroed before tracing).  A breathing
    blob pushed against the image border gives the two documented ROIs, each equal to the oracle's twin."""
    import torch
    import scipy.ndimage as ndi
    from respmon_amd import synth, dist
    from respmon_amd.base import RespiratoryMonitor
    rng = np.random.default_rng(4)
    for k, shape in enumerate([(37, 61), (64, 130), (90, 200), (33, 64)]):
        heat = ndi.gaussian_filter(rng.standard_normal(shape), 2.5)
        if k % 2 == 0:
            heat[:, 0] = heat.max(); heat[0, :] = heat.max()
        else:
            heat[-1, :] = heat.max(); heat[:, -1] = heat.max()
        u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min()))
        dev = torch.from_numpy(heat).cuda()
        new, old = dist.hip_heatmap_to_roi(dev, 150), dist.hip_heatmap_to_roi(dev, 150, clip_frame=True)
        assert new == oracle.roi_from_heatmap_u8(u8, 150) and old == oracle.roi_from_heatmap_u8(u8, 150, clip_frame=True)
        assert new != old and dist.hip_heatmap_to_roi(dev, 150) == new
    # through locate(): blob centred on the left image edge
    v8 = synth.synth_breathing(64, 96, 128, seed=3, center=(0.5, 0.0), sigma=(0.2, 0.15))
    frames = oracle.uint8_to_float(v8)
    kw = dict(pyramid_levels=6, skip_levels_at_top=2)
    ref_new = oracle.locate(frames, 10, **kw)
    ref_old = oracle.locate(frames, 10, contour_clip_frame=True, **kw)
    assert RespiratoryMonitor.locate(frames, 10, **kw) == ref_new
    RespiratoryMonitor.opencv_contours_clip_frame = True
    try:
        assert RespiratoryMonitor.locate(frames, 10, **kw) == ref_old
    finally:
        RespiratoryMonitor.opencv_contours_clip_frame = False
    assert ref_new is not None and ref_new[0] == 0 and (ref_old is None or ref_old[0] >= 1)


def test_bpm_estimate_config1_on_the_gpu(hip, golden):
    """Row f2 end to end: the 0.4 Hz brightness video through run() on the GPU ('average' mode, ROI means from rm_roi_mean)
    => mon.freq[-1] within 1 BPM of 24 and equal to the reference's estimate for the same samples (G9 case 0)."""
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    g = golden("g9_measure.npz")
    frames = synth.synth_brightness_video(64, 240, 320)
    mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=10), visualize=None, save_all_data=False,
                             motion_extraction_method="average", run_on_init=False)
    mon.sync_to_fps = lambda: None
    mon.skip_calibration(100, 80, 70, 51)
    mon.run()
    assert len(mon.freq) > 0 and abs(mon.freq[-1] - 24.0) < 1.0
    assert abs(mon.freq[-1] - float(g["freq0"][-1])) <= 1e-9 * float(g["freq0"][-1])
    assert list(mon.peak_indices) == list(g["peaks0"])


def test_dense_sum_equals_sparse_path(hip, oracle):
    """rm_dense_sum.h against the selection / value-store path, bit for bit: random geometries, every frame dtype, every
    super-tile shape, config Q at full size (every pair kept: the automatic choice takes the dense kernel from the second call on)
    and the breathing video of config P (1 % of the pairs kept: the automatic choice stays sparse)."""
    import ctypes
    import torch
    from respmon_amd import _capi, device, dist, synth
    rng = np.random.default_rng(11)

    def sum_path():
        dbg = (ctypes.c_longlong * 4)()
        _capi.check(hip, hip.rm_debug_counters(device.ctx(), dbg, device.stream_ptr()), "rm_debug_counters")
        return "fused" if dbg[3] == -1 else ("dense" if dbg[3] == 0 else "sparse")

    for n, (T, H, W, L, S) in enumerate([(5, 64, 96, 4, 2), (3, 67, 131, 5, 3), (4, 135, 240, 6, 4), (6, 48, 64, 3, 1), (2, 200, 320, 7, 5),
                                         (16, 270, 480, 9, 4), (8, 360, 640, 4, 2), (4, 540, 1936, 8, 4), (3, 300, 2000, 6, 3), (7, 700, 1300, 4, 1)]):
        dt = (np.float64, np.uint8, np.float32, np.float16)[n % 4]
        v = (rng.random((T, H, W)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((T, H, W)).astype(dt)
        buf = torch.from_numpy(v).cuda()
        kw = dict(pyramid_levels=L, skip_levels_at_top=S)
        device.debug_set("dense_rows", 0)
        sparse = dist.hip_calibrate(buf, 10, flags=256, **kw)
        assert sum_path() == "sparse"
        # the exception store (rm_xstore.h: the default for a dense selection), 1 / 2 / 4 waves per tile; then an overflowing one
        for nw in (0, 1, 2, 4):
            device.debug_set("xs_waves", nw)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, "exception store", nw)
            assert sum_path() == "dense"
        device.debug_set("xs_waves", 0)
        device.debug_set("xs_budget_words", 5000)
        assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, "exception store overflow")
        device.debug_set("xs_budget_words", 0)
        device.debug_set("xs", 0)                # ... and the store-less kernels it stands in front of
        for rows in (0, 16, 32, 64):
            device.debug_set("dense_rows", rows)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, rows)
            assert sum_path() == "dense"
        if S <= 2:      # the wave-private kernels: one wave per tile (one frame per trip / two interleaved), two and four waves per tile
            device.debug_set("dense_rows", 0)
            for split, frames in ((1, 1), (1, 2), (2, 0), (4, 0)):
                device.debug_set("dense_split", split)
                device.debug_set("dense_frames", frames)
                assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, "waves per tile", split, frames)
            device.debug_set("dense_split", 0)
            device.debug_set("dense_frames", 0)
            for tl in (1, 0):                    # the TileEval kernel of the deeper chains (large frames take it) / never
                device.debug_set("dense_t_low", tl)
                assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, "dense_t_low", tl)
            device.debug_set("dense_t_low", -1)
        device.debug_set("dense_rows", 0)        # (rows 0 above: skip <= 2 takes the wave-private k_dense_sum_w; here the workgroup kernel)
        device.debug_set("dense_wave", 0)
        assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, "workgroup kernel")
        device.debug_set("dense_wave", 1)
        assert torch.equal(dist.hip_calibrate(buf, 10, flags=128 | 1, **kw), sparse), (dt, T, H, W, L, S)
        if S <= 2:      # skip <= 2 takes the table-driven kernel: the general one must agree there too
            for rows in (16, 32, 64):
                device.debug_set("dense_rows", rows)
                device.debug_set("dense_general", 1)
                assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), sparse), (dt, T, H, W, L, S, rows, "general")
            device.debug_set("dense_general", 0)
        device.debug_set("dense_rows", 0)
        # an 8-slot value store (RM_FLAG_TINY_STORE) overflows: the dense kernel takes over inside the same call, asked for or not
        for xs in (0, 1):
            device.debug_set("xs", xs)
            for fl in (4, 4 | 256):
                assert torch.equal(dist.hip_calibrate(buf, 10, flags=fl, **kw), sparse), (dt, T, H, W, L, S, "store overflow", fl, xs)
                assert sum_path() == "dense"
    device.debug_set("dense_rows", 0)
    device.debug_set("xs", 1)
    # config Q: dense by itself, on the FIRST call of the geometry (a fresh library context: nothing is remembered between calls)
    T, H, W, L, S = 128, 720, 1280, 4, 2
    buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=1234)).cuda()
    kw = dict(pyramid_levels=L, skip_levels_at_top=S)
    fresh = ctypes.c_void_p()
    _capi.check(hip, hip.rm_ctx_create(torch.cuda.current_device(), ctypes.byref(fresh)), "rm_ctx_create")
    heat = torch.empty((H, W), dtype=torch.float64, device=buf.device)
    _capi.check(hip, hip.rm_calibrate(fresh, device.ptr(buf), device.dtype_code(buf), T, H, W, 10.0, 0.1, 1.0, 500.0, L, S, 0.7, 0,
                                      device.ptr(heat), None, device.stream_ptr()), "rm_calibrate")
    dbg = (ctypes.c_longlong * 4)()
    _capi.check(hip, hip.rm_debug_counters(fresh, dbg, device.stream_ptr()), "rm_debug_counters")
    assert dbg[3] == 0, "first call of the geometry must already take the dense kernel"
    sparse = dist.hip_calibrate(buf, 10, flags=256, **kw)
    assert sum_path() == "sparse" and torch.equal(heat, sparse)
    auto = dist.hip_calibrate(buf, 10, **kw)
    assert sum_path() == "dense" and torch.equal(auto, sparse)
    assert dist.hip_heatmap_to_roi(auto, 20) == dist.hip_heatmap_to_roi(sparse, 20)
    _capi.check(hip, hip.rm_ctx_destroy(fresh), "rm_ctx_destroy")
    # config P's video: sparse stays sparse, and the forced dense kernel agrees
    T, H, W = 256, 1080, 1920
    buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=1234)).cuda()
    a = dist.hip_calibrate(buf, 10)
    torch.cuda.synchronize()
    b = dist.hip_calibrate(buf, 10)
    assert sum_path() == "sparse" and torch.equal(a, b)
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=128), a)
    # ... and a value store too small for what the selection keeps (5 078 pairs) hands the sum to the dense kernel: same bits
    device.debug_set("store_slots", 1000)
    assert torch.equal(dist.hip_calibrate(buf, 10), a) and sum_path() == "dense"
    device.debug_set("store_slots", 0)
    assert torch.equal(dist.hip_calibrate(buf, 10), a) and sum_path() == "sparse"
    # the store-less passes (rm_tile_eval.h): kept pairs evaluated where they are summed -- whole tiles or half tiles, same bits
    device.debug_set("collapse_fused", 1)
    assert torch.equal(dist.hip_calibrate(buf, 10), a) and sum_path() == "fused"      # k_dense_sum_t
    device.debug_set("dense_tiles", 0)                                                # k_tile_sum
    for half in (-1, 0, 1):
        device.debug_set("tile_sum_half", half)
        assert torch.equal(dist.hip_calibrate(buf, 10), a) and sum_path() == "fused", half
        assert torch.equal(dist.hip_calibrate(buf, 10, flags=1), a), ("no_prune", half)
    device.debug_set("tile_sum_half", -1)
    device.debug_set("dense_tiles", 1)
    device.debug_set("collapse_fused", 0)
    # ... which are also what stands in for an overflowing value store at skip >= 3 (k_tile_sum behind the sparse kernel)
    device.debug_set("store_slots", 1000)
    assert torch.equal(dist.hip_calibrate(buf, 10), a) and sum_path() == "dense"
    device.debug_set("store_slots", 0)


def test_exception_store_few_and_many_exceptions(hip, oracle):
    """rm_xstore.h on the streams it was built for: the 720p x 128 breathing video at skip 2 (0.6 % of the values below top: the
    exceptions travel with the record headers) and full-frame noise at 1080p, skip 4, against the store-less kernels (xs = 0) and
    the value-store path, bit for bit; temporal thresholds from 0.02 to 1.0 (no exceptions at all .. everything an exception); the
    ROI of the default path is the oracle's (test_config_q_720p_full_size_vs_oracle)."""
    import torch
    from respmon_amd import device, dist, synth
    T, H, W, L, S = 128, 720, 1280, 4, 2
    buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=1234)).cuda()
    for thr in (0.7, 0.02, 0.3, 1.0, 0.0):
        kw = dict(pyramid_levels=L, skip_levels_at_top=S, temporal_threshold=thr)
        device.debug_set("xs", 0)
        want = dist.hip_calibrate(buf, 10, flags=128, **kw)
        device.debug_set("xs", 1)
        for nw in (0, 1, 2):
            device.debug_set("xs_waves", nw)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), want), (thr, nw)
        device.debug_set("xs_waves", 0)
        assert torch.equal(dist.hip_calibrate(buf, 10, **kw), want), (thr, "automatic")
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=256, pyramid_levels=L, skip_levels_at_top=S), dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S))
    del buf
    buf = torch.from_numpy(synth.synth_noise_only(64, 1080, 1920)).cuda()
    device.debug_set("xs", 0)
    want = dist.hip_calibrate(buf, 10, flags=128)
    device.debug_set("xs", 1)
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=128), want)
    device.debug_set("xs_budget_words", 200000)          # overflows: k_dense_sum_t behind k_xs_sum
    assert torch.equal(dist.hip_calibrate(buf, 10, flags=128), want)
    device.debug_set("xs_budget_words", 0)


def test_bounds_refined_one_level_down_behind_a_dense_selection(hip, oracle):
    """rm_bounds_l1.h k_bounds_up1 (round 6): skip 3 / 4 bounds from the level-(S - 1) footprint.  Forced on: the heatmap does not
    change by a bit and the selection keeps no more pairs (ragged shapes, both skips).  Automatic: on a stream of sensor noise the
    first locate() keeps most pairs; the next calls of the context refine their bounds first, keep far fewer, and return the same
    ROI -- the oracle's; the headline stream (few pairs kept) never pays for the refinement."""
    import torch
    from respmon_amd import device, dist, synth
    from respmon_amd.base import RespiratoryMonitor
    rng = np.random.default_rng(131)
    for (T, H, W, L, S) in [(4, 67, 131, 5, 3), (5, 270, 480, 7, 4), (6, 540, 1936, 8, 4), (3, 300, 2000, 6, 3)]:
        v = rng.random((T, H, W)); v[:, : H // 2] *= 0.1
        buf = torch.from_numpy(v).cuda()
        kw = dict(pyramid_levels=L, skip_levels_at_top=S)
        device.debug_set("bounds_up1", 0)
        ref = dist.hip_calibrate(buf, 10, **kw); kept0 = device.debug_counters()[2]
        device.debug_set("bounds_up1", 1)
        got = dist.hip_calibrate(buf, 10, **kw); kept1 = device.debug_counters()[2]
        assert torch.equal(got, ref) and kept1 <= kept0, (T, H, W, L, S, kept1, kept0)
        for f in (128, 256):
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=f, **kw), ref), (T, H, W, f)
    device.debug_set("bounds_up1", -1)
    v8 = synth.synth_noise_only(64, 1080, 1920)
    buf = torch.from_numpy(v8).cuda()
    rois, kept = [], []
    for k in range(4):
        rois.append(RespiratoryMonitor.locate(buf, 10))
        kept.append(device.debug_counters()[2])
    assert len(set(rois)) == 1, rois
    assert kept[1] < 0.7 * kept[0] and kept[2] == kept[1] == kept[3], kept      # the hint: refined from the second call on, and stable
    frames = oracle.uint8_to_float(v8)
    assert rois[0] == oracle.locate(frames, 10)
    buf = torch.from_numpy(synth.synth_breathing(64, 1080, 1920, seed=1234)).cuda()
    k0 = []
    for k in range(3):
        RespiratoryMonitor.locate(buf, 10); k0.append(device.debug_counters()[2])
    assert k0[1] == k0[2] and max(k0) < 8192, k0       # (a sparse stream after a dense one: one refined call at most, then the plain path)


def test_fused_collapse_equals_store_path(hip, oracle):
    """rm_tile_eval.h against the selection / value-store path (flags=256), bit for bit: skip 1..4, every frame dtype, ragged sizes,
    whole-tile and half-tile work items, exhaustive evaluation; plus the oracle's ROI on a breathing video."""
    import torch
    from respmon_amd import device, dist, synth
    rng = np.random.default_rng(23)
    try:
        for n, (T, H, W, L, S) in enumerate([(5, 64, 96, 6, 4), (3, 67, 131, 5, 3), (4, 135, 240, 6, 4), (6, 48, 64, 3, 1), (8, 360, 640, 4, 2),
                                             (16, 270, 480, 9, 4), (4, 540, 1936, 8, 4), (3, 300, 2000, 6, 3), (7, 700, 1300, 5, 3),
                                             (2, 33, 70, 6, 4), (3, 17, 129, 5, 3), (5, 1080, 1920, 9, 4), (3, 31, 193, 7, 4),
                                             (130, 40, 70, 4, 2), (101, 33, 140, 5, 3), (256, 64, 128, 6, 4)]):
            dt = (np.float64, np.uint8, np.float32, np.float16)[n % 4]
            v = (rng.random((T, H, W)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((T, H, W)).astype(dt)
            buf = torch.from_numpy(v).cuda()
            kw = dict(pyramid_levels=L, skip_levels_at_top=S)
            device.debug_set("collapse_fused", 0)
            device.debug_set("eval_fast", 0)         # the generic chain in LDS (k_eval_pairs): the reference of both newer forms
            device.debug_set("sum_sym", 0)           # ... and the sum that fetches every visit of a frame (k_masked_sum_tiles)
            device.debug_set("sum_rows", 0)
            store = dist.hip_calibrate(buf, 10, flags=256, **kw)
            device.debug_set("sum_rows", 1)          # one wave per (tile, row), unique frames staged by LDS-DMA (k_masked_sum_rows)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=256, **kw), store), (dt, T, H, W, L, S, "k_masked_sum_rows")
            device.debug_set("sum_rows", 0)
            device.debug_set("sum_sym", 1)           # unique frames loaded once, added on the way up and down (k_masked_sum_sym)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=256, **kw), store), (dt, T, H, W, L, S, "k_masked_sum_sym")
            device.debug_set("sum_sym", 0)
            device.debug_set("eval_fast", 1)         # the same flat pass with the wave-private evaluator (k_eval_pairs_fast)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=256, **kw), store), (dt, T, H, W, L, S, "k_eval_pairs_fast")
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=256 | 1, **kw), store), (dt, T, H, W, L, S, "k_eval_pairs_fast, no_prune")
            device.debug_set("collapse_fused", 1)
            device.debug_set("dense_tiles", 1)       # one wave per tile, frame after frame (k_dense_sum_t)
            assert torch.equal(dist.hip_calibrate(buf, 10, **kw), store), (dt, T, H, W, L, S, "k_dense_sum_t")
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=1, **kw), store), (dt, T, H, W, L, S, "k_dense_sum_t, no_prune")
            device.debug_set("dense_tiles", 0)       # rounds of sixteen waves per tile (k_tile_sum)
            for half in (0, 1, -1):
                device.debug_set("tile_sum_half", half)
                assert torch.equal(dist.hip_calibrate(buf, 10, **kw), store), (dt, T, H, W, L, S, half)
            assert torch.equal(dist.hip_calibrate(buf, 10, flags=1, **kw), store), (dt, T, H, W, L, S, "no_prune")
        device.debug_set("collapse_fused", 1)
        device.debug_set("sum_rows", 0)
        v8 = synth.synth_breathing(64, 270, 480, seed=5)
        fr = oracle.uint8_to_float(v8)
        buf = torch.from_numpy(fr).cuda()
        for (L, S) in ((7, 4), (6, 3)):
            heat = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
            device.debug_set("collapse_fused", 0)
            store = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
            device.debug_set("collapse_fused", 1)
            assert torch.equal(heat, store)
            assert dist.hip_heatmap_to_roi(heat, 20) == oracle.locate(fr, 10, pyramid_levels=L, skip_levels_at_top=S)
    finally:
        device.debug_set("collapse_fused", 0)
        device.debug_set("tile_sum_half", -1)
        device.debug_set("eval_fast", 1)
        device.debug_set("sum_sym", 0)
        device.debug_set("sum_rows", 0)
        device.debug_set("dense_tiles", 1)


def test_filter_first_per_level_equals_fused(hip):
    """Levels too large for LDS (4K, skip 2) take the filter-first small pyramid with one launch per level: forced here on
    geometries that also fit the one-kernel form, the two must agree bit for bit; the 4K geometry itself against the
    reference's operation order (flags=2) to rounding."""
    import torch
    from respmon_amd import dist
    rng = np.random.default_rng(13)
    for (T, H, W, L, S) in [(5, 64, 96, 4, 2), (3, 67, 131, 5, 3), (16, 270, 480, 9, 4), (6, 48, 64, 3, 1), (2, 200, 320, 7, 5), (4, 540, 1936, 8, 4)]:
        buf = torch.from_numpy(rng.random((T, H, W))).cuda()
        kw = dict(pyramid_levels=L, skip_levels_at_top=S)
        fused = dist.hip_calibrate(buf, 10, **kw)
        assert torch.equal(dist.hip_calibrate(buf, 10, flags=512, **kw), fused), (T, H, W, L, S)
    buf = torch.from_numpy(rng.random((6, 2160, 3840)).astype(np.float16)).cuda()
    kw = dict(pyramid_levels=6, skip_levels_at_top=2)
    assert _close(dist.hip_calibrate(buf, 10, **kw), dist.hip_calibrate(buf, 10, flags=2, **kw))


def test_contour_stage_device_labelling(hip, oracle):
    """rm_ccl.h on the GPU (lock-free union-find with thousands of waves in flight): labelled ROI == ROI from following every
    border == the oracle's findContours restatement; one record per 8-connected component (scipy.ndimage.label).  Sizes up to
    4K, noise densities around and beyond the percolation threshold (one component snaking through the whole image), a blob in
    16 % noise (what BASELINE configs 2 / 5 threshold to), rows that straddle the 64-pixel words."""
    import torch
    import scipy.ndimage as ndi
    from respmon_amd import dist
    rng = np.random.default_rng(2024)
    cases = []
    for (h, w, dens) in [(720, 1280, 0.16), (720, 1280, 0.45), (1080, 1920, 0.30), (2160, 3840, 0.02), (2160, 3840, 0.62),
                         (333, 1001, 0.5), (97, 4099, 0.4), (1500, 63, 0.55), (64, 64, 0.9)]:
        cases.append(rng.random((h, w)) < dens)
    for (h, w, dens) in [(720, 1280, 0.16), (2160, 3840, 0.03), (411, 777, 0.16)]:
        m = rng.random((h, w)) < dens
        yy, xx = np.mgrid[0:h, 0:w]
        m |= ((yy - 0.45 * h) / (0.11 * h)) ** 2 + ((xx - 0.6 * w) / (0.06 * w)) ** 2 < 1.0
        cases.append(m)
    spiral = np.zeros((600, 600), bool)                 # one long thin component: deep union-find chains
    for k in range(0, 290, 4):
        spiral[k, k:600 - k] = True; spiral[k:600 - k, 599 - k] = True
        spiral[599 - k, k + 2:600 - k] = True; spiral[k + 4:600 - k, k + 2] = True
    cases.append(spiral)
    # smooth blobs that span the frame among specks (what a full-frame noise video thresholds to): the area bound of the labelled path
    # settles the winner without following a border (rm_ccl.h ccl_piece_2n_minus_p)
    fire = []
    for (h, w) in [(1080, 1920), (720, 1280), (540, 1000)]:
        m = rng.random((h, w)) < 0.05
        m |= ndi.gaussian_filter(rng.standard_normal((h, w)), 40.0) > 0.0
        fire.append(len(cases)); cases.append(m)
    from respmon_amd import device
    for im, m in enumerate(cases):
        heat = torch.from_numpy(m.astype(np.float64)).cuda()
        device.debug_set("host_area_bound", 0)         # the top component's border always followed
        try:
            roi_f = dist.hip_heatmap_to_roi(heat, 20, labelling=True)
            assert dist.roi_path() == 3
        finally:
            device.debug_set("host_area_bound", 1)
        roi_l = dist.hip_heatmap_to_roi(heat, 20, labelling=True)
        assert roi_l == roi_f and dist.roi_path() in (3, 4), (m.shape, roi_l, roi_f)
        path_l = dist.roi_path()
        device.debug_set("ccl_tiles", 0)   # rows of whole words through the global-memory kernels too (the default: tile by tile in LDS, k_ccl_tile)
        for table in (0, 1):     # k_ccl_bbox without / with its per-tile LDS table (the default picks by the last component count)
            device.debug_set("ccl_table", table)
            try:
                assert dist.hip_heatmap_to_roi(heat, 20, labelling=True) == roi_l and dist.roi_path() == path_l, (m.shape, table)
            finally:
                device.debug_set("ccl_table", -1)
        n_flat = dist.contour_stats()[0]
        device.debug_set("ccl_tiles", 1)
        assert dist.hip_heatmap_to_roi(heat, 20, labelling=True) == roi_l and dist.roi_path() == path_l and dist.contour_stats()[0] == n_flat, m.shape
        # ... exactly when scipy's labels say it can: 2 N - P - 2 of the component with the largest box beats every other box bound
        lab, n = ndi.label(m, structure=np.ones((3, 3)))
        pad = np.pad(m, 1); c = pad[1:-1, 1:-1]
        cracks = (c & ~pad[:-2, 1:-1]).astype(np.int64) + (c & ~pad[2:, 1:-1]) + (c & ~pad[1:-1, :-2]) + (c & ~pad[1:-1, 2:])
        low2 = 2 * np.bincount(lab.ravel(), minlength=n + 1) - np.bincount(lab.ravel(), weights=cracks.ravel(), minlength=n + 1).astype(np.int64) - 2
        bound2 = np.array([-1] + [2 * (s[0].stop - s[0].start - 1) * (s[1].stop - s[1].start - 1) for s in ndi.find_objects(lab)])
        top = int(np.argmax(bound2))
        rivals = np.delete(bound2, [0, top])
        expect = n >= 1 and low2[top] > 0 and (rivals.size == 0 or low2[top] > rivals.max())
        assert (dist.roi_path() == 4) == bool(expect), (m.shape, dist.roi_path(), int(low2[top]), int(rivals.max()) if rivals.size else None)
        fired = locals().get("fired", 0) + (dist.roi_path() == 4)
        n_l, used = dist.contour_stats()
        assert used
        roi_h = dist.hip_heatmap_to_roi(heat, 20, labelling=False)
        assert not dist.contour_stats()[1]
        assert roi_l == roi_h, (m.shape, roi_l, roi_h)
        assert n_l == ndi.label(m, structure=np.ones((3, 3)))[1], m.shape
        if m.size <= 1300 * 800:
            assert roi_l == oracle.roi_from_heatmap_u8(np.where(m, 255, 0).astype(np.uint8), 20), m.shape
    assert fired >= 1
    # more components than the record list holds (2^18): the host follows every border of the image it has anyway
    m = rng.random((2160, 3840)) < 0.16
    heat = torch.from_numpy(m.astype(np.float64)).cuda()
    roi_l = dist.hip_heatmap_to_roi(heat, 20, labelling=True)
    assert not dist.contour_stats()[1] and roi_l == dist.hip_heatmap_to_roi(heat, 20, labelling=False)
    # the lazy stage (rm_roi.hip): after an extraction the summary records settled alone, the next one of that geometry leaves the packed
    # image and the component list on the device -- and fetches them when its own summaries leave the winner open, or the list overflows
    blob4k = np.zeros((2160, 3840), bool); blob4k[300:1500, 500:3000] = True; blob4k |= rng.random((2160, 3840)) < 0.01
    two4k = np.zeros((2160, 3840), bool); two4k[100:900, 200:1800] = True          # a solid block, and a hollow frame with the larger box:
    two4k[1000:2100, 1900:3800] = True; two4k[1003:2097, 1903:3797] = False         # its count bound is tiny, the block's box beats it
    two4k |= rng.random((2160, 3840)) < 0.002
    want = {}
    for name, img in (("blob", blob4k), ("two", two4k), ("overflow", m)):
        want[name] = dist.hip_heatmap_to_roi(torch.from_numpy(img.astype(np.float64)).cuda(), 20, labelling=False)
    for lazy in (1, 0):
        device.debug_set("label_lazy", lazy)
        try:
            seen = []
            for name in ("blob", "blob", "two", "blob", "blob", "overflow", "blob", "blob"):
                img = {"blob": blob4k, "two": two4k, "overflow": m}[name]
                assert dist.hip_heatmap_to_roi(torch.from_numpy(img.astype(np.float64)).cuda(), 20, labelling=True) == want[name], (name, lazy)
                seen.append(dist.roi_path())
            assert seen[0] == 4 and seen[1] == 4 and seen[2] == 3 and seen[5] == 2 and seen[7] == 4, seen
        finally:
            device.debug_set("label_lazy", 1)
    # the automatic rule switches a noisy geometry to the labelled path from its second extraction on, and back
    noisy = torch.from_numpy((rng.random((720, 1280)) < 0.16).astype(np.float64)).cuda()
    clean = torch.zeros((720, 1280), dtype=torch.float64, device="cuda"); clean[100:300, 200:500] = 1.0
    r0 = dist.hip_heatmap_to_roi(noisy, 20); assert not dist.contour_stats()[1]
    r1 = dist.hip_heatmap_to_roi(noisy, 20); assert dist.contour_stats()[1] and r1 == r0
    assert dist.hip_heatmap_to_roi(clean, 20) == (200, 100, 300, 200) and dist.contour_stats()[1]
    assert dist.hip_heatmap_to_roi(clean, 20) == (200, 100, 300, 200) and not dist.contour_stats()[1]


def test_streaming_tile_bounds_equal_table_form(hip):
    """k_frame_bounds_rows (a wave per eight tile rows, rows parked in a skewed LDS buffer; the automatic choice for wide levels
    with many frames: 4K x 512) against k_frame_bounds (row-extrema table) on wide, ragged levels: the same heatmap bit for bit,
    pruned and exhaustive, and the same exact extrema."""
    import torch
    from respmon_amd import device, dist
    rng = np.random.default_rng(31)
    for (T, H, W, L, S) in [(6, 300, 1100, 4, 2), (4, 203, 2100, 5, 3), (5, 160, 1040, 3, 1), (3, 540, 3840, 6, 2)]:
        buf = torch.from_numpy(rng.random((T, H, W))).cuda()
        kw = dict(pyramid_levels=L, skip_levels_at_top=S, flags=512)     # (per-level filter-first path: the bounds are a kernel of their own)
        device.debug_set("bounds_l1", 0)                                 # (skip 2 would take the level-1 bounds of rm_bounds_l1.h: tested below)
        device.debug_set("bounds_scalar", 1)
        ref, mm = dist.hip_calibrate(buf, 10, return_minmax=True, **kw)
        device.debug_set("bounds_scalar", 2)
        got, mm2 = dist.hip_calibrate(buf, 10, return_minmax=True, **kw)
        device.debug_set("bounds_scalar", 0)
        assert torch.equal(got, ref) and mm == mm2, (T, H, W, L, S)
        exhaustive = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S, flags=512 | 1)
        assert torch.equal(got, exhaustive), (T, H, W, L, S, "no prune")
        device.debug_set("bounds_l1", 2)


def test_level1_tile_bounds(hip, oracle):
    """rm_bounds_l1.h (round 6): at skip 2 the tile bounds are the extrema of the LEVEL-1 footprint (one pyrUp below the collapsed
    level), formed by a streaming kernel -- three tile columns per wave, DPP wave rotations for the horizontal neighbours, bands of
    tile rows.  (a) the bounds equal the oracle's pyrUp of the collapsed level bit for bit (ragged shapes, several chunks and
    bands); (b) the heatmap and the extrema do not change by a bit against the level-2 bounds, exhaustive evaluation, and either sum
    path; (c) on a noisy stream the selection keeps far fewer pairs."""
    import torch
    from respmon_amd import device, dist
    rng = np.random.default_rng(61)
    for (T, H, W, L, S, trb) in [(6, 300, 1100, 4, 2, 0), (5, 203, 2100, 5, 2, 2), (4, 131, 450, 4, 2, 1), (3, 540, 3840, 6, 2, 0),
                                 (6, 720, 1280, 4, 2, 0), (4, 70, 64, 4, 2, 16)]:
        v = rng.random((T, H, W))
        v[:, : H // 2, : W // 3] *= 0.05
        buf = torch.from_numpy(v).cuda()
        kw = dict(pyramid_levels=L, skip_levels_at_top=S, flags=512)
        device.debug_set("bounds_l1_rows", trb)
        device.debug_set("bounds_l1", 0)
        ref, mm = dist.hip_calibrate(buf, 10, return_minmax=True, **kw)
        kept_l2 = device.debug_counters()[2]
        device.debug_set("bounds_l1", 1)
        got, mm2 = dist.hip_calibrate(buf, 10, return_minmax=True, **kw)
        kept_l1 = device.debug_counters()[2]
        assert torch.equal(got, ref) and mm == mm2, (T, H, W, L, S, trb)
        assert kept_l1 <= kept_l2, (T, H, W, kept_l1, kept_l2)
        Th = T // 2 + 1
        h1, w1 = (H + 1) // 2, (W + 1) // 2
        h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
        nty, ntx = (H + 15) // 16, (W + 63) // 64
        c2 = device.debug_workspace("cS", (Th, h2, w2))
        lo = device.debug_workspace("tile_lo", (Th, nty, ntx)); hi = device.debug_workspace("tile_hi", (Th, nty, ntx))
        device.debug_set("bounds_l1", 2)         # (the default) packed float32 + margin
        got2 = dist.hip_calibrate(buf, 10, **kw)
        assert torch.equal(got2, ref) and device.debug_counters()[2] <= kept_l2, (T, H, W, "float32 bounds")
        lo32 = device.debug_workspace("tile_lo", (Th, nty, ntx)); hi32 = device.debug_workspace("tile_hi", (Th, nty, ntx))
        slack = np.abs(c2).max() * 2.0 ** -19 + 1e-40
        for u in range(Th):
            l1 = oracle.pyrUp(c2[u], (w1, h1))
            for ty in range(nty):
                y0, y1 = max(8 * ty - 1, 0), min(8 * ty + 8, h1 - 1)
                rows = l1[y0:y1 + 1]
                for tx in range(ntx):
                    f = rows[:, max(32 * tx - 1, 0): min(32 * tx + 32, w1 - 1) + 1]
                    assert lo[u, ty, tx] == f.min() and hi[u, ty, tx] == f.max(), (T, H, W, u, ty, tx)
                    assert f.min() - slack <= lo32[u, ty, tx] <= f.min() and f.max() <= hi32[u, ty, tx] <= f.max() + slack, (T, H, W, u, ty, tx, "float32")
        exhaustive = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S, flags=512 | 1)
        assert torch.equal(got, exhaustive), (T, H, W, "no prune")
        for f in (128, 256):
            alt = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S, flags=512 | f)
            assert torch.equal(alt, ref), (T, H, W, f)
        device.debug_set("dense_t_low", 1)
        alt = dist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S, flags=512 | 128)
        device.debug_set("dense_t_low", -1)
        assert torch.equal(alt, ref), (T, H, W, "k_dense_sum_t")
    device.debug_set("bounds_l1_rows", 0)
    device.debug_set("bounds_l1", 2)
    # (c) speckle at the scale of one level-2 pixel over a smooth breathing signal: the level-2 footprint bound keeps (nearly) every
    # pair, the level-1 bound far fewer -- and the ROI is the oracle's
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    v8 = synth.synth_breathing(32, 360, 640, seed=1234)
    frames = oracle.uint8_to_float(v8)
    buf = torch.from_numpy(frames).cuda()
    device.debug_set("bounds_l1", 0)
    dist.hip_calibrate(buf, 10, pyramid_levels=4, skip_levels_at_top=2, flags=512)
    kept_l2 = device.debug_counters()[2]
    device.debug_set("bounds_l1", 2)
    dist.hip_calibrate(buf, 10, pyramid_levels=4, skip_levels_at_top=2, flags=512)
    pairs, _, kept_l1, _ = device.debug_counters()
    assert kept_l1 <= kept_l2, (kept_l1, kept_l2, pairs)
    roi = RespiratoryMonitor.locate(buf, 10, pyramid_levels=4, skip_levels_at_top=2)
    assert roi == oracle.locate(frames, 10, pyramid_levels=4, skip_levels_at_top=2)


def test_value_store_grows_with_the_selection(hip, oracle):
    """rm_locate on streams that keep more pairs than the value store starts with.  (a) A moderately dense selection (a breathing video
    against a store of 64 slots): the first call finds the store overflowed after its host synchronisation, allocates one that holds
    the selection and runs evaluation + sum again; later calls go straight through the grown store.  (b) A dense selection (noise:
    every pair kept): the store-less sum (k_dense_sum_t) takes over; the following calls refine their bounds first (round 6) and take whichever
    path their smaller selection needs -- the same one every time.  Same ROI as the path with a slot per pair (flags=256) every time."""
    import ctypes
    import torch
    from respmon_amd import _capi, device, dist, synth
    rng = np.random.default_rng(41)
    xywh = (ctypes.c_int32 * 4)()
    dbg = (ctypes.c_longlong * 4)()

    def run(ctx, buf, L, S):
        T, H, W = buf.shape
        rc = _capi.check(hip, hip.rm_locate(ctx, device.ptr(buf), device.dtype_code(buf), T, H, W, 10.0, 0.1, 1.0, 500.0, L, S, 0.7, 20, 0, xywh,
                                            device.stream_ptr()), "rm_locate")
        _capi.check(hip, hip.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
        return None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)

    for case in ("grow", "dense"):
        if case == "grow":
            L, S = 7, 4
            fr = oracle.uint8_to_float(synth.synth_breathing(64, 270, 480, seed=5))
            buf = torch.from_numpy(fr).cuda()
        else:
            L, S = 7, 3
            buf = torch.from_numpy(rng.random((64, 544, 1024))).cuda()     # 16 x 34 tiles x 33 unique frames = 17 952 pairs, all kept
        ref = dist.hip_heatmap_to_roi(dist.hip_calibrate(buf, 10, flags=256, pyramid_levels=L, skip_levels_at_top=S), 20)
        fresh = ctypes.c_void_p()
        _capi.check(hip, hip.rm_ctx_create(torch.cuda.current_device(), ctypes.byref(fresh)), "rm_ctx_create")
        if case == "grow":
            _capi.check(hip, hip.rm_debug_set(fresh, b"store_default_slots", 64), "rm_debug_set")
        for call in range(4):
            roi = run(fresh, buf, L, S)
            assert roi == ref, (case, call, roi, ref)
            if case == "grow":
                assert 64 < dbg[2] and dbg[2] * 4 <= dbg[0]                        # more than the starting store, not a dense selection
                assert dbg[3] >= dbg[2], "the store must hold the selection (grown inside the first call)"
            elif call == 0:
                assert dbg[2] > 16384 and dbg[3] == 0                              # dense: no store in use
                kept_first = dbg[2]
            else:
                # round 6: behind a call that kept that many pairs the bounds are refined one level down first (k_bounds_up1): fewer pairs --
                # through the store if they fit it now, store-less otherwise; the same on every later call (no flip-flop between the two)
                assert dbg[2] < kept_first and (dbg[3] == 0 or dbg[3] >= dbg[2]), (call, dbg[2], dbg[3], kept_first)
                if call == 1:
                    second = (dbg[2], dbg[3] == 0)
                else:
                    assert (dbg[2], dbg[3] == 0) == second, (call, dbg[2], dbg[3], second)
        _capi.check(hip, hip.rm_ctx_destroy(fresh), "rm_ctx_destroy")


def test_small_pyramid_split_over_workgroups(hip, oracle):
    """k_small_filter_first with one, two and four workgroups per frame: same C_S and tile bounds, so the heatmap is bit-identical
    whatever the split (1080p x 64 takes two by default)."""
    import torch
    from respmon_amd import device, dist, synth
    rng = np.random.default_rng(31)
    try:
        for (T, H, W, L, S) in [(64, 1080, 1920, 9, 4), (16, 540, 960, 8, 4), (9, 300, 500, 7, 3), (12, 720, 1280, 8, 3)]:
            buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=9)).cuda() if H == 1080 else torch.from_numpy(rng.random((T, H, W))).cuda()
            kw = dict(pyramid_levels=L, skip_levels_at_top=S)
            device.debug_set("ff_parts", 1)
            one = dist.hip_calibrate(buf, 10, **kw)
            for parts in (0, 2, 4):
                device.debug_set("ff_parts", parts)
                assert torch.equal(dist.hip_calibrate(buf, 10, **kw), one), (T, H, W, L, S, parts)
    finally:
        device.debug_set("ff_parts", 0)


@pytest.mark.gpu
def test_roi_stage_skips_constant_tiles(hip, oracle):
    """rm_locate's ROI stage reads a 64 x 16 tile of its own heatmap through ONE value where the sum kernel flagged the tile as a
    constant (tile_nkept == 0): same ROI as the stage that reads every pixel, and as the oracle."""
    import torch
    from respmon_amd import device, dist, synth
    from oracle import respmon_oracle as ro
    rng = np.random.default_rng(77)
    cases = [(32, 1080, 1920, 9, 4, "breath"), (16, 256, 448, 7, 3, "breath"), (12, 128, 192, 6, 2, "noise"),
             (10, 144, 256, 6, 1, "breath"), (16, 270, 640, 7, 3, "noise")]
    try:
        for (T, H, W, L, S, kind) in cases:
            vid = synth.synth_breathing(T, H, W, seed=T + S) if kind == "breath" else rng.random((T, H, W))
            buf = torch.from_numpy(np.ascontiguousarray(vid)).cuda()
            kw = dict(pyramid_levels=L, skip_levels_at_top=S)
            got = {}
            for flag in (0, 1):
                device.debug_set("heat_const_tiles", flag)
                from respmon_amd.base import RespiratoryMonitor
                got[flag] = RespiratoryMonitor.locate(buf, 10, **kw)
            assert got[0] == got[1], (T, H, W, L, S, kind, got)
            if H <= 300:
                want = ro.locate(vid, 10, pyramid_levels=L, skip_levels_at_top=S)
                assert (got[1] is None and want is None) or tuple(got[1]) == tuple(int(v) for v in want), (T, H, W, L, S, kind, got[1], want)
    finally:
        device.debug_set("heat_const_tiles", 1)


@pytest.mark.gpu
def test_locate_submit_result(hip, oracle):
    """rm_locate_submit / rm_locate_result (two calibration buffers in flight on one stream): every ROI equals the synchronous
    rm_locate's, in either fetch order, at full size back to back, through the value-store overflow path (RM_FLAG_TINY_STORE: taken
    again inside rm_locate_result) and on a flat video (no contour); a third submission is refused."""
    import torch
    from respmon_amd import _capi, synth
    from respmon_amd.base import RespiratoryMonitor, _Backend
    be = _Backend()
    small = []
    for (T, H, W, L, S, seed) in [(16, 256, 448, 7, 3, 3), (12, 128, 192, 6, 2, 4), (10, 144, 256, 6, 1, 5), (24, 360, 640, 8, 4, 6)]:
        buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=seed)).cuda()
        small.append((buf, L, S, RespiratoryMonitor.locate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)))
    for i in range(len(small)):
        (ba, La, Sa, ra), (bb, Lb, Sb, rb) = small[i], small[(i + 1) % len(small)]
        ta = be.locate_submit(ba, 10, pyramid_levels=La, skip_levels_at_top=Sa)
        tb = be.locate_submit(bb, 10, pyramid_levels=Lb, skip_levels_at_top=Sb, flags=_capi.RM_FLAG_TINY_STORE if i % 2 else 0)
        with pytest.raises(_capi.RespmonError):
            be.locate_submit(ba, 10, pyramid_levels=La, skip_levels_at_top=Sa)
        if i % 2:
            assert be.locate_result(tb) == rb and be.locate_result(ta) == ra
        else:
            assert be.locate_result(ta) == ra and be.locate_result(tb) == rb
    flat = torch.full((16, 40, 48), 0.5, dtype=torch.float64, device="cuda")
    assert be.locate_result(be.locate_submit(flat, 10, pyramid_levels=4, skip_levels_at_top=2)) is None
    # full size, two different videos alternating, the next one submitted before the previous ROI is fetched
    vids = [torch.from_numpy(synth.synth_breathing(64, 1080, 1920, seed=s, center=c)).cuda() for s, c in ((11, (0.6, 0.4)), (12, (0.3, 0.7)))]
    want = [RespiratoryMonitor.locate(v, 10) for v in vids]
    assert want[0] != want[1] and None not in want
    tk = be.locate_submit(vids[0], 10)
    for k in range(1, 12):
        nxt = be.locate_submit(vids[k % 2], 10)
        assert be.locate_result(tk) == want[(k - 1) % 2], k
        tk = nxt
    assert be.locate_result(tk) == want[11 % 2]
    # the submissions of a context share one stream; a context may be destroyed with a submission nobody fetched
    import ctypes
    from respmon_amd import device
    lib = _capi.load()
    h = ctypes.c_void_p()
    _capi.check(lib, lib.rm_ctx_create(0, ctypes.byref(h)), "rm_ctx_create")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    v = small[0][0]
    T, H, W = v.shape
    tk1, tk2 = ctypes.c_int(-1), ctypes.c_int(-1)
    args = (device.ptr(v), device.dtype_code(v), T, H, W, 10.0, 0.1, 1.0, 500.0, small[0][1], small[0][2], 0.7, 20, 0)
    assert lib.rm_locate_submit(h, *args, ctypes.c_void_p(s1.cuda_stream), ctypes.byref(tk1)) == _capi.RM_OK
    assert lib.rm_locate_submit(h, *args, ctypes.c_void_p(s2.cuda_stream), ctypes.byref(tk2)) == _capi.RM_E_BADARG
    xywh = (ctypes.c_int32 * 4)()
    assert lib.rm_locate_result(h, 1, xywh) == _capi.RM_E_BADARG          # no such submission
    assert lib.rm_ctx_destroy(h) == _capi.RM_OK                            # waits for ticket 0's work, frees its pinned slot
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_round5_switches_leave_results_bit_identical(hip, oracle):
    """Round-5 forms against the ones they replaced, on the device: the row-record ROI kernel (k_heat_rows_u8) against the flat one,
    the store-less sum with the exact-top mask / level-1 stop / second-visit skip against evaluating every kept pair to the end, and
    the rotating wave priority of the frame-buffer kernels (a scheduling hint: the same bits whatever its setting)."""
    import torch
    from respmon_amd import device, dist, synth
    from respmon_amd.base import RespiratoryMonitor
    try:
        # ROI stage: shapes around the one-blob rule, rows of whole words (the record path) -- and the labelled path with its top candidates
        rng = np.random.default_rng(5)
        for name, img in [("blob", None), ("two blobs", None), ("ring", None), ("noise", None), ("nothing", None)]:
            z = np.zeros((96, 256))
            if name == "blob": z[20:60, 40:200] = 1
            if name == "two blobs": z[10:30, 10:90] = 1; z[50:90, 100:250] = 1
            if name == "ring": z[10:80, 20:220] = 1; z[30:60, 60:180] = 0
            if name == "noise": z = (rng.random((96, 256)) > 0.55).astype(float)
            heat = torch.from_numpy(z * 1.0).cuda()
            want = oracle.roi_from_heatmap_u8(oracle.float_to_uint8(z / z.max()) if z.max() > 0 else np.zeros(z.shape, np.uint8), 100)
            for rows in (1, 0, 1):
                device.debug_set("heat_rows", rows)
                for lab in (False, True):
                    for _ in range(2):
                        assert dist.hip_heatmap_to_roi(heat, 100, labelling=lab) == want, (name, rows, lab)
        # store-less sum on a stream of noise (every pair kept by the selection) at skip 2 and 4, odd and even T
        for (T, H, W, L, S) in [(32, 270, 480, 6, 2), (33, 272, 448, 9, 4), (64, 135, 256, 7, 3)]:
            v8 = synth.synth_noise_only(T, H, W)
            buf = torch.from_numpy(oracle.uint8_to_float(v8)).cuda()
            kw = dict(pyramid_levels=L, skip_levels_at_top=S)
            device.debug_set("dense_t_low", 1)
            device.debug_set("dense_exact_top", 0)
            ref = dist.hip_calibrate(buf, 10, flags=128, **kw)                 # RM_FLAG_DENSE_SUM: k_dense_sum_t, every kept pair to the end
            store = dist.hip_calibrate(buf, 10, flags=256, **kw)               # the value-store path
            device.debug_set("dense_exact_top", 1)
            got = dist.hip_calibrate(buf, 10, flags=128, **kw)
            assert torch.equal(ref, store) and torch.equal(got, ref), (T, H, W, L, S)
            for prio in (0, 1, 3, 2):
                device.debug_set("dc_prio", prio)
                assert torch.equal(dist.hip_calibrate(buf, 10, flags=128, **kw), ref), prio
            assert RespiratoryMonitor.locate(buf, 10, **kw) == oracle.locate(oracle.uint8_to_float(v8), 10, **kw)
    finally:
        device.debug_set("heat_rows", 1)
        device.debug_set("dense_exact_top", 1)
        device.debug_set("dense_t_low", -1)
        device.debug_set("dc_prio", 2)
