"""GPU parity of the ROI motion-extraction path (reference base.py:354-407): Shi-Tomasi corners, pyramidal LK,
mean flow and PCA reduction against the CPU oracle.  Bit-exact corner coordinates / status; flow vectors within
1e-4 relative (in practice identical: the kernels keep OpenCV's accumulation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    assert torch.cuda.is_available()
    from respmon_amd.base import _Backend
    return _Backend()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_corners_and_flow_config3(be, oracle):
    """BASELINE config 3: 256x256 ROI, up to 1000 Shi-Tomasi points, sub-pixel motion at 30 fps."""
    from respmon_amd import synth
    render = synth.synth_texture(256, 256, seed=4321)
    a = render(0.0, 0.0)
    pts = be.good_features_to_track(_dev(a), maxCorners=1000, qualityLevel=0.01, minDistance=7, blockSize=7)
    ref = oracle.goodFeaturesToTrack(a, 1000, 0.01, 7, blockSize=7)
    assert pts is not None and np.array_equal(pts, ref) and len(pts) >= 300
    prev = a
    cur_pts = pts
    for t in range(1, 6):
        dx, dy = 1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)
        b = render(dx, dy)
        p1, st = be.calc_optical_flow_pyr_lk(_dev(prev), _dev(b), cur_pts, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        r1, rs, _ = oracle.calcOpticalFlowPyrLK(prev, b, cur_pts, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        assert np.array_equal(st, rs)
        good = rs.ravel() == 1
        flow, rflow = (p1 - cur_pts).reshape(-1, 2)[good], (r1 - cur_pts).reshape(-1, 2)[good]
        assert np.abs(flow - rflow).max() <= 1e-4 * max(np.abs(rflow).max(), 1e-3)   # north_star gate
        assert np.array_equal(p1, r1)                                                # and in fact identical
        mean, ng = be.mean_flow(cur_pts, p1, st)
        assert ng == good.sum() and np.array_equal(mean, np.mean(cur_pts[rs == 1] - r1[rs == 1], axis=0))
        prev, cur_pts = b, r1[rs == 1].reshape(-1, 1, 2)


def test_config3_at_1000_points(be, oracle):
    """BASELINE config 3 at its stated point count: minDistance=3 lets the texture yield maxCorners = 1000 corners (minDistance=7
    stops at 655); five LK steps on all of them, points, status and mean flow bit-exact against the oracle."""
    from respmon_amd import synth
    render = synth.synth_texture(256, 256, seed=4321)
    a = render(0.0, 0.0)
    pts = be.good_features_to_track(_dev(a), maxCorners=1000, qualityLevel=0.01, minDistance=3, blockSize=7)
    ref = oracle.goodFeaturesToTrack(a, 1000, 0.01, 3, blockSize=7)
    assert pts is not None and len(pts) == 1000 and np.array_equal(pts, ref)
    prev, cur_pts = a, pts
    tracked = []
    for t in range(1, 6):
        dx, dy = 1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)
        b = render(dx, dy)
        p1, st = be.calc_optical_flow_pyr_lk(_dev(prev), _dev(b), cur_pts, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        r1, rs, _ = oracle.calcOpticalFlowPyrLK(prev, b, cur_pts, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        assert np.array_equal(st, rs) and np.array_equal(p1, r1)
        mean, ng = be.mean_flow(cur_pts, p1, st)
        assert ng == int((rs.ravel() == 1).sum()) and np.array_equal(mean, np.mean(cur_pts[rs == 1] - r1[rs == 1], axis=0))
        tracked.append(ng)
        prev, cur_pts = b, r1[rs == 1].reshape(-1, 1, 2)
    assert tracked[0] >= 950, tracked   # (nearly) all of the 1000 points are really tracked


def test_small_roi_like_the_reference(be, oracle):
    from respmon_amd import synth
    render = synth.synth_texture(96, 112, seed=99)
    a, b = render(0, 0)[10:61, 20:90], render(0.4, 0.2)[10:61, 20:90]   # 70x51, two LK levels
    pts = be.good_features_to_track(_dev(a), maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7)
    ref = oracle.goodFeaturesToTrack(a, 100, 0.3, 7, blockSize=7)
    assert (pts is None) == (ref is None)
    if ref is not None:
        assert np.array_equal(pts, ref)
        p1, st = be.calc_optical_flow_pyr_lk(_dev(a), _dev(b), pts, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        r1, rs, _ = oracle.calcOpticalFlowPyrLK(a, b, ref, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        assert np.array_equal(p1, r1) and np.array_equal(st, rs)
    flat = np.full((40, 40), 9, np.uint8)
    assert be.good_features_to_track(_dev(flat), maxCorners=10, qualityLevel=0.3, minDistance=7, blockSize=7) is None


def test_pca_reduce_golden_and_random(be, oracle, golden):
    g = golden("g5_extract_motion.npz")
    md, vals = g["motion_data_f32"], g["values"]
    for k in range(2, len(md) + 1):
        assert abs(be.pca_reduce(md[:k]) - vals[k]) <= 1e-9 * max(1.0, abs(vals[k]))
    rng = np.random.default_rng(5)
    for _ in range(50):
        n = int(rng.integers(2, 129))
        m = (rng.standard_normal((n, 2)) * rng.uniform(0.01, 2, 2) + rng.uniform(-1, 1, 2)).astype(np.float32)
        ref = oracle.pca_first_component([list(r) for r in m])
        assert abs(be.pca_reduce(m) - ref) <= 1e-9 * max(1.0, abs(ref))


def test_extract_motion_flow_state_machine(oracle):
    """RespiratoryMonitor in 'flow' mode on a moving texture against the oracle's extract_motion restatement."""
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    render = synth.synth_texture(120, 160, seed=7)
    T = 40
    frames = np.stack([render(1.5 * np.sin(2 * np.pi * 0.4 * t / 10), 0.5 * np.sin(2 * np.pi * 0.4 * t / 10 + 1.0)) for t in range(T)])
    mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=10), visualize=None, save_all_data=False,
                             motion_extraction_method="flow", run_on_init=False)
    mon.sync_to_fps = lambda: None
    mon.skip_calibration(30, 20, 90, 70)
    mon.run()
    state = oracle.FlowState()
    ref = []
    for t in range(T):
        crop = oracle.uint8_to_float(frames[t])[20:90, 30:120]
        ref.append(oracle.extract_motion_flow(state, crop))
    got = np.array(mon.data, dtype=np.float64)
    ref = np.array(ref, dtype=np.float64)
    assert len(got) == T and got[0] == 0.0 and got[1] == 0.0
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)
    assert np.array_equal(np.array(mon.motion_data, dtype=np.float32), np.array(state.motion_data, dtype=np.float32))


def test_flow_against_float64_brute_force_and_component_boxes(be, oracle):
    """Cross-checks of the PRODUCT that involve neither cv2 nor the oracle's restatement of it (VERDICT r1 Next 6b):
    (i) rm_calc_optical_flow_pyr_lk against a plain float64 Lucas-Kanade on the config-3 texture, within 1e-2 px;
    (ii) rm_heatmap_to_roi against scipy.ndimage's connected components: the ROI is the bounding box of one 8-connected
    component of the thresholded heatmap, and no other component has a larger box area than ... the polygon-area winner's
    component is among the components (sanity: the box of the largest-pixel-count component when blobs are convex)."""
    import torch
    import scipy.ndimage as ndi
    from tests import lk_bruteforce as bf
    from respmon_amd import dist
    a, b, pts, (dx, dy) = bf.config3_case(oracle)
    p1, st = be.calc_optical_flow_pyr_lk(_dev(a), _dev(b), pts.reshape(-1, 1, 2), winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
    ok = st.ravel() == 1
    assert ok.sum() >= 100
    ref = bf.lk_float(a, b, pts[ok])
    err = np.linalg.norm(p1.reshape(-1, 2)[ok] - ref, axis=1)
    assert err.max() <= 1e-2 and np.median(err) <= 3e-3
    assert np.median(np.linalg.norm(p1.reshape(-1, 2)[ok] - pts[ok] - np.array([dx, dy]), axis=1)) <= 0.05
    rng = np.random.default_rng(17)
    for k in range(12):
        H, W = int(rng.integers(40, 200)), int(rng.integers(40, 300))
        heat = ndi.gaussian_filter(rng.standard_normal((H, W)), float(rng.uniform(2.0, 5.0)))
        thr = int(rng.integers(120, 200))
        roi = dist.hip_heatmap_to_roi(torch.from_numpy(heat).cuda(), thr)
        u8 = (np.floor((heat - heat.min()) / (heat.max() - heat.min()) * 255)).astype(np.int64)   # truncating float_to_uint8, no oracle
        lab, n = ndi.label(u8 > thr, structure=np.ones((3, 3)))
        boxes = [(sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start) for sl in ndi.find_objects(lab)]
        if n == 0:
            assert roi is None
            continue
        assert roi in boxes                                  # a true component's bounding box, found without border following
        # smooth blobs: the component with the largest polygon area is, up to ties among near-equal blobs, the one whose filled
        # region is largest -- its box must not be smaller than HALF the largest component's (a gross mis-selection would be)
        sizes = ndi.sum(np.ones_like(lab), lab, index=np.arange(1, n + 1))
        assert sizes[boxes.index(roi)] >= 0.5 * sizes.max()


def test_flow_step_resident_equals_four_calls(oracle):
    """rm_flow_begin / rm_flow_step (crops, points and status resident on the device, one C-ABI call per frame) against the four
    calls of the reference's order (rm_roi_to_uint8, rm_calc_optical_flow_pyr_lk, rm_mean_flow with points through host memory):
    identical data trace, motion_data and surviving points, frame after frame -- also while points are being lost (large shifts
    push corners out of the ROI) and after all of them are gone (NaN from then on, base.py:385-386)."""
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    render = synth.synth_texture(120, 160, seed=11)
    T = 36
    for amp, roi in [(1.5, (30, 20, 90, 70)), (9.0, (40, 30, 60, 50)), (40.0, (50, 40, 24, 20))]:
        frames = np.stack([render(amp * np.sin(2 * np.pi * 0.4 * t / 10), 0.6 * amp * np.sin(2 * np.pi * 0.4 * t / 10 + 1.0))
                           for t in range(T)])
        traces = []
        for fused in (True, False):
            mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=10), visualize=None, save_all_data=False,
                                     motion_extraction_method="flow", run_on_init=False)
            mon.fused_flow_step = fused
            mon.sync_to_fps = lambda: None
            mon.skip_calibration(*roi)
            pts_per_frame = []
            step = mon.extract_motion
            def spy(step=step, mon=mon, out=pts_per_frame):
                v = step()
                p = mon.motion_key_points
                out.append(None if p is None else np.array(p, dtype=np.float32).reshape(-1, 2).copy())
                return v
            mon.extract_motion = spy
            mon.run()
            traces.append((np.array(mon.data, dtype=np.float64), np.array(mon.motion_data, dtype=np.float32), pts_per_frame))
        (d1, m1, p1), (d0, m0, p0) = traces
        assert len(d1) == T and np.array_equal(d1, d0, equal_nan=True), amp
        assert np.array_equal(m1, m0), amp
        assert len(p1) == len(p0)
        for a, b in zip(p1, p0):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), amp
        x, y, w, h = roi
        state = oracle.FlowState()
        ref = np.array([oracle.extract_motion_flow(state, oracle.uint8_to_float(frames[t])[y:y + h, x:x + w]) for t in range(T)])
        assert np.allclose(d1, ref, rtol=1e-9, atol=1e-12, equal_nan=True), amp


def test_two_flow_sessions_interleaved_on_one_gpu(be):
    """Two tracking sessions on one GPU and one library context (two RespiratoryMonitor objects in a process, or bench.py's
    flow measurement beside a monitor): each owns an rm_flow_state, so interleaving their frames changes nothing -- the previous
    crop, its pyramid and the tracked points of one session are never seen by the other."""
    from respmon_amd import synth
    ra, rb = synth.synth_texture(96, 128, seed=11), synth.synth_texture(96, 128, seed=12)     # same ROI size: the old shared state could not tell them apart
    fa = [ra(0.8 * np.sin(0.3 * t), 0.4 * np.cos(0.2 * t)) for t in range(8)]
    fb = [rb(-0.6 * np.sin(0.25 * t), 0.7 * np.sin(0.4 * t)) for t in range(8)]

    def alone(frames):
        st = be.flow_state()
        pts = be.flow_begin(st, _dev(frames[0]), 0, 0, 128, 96, 100, 0.3, 7, 7)
        out = [pts]
        for f in frames[1:]:
            out.append(be.flow_step(st, _dev(f), 0, 0, 128, 96, (15, 15), 2, (3, 10, 0.03)))
        out.append(be.flow_points(st, 100))
        return out

    want_a, want_b = alone(fa), alone(fb)
    sa, sb = be.flow_state(), be.flow_state()
    got_a = [be.flow_begin(sa, _dev(fa[0]), 0, 0, 128, 96, 100, 0.3, 7, 7)]
    got_b = [be.flow_begin(sb, _dev(fb[0]), 0, 0, 128, 96, 100, 0.3, 7, 7)]
    for t in range(1, 8):
        got_a.append(be.flow_step(sa, _dev(fa[t]), 0, 0, 128, 96, (15, 15), 2, (3, 10, 0.03)))
        got_b.append(be.flow_step(sb, _dev(fb[t]), 0, 0, 128, 96, (15, 15), 2, (3, 10, 0.03)))
    got_a.append(be.flow_points(sa, 100)); got_b.append(be.flow_points(sb, 100))
    assert want_a[0] is not None and want_b[0] is not None and not np.array_equal(want_a[0], want_b[0])
    for got, want in ((got_a, want_a), (got_b, want_b)):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[-1], want[-1])
        for g, w in zip(got[1:-1], want[1:-1]):
            assert np.array_equal(g[0], w[0]) and g[1] == w[1]
