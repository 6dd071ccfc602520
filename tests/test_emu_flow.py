"""Flow-path kernels (goodFeaturesToTrack / calcOpticalFlowPyrLK / mean flow / PCA) compiled for the host
emulation and compared with the oracle -- catches logic bugs without a GPU (see tests/emu/build.py)."""
import numpy as np
import pytest

from respmon_amd import synth


@pytest.fixture(scope="module")
def emu():
    from tests.emu_harness import Emu
    return Emu()


def test_emu_gftt_and_lk_bit_exact(emu, oracle):
    render = synth.synth_texture(96, 112, seed=4321)
    a = render(0.0, 0.0)
    for (n, q) in [(100, 0.05), (20, 0.3), (400, 0.01)]:
        ce = emu.good_features(a, n, q, 7, 7)
        co = oracle.goodFeaturesToTrack(a, n, q, 7, blockSize=7)
        assert np.array_equal(ce, co)
    pts = oracle.goodFeaturesToTrack(a, 100, 0.05, 7, blockSize=7)
    for (dx, dy) in [(0.7, -0.4), (-3.0, 2.0), (9.0, 0.0)]:
        b = render(dx, dy)
        pe, se = emu.pyr_lk(a, b, pts)
        po, so, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        assert np.array_equal(se, so)
        assert np.array_equal(pe, po)
        m, ng = emu.mean_flow(pts, po, so)
        if ng:
            assert np.array_equal(m, np.mean(pts[so == 1] - po[so == 1], axis=0))
    # points near / outside the border lose status exactly like the oracle
    edge = np.array([[[1.0, 1.0]], [[110.5, 94.0]], [[-30.0, 5.0]], [[50.0, 50.0]]], np.float32)
    pe, se = emu.pyr_lk(a, render(0.5, 0.5), edge)
    po, so, _ = oracle.calcOpticalFlowPyrLK(a, render(0.5, 0.5), edge, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
    assert np.array_equal(se, so) and np.array_equal(pe[so.ravel() == 1], po[so.ravel() == 1])
    # small ROI like the reference's 70x51: two pyramid levels only
    small = a[10:61, 20:90]
    sp = oracle.goodFeaturesToTrack(small, 100, 0.3, 7, blockSize=7)
    if sp is not None:
        pe, se = emu.pyr_lk(small, render(0.4, 0.2)[10:61, 20:90], sp)
        po, so, _ = oracle.calcOpticalFlowPyrLK(small, render(0.4, 0.2)[10:61, 20:90], sp, None, winSize=(15, 15), maxLevel=2,
                                                criteria=(3, 10, 0.03))
        assert np.array_equal(pe, po) and np.array_equal(se, so)


def test_emu_pca_matches_numpy_eig_signs(emu, oracle, golden):
    rng = np.random.default_rng(11)
    for trial in range(300):
        n = int(rng.integers(2, 129))
        ang = rng.uniform(0, np.pi)
        s1, s2 = rng.uniform(0.05, 3.0), rng.uniform(0.001, 1.0)
        base = rng.standard_normal((n, 2)) * [s1, s2]
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        md = (base @ rot.T + rng.uniform(-1, 1, 2)).astype(np.float32)
        ref = oracle.pca_first_component([list(r) for r in md])
        got = emu.pca_reduce(md)
        assert abs(got - ref) <= 1e-9 * max(1.0, abs(ref)), (trial, n, got, ref)
    # axis-aligned / degenerate covariances
    for md in ([[1, 0], [2, 0], [3, 0]], [[0, 1], [0, 2], [0, 5]], [[1, 1], [2, 2], [3, 3]], [[1, -1], [2, -2], [4, -4]],
               [[0, 0], [0, 0]], [[1, 2], [1, 2], [1, 2]]):
        md = np.array(md, np.float32)
        assert abs(emu.pca_reduce(md) - oracle.pca_first_component([list(r) for r in md])) < 1e-9
    # the reference's own sequence (scripted LK, G5)
    g = golden("g5_extract_motion.npz")
    md = g["motion_data_f32"]
    vals = g["values"]
    for k in range(2, len(md) + 1):
        assert abs(emu.pca_reduce(md[:k]) - vals[k]) <= 1e-9 * max(1.0, abs(vals[k]))


def test_emu_roi_mean(emu):
    rng = np.random.default_rng(0)
    for dt in (np.float64, np.uint8):
        f = (rng.random((60, 80)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((60, 80))
        ref = (f[12:43, 9:70] * (1. / 255)).mean() if dt == np.uint8 else f[12:43, 9:70].mean()
        assert abs(emu.roi_mean(f, 9, 12, 61, 31) - ref) <= 1e-13 * abs(ref)


def test_emu_flow_step_resident_equals_four_calls(emu, oracle):
    """rm_flow_begin / rm_flow_step (state resident on the device, one call per frame) against the reference's four steps run
    through the single entry points, and against the oracle's calcOpticalFlowPyrLK: same corners, same mean flow, same surviving
    points frame after frame, also while large shifts push points out of the ROI."""
    render = synth.synth_texture(80, 100, seed=99)
    x, y, w, h = 12, 9, 70, 51
    for amp in (1.0, 14.0):
        frames = [render(amp * np.sin(0.5 * t), 0.5 * amp * np.cos(0.4 * t)) for t in range(7)]
        # the crop is float_to_uint8(uint8_to_float(frame)[roi]) (base.py:364, 371: truncation loses 24 of the 256 levels)
        crop = lambda f: np.ascontiguousarray(oracle.float_to_uint8(oracle.uint8_to_float(f)[y:y + h, x:x + w]))
        p = emu.flow_begin(frames[0], x, y, w, h, 100, 0.3, 7, 7)
        q = emu.good_features(crop(frames[0]), 100, 0.3, 7, 7)
        assert (p is None) == (q is None) and (p is None or np.array_equal(p, q))
        if p is None:
            continue
        assert np.array_equal(emu.flow_points(), q)
        for t in range(1, 7):
            m, ng = emu.flow_step(frames[t], x, y, w, h)
            if len(q) == 0:
                assert ng == 0
                continue
            p1, st = emu.pyr_lk(crop(frames[t - 1]), crop(frames[t]), q)
            po, so, _ = oracle.calcOpticalFlowPyrLK(crop(frames[t - 1]), crop(frames[t]), q, None, winSize=(15, 15), maxLevel=2,
                                                    criteria=(3, 10, 0.03))
            assert np.array_equal(st, so)
            m4, ng4 = emu.mean_flow(q, p1, st)
            assert ng == ng4 and np.array_equal(m, m4), (amp, t)
            q = p1[st.ravel() == 1].reshape(-1, 1, 2)
            assert np.array_equal(emu.flow_points(), q), (amp, t)


def test_emu_two_flow_states_interleaved(emu):
    """rm_flow_state: two sessions interleaved on one context give what each gives alone (index / buffer logic of the handle)."""
    from respmon_amd import synth
    ra, rb = synth.synth_texture(60, 72, seed=3), synth.synth_texture(60, 72, seed=4)
    fa = [ra(0.5 * t, 0.2 * t) for t in range(4)]
    fb = [rb(-0.4 * t, 0.3 * t) for t in range(4)]

    def run(frames_list):
        states = [emu.flow_state() for _ in frames_list]
        out = [[emu.flow_begin(fr[0], 0, 0, 72, 60, 40, 0.3, 5, 7, state=st)] for fr, st in zip(frames_list, states)]
        for t in range(1, 4):
            for o, fr, st in zip(out, frames_list, states):
                o.append(emu.flow_step(fr[t], 0, 0, 72, 60, state=st))
        for o, st in zip(out, states):
            o.append(emu.flow_points(100, state=st))
        return out

    (alone_a,), (alone_b,) = run([fa]), run([fb])
    both_a, both_b = run([fa, fb])
    for got, want in ((both_a, alone_a), (both_b, alone_b)):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[-1], want[-1])
        for g, w in zip(got[1:-1], want[1:-1]):
            assert np.array_equal(g[0], w[0]) and g[1] == w[1]
