"""The oracle's restatement of the cv2 calls against goldens captured from a REAL OpenCV (tools/capture_cv2_goldens.py).
No OpenCV exists in the build container or on the GPU boxes, so these tests skip until somebody runs the capture script where
`import cv2` works and commits tests/golden/cv2_*.npz -- that one command pins the cv2 leg of the oracle."""
import os

import numpy as np
import pytest

GOLDEN = os.environ.get("RESPMON_CV2_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")   # (override: dry runs of the tests)
SKIP = "tests/golden/%s is absent: run `python tools/capture_cv2_goldens.py` in an environment with OpenCV (cv2) and commit the files"


def load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(SKIP % name)
    return np.load(path)


def close(a, b, what):
    """Bit-exact against OpenCV 3.x (whose generic code paths the oracle restates); newer builds vectorise some float64 loops in
    another operation order, so other versions get 4 ulp."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, what
    assert np.all(np.abs(a - b) <= 4 * np.spacing(np.maximum(np.abs(a), np.abs(b)))), what


def test_pyr_down_up(oracle):
    g = load("cv2_pyr.npz")
    v3 = str(g["cv2_version"]).startswith("3.")
    for i in range(int(g["n"])):
        a = g["in_%d" % i]
        d = oracle.pyrDown(a)
        u = oracle.pyrUp(g["down_%d" % i], (a.shape[1], a.shape[0]))
        if v3:
            assert np.array_equal(d, g["down_%d" % i]) and np.array_equal(u, g["up_%d" % i]), (i, a.shape)
        else:
            close(d, g["down_%d" % i], ("pyrDown", i, a.shape))
            close(u, g["up_%d" % i], ("pyrUp", i, a.shape))


def test_cvtcolor_threshold(oracle):
    g = load("cv2_misc.npz")
    assert np.array_equal(oracle.cvtColor_bgr2gray(g["bgr"]), g["gray"])
    for t in (0, 20, 127, 254, 255):
        assert np.array_equal(oracle.threshold(g["thr_in"], t, 255)[1], g["thr_%d" % t]), t


def test_contours_roi(oracle):
    g = load("cv2_contours.npz")
    ver = tuple(int(x) for x in str(g["cv2_version"]).split(".")[:2])
    clip = ver < (3, 2)                       # <= 3.1 zeroes the 1-pixel frame before tracing
    for i in range(int(g["n"])):
        img = g["img_%d" % i]
        contours = oracle.findContours(img, clip_frame=clip)
        assert len(contours) == int(g["n_contours_%d" % i]), i
        assert np.array_equal(np.array([oracle.contourArea(c) for c in contours]), g["areas_%d" % i]), i      # same list order, same areas
        assert np.array_equal(np.array([oracle.boundingRect(c) for c in contours]).reshape(-1, 4), g["rects_%d" % i]), i
        roi = oracle.roi_from_heatmap_u8(img, 20, clip_frame=clip)
        want = tuple(int(v) for v in g["roi_%d" % i])
        assert (roi is None and want[2] < 0) or roi == want, (i, roi, want)


def test_corners_and_lk(oracle):
    g = load("cv2_flow.npz")
    for i in range(int(g["n"])):
        frames = g["frames_%d" % i]
        mc, q, md, bs = g["params_%d" % i]
        pts = oracle.goodFeaturesToTrack(frames[0], int(mc), float(q), float(md), blockSize=int(bs))
        want = g["corners_%d" % i]
        assert (pts is None and len(want) == 0) or np.array_equal(pts, want), i
        p = pts
        for t in range(1, 4):
            key = "lk_p1_%d_%d" % (i, t)
            if key not in g.files:
                break
            p1, st, _ = oracle.calcOpticalFlowPyrLK(frames[t - 1], frames[t], p, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
            assert np.array_equal(st, g["lk_st_%d_%d" % (i, t)]), (i, t)
            good = st.ravel() == 1
            flow, ref = (p1 - p).reshape(-1, 2)[good], (g[key] - p).reshape(-1, 2)[good]
            assert np.abs(flow - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-3), (i, t)      # north_star gate (SIMD builds reorder the float sums)
            p = g[key][st == 1].reshape(-1, 1, 2)
