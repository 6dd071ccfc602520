"""The HIP kernels (through the C-ABI) against goldens captured from a real OpenCV (tools/capture_cv2_goldens.py); skipped until
tests/golden/cv2_*.npz exist -- see tests/test_cv2_goldens.py."""
import numpy as np
import pytest

from tests.test_cv2_goldens import close, load

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_hip_pyramid_steps_vs_cv2():
    g = load("cv2_pyr.npz")
    from respmon_amd import pyramid
    for i in range(int(g["n"])):
        a = g["in_%d" % i]
        if min(a.shape) < 1:
            continue
        pyr = pyramid.create_gaussian_image_pyramid(a, 2)
        close(np.asarray(pyr[1]), g["down_%d" % i], ("pyrDown", i, a.shape))
        lap = pyramid.create_laplacian_image_pyramid(a, 2)     # L_0 = a - pyrUp(pyrDown(a)): exercises pyrUp at the parent's size
        close(np.asarray(lap[0]), a - g["up_%d" % i], ("pyrUp", i, a.shape))


def test_hip_roi_stage_vs_cv2():
    g = load("cv2_contours.npz")
    from respmon_amd import dist
    ver = tuple(int(x) for x in str(g["cv2_version"]).split(".")[:2])
    for i in range(int(g["n"])):
        img = g["img_%d" % i].astype(np.float64)
        if img.max() == img.min():
            continue                           # a flat heatmap normalises to NaN: no contour by construction
        want = tuple(int(v) for v in g["roi_%d" % i])
        for labelling in (False, True):
            roi = dist.hip_heatmap_to_roi(_dev(img), 20, clip_frame=ver < (3, 2), labelling=None if ver < (3, 2) else labelling)
            assert (roi is None and want[2] < 0) or roi == want, (i, labelling, roi, want)


def test_hip_corners_and_lk_vs_cv2():
    g = load("cv2_flow.npz")
    from respmon_amd.base import _Backend
    be = _Backend()
    for i in range(int(g["n"])):
        frames = g["frames_%d" % i]
        mc, q, md, bs = g["params_%d" % i]
        pts = be.good_features_to_track(_dev(frames[0]), int(mc), float(q), float(md), int(bs))
        want = g["corners_%d" % i]
        assert (pts is None and len(want) == 0) or np.array_equal(pts, want), i
        p = pts
        for t in range(1, 4):
            key = "lk_p1_%d_%d" % (i, t)
            if key not in g.files:
                break
            p1, st = be.calc_optical_flow_pyr_lk(_dev(frames[t - 1]), _dev(frames[t]), p, (15, 15), 2, (3, 10, 0.03))
            assert np.array_equal(st, g["lk_st_%d_%d" % (i, t)]), (i, t)
            good = st.ravel() == 1
            flow, ref = (p1 - p).reshape(-1, 2)[good], (g[key] - p).reshape(-1, 2)[good]
            assert np.abs(flow - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-3), (i, t)
            p = g[key][st == 1].reshape(-1, 1, 2)
