"""Analytic known-answer tests anchoring the oracle's restatement of the OpenCV calls
(pyrDown / pyrUp / threshold / findContours / contourArea / boundingRect / BGR2GRAY).
cv2 itself is absent ("parity unpinned"): these KATs + an independent scipy implementation
are what pins oracle/cvref.c."""
import numpy as np
import scipy.ndimage as ndi


def test_pyrdown_matches_independent_scipy(oracle):
    rng = np.random.default_rng(0)
    k = np.array([1, 4, 6, 4, 1]) / 16.0
    for shape in [(21, 30), (8, 8), (5, 8), (3, 3), (2, 5), (1, 1), (1, 7), (68, 120), (9, 15)]:
        a = rng.random(shape)
        d = oracle.pyrDown(a)
        assert d.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
        ref = ndi.correlate1d(ndi.correlate1d(a, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")[::2, ::2]
        assert np.abs(d - ref).max() < 1e-15


def test_pyrdown_impulse_and_constant(oracle):
    a = np.zeros((17, 17)); a[8, 8] = 256.0
    d = oracle.pyrDown(a)
    # the impulse at an even position feeds taps 6 (centre) and 1 (distance 2)
    assert d[4, 4] == 36.0 and d[4, 3] == 6.0 and d[3, 3] == 1.0 and d[4, 2] == 0.0
    c = np.full((11, 14), 0.3)
    assert np.ptp(oracle.pyrDown(c)) < 1e-16


def _pyrup_ref(src, dh, dw):
    """Independent statement: zero-insert x2, correlate with [1 4 6 4 1]/8 per axis where the
    left/top border reflects (101) and the right/bottom border replicates, crop to (dh, dw)."""
    sh, sw = src.shape

    def up1d(v):
        n = len(v)
        out = np.zeros(2 * n)
        for x in range(n):
            l = v[x - 1] if x > 0 else (v[1] if n > 1 else v[0])
            r = v[x + 1] if x < n - 1 else v[n - 1]
            out[2 * x] = (l + 6 * v[x] + r) / 8.0
            out[2 * x + 1] = (4 * v[x] + 4 * r) / 8.0
        return out
    tmp = np.stack([up1d(row) for row in src])
    full = np.stack([up1d(col) for col in tmp.T]).T
    return full[:dh, :dw]


def test_pyrup_matches_independent_statement(oracle):
    rng = np.random.default_rng(1)
    for (sh, sw, dh, dw) in [(5, 8, 9, 15), (5, 8, 10, 16), (3, 3, 5, 6), (1, 1, 1, 1), (1, 1, 2, 2),
                             (2, 2, 3, 3), (34, 60, 68, 120), (68, 120, 135, 240), (1, 4, 2, 7)]:
        s = rng.random((sh, sw))
        u = oracle.pyrUp(s, (dw, dh))
        assert u.shape == (dh, dw)
        assert np.abs(u - _pyrup_ref(s, dh, dw)).max() < 1e-15


def test_pyrup_constant_and_bad_size(oracle):
    c = np.full((5, 7), 3.25)
    assert np.ptp(oracle.pyrUp(c, (14, 10))) == 0.0  # 3.25*k sums are exact in binary
    import pytest
    with pytest.raises(ValueError):
        oracle.pyrUp(c, (16, 10))


def test_threshold_is_strict(oracle):
    a = np.array([[19, 20, 21, 255, 0]], dtype=np.uint8)
    _, t = oracle.threshold(a, 20, 255)
    assert t.tolist() == [[0, 0, 255, 255, 0]]


def test_contours_rectangle_pixel_line(oracle):
    img = np.zeros((20, 30), np.uint8)
    img[4:10, 5:17] = 255
    cs = oracle.findContours(img)
    assert len(cs) == 1
    assert cs[0].reshape(-1, 2).tolist() == [[5, 4], [5, 9], [16, 9], [16, 4]]
    assert oracle.contourArea(cs[0]) == (12 - 1) * (6 - 1)
    assert oracle.boundingRect(cs[0]) == (5, 4, 12, 6)
    img[:] = 0; img[7, 11] = 1
    cs = oracle.findContours(img)
    assert len(cs) == 1 and oracle.contourArea(cs[0]) == 0.0 and oracle.boundingRect(cs[0]) == (11, 7, 1, 1)
    img[:] = 0; img[3, 2:9] = 9
    cs = oracle.findContours(img)
    assert len(cs) == 1 and oracle.contourArea(cs[0]) == 0.0 and oracle.boundingRect(cs[0]) == (2, 3, 7, 1)
    img[:] = 0
    for i in range(5):
        img[2 + i, 3 + i] = 255  # 8-connected diagonal chain is ONE component
    cs = oracle.findContours(img)
    assert len(cs) == 1 and oracle.boundingRect(cs[0]) == (3, 2, 5, 5)


def test_contours_frame_pixels_count_and_order_and_nesting(oracle):
    img = np.zeros((12, 16), np.uint8)
    img[0:4, 0:5] = 255          # touches the image frame (OpenCV >= 3.2 rule: kept whole)
    img[8:12, 10:16] = 255
    cs = oracle.findContours(img)
    assert len(cs) == 2
    # cv2 order = reverse raster discovery: the lower blob comes first
    assert oracle.boundingRect(cs[0]) == (10, 8, 6, 4)
    assert oracle.boundingRect(cs[1]) == (0, 0, 5, 4)
    # tie on area -> max() keeps the first in cv2 order (the later-discovered blob)
    img[:] = 0; img[1:4, 1:5] = 255; img[7:10, 9:13] = 255
    cs = oracle.findContours(img)
    best = max(cs, key=oracle.contourArea)
    assert oracle.boundingRect(best) == (9, 7, 4, 3)
    # a blob inside a hole of another blob is not an EXTERNAL contour
    img = np.zeros((15, 15), np.uint8)
    img[1:14, 1:14] = 255; img[4:11, 4:11] = 0; img[6:9, 6:9] = 255
    cs = oracle.findContours(img)
    assert len(cs) == 1 and oracle.boundingRect(cs[0]) == (1, 1, 13, 13)
    assert oracle.contourArea(cs[0]) == 144.0  # holes are not subtracted


def test_contours_bboxes_match_scipy_label(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        img = (ndi.gaussian_filter(rng.standard_normal((40, 56)), 2.0) > 0.12).astype(np.uint8) * 255
        lab, n = ndi.label(img, structure=np.ones((3, 3)))
        filled = ndi.binary_fill_holes(img)  # nested blobs disappear into their parents
        lab_f, n_f = ndi.label(filled, structure=np.ones((3, 3)))
        boxes = set()
        for sl in ndi.find_objects(lab_f):
            boxes.add((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
        cs = oracle.findContours(img)
        got = set(oracle.boundingRect(c) for c in cs)
        assert got == boxes
        # Pick's theorem on the filled component: polygon through boundary pixel centres
        for c in cs:
            full = oracle.findContours(img, method=oracle.CHAIN_APPROX_NONE)
            break
        assert sorted(oracle.contourArea(c) for c in cs) == sorted(oracle.contourArea(c) for c in full)


def test_bgr2gray(oracle):
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    g = oracle.cvtColor_bgr2gray(px)
    exp = [(b * 1868 + gg * 9617 + r * 4899 + 8192) >> 14 for b, gg, r in px[0].astype(int)]
    assert g[0].tolist() == exp and g[0, 0] == 255 and g[0, 2] == 29 and g[0, 3] == 150 and g[0, 4] == 76


def test_find_contours_version_switch(oracle):
    """OpenCV <= 3.1 zeroes the 1-pixel image frame before tracing, >= 3.2 traces on a padded copy.  A
    rectangle touching the top-left corner gives the two documented results; a blob away from the frame gives the same."""
    img = np.zeros((12, 16), np.uint8)
    img[0:5, 0:7] = 255                       # rows 0..4, cols 0..6, touching the frame
    c_new = oracle.findContours(img)
    c_old = oracle.findContours(img, clip_frame=True)
    assert len(c_new) == 1 and len(c_old) == 1
    assert oracle.boundingRect(c_new[0]) == (0, 0, 7, 5) and oracle.contourArea(c_new[0]) == 6 * 4      # (w-1)(h-1)
    assert oracle.boundingRect(c_old[0]) == (1, 1, 6, 4) and oracle.contourArea(c_old[0]) == 5 * 3      # clipped to [1, W-2] x [1, H-2]
    assert oracle.roi_from_heatmap_u8(img, 20) == (0, 0, 7, 5) and oracle.roi_from_heatmap_u8(img, 20, clip_frame=True) == (1, 1, 6, 4)
    # a one-pixel line ON the frame disappears entirely under the <= 3.1 rule
    line = np.zeros((8, 9), np.uint8); line[:, 8] = 255
    assert oracle.roi_from_heatmap_u8(line, 20) == (8, 0, 1, 8) and oracle.roi_from_heatmap_u8(line, 20, clip_frame=True) is None
    inner = np.zeros((12, 16), np.uint8); inner[3:8, 4:11] = 255
    assert oracle.roi_from_heatmap_u8(inner, 20) == oracle.roi_from_heatmap_u8(inner, 20, clip_frame=True) == (4, 3, 7, 5)
    assert img[0, 0] == 255                   # the caller's image is not modified (the reference passes a copy, base.py:567)
