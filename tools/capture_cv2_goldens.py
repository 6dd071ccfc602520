#!/usr/bin/env python
"""Capture real-OpenCV goldens for the cv2 calls on the respmon hot path (run this wherever `import cv2` works):

    python tools/capture_cv2_goldens.py [out_dir]          # default: tests/golden  ->  cv2_*.npz

Imports cv2 and numpy only -- never the reference, never this repository's packages -- so it can run in any environment that
has an OpenCV build (the reference names "opencv3" from conda channel menpo, README.md:12).  Each file stores the INPUTS and what
cv2 returned for them, plus cv2.__version__; tests/test_cv2_goldens.py (CPU: the oracle's restatement) and
tests/test_gpu_cv2_goldens.py (GPU: the HIP kernels through the C-ABI) consume whatever files are present and skip otherwise.

Call sites covered (reference file:line):
  pyramid.py:14      cv2.pyrDown(float64 image)                              -> cv2_pyr.npz
  pyramid.py:25,55   cv2.pyrUp(float64 image, dstsize=(w, h))                -> cv2_pyr.npz
  base.py:230        cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)                 -> cv2_misc.npz
  base.py:566        cv2.threshold(uint8, thresh, 255, THRESH_BINARY)        -> cv2_misc.npz
  base.py:568-575    cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), max(key=cv2.contourArea), cv2.boundingRect -> cv2_contours.npz
  base.py:365        cv2.goodFeaturesToTrack(uint8, **feature_params)        -> cv2_flow.npz
  base.py:371-372    cv2.calcOpticalFlowPyrLK(prev, cur, p0, None, **lk)     -> cv2_flow.npz
"""
import os
import sys

import numpy as np


def texture(h, w, seed):
    """Smooth random texture with corners (sum of random sinusoids and Gaussian spots), float64 in [0, 1] as a function of a
    sub-pixel shift."""
    rng = np.random.default_rng(seed)
    kx, ky = rng.uniform(-0.9, 0.9, 12), rng.uniform(-0.9, 0.9, 12)
    ph, am = rng.uniform(0, 2 * np.pi, 12), rng.uniform(0.3, 1.0, 12)
    sx, sy, ss, sa = rng.uniform(0, w, 40), rng.uniform(0, h, 40), rng.uniform(1.5, 4.0, 40), rng.uniform(-1, 1, 40)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def render(dx, dy):
        x, y = xx - dx, yy - dy
        img = np.zeros((h, w))
        for i in range(12):
            img += am[i] * np.sin(kx[i] * x + ky[i] * y + ph[i])
        for i in range(40):
            img += 3.0 * sa[i] * np.exp(-0.5 * ((x - sx[i]) ** 2 + (y - sy[i]) ** 2) / ss[i] ** 2)
        img = (img - img.min()) / (img.max() - img.min())
        return np.clip(np.round(255 * img), 0, 255).astype(np.uint8)
    return render


def main():
    import cv2
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(out, exist_ok=True)
    ver = np.array(cv2.__version__)
    rng = np.random.default_rng(20260930)

    # ---- pyrDown / pyrUp on float64: odd, even, 1- and 2-wide sizes, values of either sign
    pyr = {"cv2_version": ver}
    sizes = [(9, 15), (1, 7), (7, 1), (2, 2), (3, 3), (16, 16), (33, 47), (5, 2), (68, 120), (1, 1), (2, 9)]
    for i, (h, w) in enumerate(sizes):
        a = rng.standard_normal((h, w)) * (10.0 ** rng.integers(-3, 4))
        if i % 3 == 0:
            a = rng.integers(0, 256, (h, w)).astype(np.float64) * (1.0 / 255)      # what uint8_to_float produces
        pyr["in_%d" % i] = a
        d = cv2.pyrDown(a)
        pyr["down_%d" % i] = d
        pyr["up_%d" % i] = cv2.pyrUp(d, dstsize=(w, h))                            # pyramid.py:25: back to the parent's size
    pyr["n"] = np.array(len(sizes))
    np.savez_compressed(os.path.join(out, "cv2_pyr.npz"), **pyr)

    # ---- cvtColor / threshold
    misc = {"cv2_version": ver}
    bgr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    misc["bgr"] = bgr
    misc["gray"] = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
    u8 = rng.integers(0, 256, (29, 41), dtype=np.uint8)
    misc["thr_in"] = u8
    for t in (0, 20, 127, 254, 255):
        misc["thr_%d" % t] = cv2.threshold(u8, t, 255, cv2.THRESH_BINARY)[1]
    np.savez_compressed(os.path.join(out, "cv2_misc.npz"), **misc)

    # ---- findContours(EXTERNAL, SIMPLE) -> max(key=contourArea) -> boundingRect   (base.py:568-575)
    cont = {"cv2_version": ver}
    masks = []
    for (h, w, dens) in [(40, 131, 0.15), (64, 64, 0.35), (57, 200, 0.5), (33, 129, 0.62), (20, 70, 0.05)]:
        masks.append(rng.random((h, w)) < dens)
    ring = np.zeros((60, 150), bool)
    ring[5:55, 10:90] = True; ring[12:48, 20:80] = False; ring[20:40, 30:70] = True; ring[25:35, 40:60] = False
    ring[:, 100:] = rng.random((60, 50)) < 0.3
    masks.append(ring)
    frame = np.zeros((30, 40), bool)                      # blobs that touch the image frame (the <= 3.1 / >= 3.2 difference)
    frame[0:6, 0:9] = True; frame[24:30, 30:40] = True; frame[10:20, 15:25] = True
    masks.append(frame)
    ties = np.zeros((30, 60), bool)                       # equal-area contours: which one max() keeps
    ties[3:9, 3:13] = True; ties[3:9, 30:40] = True; ties[18:24, 10:20] = True
    masks.append(ties)
    full = np.ones((17, 66), bool); full[8, 33] = False
    masks.append(full)
    masks.append(np.zeros((12, 12), bool))                # no contour at all
    single = np.zeros((9, 9), bool); single[4, 4] = True  # zero-area contour only
    masks.append(single)
    for i, m in enumerate(masks):
        img = np.where(m, 255, 0).astype(np.uint8)
        res = cv2.findContours(img.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
        contours = res[0] if len(res) == 2 else res[1]    # OpenCV 3.x returns (image, contours, hierarchy)
        cont["img_%d" % i] = img
        cont["n_contours_%d" % i] = np.array(len(contours))
        cont["areas_%d" % i] = np.array([cv2.contourArea(c) for c in contours], dtype=np.float64)
        cont["rects_%d" % i] = np.array([cv2.boundingRect(c) for c in contours], dtype=np.int64).reshape(-1, 4)
        if len(contours):
            best = max(contours, key=cv2.contourArea)     # base.py:571
            cont["roi_%d" % i] = np.array(cv2.boundingRect(best), dtype=np.int64)
        else:
            cont["roi_%d" % i] = np.array([-1, -1, -1, -1], dtype=np.int64)
    cont["n"] = np.array(len(masks))
    np.savez_compressed(os.path.join(out, "cv2_contours.npz"), **cont)

    # ---- goodFeaturesToTrack + calcOpticalFlowPyrLK   (base.py:91-98 parameters; BASELINE config 3 sizes)
    flow = {"cv2_version": ver}
    cases = [(256, 256, 4321, dict(maxCorners=1000, qualityLevel=0.01, minDistance=3, blockSize=7)),
             (256, 256, 4321, dict(maxCorners=1000, qualityLevel=0.01, minDistance=7, blockSize=7)),
             (51, 70, 99, dict(maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7))]
    lk = dict(winSize=(15, 15), maxLevel=2, criteria=(cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 10, 0.03))
    for i, (h, w, seed, fp) in enumerate(cases):
        render = texture(h, w, seed)
        frames = [render(1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)) for t in range(4)]
        flow["frames_%d" % i] = np.stack(frames)
        flow["params_%d" % i] = np.array([fp["maxCorners"], fp["qualityLevel"], fp["minDistance"], fp["blockSize"]], dtype=np.float64)
        p0 = cv2.goodFeaturesToTrack(frames[0], mask=None, **fp)
        flow["corners_%d" % i] = np.zeros((0, 1, 2), np.float32) if p0 is None else p0
        p = p0
        for t in range(1, 4):
            if p is None or len(p) == 0:
                break
            p1, st, err = cv2.calcOpticalFlowPyrLK(frames[t - 1], frames[t], p, None, **lk)
            flow["lk_p1_%d_%d" % (i, t)] = p1
            flow["lk_st_%d_%d" % (i, t)] = st
            p = p1[st == 1].reshape(-1, 1, 2)
    flow["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(out, "cv2_flow.npz"), **flow)
    print("cv2 %s: wrote cv2_pyr.npz, cv2_misc.npz, cv2_contours.npz, cv2_flow.npz to %s" % (cv2.__version__, out))


if __name__ == "__main__":
    main()
