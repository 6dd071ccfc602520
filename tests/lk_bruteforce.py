"""TEST INFRASTRUCTURE: a plain float64 pyramidal Lucas-Kanade tracker, written independently of OpenCV's fixed-point one
(no W_BITS weights, no DESCALE, no Scharr kernels, no early-exit rules): bilinear sampling with scipy's map_coordinates,
central-difference gradients, Gauss-Newton on the 15 x 15 window, coarse to fine.  It bounds the error of the restated /
HIP calcOpticalFlowPyrLK without a cv2 binary (Next 6b): on a smooth texture the two must agree to ~1e-3 px."""
import numpy as np
import scipy.ndimage as ndi


def _pyr_down(img):
    k = np.array([1, 4, 6, 4, 1]) / 16.0
    t = ndi.correlate1d(img, k, axis=0, mode='mirror')
    t = ndi.correlate1d(t, k, axis=1, mode='mirror')
    return t[::2, ::2]


def lk_float(prev, nxt, pts, win=15, levels=2, iters=30, eps=1e-4):
    P = [np.asarray(prev, dtype=np.float64)]
    N = [np.asarray(nxt, dtype=np.float64)]
    for _ in range(levels):
        P.append(_pyr_down(P[-1]))
        N.append(_pyr_down(N[-1]))
    half = win // 2
    oy, ox = np.mgrid[-half:half + 1, -half:half + 1]
    grads = [np.gradient(I) for I in P]
    out = np.zeros((len(pts), 2))
    for i, (x, y) in enumerate(np.asarray(pts, dtype=np.float64).reshape(-1, 2)):
        g = np.zeros(2)
        for l in range(levels, -1, -1):
            I, J = P[l], N[l]
            gy, gx = grads[l]
            cy, cx = y / 2 ** l + oy, x / 2 ** l + ox
            Iw = ndi.map_coordinates(I, [cy, cx], order=1, mode='nearest')
            Ix = ndi.map_coordinates(gx, [cy, cx], order=1, mode='nearest')
            Iy = ndi.map_coordinates(gy, [cy, cx], order=1, mode='nearest')
            A = np.array([[np.sum(Ix * Ix), np.sum(Ix * Iy)], [np.sum(Ix * Iy), np.sum(Iy * Iy)]])
            v = np.zeros(2)
            for _ in range(iters):
                Jw = ndi.map_coordinates(J, [cy + g[1] + v[1], cx + g[0] + v[0]], order=1, mode='nearest')
                d = Iw - Jw
                dv = np.linalg.solve(A, np.array([np.sum(d * Ix), np.sum(d * Iy)]))
                v += dv
                if dv @ dv < eps * eps:
                    break
            g = (g + v) * (2 if l > 0 else 1)
        out[i] = [x + g[0], y + g[1]]
    return out


def config3_case(oracle, t=3, n=150):
    """BASELINE config 3 texture at frame t: (prev, next, interior tracked corners [n,2], true shift)."""
    from respmon_amd import synth
    render = synth.synth_texture(256, 256, seed=4321)
    a = render(0.0, 0.0)
    dx, dy = 1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)
    b = render(dx, dy)
    pts = oracle.goodFeaturesToTrack(a, 1000, 0.01, 7, blockSize=7).reshape(-1, 2)
    m = (pts[:, 0] > 20) & (pts[:, 0] < 235) & (pts[:, 1] > 20) & (pts[:, 1] < 235)
    return a, b, pts[m][:n].astype(np.float32), (dx, dy)
