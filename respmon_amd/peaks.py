"""Peak helpers for measure()/find_peaks() (reference base.py:312-338 calls peakutils.indexes and
peakutils.gaussian_fit).  peakutils is an un-pinned pip dependency that is not installable in the build
image, so these are restatements of its published algorithm (PeakUtils 1.x) -- parity unpinned.
Host-side: a 128-sample 1-D signal, microseconds of CPU; SURVEY.md section 8(f) row f2 ("next")."""
import numpy as np


def indexes(y, thres=0.3, min_dist=1):
    """First-difference peak finder with plateau handling, relative threshold and greedy min-distance
    suppression (highest peaks first)."""
    y = np.asarray(y, dtype=float)
    if y.size < 3:
        return np.array([], dtype=int)
    level = thres * (np.max(y) - np.min(y)) + np.min(y)
    min_dist = int(min_dist)
    dy = np.diff(y)
    flat, = np.where(dy == 0)
    if len(flat) == len(y) - 1:
        return np.array([], dtype=int)
    if len(flat):
        breaks, = np.add(np.where(np.diff(flat) != 1), 1)
        runs = np.split(flat, breaks)
        if runs[0][0] == 0:
            dy[runs[0]] = dy[runs[0][-1] + 1]
            runs.pop(0)
        if len(runs) and runs[-1][-1] == len(dy) - 1:
            dy[runs[-1]] = dy[runs[-1][0] - 1]
            runs.pop(-1)
        for run in runs:
            mid = np.median(run)
            dy[run[run < mid]] = dy[run[0] - 1]
            dy[run[run >= mid]] = dy[run[-1] + 1]
    peaks = np.where((np.hstack([dy, 0.]) < 0.) & (np.hstack([0., dy]) > 0.) & (y > level))[0]
    if peaks.size > 1 and min_dist > 1:
        order = peaks[np.argsort(y[peaks])][::-1]
        removed = np.ones(y.size, dtype=bool)
        removed[peaks] = False
        for p in order:
            if not removed[p]:
                removed[max(0, p - min_dist):p + min_dist + 1] = True
                removed[p] = False
        peaks = np.arange(y.size)[~removed]
    return peaks


def gaussian(x, ampl, center, dev):
    return ampl * np.exp(-(x - float(center)) ** 2 / (2.0 * dev ** 2))


def gaussian_fit(x, y):
    """Least-squares Gaussian fit -> (amplitude, centre, deviation); raises RuntimeError when it fails."""
    from scipy import optimize
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    if len(x) < 3:
        raise RuntimeError("not enough samples for a Gaussian fit")
    initial = [np.max(y), x[0], (x[1] - x[0]) * 5]
    params, _ = optimize.curve_fit(gaussian, x, y, initial)
    return params
