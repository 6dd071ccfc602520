"""
oracle/ref_loader.py -- TEST INFRASTRUCTURE ONLY; usable ONLY in the build container.

Imports the REAL reference modules (base, transforms, pyramid, tools) from
/root/reference so that their numpy/scipy glue can be run authentically and its
outputs captured as golden vectors (oracle/make_golden.py -> tests/golden/).

The reference cannot be imported as-is: it needs cv2, pywt, peakutils and
pyqtgraph, none of which is installed or installable here.  This loader puts
stand-in modules into sys.modules first:

  cv2        -> thin namespace whose pyrDown/pyrUp/threshold/findContours/
                contourArea/boundingRect/goodFeaturesToTrack/calcOpticalFlowPyrLK/
                cvtColor route to oracle/cvref*.c (the build's own restatement;
                PARITY UNPINNED).  Every call site the reference's hot path
                reaches is covered; anything else raises.
  pywt, peakutils, pyqtgraph(.Qt) -> inert stubs (off the hot path).

Nothing here is copied from the reference and nothing from the reference is
written anywhere: only input/output ARRAYS of its functions are saved.
The GPU box has no /root/reference; nothing that runs there imports this file.
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "transforms.py"))


def _make_cv2_stub():
    from oracle import respmon_oracle as O

    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "3.4-restated-by-oracle/cvref.c"
    cv2.THRESH_BINARY = O.THRESH_BINARY
    cv2.RETR_EXTERNAL = O.RETR_EXTERNAL
    cv2.CHAIN_APPROX_SIMPLE = O.CHAIN_APPROX_SIMPLE
    cv2.CHAIN_APPROX_NONE = O.CHAIN_APPROX_NONE
    cv2.TERM_CRITERIA_COUNT = 1
    cv2.TERM_CRITERIA_EPS = 2
    cv2.COLOR_BGR2GRAY = 6
    cv2.CAP_PROP_FPS = 5
    cv2.CAP_PROP_FRAME_WIDTH = 3
    cv2.CAP_PROP_FRAME_HEIGHT = 4
    cv2.pyrDown = O.pyrDown
    cv2.pyrUp = lambda img, dstsize=None: O.pyrUp(img, dstsize)
    cv2.threshold = O.threshold

    def find_contours(img, mode, method):
        # OpenCV 3.x returns (image, contours, hierarchy) -- base.py:568 unpacks three values
        return img, O.findContours(img, mode, method), None

    cv2.findContours = find_contours
    cv2.contourArea = O.contourArea
    cv2.boundingRect = O.boundingRect
    cv2.goodFeaturesToTrack = lambda img, mask=None, **kw: O.goodFeaturesToTrack(img, mask=mask, **kw)
    cv2.calcOpticalFlowPyrLK = lambda a, b, p, n, **kw: O.calcOpticalFlowPyrLK(a, b, p, n, **kw)

    def cvt_color(frame, code):
        assert code == cv2.COLOR_BGR2GRAY
        return O.cvtColor_bgr2gray(frame)

    cv2.cvtColor = cvt_color

    def _unsupported(name):
        def f(*a, **k):
            raise NotImplementedError("cv2.%s is outside the hot path and not restated" % name)
        return f

    for name in ("VideoWriter", "VideoWriter_fourcc", "drawContours", "rectangle", "imwrite"):
        setattr(cv2, name, _unsupported(name))
    cv2.VideoCapture = _unsupported("VideoCapture")  # tests replace this with a fake capture
    return cv2


def _inert(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_LOADED = None


def load():
    """Returns a namespace with the reference's modules: .base .transforms .pyramid .tools .cv2"""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    saved_path = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in
                  ("cv2", "pywt", "pywt.data", "peakutils", "pyqtgraph", "pyqtgraph.Qt",
                   "base", "transforms", "pyramid", "tools", "prototypes", "prototypes.parabolic",
                   "prototypes.wavelets")}
    cv2 = _make_cv2_stub()
    pywt = _inert("pywt", Modes=types.SimpleNamespace(smooth="smooth"), Wavelet=lambda *a, **k: None,
                  dwt=None, waverec=None)
    pywt_data = _inert("pywt.data", ecg=lambda: np.zeros(1024))
    pywt.data = pywt_data
    peakutils = _inert("peakutils", indexes=lambda y, thres=0.3, min_dist=1: np.array([], dtype=int),
                       gaussian_fit=None, gaussian=None)
    qt = _inert("pyqtgraph.Qt", QtGui=types.SimpleNamespace())
    pg = _inert("pyqtgraph", Qt=qt, QtGui=types.SimpleNamespace())
    sys.modules.update({"cv2": cv2, "pywt": pywt, "pywt.data": pywt_data, "peakutils": peakutils,
                        "pyqtgraph": pg, "pyqtgraph.Qt": qt})
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    try:
        for name in ("tools", "pyramid", "transforms", "base"):
            sys.modules.pop(name, None)
        mods = {name: importlib.import_module(name) for name in ("tools", "pyramid", "transforms", "base")}
    finally:
        sys.path[:] = saved_path
    ns = types.SimpleNamespace(cv2=cv2, peakutils=peakutils, **mods)
    # keep the reference modules importable only through this namespace
    for k, v in saved_mods.items():
        if k in ("cv2", "pywt", "pywt.data", "peakutils", "pyqtgraph", "pyqtgraph.Qt"):
            continue  # the reference modules hold references to the stubs; leave them registered
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    _LOADED = ns
    return ns
