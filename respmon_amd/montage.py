"""The calibration image of RespiratoryMonitor.locate(save_calibration_image=True), reference base.py:577-596
(SURVEY 8f row f1): a 2 x 3 montage

    avg_original | avg_raw      | avg          time means: the frames, the raw band-passed video, the masked one
    thresh       | contour_img  | drawn        threshold image, contours on the mean frame, ROI rectangle on mean + avg

The DATA of the panels (three time averages, normalisation, float_to_uint8, threshold) comes from the device
through the C-ABI and is bit-comparable with the oracle.  The two drawing calls (cv2.drawContours with thickness 3,
cv2.rectangle with thickness 2) and the PNG encoder are OpenCV code the reference only uses for this debug
picture; they are restated approximately on the host (scipy morphology, zlib) -- "parity unpinned", like every
cv2 call, and irrelevant to the ROI.
"""
import ctypes
import os
import struct
import zlib

import numpy as np

from . import _capi, device


def time_average(video):
    """np.average(video, axis=0) on the device (rm_time_average): float64 [H,W] tensor."""
    t = device.require_gpu()
    lib = _capi.load()
    v = device.to_device(video)
    if v.dim() == 4:     # a [T,H,W,3] uint8 frame buffer: the average of its gray frames (base.py:230)
        from . import transforms
        v = transforms.bgr_buffer_to_gray(v)
    T = v.shape[0]
    out = t.empty(tuple(v.shape[1:]), dtype=t.float64, device=v.device)
    _capi.check(lib, lib.rm_time_average(device.ctx(), device.ptr(v), device.dtype_code(v), T, out.numel(), device.ptr(out),
                                         device.stream_ptr()), "rm_time_average")
    return out


def _to_u8(x):
    from . import transforms
    return transforms.float_to_uint8(x)


def calibration_panels(calibration_video_data, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9,
                       skip_levels_at_top=4, temporal_threshold=0.7, threshold=20):
    """The six uint8 panels (numpy, [H,W]) and the ROI.  Materialising path: two [T,H,W] float64 arrays live on the
    device while it runs (8.5 GB at 1080p x 256) -- a debug facility, not the calibration path."""
    from . import transforms
    t = device.require_gpu()
    vid = device.to_device(calibration_video_data)
    masked, raw = transforms.eulerian_magnification_bandpass(vid, fps, freq_min, freq_max, amplification,
                                                             pyramid_levels=pyramid_levels, skip_levels_at_top=skip_levels_at_top,
                                                             threshold=temporal_threshold)
    avg_frame = time_average(masked)                                              # base.py:562
    del masked
    avg_raw_frame = time_average(raw)                                             # base.py:587
    del raw
    total_avg = _to_u8(time_average(vid)).cpu().numpy()                           # base.py:579, 589 (the same image twice)

    def norm_u8(a):                                                                # base.py:563-564, 588
        a = a.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            n = (a - a.min()) / (a.max() - a.min())
        return _to_u8(n)

    avg = norm_u8(avg_frame)
    avg_raw = norm_u8(avg_raw_frame)
    thresh = np.where(avg > threshold, 255, 0).astype(np.uint8)                    # cv2.threshold THRESH_BINARY, base.py:566
    from .dist import hip_heatmap_to_roi
    roi = hip_heatmap_to_roi(avg_frame, threshold)                                 # base.py:568-575
    contour_img = draw_external_contours(total_avg, thresh, value=0, thickness=3)  # base.py:580-581: colour (0,255,0) on gray -> 0
    drawn = (total_avg + avg).astype(np.uint8)                                     # base.py:583: uint8 addition wraps
    if roi is not None:
        x, y, w, h = roi
        drawn = draw_rectangle(drawn, x, y, x + w, y + h, value=255, thickness=2)
    return dict(avg_original=total_avg, avg_raw=avg_raw, avg=avg, thresh=thresh, contour_img=contour_img, drawn=drawn), roi


def draw_external_contours(img, binary, value=0, thickness=3):
    """Stand-in for cv2.drawContours(img, contours, -1, colour, 3) with RETR_EXTERNAL contours: the outer boundary
    pixels of every 8-connected component (holes filled), thickened."""
    import scipy.ndimage as ndi
    fg = binary != 0
    eight = np.ones((3, 3), bool)
    lab, n = ndi.label(fg, structure=eight)
    filled = np.zeros_like(fg)
    for i in range(1, n + 1):
        filled |= ndi.binary_fill_holes(lab == i)
    four = ndi.generate_binary_structure(2, 1)
    border = filled & ~ndi.binary_erosion(filled, structure=four, border_value=0)
    if thickness > 1:
        border = ndi.binary_dilation(border, structure=eight, iterations=(thickness - 1) // 2)
    out = img.copy()
    out[border] = value
    return out


def draw_rectangle(img, x0, y0, x1, y1, value=255, thickness=2):
    """Stand-in for cv2.rectangle(img, (x0,y0), (x1,y1), value, thickness): the four edges, `thickness` pixels wide."""
    out = img.copy()
    H, W = out.shape
    lo, hi = (thickness - 1) // 2, thickness // 2
    for (ya, yb, xa, xb) in [(y0 - lo, y0 + hi, x0 - lo, x1 + hi), (y1 - lo, y1 + hi, x0 - lo, x1 + hi),
                             (y0 - lo, y1 + hi, x0 - lo, x0 + hi), (y0 - lo, y1 + hi, x1 - lo, x1 + hi)]:
        out[max(ya, 0):min(yb + 1, H), max(xa, 0):min(xb + 1, W)] = value
    return out


def montage(panels):
    row0 = np.hstack((panels["avg_original"], panels["avg_raw"], panels["avg"]))    # base.py:590
    row1 = np.hstack((panels["thresh"], panels["contour_img"], panels["drawn"]))    # base.py:591
    return np.vstack((row0, row1))                                                  # base.py:592


def write_png_gray(path, img):
    """8-bit grayscale PNG (what cv2.imwrite produces for a single-channel uint8 image), zlib only."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png_gray(path):
    """Inverse of write_png_gray (tests)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w + 1)
    return rows[:, 1:].copy()


def save_calibration_image(calibration_video_data, fps, directory=".", **kw):
    """base.py:577-596: writes calibration<i>.png (first unused i) into `directory`; returns (path, roi)."""
    panels, roi = calibration_panels(calibration_video_data, fps, **kw)
    i = 0
    while os.path.exists(os.path.join(directory, "calibration%s.png" % i)):     # base.py:593-595
        i += 1
    path = os.path.join(directory, "calibration%s.png" % i)
    write_png_gray(path, montage(panels))
    return path, roi
