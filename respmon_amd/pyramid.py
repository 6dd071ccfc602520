"""Device-backed counterparts of the reference's pyramid.py (same function names / arguments).
Inputs may be numpy arrays (results come back as numpy, like the reference) or CUDA tensors."""
import ctypes

from . import _capi, device


def _level_shapes(T, H, W, levels):
    shapes = [(T, H, W)]
    for _ in range(1, levels):
        _, h, w = shapes[-1]
        shapes.append((T, (h + 1) // 2, (w + 1) // 2))
    return shapes


def create_gaussian_image_pyramid(image, pyramid_levels):
    """reference pyramid.py:9-17: [float64 copy of the image, then (levels - 1) x cv2.pyrDown]."""
    t = device.require_gpu()
    lib = _capi.load()
    cur = device.to_device(image, t.float64).clone()
    out = [cur]
    for _ in range(1, pyramid_levels):
        h, w = cur.shape
        nxt = t.empty(((h + 1) // 2, (w + 1) // 2), dtype=t.float64, device=cur.device)
        _capi.check(lib, lib.rm_pyr_down(device.ctx(), device.ptr(cur), _capi.RM_F64, 1, h, w, device.ptr(nxt), device.stream_ptr()),
                    "rm_pyr_down")
        out.append(nxt)
        cur = nxt
    return [device.like_input(lv, image) for lv in out]


def create_laplacian_image_pyramid(image, pyramid_levels):
    """reference pyramid.py:20-28: L_i = G_i - pyrUp(G_{i+1}, dstsize = size of G_i); last level = coarsest Gaussian."""
    t = device.require_gpu()
    frame = device.to_device(image, t.float64)
    levels = create_laplacian_video_pyramid(frame.unsqueeze(0), pyramid_levels)
    return [device.like_input(lv[0], image) for lv in levels]


def collapse_laplacian_pyramid(image_pyramid):
    """reference pyramid.py:51-57: img = pop(); while levels remain: img = pyrUp(img, size of next) + next."""
    t = device.require_gpu()
    levels = [device.to_device(lv, t.float64).unsqueeze(0).clone() for lv in image_pyramid]
    return device.like_input(collapse_laplacian_video_pyramid(levels)[0], image_pyramid[0])


def create_laplacian_video_pyramid(video, pyramid_levels):
    """reference pyramid.py:31-48: list of `pyramid_levels` float64 arrays [T,h_l,w_l];
    L_i = G_i - pyrUp(G_{i+1}), last level = coarsest Gaussian."""
    t = device.require_gpu()
    lib = _capi.load()
    vid = device.to_device(video)
    T, H, W = vid.shape
    levels = [t.empty(s, dtype=t.float64, device=vid.device) for s in _level_shapes(T, H, W, pyramid_levels)]
    ptrs = (ctypes.c_void_p * pyramid_levels)(*[lv.data_ptr() for lv in levels])
    _capi.check(lib, lib.rm_create_laplacian_video_pyramid(device.ctx(), device.ptr(vid), device.dtype_code(vid), T, H, W,
                                                           pyramid_levels, ptrs, device.stream_ptr()),
                "rm_create_laplacian_video_pyramid")
    return [device.like_input(lv, video) for lv in levels]


def collapse_laplacian_video_pyramid(pyramid):
    """reference pyramid.py:60-69: img = pyrUp(img) + level, coarsest to finest, per frame.
    Like the reference the result is written into (and returned as) pyramid[0] when it is a tensor."""
    t = device.require_gpu()
    lib = _capi.load()
    levels = [device.to_device(lv, t.float64) for lv in pyramid]
    T, H, W = levels[0].shape
    n = len(levels)
    ptrs = (ctypes.c_void_p * n)(*[lv.data_ptr() for lv in levels])
    out = levels[0]
    _capi.check(lib, lib.rm_collapse_laplacian_video_pyramid(device.ctx(), ptrs, T, H, W, n, device.ptr(out), device.stream_ptr()),
                "rm_collapse_laplacian_video_pyramid")
    res = device.like_input(out, pyramid[0])
    if not isinstance(pyramid[0], t.Tensor):
        pyramid[0][...] = res  # the reference mutates pyramid[0] in place (pyramid.py:65)
        return pyramid[0]
    return res
