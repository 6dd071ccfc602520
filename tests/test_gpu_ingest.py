"""Deterministic GPU tests of the ingest rows (reference base.py:227-233): cv2.cvtColor(BGR2GRAY) on the device,
uint8_to_float into every calibration-buffer dtype (`store_frame`), and the per-call contour options of the ROI stage."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from respmon_amd import _capi
    return _capi.load()


def test_bgr_to_gray_every_sampled_triple_and_a_1080p_frame(hip, oracle):
    """rm_bgr_to_gray against the oracle's cvtColor_bgr2gray (OpenCV's fixed-point weights, base.py:230): every (B, G, R) on the
    lattice {0, 5, .., 255}^3 plus the lattice's neighbours of the rounding boundaries, and a random 1080p frame."""
    from respmon_amd.base import _Backend
    be = _Backend()
    v = np.arange(0, 256, 5, dtype=np.uint8)
    v = np.unique(np.concatenate([v, [1, 2, 3, 127, 128, 129, 253, 254, 255]])).astype(np.uint8)
    b, g, r = np.meshgrid(v, v, v, indexing="ij")
    lattice = np.stack([b.ravel(), g.ravel(), r.ravel()], axis=1)          # [N, 3]
    n = lattice.shape[0]
    w = 512
    h = (n + w - 1) // w
    img = np.zeros((h * w, 3), dtype=np.uint8)
    img[:n] = lattice
    img = img.reshape(h, w, 3)
    got = be.bgr_to_gray(img).cpu().numpy()
    assert np.array_equal(got, oracle.cvtColor_bgr2gray(img))
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    got = be.bgr_to_gray(frame).cpu().numpy()
    assert np.array_equal(got, oracle.cvtColor_bgr2gray(frame))
    # all 2^24 triples: the weights are exact integers, so a closed form over the full cube is cheap on both sides
    bb, gg, rr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(0, 256, 1, dtype=np.uint8), indexing="ij")
    cube = np.stack([bb, gg, rr], axis=-1).reshape(4096, 4096, 3)
    want = oracle.cvtColor_bgr2gray(cube)
    got = be.bgr_to_gray(cube).cpu().numpy()
    assert np.array_equal(got, want)
    # ... and every triple at each of the four byte positions a pixel can take inside the words the kernels read (k_bgr_to_gray_quads and
    # the RM_BGR8 frame-buffer chain share bgr_gray_x8: whole-word dot products, rm_kernels.h)
    flat = cube.reshape(-1, 3)
    for shift in (1, 2, 3):
        rolled = np.roll(flat, shift, axis=0).reshape(4096, 4096, 3)
        got = be.bgr_to_gray(rolled).cpu().numpy()
        assert np.array_equal(got.reshape(-1), np.roll(want.reshape(-1), shift))
    odd = np.ascontiguousarray(flat[:4099 * 3].reshape(3, 4099, 3))      # a pixel count that is not a multiple of four: scalar tail
    assert np.array_equal(be.bgr_to_gray(odd).cpu().numpy(), oracle.cvtColor_bgr2gray(odd))


def test_bgr_frame_buffer_equals_its_gray_buffer(hip, oracle):
    """RM_BGR8 -- north_star's [T,H,W,C] frame buffer, frames as cv2.VideoCapture delivers them (base.py:229): the calibration applies
    base.py:230-231 while it reads the buffer.  Heatmaps bit-identical to the same call on the gray uint8 buffer the ORACLE's cvtColor
    makes of it: the fused register chain (W % 16 == 0, depths 1..4, one / several strips and segments), the whole-buffer conversion in
    front of the other chains (ragged widths, skip 0, per-level flags) and forced (`bgr_unfused`); ROI == oracle through locate(), the
    two-call form and the materialising form; the gray-only entry points refuse the code."""
    import torch
    from respmon_amd import _capi, device, dist, synth, transforms
    from respmon_amd.base import RespiratoryMonitor, _Backend
    rng = np.random.default_rng(43)
    cases = [(8, 270, 480, 5, 2, 0), (4, 135, 2000, 6, 4, 0), (3, 540, 960, 7, 3, 0), (5, 97, 1936, 4, 1, 0), (16, 1080, 1920, 9, 4, 0),
             (4, 67, 131, 5, 3, 0), (3, 48, 64, 3, 0, 0), (3, 64, 96, 4, 2, _capi.RM_FLAG_UNFUSED_DOWN), (2, 120, 160, 5, 2, _capi.RM_FLAG_FILTER_LAPLACIANS),
             (6, 720, 1280, 4, 2, 0)]
    for (T, H, W, L, S, flags) in cases:
        bgr = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
        gray = np.stack([oracle.cvtColor_bgr2gray(f) for f in bgr])
        kw = dict(pyramid_levels=L, skip_levels_at_top=S, flags=flags, return_minmax=True)
        dev_b, dev_g = torch.from_numpy(bgr).cuda(), torch.from_numpy(gray).cuda()
        assert torch.equal(transforms.bgr_buffer_to_gray(dev_b), dev_g)
        want, mm_w = dist.hip_calibrate(dev_g, 10.0, **kw)
        got, mm_g = dist.hip_calibrate(dev_b, 10.0, **kw)
        assert torch.equal(got, want) and mm_g == mm_w, (T, H, W, L, S, flags)
        device.debug_set("bgr_unfused", 1)
        try:
            got2, _ = dist.hip_calibrate(dev_b, 10.0, **kw)
        finally:
            device.debug_set("bgr_unfused", 0)
        assert torch.equal(got2, want), (T, H, W, L, S, flags)
    # a breathing video whose colour planes differ
    v8 = synth.synth_breathing(64, 270, 480, seed=5)
    bgr = np.stack([np.clip(v8.astype(np.int32) + 12, 0, 255), v8, np.clip(v8.astype(np.int32) - 9, 0, 255)], axis=-1).astype(np.uint8)
    gray = np.stack([oracle.cvtColor_bgr2gray(f) for f in bgr])
    want = oracle.locate(oracle.uint8_to_float(gray), 10.0, pyramid_levels=6, skip_levels_at_top=2)
    assert want is not None
    dev_b = torch.from_numpy(bgr).cuda()
    assert RespiratoryMonitor.locate(dev_b, 10.0, pyramid_levels=6, skip_levels_at_top=2) == want
    assert RespiratoryMonitor.locate(bgr, 10.0, pyramid_levels=6, skip_levels_at_top=2) == want       # a host array: uploaded as it is
    be = _Backend()
    tk = be.locate_submit(dev_b, 10.0, pyramid_levels=6, skip_levels_at_top=2)
    assert be.locate_result(tk) == want
    m_b, r_b = transforms.eulerian_magnification_bandpass(dev_b[:16], 10.0, 0.1, 1.0, 500, pyramid_levels=4, skip_levels_at_top=2)
    m_g, r_g = transforms.eulerian_magnification_bandpass(torch.from_numpy(gray[:16]).cuda(), 10.0, 0.1, 1.0, 500, pyramid_levels=4, skip_levels_at_top=2)
    assert torch.equal(m_b, m_g) and torch.equal(r_b, r_g)
    # the calibration image panels (base.py:577-596) of a BGR buffer are those of its gray buffer
    from respmon_amd import montage
    p_b, roi_b = montage.calibration_panels(dev_b[:32], 10.0, pyramid_levels=5, skip_levels_at_top=2)
    p_g, roi_g = montage.calibration_panels(torch.from_numpy(gray[:32]).cuda(), 10.0, pyramid_levels=5, skip_levels_at_top=2)
    assert roi_b == roi_g and all(np.array_equal(p_b[k], p_g[k]) for k in p_g)
    out = torch.empty((1, 135, 240), dtype=torch.float64, device="cuda")
    assert hip.rm_pyr_down(device.ctx(), device.ptr(dev_b), _capi.RM_BGR8, 1, 270, 480, device.ptr(out), device.stream_ptr()) == _capi.RM_E_BADARG
    with pytest.raises(TypeError):
        device.buffer_dtype_code(torch.zeros((2, 8, 8, 4), dtype=torch.uint8, device="cuda"))


def test_store_frame_every_buffer_dtype(hip, oracle):
    """calibration_buffer[idx] = uint8_to_float(gray) (base.py:231, 431) for the four device buffer dtypes: float64 holds
    k * (1 / 255) bit for bit, float32 / float16 its correctly rounded value, uint8 the level itself; all 256 levels + a random frame."""
    import torch
    from respmon_amd.base import _Backend
    be = _Backend()
    rng = np.random.default_rng(9)
    levels = np.arange(256, dtype=np.uint8).reshape(16, 16)
    frame = rng.integers(0, 256, size=(270, 480), dtype=np.uint8)
    for g8 in (levels, frame):
        ref64 = oracle.uint8_to_float(g8)
        gray = torch.from_numpy(g8).cuda()
        for tdt, ndt in ((torch.float64, np.float64), (torch.float32, np.float32), (torch.float16, np.float16), (torch.uint8, np.uint8)):
            buf = torch.zeros((3,) + g8.shape, dtype=tdt, device="cuda")
            be.store_frame(buf, 1, gray)
            torch.cuda.synchronize()
            got = buf.cpu().numpy()
            assert not got[0].any() and not got[2].any()
            want = g8 if ndt == np.uint8 else ref64.astype(ndt)
            assert np.array_equal(got[1], want), ndt
    # a 'bgr8' buffer ([T,H,W,3] uint8): the captured frame when the caller has it, else the gray value in three planes -- the
    # calibration converts either back to the same gray frame (no coupling to "the tensor bgr_to_gray returned last")
    bgr = rng.integers(0, 256, size=(270, 480, 3), dtype=np.uint8)
    gray = be.bgr_to_gray(bgr)
    buf = torch.zeros((3, 270, 480, 3), dtype=torch.uint8, device="cuda")
    be.store_frame(buf, 0, gray, bgr=be.last_bgr)
    be.store_frame(buf, 1, gray)                                     # (no BGR tensor: any gray frame)
    be.store_frame(buf, 2, torch.from_numpy(frame).cuda())           # (a gray frame that never was BGR)
    torch.cuda.synchronize()
    assert np.array_equal(buf[0].cpu().numpy(), bgr)
    assert np.array_equal(buf[1].cpu().numpy(), np.repeat(gray.cpu().numpy()[:, :, None], 3, axis=2))
    for k in (0, 1):
        assert np.array_equal(be.bgr_to_gray(buf[k].cpu().numpy()).cpu().numpy(), gray.cpu().numpy()), k
    assert np.array_equal(be.bgr_to_gray(buf[2].cpu().numpy()).cpu().numpy(), frame)
    # the float64 round trip reproduces the reference's 24 lossy levels (transforms.py:26-29) through the device helpers
    buf = torch.zeros((1, 16, 16), dtype=torch.float64, device="cuda")
    be.store_frame(buf, 0, torch.from_numpy(levels).cuda())
    back = torch.empty((16, 16), dtype=torch.uint8, device="cuda")
    from respmon_amd import _capi, device
    _capi.check(hip, hip.rm_float_to_uint8(device.ctx(), device.ptr(buf[0]), device.ptr(back), 256, device.stream_ptr()), "rm_float_to_uint8")
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy(), oracle.float_to_uint8(oracle.uint8_to_float(levels)))
    assert int((back.cpu().numpy() != levels).sum()) == 24


def test_contour_options_are_restored_after_a_per_call_override(hip):
    """dist.hip_heatmap_to_roi(clip_frame=..., labelling=...) overrides the context's contour options for ONE call and puts back what
    the context had (rm_get_contour_clip_frame / rm_get_contour_labelling)."""
    import torch
    from respmon_amd import _capi, device, dist
    ctx = device.ctx()
    heat = torch.zeros((40, 64), dtype=torch.float64, device="cuda")
    heat[0:12, 0:20] = 1.0                       # a blob that touches the image frame: the two findContours rules differ
    on, mode = ctypes.c_int(-7), ctypes.c_int(-7)
    try:
        for ctx_clip, ctx_mode in ((0, -1), (1, 1), (1, 0), (0, 1)):
            _capi.check(hip, hip.rm_set_contour_clip_frame(ctx, ctx_clip), "set clip")
            _capi.check(hip, hip.rm_set_contour_labelling(ctx, ctx_mode), "set labelling")
            plain = dist.hip_heatmap_to_roi(heat, 20)
            for clip in (False, True):
                for lab in (None, False, True):
                    roi = dist.hip_heatmap_to_roi(heat, 20, clip_frame=clip, labelling=lab)
                    assert roi == ((1, 1, 19, 11) if (clip or ctx_clip) else (0, 0, 20, 12)), (ctx_clip, ctx_mode, clip, lab, roi)
                    _capi.check(hip, hip.rm_get_contour_clip_frame(ctx, ctypes.byref(on)), "get clip")
                    _capi.check(hip, hip.rm_get_contour_labelling(ctx, ctypes.byref(mode)), "get labelling")
                    assert (on.value, mode.value) == (ctx_clip, ctx_mode), "the context's own options must survive a per-call override"
            assert dist.hip_heatmap_to_roi(heat, 20) == plain
    finally:
        hip.rm_set_contour_clip_frame(ctx, 0)
        hip.rm_set_contour_labelling(ctx, -1)


def test_simple_shape_shortcut_equals_border_following(hip, oracle):
    """The host stage's one-blob shortcut on the packed rows (rm_contour.cpp simple_shape_bits_rows) against following every border and
    the oracle: hand-made shapes around the rule, random smooth heatmaps at 1080p / 720p / 4K widths."""
    import torch
    import scipy.ndimage as ndi
    from respmon_amd import device, dist
    from tests.test_emu_calibration import _roi_shapes
    rng = np.random.default_rng(19)
    cases = [(n, a) for n, a in _roi_shapes(rng)]
    for (H, W) in ((270, 1920), (180, 1280), (135, 3840), (97, 1000)):
        for sig in (30.0, 12.0, 4.0):
            cases.append(("smooth %dx%d s%.0f" % (H, W, sig), ndi.gaussian_filter(rng.standard_normal((H, W)), sig)))
        yy, xx = np.mgrid[0:H, 0:W]
        cases.append(("ellipse %dx%d" % (H, W), np.exp(-0.5 * (((yy - 0.6 * H) / (0.1 * H)) ** 2 + ((xx - 0.4 * W) / (0.08 * W)) ** 2))))
    thr = 100
    try:
        for name, img in cases:
            heat = np.ascontiguousarray(img, dtype=np.float64)
            ref_u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min())) if heat.max() > heat.min() else np.zeros(heat.shape, np.uint8)
            want = oracle.roi_from_heatmap_u8(ref_u8, thr)
            dev = torch.from_numpy(heat).cuda()
            device.debug_set("host_simple_shape", 0)    # every border followed on the host
            slow = dist.hip_heatmap_to_roi(dev, thr, labelling=False)
            device.debug_set("host_simple_shape", 1)
            for attempt in range(2):
                short = dist.hip_heatmap_to_roi(dev, thr, labelling=False)
                assert short == slow == want, (name, heat.shape, attempt, short, slow, want)
    finally:
        device.debug_set("host_simple_shape", 1)
