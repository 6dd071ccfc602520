"""Kernel-logic tests WITHOUT a GPU: the product sources in respmon_amd/csrc are compiled with g++
against a host emulation of the HIP subset they use (tests/emu, test infrastructure only) and
driven through the same C-ABI as the real library.  They catch index / border / ordering bugs
before GPU time is spent; the parity tests proper are the `-m gpu` tests."""
import ctypes
import numpy as np
import pytest

from respmon_amd import _capi, synth


@pytest.fixture(scope="module")
def emu():
    from tests.emu_harness import Emu
    return Emu()


def test_emu_pyr_ops_bit_exact(emu, oracle):
    rng = np.random.default_rng(0)
    for shape in [(2, 21, 30), (1, 5, 8), (1, 1, 1), (1, 2, 3), (1, 9, 15), (1, 70, 131)]:
        a = rng.random(shape)
        d = emu.pyr_down(a)
        for t in range(shape[0]):
            assert np.array_equal(d[t], oracle.pyrDown(a[t]))
        u = emu.pyr_up(d, shape[1], shape[2])
        assert np.array_equal(u[0], oracle.pyrUp(d[0], (shape[2], shape[1])))
        lap = emu.pyr_up(d, shape[1], shape[2], mode=1, other=a)
        assert np.array_equal(lap[0], a[0] - oracle.pyrUp(d[0], (shape[2], shape[1])))
    u8 = (rng.random((1, 9, 15)) * 255).astype(np.uint8)
    assert np.array_equal(emu.pyr_down(u8)[0], oracle.pyrDown(oracle.uint8_to_float(u8[0])))
    f32 = rng.random((1, 12, 7)).astype(np.float32)
    assert np.array_equal(emu.pyr_down(f32)[0], oracle.pyrDown(f32[0].astype(np.float64)))


def test_emu_temporal_operator_and_filter(emu, oracle, golden):
    g = golden("g1_temporal_fft.npz")
    for i in range(int(g["ncases"])):
        n, fps, fmin, fmax, amp = g["meta%d" % i]
        n = int(n)
        if n > 256:
            continue
        M, lo, hi = emu.operator(n, fps, fmin, fmax)
        assert (lo, hi) == oracle.band_bounds(n, fps, fmin, fmax)
        if "M%d" % i in g.files:
            assert np.abs(M - g["M%d" % i]).max() < 1e-15
        y = emu.temporal(g["x%d" % i], fps, fmin, fmax, amp)
        ref = g["y%d" % i]
        assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
        # the form large levels take (k_temporal_sym_px: a wave per 16 pixel columns, operator fragments requested a group ahead) and the
        # VALU twin, forced on the same data: both within rounding of the golden as well
        for knob in ("temporal_wide", "temporal_valu"):
            emu.debug_set(knob, 1)
            try:
                y2 = emu.temporal(g["x%d" % i], fps, fmin, fmax, amp)
            finally:
                emu.debug_set(knob, -1 if knob == "temporal_wide" else 0)
            assert np.abs(y2 - ref).max() <= 1e-12 * np.abs(ref).max(), (knob, n)
    # lengths around the group size of the fragment pipeline (4 K-steps of 4 frames: T / 2 + 1 unique frames), odd and tiny ones, wide bands
    rng = np.random.default_rng(12)
    for (T, fps, fmin, fmax) in [(2, 10, 0.1, 6.0), (3, 10, 0.1, 6.0), (7, 10, 0.5, 3.0), (30, 10, 0.1, 1.0), (31, 30, 0.5, 9.0), (33, 10, 0.1, 5.0),
                                 (62, 10, 0.1, 1.0), (64, 10, 0.1, 4.9), (66, 10, 0.3, 1.0), (97, 5.01, 0.1, 1.0), (126, 10, 0.1, 1.0), (130, 10, 0.1, 2.5)]:
        x = rng.standard_normal((T, 37))
        want = oracle.temporal_bandpass_filter_fft(x, fps, freq_min=fmin, freq_max=fmax, amplification_factor=50)
        scale = max(np.abs(want).max(), 1e-300)
        for knob in (None, "temporal_wide", "temporal_valu"):
            if knob:
                emu.debug_set(knob, 1)
            try:
                got = emu.temporal(x, fps, fmin, fmax, 50.0)
            finally:
                if knob:
                    emu.debug_set(knob, -1 if knob == "temporal_wide" else 0)
            assert np.abs(got - want).max() <= 1e-12 * scale, (T, fps, fmin, fmax, knob)


def test_emu_eulerian_and_fused_calibrate_match_golden(emu, golden):
    g = golden("g3_eulerian.npz")
    vid8 = g["vid_u8"]
    for i in range(int(g["ncases"])):
        L, S, fps = g["meta%d" % i]
        masked, raw, mm = emu.eulerian(vid8, fps, 0.1, 1.0, 500.0, int(L), int(S))
        ref = g["raw%d" % i]
        assert np.abs(raw - ref).max() <= 1e-12 * np.abs(ref).max()   # north_star gate: 1e-4 relative
        assert np.allclose(mm, [ref.min(), ref.max()], rtol=1e-12, atol=0)
        heat, mm2 = emu.calibrate(vid8, fps, levels=int(L), skip=int(S))
        avg = g["avg%d" % i]
        assert np.abs(heat - avg).max() <= 1e-12 * np.abs(avg).max()
        assert np.array_equal(mm, mm2)                                 # fused == materialised min/max
        heat_np, mm3 = emu.calibrate(vid8, fps, levels=int(L), skip=int(S), flags=1)
        assert np.array_equal(heat, heat_np) and np.array_equal(mm2, mm3)  # pruning never changes a bit
        # the reference's operation order (Laplacians first, then the temporal filter), per-level launches ...
        heat_us, mm5 = emu.calibrate(vid8, fps, levels=int(L), skip=int(S), flags=16)
        heat_fl, mm6 = emu.calibrate(vid8, fps, levels=int(L), skip=int(S), flags=64)
        assert np.array_equal(heat_fl, heat_us) and np.array_equal(mm6, mm5)  # ... == LDS-resident small pyramid in that order
        # the default filters G_S first (linearity) and builds the Laplacians of the filtered images: equal to rounding
        assert np.abs(heat - heat_us).max() <= 1e-12 * np.abs(heat_us).max() and np.allclose(mm2, mm5, rtol=1e-12, atol=0)
        heat_ts, mm4 = emu.calibrate(vid8, fps, levels=int(L), skip=int(S), flags=4)
        assert np.array_equal(heat, heat_ts) and np.array_equal(mm2, mm4)  # value-store overflow path
        assert np.array_equal(np.average(masked, axis=0), heat)        # fused == materialised heatmap


def test_emu_locate_matches_golden_roi(emu, golden):
    g = golden("g4_locate.npz")
    for i in range(int(g["ncases"])):
        T, H, W, seed, L, S, fps = (int(v) for v in g["meta%d" % i])
        v8 = synth.synth_breathing(T, H, W, seed=seed)
        assert emu.locate(v8, fps, levels=L, skip=S) == tuple(int(v) for v in g["roi%d" % i])
        heat, _ = emu.calibrate(v8, fps, levels=L, skip=S)
        roi, u8, binary = emu.heatmap_to_roi(heat)
        assert np.array_equal(u8, g["avg_u8_%d" % i])
        assert np.array_equal(binary, np.where(u8 > 20, 255, 0).astype(np.uint8))
    assert emu.locate(np.full((16, 40, 48), 0.5), 10, levels=4, skip=2) is None  # base.py:569-570


def test_emu_locate_submit_result(emu, golden):
    """rm_locate_submit / rm_locate_result: two buffers in flight, results fetched in either order, equal rm_locate on every golden
    case; a third submission is refused (RM_E_BUSY); a selection that overflows the value store (RM_FLAG_TINY_STORE) is taken again
    inside rm_locate_result; a flat video reports no contour."""
    g = golden("g4_locate.npz")
    vids, want = [], []
    for i in range(int(g["ncases"])):
        T, H, W, seed, L, S, fps = (int(v) for v in g["meta%d" % i])
        vids.append((synth.synth_breathing(T, H, W, seed=seed), fps, L, S))
        want.append(tuple(int(v) for v in g["roi%d" % i]))
    n = len(vids)
    for i in range(min(n, 3)):
        j = (i + 1) % n
        (va, fa, La, Sa), (vb, fb, Lb, Sb) = vids[i], vids[j]
        ta = emu.locate_submit(va, fa, levels=La, skip=Sa)
        tb = emu.locate_submit(vb, fb, levels=Lb, skip=Sb, flags=4 if i % 2 else 0)
        assert emu.locate_submit(va, fa, levels=La, skip=Sa) == _capi.RM_E_BUSY
        if i % 2:
            assert emu.locate_result(tb) == want[j] and emu.locate_result(ta) == want[i]
        else:
            assert emu.locate_result(ta) == want[i] and emu.locate_result(tb) == want[j]
        assert emu.locate(va, fa, levels=La, skip=Sa) == want[i]     # the synchronous entry between submissions
    # a submission in flight on one stream: work for the same context on ANOTHER stream is refused (the context's workspaces are
    # protected by stream order only), the same call on the submission's stream goes through
    va, fa, La, Sa = vids[0]
    tk = emu.locate_submit(va, fa, levels=La, skip=Sa)
    heat = np.zeros((va.shape[1], va.shape[2])); xy = np.zeros(4, np.int32)
    other = ctypes.c_void_p(0x40)
    rc = emu.lib.rm_heatmap_to_roi(emu.ctx, heat.ctypes.data_as(ctypes.c_void_p), heat.shape[0], heat.shape[1], 20, xy.ctypes.data_as(ctypes.c_void_p), None, None, other)
    assert rc == _capi.RM_E_BUSY and b"in flight" in emu.lib.rm_last_error_string()
    rc = emu.lib.rm_calibrate(emu.ctx, va.ctypes.data_as(ctypes.c_void_p), 0, va.shape[0], va.shape[1], va.shape[2], 10.0, 0.1, 1.0, 500.0, La, Sa, 0.7, 0,
                              heat.ctypes.data_as(ctypes.c_void_p), None, other)
    assert rc == _capi.RM_E_BUSY
    assert emu.locate(va, fa, levels=La, skip=Sa) == want[0]
    assert emu.locate_result(tk) == want[0]
    flat = emu.locate_submit(np.full((16, 40, 48), 0.5), 10, levels=4, skip=2)
    assert emu.locate_result(flat) is None
    # two buffers in flight through the LABELLED ROI stage, lazy (image and component list stay on the device until the summary records
    # leave the winner open): the fetch of the first buffer's image then runs behind the second buffer's kernels -- per-slot buffers
    rng = np.random.default_rng(91)
    noisy = []
    for k in range(6):
        v = rng.random((8, 48, 128))
        if k % 2 == 0:
            v[:, 8:40, 20:100] += 0.6 * np.sin(np.arange(8) * 0.9)[:, None, None]     # a breathing block among the noise: the summaries settle it
        noisy.append(v)
    emu.ck(emu.lib.rm_set_contour_labelling(emu.ctx, 1), "labelling")
    try:
        sync = [emu.locate(v, 10, levels=3, skip=1) for v in noisy]
        paths = set()
        for i in range(len(noisy) - 1):
            ta = emu.locate_submit(noisy[i], 10, levels=3, skip=1)
            tb = emu.locate_submit(noisy[i + 1], 10, levels=3, skip=1)
            assert emu.locate_result(ta) == sync[i]; paths.add(emu.roi_path())
            assert emu.locate_result(tb) == sync[i + 1]; paths.add(emu.roi_path())
        assert emu.contour_stats()[1] == 1
    finally:
        emu.lib.rm_set_contour_labelling(emu.ctx, -1)
    xywh = np.zeros(4, np.int32)
    assert emu.lib.rm_locate_result(emu.ctx, flat[0], xywh.ctypes.data_as(ctypes.c_void_p)) == _capi.RM_E_BADARG   # fetched already


def test_emu_ragged_shapes_and_degenerate_levels(emu, oracle):
    rng = np.random.default_rng(5)
    cases = [(8, 17, 23, 3, 1), (8, 5, 9, 3, 1), (6, 33, 70, 5, 3), (4, 16, 64, 2, 1), (4, 1, 9, 2, 1),
             (5, 30, 31, 4, 3), (4, 20, 20, 1, 0), (4, 20, 20, 2, 0), (6, 40, 130, 4, 1)]
    for (T, H, W, L, S) in cases:
        v = rng.random((T, H, W))
        heat, mm = emu.calibrate(v, 10.0, levels=L, skip=S)
        mo, ro = oracle.eulerian_magnification_bandpass(v, 10.0, 0.1, 1.0, 500.0, pyramid_levels=L, skip_levels_at_top=S)
        ho = np.average(mo, axis=0)
        assert np.abs(heat - ho).max() <= 1e-12 * max(np.abs(ho).max(), 1e-300), (T, H, W, L, S)
        assert np.allclose(mm, [ro.min(), ro.max()], rtol=1e-12, atol=0)


def test_emu_contour_stage_matches_oracle(emu, oracle):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(9)
    shapes = [(37, 61), (37, 61), (37, 61), (20, 64), (9, 128), (33, 130), (40, 7), (5, 200), (64, 65), (3, 63), (1, 70), (31, 1)]
    for k in range(12):   # widths around the 64-pixel words of the bit-packed binary image, rows straddling words
        heat = ndi.gaussian_filter(rng.standard_normal(shapes[k]), 2.5 if min(shapes[k]) > 8 else 0.8)
        if k % 3 == 0:
            heat[:, 0] = heat.max()      # blobs touching the frame
        if k % 4 == 1:
            heat[-1, :] = heat.max(); heat[:, -1] = heat.max()
        roi, u8, binary = emu.heatmap_to_roi(heat, threshold=150)
        ref_u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min()))
        assert np.array_equal(u8, ref_u8)
        assert roi == oracle.roi_from_heatmap_u8(ref_u8, 150)
        # cv2.findContours as OpenCV <= 3.1 did it (1-pixel frame zeroed first): rm_set_contour_clip_frame vs the oracle's twin
        roi_old, _, _ = emu.heatmap_to_roi(heat, threshold=150, clip_frame=True)
        assert roi_old == oracle.roi_from_heatmap_u8(ref_u8, 150, clip_frame=True)
        assert emu.heatmap_to_roi(heat, threshold=150)[0] == roi       # the option does not stick


def test_emu_contour_stage_reuses_its_working_copy(emu, oracle):
    """Same geometry, different images back to back: the host tracer keeps its working copy between calls and
    clears only what the previous image (and its border marks) touched."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(21)
    for k in range(10):
        heat = ndi.gaussian_filter(rng.standard_normal((45, 203)), 1.5 + 0.4 * (k % 4))
        if k == 4:
            heat[:] = 0.0                       # flat heatmap in between: NaN -> no contour
        if k == 7:
            heat[10:30, 60:190] = heat.max() + 1.0   # one big rectangle
        roi, u8, binary = emu.heatmap_to_roi(heat, threshold=120 + 10 * (k % 3))
        with np.errstate(invalid="ignore", divide="ignore"):
            ref_u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min()))
        assert np.array_equal(u8, ref_u8)
        assert roi == oracle.roi_from_heatmap_u8(ref_u8, 120 + 10 * (k % 3)), k


def test_emu_fused_down_chain_equals_per_level(emu):
    """The marching fused pyrDown chain (rm_down_chain.h) must equal the per-level kernel bit for bit,
    for every frame dtype, vector and scalar load paths, and many-strip / many-segment decompositions."""
    rng = np.random.default_rng(3)
    shapes = [(3, 64, 96, 4, 2), (2, 67, 131, 5, 3), (2, 48, 64, 3, 1), (1, 135, 240, 6, 4),
              (9, 32, 48, 4, 2), (2, 33, 47, 4, 2), (1, 200, 320, 7, 5), (1, 40, 800, 4, 2),
              (1, 48, 704, 6, 4), (1, 36, 401, 5, 3), (1, 70, 1936, 6, 4)]
    # (the CPU suite has minutes, not hours: every dtype sees the geometries that differ per dtype -- vector width, strip
    #  count, odd sizes --, float64 sees them all; tests/test_gpu_calibration.py runs the full matrix on the device)
    ncase = 0
    for dt in (np.float64, np.uint8, np.float32, np.float16):
        narrow = shapes[1:2] + shapes[3:4] + shapes[7:] if dt == np.uint8 else shapes[1:2] + shapes[3:4] + shapes[7:10]   # (the 1936-wide case: float64 and uint8)
        for (T, H, W, L, S) in (shapes if dt == np.float64 else narrow):
            v = (rng.random((T, H, W)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((T, H, W)).astype(dt)
            # flags=2: one kernel per pyramid level, the reference's operation order.  With the same order (64) the fused pyrDown
            # chain + LDS-resident small pyramid reproduce it bit for bit, tiny strips (8) included; the default (temporal filter
            # on G_S first, Laplacians of the filtered images -- linearity) agrees to rounding and is strip-independent too
            # (tiny strips: alternately on the reference-order and on the default path -- the strip geometry only shapes G_S)
            ncase += 1
            per_level, _ = emu.calibrate(v, 10.0, levels=L, skip=S, flags=2)
            fused, _ = emu.calibrate(v, 10.0, levels=L, skip=S, flags=64 | (8 if ncase & 1 else 0))
            assert np.array_equal(fused, per_level), (dt, T, H, W, L, S)
            ff, _ = emu.calibrate(v, 10.0, levels=L, skip=S, flags=0 if ncase & 1 else 8)
            assert np.abs(ff - per_level).max() <= 1e-12 * np.abs(per_level).max(), (dt, T, H, W, L, S)
            if dt == np.float64 and ncase <= 3:
                ff_other, _ = emu.calibrate(v, 10.0, levels=L, skip=S, flags=8 if ncase & 1 else 0)
                assert np.array_equal(ff, ff_other), (dt, T, H, W, L, S)


def test_emu_bgr_frame_buffer_equals_its_gray_buffer(emu, oracle):
    """RM_BGR8 ([T,H,W,3] uint8, frames as cv2.VideoCapture delivers them): the calibration applies base.py:230-231 (cvtColor(BGR2GRAY),
    uint8_to_float) while it reads the buffer -- bit-identical to the same call on the gray uint8 buffer the ORACLE's cvtColor makes of it,
    on the fused register chain (W % 16 == 0, depths 1..4, tiny strips), on the whole-buffer conversion in front of the other chains
    (ragged widths, skip 0, the per-level flags), for rm_locate, the two-call form, the frame-sharded stages and the materialising form."""
    rng = np.random.default_rng(41)
    cases = [(3, 64, 96, 4, 2, 0), (2, 67, 131, 5, 3, 0), (1, 135, 240, 6, 4, 0), (1, 40, 1936, 4, 1, 0), (2, 70, 112, 6, 3, 8),
             (2, 33, 48, 3, 0, 0), (2, 48, 64, 4, 2, 2), (1, 36, 401, 5, 3, 64)]
    for (T, H, W, L, S, flags) in cases:
        bgr = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
        gray = np.stack([oracle.cvtColor_bgr2gray(f) for f in bgr])
        want, mm_w = emu.calibrate(gray, 10.0, levels=L, skip=S, flags=flags)
        got, mm_g = emu.calibrate(bgr, 10.0, levels=L, skip=S, flags=flags)
        assert np.array_equal(got, want) and np.array_equal(mm_g, mm_w), (T, H, W, L, S, flags)
        emu.debug_set("bgr_unfused", 1)     # the whole-buffer conversion also where the fused chain applies
        try:
            got2, _ = emu.calibrate(bgr, 10.0, levels=L, skip=S, flags=flags)
        finally:
            emu.debug_set("bgr_unfused", 0)
        assert np.array_equal(got2, want), (T, H, W, L, S, flags)
    # a breathing video whose colour planes differ: ROI through every entry point that takes a frame buffer
    v8 = synth.synth_breathing(24, 96, 160, seed=5)
    bgr = np.stack([np.clip(v8.astype(np.int32) + 12, 0, 255), v8, np.clip(v8.astype(np.int32) - 9, 0, 255)], axis=-1).astype(np.uint8)
    gray = np.stack([oracle.cvtColor_bgr2gray(f) for f in bgr])
    kw = dict(levels=5, skip=2)
    want = oracle.locate(oracle.uint8_to_float(gray), 10.0, pyramid_levels=5, skip_levels_at_top=2)
    assert want is not None
    assert emu.locate(bgr, 10.0, **kw) == want
    tk = emu.locate_submit(bgr, 10.0, **kw)
    assert emu.locate_result(tk) == want
    roi_s, heat_s, _ = emu.locate_sharded(bgr, 3, **kw)
    assert roi_s == want
    m_b, r_b, mm_b = emu.eulerian(bgr[:8], 10.0, 0.1, 1.0, 500.0, 4, 2)
    m_g, r_g, mm_g = emu.eulerian(gray[:8], 10.0, 0.1, 1.0, 500.0, 4, 2)
    assert np.array_equal(m_b, m_g) and np.array_equal(r_b, r_g) and np.array_equal(mm_b, mm_g)
    # gray-only entry points refuse the code
    from tests.emu_harness import ptr
    out = np.empty((1, 32, 48))
    assert emu.lib.rm_pyr_down(emu.ctx, ptr(bgr), _capi.RM_BGR8, 1, 64, 96, ptr(out), None) == _capi.RM_E_BADARG


def test_emu_frame_sharded_stages(emu, oracle):
    """rm_shard_* (Mode A): one rank == rm_calibrate bit for bit; several emulated ranks give the same ROI and
    extrema, the heatmap up to the association of the time sum."""
    cases = [(24, 40, 56, 5, 2, 3), (17, 33, 47, 4, 1, 4)]
    for (T, H, W, L, S, seed) in cases:
        frames = oracle.uint8_to_float(synth.synth_breathing(T, H, W, seed=seed))
        heat_ref, mm_ref = emu.calibrate(frames, 10.0, levels=L, skip=S)
        roi_ref = emu.locate(frames, 10.0, levels=L, skip=S)
        assert roi_ref == oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
        roi1, heat1, mm1 = emu.locate_sharded(frames, 1, levels=L, skip=S)
        assert roi1 == roi_ref and np.array_equal(heat1, heat_ref) and tuple(mm1) == tuple(mm_ref)
        for world in (2, 3):
            roi, heat, mm = emu.locate_sharded(frames, world, levels=L, skip=S)
            assert roi == roi_ref and tuple(mm) == tuple(mm_ref)
            assert np.abs(heat - heat_ref).max() <= 1e-12 * np.abs(heat_ref).max()
    # nothing filtered (skip >= levels - 1): all-zero heatmap, no contour -- same as rm_locate
    frames = oracle.uint8_to_float(synth.synth_breathing(8, 16, 16, seed=1))
    roi, heat, mm = emu.locate_sharded(frames, 2, levels=3, skip=2)
    assert roi is None and not heat.any() and emu.locate(frames, 10.0, levels=3, skip=2) is None


def test_emu_shard_collapse_after_single_level_calibrate(emu, oracle):
    """a calibration that filters ONE level (levels - 2 == skip) used to leave the context believing its
    reduction state was freshly reset; an rm_shard_collapse of a foreign small pyramid on the same context then reduced
    min / max into stale state.  Every public sequence must give the extrema of the buffer it was handed."""
    import ctypes
    from tests.emu_harness import DT, ptr
    lib, ctx = emu.lib, emu.ctx
    a = oracle.uint8_to_float(synth.synth_breathing(20, 40, 56, seed=31))
    b = oracle.uint8_to_float(synth.synth_breathing(20, 40, 56, seed=32)) * 0.5
    L, S = 5, 2
    T, H, W = b.shape
    n = ctypes.c_size_t()
    emu.ck(lib.rm_shard_layout(H, W, L, S, ctypes.byref(n)), "layout")
    NP = int(n.value)
    other = emu.new_ctx()
    lap = np.empty((T, NP))
    emu.ck(lib.rm_shard_pyramid(other, ptr(b), DT[b.dtype], T, H, W, L, S, 0, ptr(lap), None), "shard_pyramid")
    _, mm_ref = emu.calibrate(b, 10.0, levels=L, skip=S)
    for prelude in ("single_level_calibrate", "threshold_mask", "shard_pyramid_then_mask"):
        if prelude == "single_level_calibrate":
            emu.calibrate(a, 10.0, levels=3, skip=1)           # levels - 2 == skip: one filtered level
        elif prelude == "threshold_mask":
            emu.threshold_mask(np.linspace(-900.0, 900.0, 64))
        else:
            scratch = np.empty((T, NP))
            emu.ck(lib.rm_shard_pyramid(ctx, ptr(a), DT[a.dtype], T, H, W, L, S, 0, ptr(scratch), None), "shard_pyramid")
            emu.threshold_mask(np.linspace(-900.0, 900.0, 64))   # dirties the state after front_pyramid reset it
        mm = np.empty(2)
        emu.ck(lib.rm_shard_collapse(ctx, ptr(lap), T, 0, T, H, W, 10.0, 0.1, 1.0, 500.0, L, S, 0.7, 0, ptr(mm), None), "shard_collapse")
        assert (-mm[0], mm[1]) == tuple(mm_ref), prelude
    lib.rm_ctx_destroy(other)


def test_emu_iir_filter_and_threshold_mask(emu, oracle, golden):
    """rm_lfilter == scipy.signal.lfilter in the reference's temporal_bandpass_filter (G8, authentic
    scipy), and rm_threshold_mask == transforms.py:184-192."""
    import scipy.signal
    g = golden("g8_iir.npz")
    for i in range(int(g["ncases"])):
        n, fps, fmin, fmax, amp = g["meta%d" % i]
        b, a = scipy.signal.butter(6, [fmin / (0.5 * fps), fmax / (0.5 * fps)], btype='band', output='ba')
        y = emu.lfilter(b, a, g["x%d" % i], scale=amp)
        ref = g["y%d" % i]
        assert np.abs(y - ref).max() <= 1e-9 * np.abs(ref).max()      # north_star gate: 1e-4 relative
    rng = np.random.default_rng(4)
    for (b, a) in [([0.5], [2.0]), ([1.0, -0.3], [1.0]), ([0.2, 0.1, 0.05], [1.5, -0.4, 0.1]), ([1.0], [1.0, -0.9])]:
        x = rng.standard_normal((50, 7))
        assert np.abs(emu.lfilter(b, a, x) - scipy.signal.lfilter(b, a, x, axis=0)).max() < 1e-12
    raw = rng.standard_normal((5, 9, 11))
    masked, mm = emu.threshold_mask(raw, 0.7)
    top = raw.max() - (raw.max() - raw.min()) * 0.7
    assert tuple(mm) == (raw.min(), raw.max())
    assert np.array_equal(masked, np.where(raw >= top, raw.min(), raw))


def test_emu_sparse_heatmap_exchange(emu, oracle):
    """rm_heat_sparse_pack / rm_heat_sparse_merge_roi (Mode B between GPUs): the fused heatmap is the rank-ordered sum
    of the per-rank heatmaps bit for bit whenever the packets hold their tiles, and the call says so when they do not.
    At sizes the emulator can run the pruning keeps every tile (it needs 1080p-scale footprints to bite), so this
    exercises the packet / index / fallback plumbing; the sparse path proper is covered on the GPU
    (tests/test_gpu_sharded.py, 128 x 1080p)."""
    from respmon_amd import _capi
    rng = np.random.default_rng(8)
    H, W, T = 70, 200, 24            # 5 x 4 tiles, the last row / column of tiles partial

    def video(seed, blob):
        v = np.full((T, H, W), 0.5) + 0.001 * np.random.default_rng(seed).standard_normal((T, H, W))
        y0, x0 = blob
        v[:, y0:y0 + 12, x0:x0 + 30] += 0.2 * np.sin(2 * np.pi * 0.4 * np.arange(T) / 10.0)[:, None, None]
        return v
    vids = [video(1, (10, 20)), video(2, (40, 150)), video(3, (55, 90))]
    rc, roi, fused, heats, counts = emu.sparse_exchange(vids, cap=20, levels=5, skip=2)
    dense = heats[0] + heats[1] + heats[2]
    if rc == _capi.RM_SPARSE_FALLBACK:
        assert max(counts) > 20
    else:
        assert np.array_equal(fused, dense), counts
        u8 = oracle.float_to_uint8((dense - dense.min()) / (dense.max() - dense.min()))
        assert roi == oracle.roi_from_heatmap_u8(u8, 20)
    rc2, _, _, _, counts2 = emu.sparse_exchange(vids, cap=1, levels=5, skip=2)
    assert rc2 == _capi.RM_SPARSE_FALLBACK or max(counts2) <= 1
    # no pruning bookkeeping (skip 0 takes the plain path) -> fallback
    rc3, _, _, _, counts3 = emu.sparse_exchange([v[:6, :20, :24] for v in vids], cap=8, levels=2, skip=0)
    assert rc3 == _capi.RM_SPARSE_FALLBACK


def _handmade_packets(H, W, world, cap, seed, avg_T=0):
    """Packets as rm_heat_sparse_pack lays them out (include/respmon_hip.h) + the dense per-rank heatmaps they stand for."""
    rng = np.random.default_rng(seed)
    tiles_x, tiles_y = (W + 63) // 64, (H + 15) // 16
    pd = 4 + cap + cap * 1024
    packets, dense = np.zeros((world, pd)), []
    for r in range(world):
        bg = float(rng.uniform(0.1, 0.9))
        heat = np.full((H, W), bg)
        tiles = rng.choice(tiles_x * tiles_y, size=min(cap, 1 + r), replace=False)
        packets[r, :1].view(np.uint32)[0] = len(tiles)
        packets[r, 1] = bg
        for j, tl in enumerate(tiles):
            ty, tx = divmod(int(tl), tiles_x)
            vals = np.full((16, 64), bg)
            hh, ww = min(16, H - 16 * ty), min(64, W - 64 * tx)
            vals[:hh, :ww] = rng.uniform(0.0, 2.0, (hh, ww))
            heat[16 * ty:16 * ty + hh, 64 * tx:64 * tx + ww] = vals[:hh, :ww]
            packets[r, 4 + j] = tl
            packets[r, 4 + cap + 1024 * j:4 + cap + 1024 * (j + 1)] = vals.ravel()
        dense.append(heat)
    acc = dense[0].copy()
    for h in dense[1:]:
        acc = acc + h
    if avg_T:
        acc = acc / avg_T
    return packets.ravel(), acc


@pytest.mark.parametrize("H,W,avg_T", [(70, 200, 0), (33, 201, 0), (48, 130, 7)])
def test_emu_sparse_merge_handmade_packets(emu, oracle, H, W, avg_T):
    """k_sparse_index / k_sparse_merge on hand-made packets: tiles nobody sent take the constant path (16-byte stores for
    even W, scalar for odd W), sent tiles the per-pixel one; the result is the rank-ordered sum bit for bit."""
    from tests.emu_harness import ptr
    from respmon_amd import _capi
    world, cap = 3, 6
    allp, want = _handmade_packets(H, W, world, cap, seed=H * W, avg_T=avg_T)
    fused = np.full((H, W), -1.0); xywh = np.zeros(4, np.int32)
    rc = emu.lib.rm_heat_sparse_merge_roi(emu.ctx, ptr(allp), world, H, W, cap, 20, avg_T, ptr(fused), ptr(xywh), None)
    assert rc in (_capi.RM_OK, _capi.RM_NO_CONTOUR)
    assert np.array_equal(fused, want)
    u8 = oracle.float_to_uint8((want - want.min()) / (want.max() - want.min()))
    assert (tuple(int(v) for v in xywh) if rc == _capi.RM_OK else None) == oracle.roi_from_heatmap_u8(u8, 20)
    n = ctypes.c_int(-1)
    assert emu.lib.rm_heat_sparse_tiles_needed(emu.ctx, ctypes.byref(n)) == 0 and n.value == world


def test_emu_error_conventions(emu):
    """include/respmon_hip.h conventions: bad arguments give a negative code plus a message, nothing aborts, and
    unsupported shapes say so (checked on the host-emulated build: the argument checks are host code)."""
    from tests.emu_harness import ptr
    from respmon_amd import _capi
    lib, ctx = emu.lib, emu.ctx
    f = np.zeros((4, 8, 8))
    heat = np.zeros((8, 8)); xywh = np.zeros(4, np.int32); mm = np.zeros(2); pk = np.zeros(64)
    bad = [
        ("rm_calibrate T=0", lib.rm_calibrate(ctx, ptr(f), _capi.RM_F64, 0, 8, 8, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 0, ptr(heat), None, None)),
        ("rm_calibrate NULL frames", lib.rm_calibrate(ctx, None, _capi.RM_F64, 4, 8, 8, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 0, ptr(heat), None, None)),
        ("rm_calibrate dtype", lib.rm_calibrate(ctx, ptr(f), 9, 4, 8, 8, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 0, ptr(heat), None, None)),
        ("rm_calibrate fps", lib.rm_calibrate(ctx, ptr(f), _capi.RM_F64, 4, 8, 8, 0.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 0, ptr(heat), None, None)),
        ("rm_locate NULL out", lib.rm_locate(ctx, ptr(f), _capi.RM_F64, 4, 8, 8, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 20, 0, None, None)),
        ("rm_pyr_up dstsize", lib.rm_pyr_up(ctx, ptr(f), 4, 8, 8, ptr(np.zeros((4, 20, 20))), 20, 20, 0, None, None)),
        ("rm_pyr_up mode", lib.rm_pyr_up(ctx, ptr(f), 4, 8, 8, ptr(np.zeros((4, 16, 16))), 16, 16, 5, None, None)),
        ("rm_heatmap_to_roi NULL", lib.rm_heatmap_to_roi(ctx, None, 8, 8, 20, ptr(xywh), None, None, None)),
        ("rm_roi_mean outside", lib.rm_roi_mean(ctx, ptr(heat), _capi.RM_F64, 8, 8, 4, 4, 8, 8, ptr(mm), None)),
        ("rm_temporal in place", lib.rm_temporal_bandpass_filter_fft(ctx, ptr(f), 4, 64, 10.0, 0.1, 1.0, 50.0, ptr(f), None)),
        ("rm_lfilter a0", lib.rm_lfilter(ctx, ptr(f), 4, 64, ptr(np.ones(2)), ptr(np.zeros(2)), 2, 1.0, ptr(np.zeros((4, 8, 8))), None)),
        ("rm_lfilter ncoef", lib.rm_lfilter(ctx, ptr(f), 4, 64, ptr(np.ones(40)), ptr(np.ones(40)), 40, 1.0, ptr(np.zeros((4, 8, 8))), None)),
        ("rm_shard_pyramid skip 0", lib.rm_shard_pyramid(ctx, ptr(f), _capi.RM_F64, 4, 8, 8, 3, 0, 0, ptr(np.zeros((4, 64))), None)),
        ("rm_shard_collapse range", lib.rm_shard_collapse(ctx, ptr(np.zeros((4, 16))), 4, 3, 2, 8, 8, 10.0, 0.1, 1.0, 500.0, 3, 1, 0.7, 0, ptr(mm), None)),
        ("rm_heat_sparse_pack cap", lib.rm_heat_sparse_pack(ctx, ptr(heat), 8, 8, 0, ptr(pk), None)),
        ("rm_threshold_mask n=0", lib.rm_threshold_mask(ctx, ptr(heat), 0, 0.7, ptr(heat), ptr(mm), None)),
        ("rm_ctx_create NULL", lib.rm_ctx_create(0, None)),
    ]
    for what, rc in bad:
        assert rc < 0, what
        assert len(lib.rm_last_error_string()) > 0, what
    # rm_shard_heat before any rm_shard_collapse on a fresh context
    c2 = emu.new_ctx()
    assert lib.rm_shard_heat(c2, ptr(mm), 0.7, ptr(heat), None) < 0 and b"rm_shard_collapse" in lib.rm_last_error_string()
    lib.rm_ctx_destroy(c2)
    assert lib.rm_ctx_destroy(None) == 0                      # destroying nothing is not an error
    # a flat buffer: NaN after normalisation -> no contour is a RESULT (positive code), not an error (base.py:569-570)
    flat = np.full((8, 16, 16), 0.25)
    assert lib.rm_locate(ctx, ptr(flat), _capi.RM_F64, 8, 16, 16, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 20, 0, ptr(xywh), None) == _capi.RM_NO_CONTOUR
    # T beyond the supported maximum is reported, not truncated
    assert lib.rm_calibrate(ctx, ptr(f), _capi.RM_F64, 5000, 8, 8, 10.0, 0.1, 1.0, 500.0, 4, 2, 0.7, 0, ptr(heat), None, None) == -4


def test_emu_contour_stage_many_components_and_nesting(emu, oracle):
    """The run-based raster scan of the host contour stage (rm_contour.cpp scan_runs) on images with thousands of
    components, holes, and components nested inside holes (RETR_EXTERNAL must skip those), against the oracle's
    independent findContours restatement."""
    rng = np.random.default_rng(33)
    masks = []
    for (h, w, dens) in [(40, 131, 0.15), (64, 64, 0.35), (57, 200, 0.5), (33, 129, 0.62), (20, 70, 0.05), (48, 192, 0.45)]:
        masks.append(rng.random((h, w)) < dens)
    ring = np.zeros((60, 150), bool)                      # ring with an island in its hole, twice nested, plus noise outside
    ring[5:55, 10:90] = True; ring[12:48, 20:80] = False; ring[20:40, 30:70] = True; ring[25:35, 40:60] = False
    ring[28:32, 45:55] = True
    ring[:, 100:] = rng.random((60, 50)) < 0.3
    masks.append(ring)
    full = np.ones((17, 66), bool); full[8, 33] = False   # everything foreground but one pixel; touches all four frame edges
    masks.append(full)
    for m in masks:
        heat = m.astype(np.float64)
        roi, u8, binary = emu.heatmap_to_roi(heat, threshold=20)
        assert np.array_equal(binary != 0, m)
        assert roi == oracle.roi_from_heatmap_u8(np.where(m, 255, 0).astype(np.uint8), 20), m.shape


def test_emu_banded_tile_bounds(emu, oracle):
    """k_frame_bounds in bands of tile rows (what 4K levels need: their row-extrema table exceeds LDS): forced here with a
    tiny table budget; the heatmap must not change by a bit."""
    v = oracle.uint8_to_float(synth.synth_breathing(12, 150, 200, seed=17))
    for (L, S) in [(5, 2), (4, 1), (6, 3)]:
        emu.debug_set("bounds_table_bytes", 0)
        emu.debug_set("no_fused_bounds", 0)
        ref, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=64)   # reference order: bounds taken inside the small-collapse kernel
        emu.debug_set("no_fused_bounds", 1)                            # the separate k_frame_bounds, whole frame at once
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=64)
        assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (L, S)
        for budget in ((600, 1) if L == 5 else (2000,)):             # a few tile rows per band ... one tile row per band (minimum)
            emu.debug_set("bounds_table_bytes", budget)
            got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=64)
            assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (L, S, budget)
    emu.debug_set("bounds_table_bytes", 0)
    # wide levels (>= 256 columns) take k_frame_bounds_rows -- a wave per level-S row, coalesced loads, skewed LDS row buffers --:
    # same bounds as the thread-per-(row, tile column) kernel, in whole frames and in bands, ragged widths and heights
    rng = np.random.default_rng(23)
    for ci, (T, H, W, L, S, budget) in enumerate([(3, 24, 1100, 4, 2, 0), (2, 37, 1100, 4, 2, 9000), (3, 21, 600, 3, 1, 0), (1, 34, 2100, 5, 3, 0)]):
        v2 = rng.random((T, H, W))
        emu.debug_set("bounds_table_bytes", budget)
        emu.debug_set("bounds_scalar", 1)
        ref, mm = emu.calibrate(v2, 10.0, levels=L, skip=S, flags=64)
        emu.debug_set("bounds_scalar", 2)                    # (2: the streaming kernel whatever the image size)
        got, mm2 = emu.calibrate(v2, 10.0, levels=L, skip=S, flags=64)
        emu.debug_set("bounds_scalar", 0)
        assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (T, H, W, L, S)
        if ci == 0:
            exhaustive, mm3 = emu.calibrate(v2, 10.0, levels=L, skip=S, flags=64 | 1)
            assert np.array_equal(got, exhaustive) and tuple(mm) == tuple(mm3), (T, H, W, L, S, "no prune")
    emu.debug_set("bounds_table_bytes", 0)
    emu.debug_set("no_fused_bounds", 0)


def test_emu_dense_sum_equals_sparse_path(emu):
    """rm_dense_sum.h: the masked time sum that recomputes every (tile, frame) pair frame after frame (flags=128) against the
    selection / value-store path (flags=256), bit for bit -- every super-tile shape (RM_DENSE_ROWS), skip 1..5, shards of the
    frame range, exhaustive evaluation (flags | 1), and the automatic choice (second call of a geometry that kept every pair)."""
    rng = np.random.default_rng(5)
    emu.debug_set("xs", 0)            # (the store-less kernels themselves: the exception store that stands in front of them has its own test below)
    for n, (T, H, W, L, S) in enumerate([(4, 64, 96, 4, 2), (3, 67, 131, 5, 3), (3, 48, 64, 3, 1), (2, 100, 160, 7, 5),
                                         (3, 70, 300, 4, 2)]):     # (the GPU twin in tests/test_gpu_calibration.py runs more and larger ones)
        v = rng.random((T, H, W))
        emu.debug_set("dense_rows", 0)
        sparse, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
        for rows in ((16, 32, 64) if n < 2 else ((16, 32, 64)[n % 3],)):
            emu.debug_set("dense_rows", rows)
            dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
            assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, rows)
        emu.debug_set("dense_rows", 0)                              # skip <= 2: the wave-private kernel (k_dense_sum_w)
        dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
        assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "wave-private tiles")
        dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128 | 1)
        assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "no_prune")
        emu.debug_set("dense_wave", 0)
        dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
        emu.debug_set("dense_wave", 1)
        assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "workgroup kernel, automatic shape")
        if S <= 2:      # skip <= 2 takes the table-driven kernel: the general one must agree there too
            emu.debug_set("dense_general", 1)
            dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
            emu.debug_set("dense_general", 0)
            assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "general kernel")
            emu.debug_set("dense_t_low", 1)                         # ... and the TileEval kernel of the deeper chains (taken on large frames)
            dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
            emu.debug_set("dense_t_low", -1)
            assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_dense_sum_t at skip <= 2")
        auto, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)        # decided on the device from this call's own selection
        assert np.array_equal(auto, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "auto")
        tiny, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=4)   # an 8-slot value store overflows: the dense kernel takes over
        assert np.array_equal(tiny, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "store overflow")
        tiny, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=4 | 256)
        assert np.array_equal(tiny, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "store overflow, sparse asked for")
    emu.debug_set("dense_rows", 0)
    # k_dense_sum_w on ragged geometries: widths / heights that are not multiples of the 64 x 16 tiles, odd level sizes (the
    # virtual rows / columns of the footprints meet every border rule), single-tile images, levels of 2 rows
    for (T, H, W, L, S) in [(3, 33, 70, 4, 2), (4, 17, 129, 3, 1), (2, 5, 7, 4, 2), (3, 31, 193, 4, 2), (2, 8, 8, 4, 2),
                            (2, 3, 3, 3, 1), (3, 47, 65, 3, 2)]:
        v = rng.random((T, H, W))
        sparse, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
        emu.debug_set("dense_split", 1)                             # (one wave per tile: k_dense_sum_w)
        for fr in (1, 2):                                           # one frame per trip / two interleaved
            emu.debug_set("dense_frames", fr)
            dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
            assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, fr, "wave-private tiles, ragged")
        emu.debug_set("dense_frames", 0)
        for nw in ((2, 4) if (T, H, W) in ((3, 33, 70), (4, 17, 129), (5, 50, 66), (3, 47, 65)) else ()):   # NW waves per tile, each evaluating every NW-th frame
            emu.debug_set("dense_split", nw)
            dense, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=128)
            assert np.array_equal(dense, sparse) and tuple(mm) == tuple(mm2), (T, H, W, L, S, nw, "frame-split tiles, ragged")
        emu.debug_set("dense_split", 0)
    # rm_locate leaves the overflow rescue to the host: the sparse kernel stands down, the ROI stage's synchronisation finds the
    # pinned word set, the dense kernel is enqueued then and the ROI extracted again (skip 3: dense is never the automatic choice)
    v = synth.synth_breathing(10, 120, 160, seed=5).astype(np.float64) / 255
    want = emu.locate(v, 10.0, levels=5, skip=3)
    emu.debug_set("store_slots", 2)
    got = emu.locate(v, 10.0, levels=5, skip=3)
    emu.debug_set("store_slots", 0)
    assert want is not None and got == want
    v = rng.random((11, 70, 150))
    for world in (2, 3):                      # frame shards: partial sums over [t0, t1) with the global extrema
        a = emu.locate_sharded(v, world, levels=4, skip=2, flags=256)
        b = emu.locate_sharded(v, world, levels=4, skip=2, flags=128)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), world
    emu.debug_set("xs", 1)


def test_emu_filter_first_per_level_equals_fused(emu):
    """The filter-first small pyramid with one launch per level (what levels too large for LDS take: 4K, skip 2) computes what
    k_small_filter_first computes, bit for bit; both agree with the reference's operation order (flags=2) to rounding."""
    rng = np.random.default_rng(7)
    for (T, H, W, L, S) in [(5, 64, 96, 4, 2), (3, 67, 131, 5, 3), (3, 135, 130, 6, 4), (4, 48, 64, 3, 1), (2, 100, 160, 7, 5), (9, 33, 47, 4, 2)]:
        v = rng.random((T, H, W))
        fused, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
        ref, _ = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256 | 2)
        per_level, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256 | 512)
        assert np.array_equal(per_level, fused) and tuple(mm) == tuple(mm2), (T, H, W, L, S)
        assert np.abs(fused - ref).max() <= 1e-12 * np.abs(ref).max(), (T, H, W, L, S)


def test_emu_contour_stage_device_labelling(emu, oracle):
    """rm_ccl.h: the device labels the 8-connected components of the thresholded image and the host follows only the borders
    whose bounding-box bound can win.  Same ROI as following every border (mode 0) and as the oracle's findContours
    restatement: noise at several densities, widths that are not multiples of the 64-pixel words (runs must not cross row
    ends), nested components, equal-area ties (the LAST discovered border wins), zero-area components only, empty image."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(77)
    masks = []
    for (h, w, dens) in [(40, 131, 0.15), (64, 64, 0.35), (57, 200, 0.5), (33, 129, 0.62), (20, 70, 0.05), (48, 192, 0.45),
                         (31, 63, 0.3), (9, 257, 0.4), (70, 5, 0.5), (1, 90, 0.5), (50, 1, 0.5), (2, 2, 1.0)]:
        masks.append(rng.random((h, w)) < dens)
    for k in range(6):   # a blob in noise: what configs 2 / 5 look like
        m = rng.random((60, 150)) < 0.16
        blob = ndi.gaussian_filter(rng.standard_normal((60, 150)), 4.0) > 0.06
        masks.append(m | blob)
    ring = np.zeros((60, 150), bool)
    ring[5:55, 10:90] = True; ring[12:48, 20:80] = False; ring[20:40, 30:70] = True; ring[25:35, 40:60] = False
    ring[28:32, 45:55] = True
    ring[:, 100:] = rng.random((60, 50)) < 0.3
    masks.append(ring)
    ties = np.zeros((40, 100), bool)          # three 6x6 squares (equal areas) and a diagonal line with a larger box, area 0
    for (y, x) in [(3, 5), (3, 40), (20, 22)]:
        ties[y:y + 6, x:x + 6] = True
    ties[np.arange(10, 30), np.arange(60, 80)] = True
    masks.append(ties)
    dots = np.zeros((20, 70), bool); dots[::3, ::5] = True     # isolated pixels only: every area is 0
    masks.append(dots)
    lines = np.zeros((20, 70), bool); lines[4, 3:40] = True; lines[9, 10:66] = True; lines[12:19, 68] = True
    masks.append(lines)
    full = np.ones((17, 66), bool); full[8, 33] = False
    masks.append(full)
    masks.append(np.zeros((12, 40), bool))
    # the area bound of the labelled path (rm_ccl.h ccl_piece_2n_minus_p): components whose BOX is the largest but whose contour
    # encloses little -- a 1-pixel U, a spiral, a frame-shaped ring around everything, a comb -- next to solid rivals that win or lose;
    # pieces that end exactly at word / row ends; a solid blob that spans the frame among specks (the shortcut must fire there)
    u_shape = np.zeros((50, 140), bool); u_shape[5:45, 10] = True; u_shape[5:45, 60] = True; u_shape[44, 10:61] = True; u_shape[10:22, 80:95] = True
    masks.append(u_shape)
    spiral = np.zeros((64, 192), bool)
    for k, (a, b) in enumerate([(2, 60), (6, 56), (10, 52), (14, 48)]):
        spiral[a, a:190 - a] = True; spiral[b, a + 4:190 - a] = True; spiral[a:b + 1, 189 - a] = True; spiral[a + 4:b + 1, a + 4] = True
    spiral[20:44, 60:130] = True
    masks.append(spiral)
    comb = np.zeros((40, 128), bool); comb[2, :] = True; comb[2:38, ::2] = True; comb[30:39, 1:20] = True
    masks.append(comb)
    frame_ring = np.zeros((48, 130), bool); frame_ring[0, :] = True; frame_ring[-1, :] = True; frame_ring[:, 0] = True; frame_ring[:, -1] = True
    frame_ring[10:30, 20:90] = True
    masks.append(frame_ring)
    wordends = np.zeros((6, 128), bool); wordends[1, 60:64] = True; wordends[2, 64:70] = True; wordends[3, 0:128] = True; wordends[4, 127] = True; wordends[5, 0] = True
    masks.append(wordends)
    fire = []
    for k in range(4):
        m = rng.random((72, 200)) < 0.1
        m |= ndi.gaussian_filter(rng.standard_normal((72, 200)), 9.0) > -0.02
        masks.append(m); fire.append(len(masks) - 1)
    # rows of whole words taller than one 64 x 32 tile (rm_ccl.h k_ccl_tile / k_ccl_seam / k_ccl_fold): specks, blobs that cross the
    # seams in every direction, a tile holding as many one-pixel components as it can
    for (h, w, dens) in [(70, 128, 0.2), (100, 256, 0.45), (33, 192, 0.6), (65, 64, 0.3), (96, 128, 0.03), (1, 128, 0.5), (7, 64, 0.6), (32, 64, 0.97)]:
        masks.append(rng.random((h, w)) < dens)
    for k in range(3):
        m = rng.random((90, 192)) < 0.12
        masks.append(m | (ndi.gaussian_filter(rng.standard_normal((90, 192)), 5.0) > 0.04))
    dots64 = np.zeros((66, 128), bool); dots64[::2, ::2] = True
    masks.append(dots64)
    diag = np.zeros((80, 192), bool); diag[np.arange(80), np.arange(80) + 30] = True; diag[np.arange(80), 150 - np.arange(80)] = True
    masks.append(diag)
    snake = np.zeros((70, 128), bool); snake[::4, :] = True; snake[2::8, 127] = True; snake[1::8, 127] = True; snake[3::8, 127] = True
    snake[6::8, 0] = True; snake[5::8, 0] = True; snake[7::8, 0] = True
    masks.append(snake)
    n_fired = 0
    for im, m in enumerate(masks):
        heat = m.astype(np.float64)
        if not m.any() or m.all():
            heat = heat + 0.0   # flat heatmap: NaN normalisation -> nothing above the threshold
        emu.debug_set("host_area_bound", 0)      # the top component's border always followed
        try:
            roi_f, _, _ = emu.heatmap_to_roi(heat, threshold=20, labelling=1)
            path_f = emu.roi_path()
        finally:
            emu.debug_set("host_area_bound", 1)
        roi_l, u8, binary = emu.heatmap_to_roi(heat, threshold=20, labelling=1)
        path_l = emu.roi_path()
        assert roi_l == roi_f and path_f != 4, (m.shape, roi_l, roi_f, path_f)
        emu.debug_set("ccl_tiles", 0)   # rows of whole words through the global-memory kernels too
        for table in (0, 1):     # k_ccl_bbox without / with its per-tile LDS table (the default picks by the last component count)
            emu.debug_set("ccl_table", table)
            try:
                assert emu.heatmap_to_roi(heat, threshold=20, labelling=1)[0] == roi_l and emu.roi_path() == path_l, (m.shape, table)
            finally:
                emu.debug_set("ccl_table", -1)
        n_flat, _ = emu.contour_stats()
        emu.debug_set("ccl_tiles", 1)
        assert emu.heatmap_to_roi(heat, threshold=20, labelling=1)[0] == roi_l and emu.contour_stats()[0] == n_flat, (m.shape, "components counted")
        n_fired += path_l == 4
        if im in fire:
            assert path_l == 4, (im, path_l)
        n_l, used = emu.contour_stats()
        roi_h, _, _ = emu.heatmap_to_roi(heat, threshold=20, labelling=0)
        n_h, used_h = emu.contour_stats()
        assert used == 1 and used_h == 0
        assert roi_l == roi_h, (m.shape, roi_l, roi_h)
        if m.any() and not m.all():
            assert np.array_equal(binary != 0, m)
            assert roi_l == oracle.roi_from_heatmap_u8(np.where(m, 255, 0).astype(np.uint8), 20), m.shape
            import scipy.ndimage as ndi2
            assert n_l == ndi2.label(m, structure=np.ones((3, 3)))[1]      # every 8-connected component has one record
            assert n_h <= n_l                                              # RETR_EXTERNAL skips nested components
    assert n_fired >= len(fire)
    # the lazy stage: once the summary records alone settled an extraction, the next one of that geometry keeps the packed image and the
    # component list on the device -- and fetches them when ITS summaries leave the winner open (two rivals whose boxes beat the top's bound),
    # when the list overflows is covered on the GPU; the eager stage (label_lazy 0) must agree everywhere
    settled_img = masks[fire[0]]
    open_img = np.zeros_like(settled_img); open_img[5:35, 10:60] = True; open_img[40:70, 100:152] = True; open_img[::7, ::9] = True
    empty_img = np.zeros_like(settled_img)
    for seq in ([settled_img, settled_img, open_img, settled_img, empty_img, settled_img], [open_img, settled_img, open_img, open_img]):
        got, paths = [], []
        for lazy in (1, 0):
            emu.debug_set("label_lazy", lazy)
            try:
                rois, pp = [], []
                for m in seq:
                    rois.append(emu.heatmap_to_roi(m.astype(np.float64), threshold=20, labelling=1)[0]); pp.append(emu.roi_path())
            finally:
                emu.debug_set("label_lazy", 1)
            got.append(rois); paths.append(pp)
        assert got[0] == got[1] and paths[0] == paths[1], (got, paths)
        for m, roi in zip(seq, got[0]):
            assert roi == (oracle.roi_from_heatmap_u8(np.where(m, 255, 0).astype(np.uint8), 20) if m.any() else None)
        assert 4 in paths[0] and 3 in paths[0]
    # the bound itself, to the unit: a solid a x b rectangle has 2 N - P - 2 = 2 (a - 1)(b - 1) - 4, and the shortcut fires exactly when
    # that exceeds a rival's box bound 2 (c - 1)(d - 1) -- rectangles across word ends, at the frame, in rows that are no whole words
    for (W_, x0) in ((200, 58), (130, 0), (192, 120), (77, 60)):
        for (c, d, fires) in ((4, 30, True), (9, 12, False)):        # rival bounds 174 < 176 and 176 >= 176 against a 10 x 11 top
            m = np.zeros((48, W_), bool)
            m[3:13, x0:x0 + 11] = True                                # 10 rows x 11 columns: twice its area 180, lower bound 176
            m[20:20 + c, 5:5 + d] = True
            heat = m.astype(np.float64)
            for table in (-1, 0, 1):
                emu.debug_set("ccl_table", table)
                try:
                    roi = emu.heatmap_to_roi(heat, threshold=20, labelling=1)[0]
                finally:
                    emu.debug_set("ccl_table", -1)
                assert (emu.roi_path() == 4) == fires and roi == (x0, 3, 11, 10), (W_, x0, c, d, table, emu.roi_path(), roi)
    # the automatic rule: a geometry whose last extraction met many components switches to the labelled path
    noisy = (rng.random((64, 256)) < 0.2).astype(np.float64)
    emu.heatmap_to_roi(noisy, threshold=20)
    assert emu.contour_stats()[1] == 0 and emu.contour_stats()[0] > 512
    r2 = emu.heatmap_to_roi(noisy, threshold=20)[0]
    assert emu.contour_stats()[1] == 1
    assert r2 == emu.heatmap_to_roi(noisy, threshold=20, labelling=0)[0]


def test_emu_fused_collapse_equals_store_path(emu, oracle):
    """rm_tile_eval.h (round 4): the store-less collapse passes -- k_eval_c (exact extrema from the C pairs) + k_tile_sum (every kept
    pair evaluated where it is summed, tile by tile, whole tiles or half tiles) -- against the selection / value-store path
    (flags=256), bit for bit: skip 1..4, ragged geometries whose virtual footprints meet every border rule (top row, rows past the
    bottom, left / right columns, 2-row / 2-column levels), exhaustive evaluation, frame shards, and the oracle's ROI."""
    rng = np.random.default_rng(11)
    cases = [(3, 64, 96, 6, 4), (3, 67, 131, 5, 3), (3, 48, 64, 3, 1), (3, 70, 130, 4, 2), (2, 31, 193, 7, 4)]   # (the GPU twin of this test, tests/test_gpu_calibration.py, runs more and larger geometries)
    try:
        for (T, H, W, L, S) in cases:
            v = rng.random((T, H, W))
            emu.debug_set("collapse_fused", 0)
            emu.debug_set("eval_fast", 0)            # the generic chain in LDS (k_eval_pairs): the reference of both newer forms
            emu.debug_set("sum_sym", 0)              # ... and the sum that fetches every visit of a frame (k_masked_sum_tiles)
            emu.debug_set("sum_rows", 0)
            store, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            emu.debug_set("sum_rows", 1)             # one wave per (tile, row), unique frames staged through LDS (k_masked_sum_rows)
            rows, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            assert np.array_equal(rows, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_masked_sum_rows")
            emu.debug_set("sum_rows", 0)
            emu.debug_set("sum_sym", 1)              # unique frames loaded once, added on the way up and down (k_masked_sum_sym)
            sym, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            assert np.array_equal(sym, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_masked_sum_sym")
            emu.debug_set("sum_sym", 0)
            tiny, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=4)   # an 8-slot store overflows: k_tile_sum stands in at skip >= 3
            assert np.array_equal(tiny, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "store overflow")
            emu.debug_set("eval_fast", 1)            # the same flat pass with the wave-private evaluator (k_eval_pairs_fast)
            fast, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            assert np.array_equal(fast, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_eval_pairs_fast")
            fast, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256 | 1)
            assert np.array_equal(fast, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_eval_pairs_fast, no_prune")
            emu.debug_set("collapse_fused", 1)
            emu.debug_set("dense_tiles", 1)          # one wave per tile, frame after frame (k_dense_sum_t)
            dense_t, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)
            assert np.array_equal(dense_t, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_dense_sum_t")
            emu.debug_set("dense_tiles", 0)          # rounds of sixteen waves per tile (k_tile_sum)
            for half in (0, 1):     # whole-tile / half-tile work items (chosen by the number of heavy tiles otherwise)
                emu.debug_set("tile_sum_half", half)
                fused, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)
                assert np.array_equal(fused, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "fused", half)
            emu.debug_set("tile_sum_half", -1)
            fused, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=1)
            assert np.array_equal(fused, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "fused, no_prune")
            emu.debug_set("collapse_fused", 0)
            auto, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)
            assert np.array_equal(auto, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "default")
        # more kept unique frames than one batch of k_masked_sum_sym holds (48): several batches up, the same ones down again
        for (T, H, W, L, S) in [(101, 20, 70, 5, 3)]:
            v = rng.random((T, H, W))
            emu.debug_set("collapse_fused", 0)
            emu.debug_set("sum_sym", 0)
            emu.debug_set("sum_rows", 0)
            store, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            emu.debug_set("sum_sym", 1)
            sym, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            emu.debug_set("sum_sym", 0)
            assert np.array_equal(sym, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_masked_sum_sym, batches")
            emu.debug_set("sum_rows", 1)
            rows, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=256)
            assert np.array_equal(rows, store) and tuple(mm) == tuple(mm2), (T, H, W, L, S, "k_masked_sum_rows, chunks")
        emu.debug_set("sum_rows", 0)
        # a breathing video: few heavy tiles (half-tile work items), pruned pairs in between, and the oracle's ROI
        from respmon_amd import synth
        v8 = synth.synth_breathing(16, 80, 128, seed=3)
        fr = oracle.uint8_to_float(v8)
        for (L, S) in [(6, 4)]:
            emu.debug_set("collapse_fused", 0)
            store, mm = emu.calibrate(fr, 10.0, levels=L, skip=S, flags=256)
            emu.debug_set("collapse_fused", 1)
            emu.debug_set("dense_tiles", 1)
            dense_t, mm2 = emu.calibrate(fr, 10.0, levels=L, skip=S)
            assert np.array_equal(dense_t, store) and tuple(mm) == tuple(mm2), (L, S, "breathing video, k_dense_sum_t")
            emu.debug_set("dense_tiles", 0)
            for half in (0, 1):
                emu.debug_set("tile_sum_half", half)
                fused, mm2 = emu.calibrate(fr, 10.0, levels=L, skip=S)
                assert np.array_equal(fused, store) and tuple(mm) == tuple(mm2), (L, S, "breathing video", half)
            emu.debug_set("tile_sum_half", -1)
            assert emu.locate(fr, 10.0, levels=L, skip=S) == oracle.locate(fr, 10, pyramid_levels=L, skip_levels_at_top=S)
            # frame shards: partial time sums of the store-less path equal the store path's
            for world in (3,):
                emu.debug_set("collapse_fused", 0)
                r0, h0, m0 = emu.locate_sharded(fr, world, levels=L, skip=S, flags=256)
                emu.debug_set("collapse_fused", 1)
                r1, h1, m1 = emu.locate_sharded(fr, world, levels=L, skip=S)
                assert r0 == r1 and np.array_equal(h0, h1) and tuple(m0) == tuple(m1), (L, S, world, "sharded")
    finally:
        emu.debug_set("collapse_fused", 0)
        emu.debug_set("tile_sum_half", -1)
        emu.debug_set("eval_fast", 1)
        emu.debug_set("sum_sym", 0)
        emu.debug_set("sum_rows", 0)
        emu.debug_set("dense_tiles", 1)


def _roi_shapes(rng):
    """Thresholded-image shapes around the simple-shape rule of the host contour stage (simple_shape_bits_rows): one blob, a ring (two runs per row),
    two blobs side by side / one above the other with a gap, runs that touch diagonally, runs that miss each other by a pixel, a
    run crossing the 64-pixel words, a blob on the image frame, a single pixel, nothing, noise."""
    import scipy.ndimage as ndi
    out = []
    for (H, W) in ((24, 64), (33, 128), (40, 192), (30, 100)):
        z = np.zeros((H, W))
        a = z.copy(); a[5:15, 10:50] = 1; out.append(("rectangle", a))
        a = z.copy(); a[4:20, 8:40] = 1; a[8:16, 16:32] = 0; out.append(("ring", a))
        a = z.copy(); a[5:15, 4:20] = 1; a[5:15, 30:60] = 1; out.append(("two side by side", a))
        a = z.copy(); a[2:8, 10:30] = 1; a[12:20, 10:30] = 1; out.append(("two, one above the other", a))
        a = z.copy()
        for y in range(3, 20):
            a[y, 2 * y:2 * y + 2] = 1
        out.append(("diagonal staircase (runs touch diagonally)", a))
        a = z.copy()
        for y in range(3, 20):
            a[y, 3 * y:3 * y + 2] = 1
        out.append(("broken staircase (runs miss by a pixel)", a))
        a = z.copy(); a[6:10, 60:W - 3] = 1; out.append(("run across word boundaries", a))
        a = z.copy(); a[0:6, 0:30] = 1; out.append(("on the frame", a))
        a = z.copy(); a[H - 1, W - 1] = 1; out.append(("single pixel in the corner", a))
        a = z.copy(); a[7, 5:W - 5] = 1; out.append(("one row", a))
        out.append(("nothing", z.copy() + 0.0))
        a = z.copy(); a[3:12, 5:25] = 1; a[8, 10:20] = 0; a[9, 5:12] = 0; out.append(("notched blob", a))
        out.append(("smooth random", (ndi.gaussian_filter(rng.standard_normal((H, W)), 4.0) > 0.05).astype(float)))
        out.append(("noise", (rng.random((H, W)) > 0.6).astype(float)))
        g = np.exp(-0.5 * (((np.arange(H)[:, None] - H * 0.55) / (H * 0.2)) ** 2 + ((np.arange(W)[None, :] - W * 0.4) / (W * 0.15)) ** 2))
        out.append(("gaussian blob", g))
    return out


def test_emu_simple_shape_shortcut_equals_border_following(emu, oracle):
    """The host stage's one-blob shortcut on the packed rows (rm_contour.cpp simple_shape_bits_rows: one run per row, neighbouring runs
    touching -> one hole-free component, its bounding box is the ROI) against following every border and the oracle, shape by shape."""
    rng = np.random.default_rng(17)
    try:
        for name, img in _roi_shapes(rng):
            heat = img * 1.0
            thr = 100
            ref_u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min())) if heat.max() > heat.min() else np.zeros(heat.shape, np.uint8)
            want = oracle.roi_from_heatmap_u8(ref_u8, thr)
            emu.debug_set("host_simple_shape", 0)       # every border followed on the host
            slow, u8s, _ = emu.heatmap_to_roi(heat, threshold=thr)
            emu.debug_set("host_simple_shape", 1)
            # heat_rows 1: a workgroup per image row + one record per row (k_heat_rows_u8, rows of whole words only); 0: the flat kernel
            # + row flags (k_heat_to_u8).  Twice each: the second extraction finds what the first one left in the pinned areas
            for rows_mode in (1, 0, 1):
                emu.debug_set("heat_rows", rows_mode)
                for attempt in range(2):
                    short, u8f, binf = emu.heatmap_to_roi(heat, threshold=thr)
                    assert short == slow == want, (name, heat.shape, rows_mode, attempt, short, slow, want)
                    assert np.array_equal(u8f, u8s) and np.array_equal(binf, np.where(u8s > thr, 255, 0).astype(np.uint8)), (name, attempt)
                assert emu.heatmap_to_roi(heat, threshold=thr, clip_frame=True)[0] == oracle.roi_from_heatmap_u8(ref_u8, thr, clip_frame=True), (name, rows_mode)
    finally:
        emu.debug_set("host_simple_shape", 1)
        emu.debug_set("heat_rows", 1)


def test_emu_small_pyramid_split_over_workgroups(emu, oracle):
    """k_small_filter_first with one, two and three workgroups per frame (each takes the last pyrUp step, the subtraction, the copy-out
    and the tile bounds of ITS band of tile rows): C_S and the bounds are the same numbers, so heatmap and extrema are bit-identical
    whatever the split; the lattice samples differ (one pair per band), which may only change how many pairs are evaluated."""
    rng = np.random.default_rng(29)
    try:
        for (T, H, W, L, S) in [(4, 160, 96, 8, 4), (3, 130, 70, 7, 3), (3, 97, 64, 6, 2)]:
            v = rng.random((T, H, W))
            emu.debug_set("ff_parts", 1)
            one, mm = emu.calibrate(v, 10.0, levels=L, skip=S)
            roi1 = emu.locate(v, 10.0, levels=L, skip=S)
            for parts in (2, 3):
                emu.debug_set("ff_parts", parts)
                got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)
                assert np.array_equal(got, one) and tuple(mm) == tuple(mm2), (T, H, W, L, S, parts)
                assert emu.locate(v, 10.0, levels=L, skip=S) == roi1
        fr = oracle.uint8_to_float(__import__("respmon_amd.synth", fromlist=["x"]).synth_breathing(12, 144, 128, seed=4))
        emu.debug_set("ff_parts", 2)
        assert emu.locate(fr, 10.0, levels=7, skip=4) == oracle.locate(fr, 10, pyramid_levels=7, skip_levels_at_top=4)
    finally:
        emu.debug_set("ff_parts", 0)


def _level1_footprint_extrema(oracle, c2, H, W):
    """min / max of pyrUp(C_2[u]) over every 64 x 16 tile's level-1 footprint (rows 8 ty - 1 .. 8 ty + 8 by columns 32 tx - 1 ..
    32 tx + 32, clipped to the image): what rm_bounds_l1.h must write, from the oracle's pyrUp."""
    h1, w1 = (H + 1) // 2, (W + 1) // 2
    nty, ntx = (H + 15) // 16, (W + 63) // 64
    lo = np.empty((c2.shape[0], nty, ntx)); hi = np.empty_like(lo)
    for u in range(c2.shape[0]):
        l1 = oracle.pyrUp(c2[u], (w1, h1))
        for ty in range(nty):
            y0, y1 = max(8 * ty - 1, 0), min(8 * ty + 8, h1 - 1)
            for tx in range(ntx):
                x0, x1 = max(32 * tx - 1, 0), min(32 * tx + 32, w1 - 1)
                f = l1[y0:y1 + 1, x0:x1 + 1]
                lo[u, ty, tx] = f.min(); hi[u, ty, tx] = f.max()
    return lo.reshape(c2.shape[0], -1), hi.reshape(c2.shape[0], -1)


def test_emu_level1_tile_bounds(emu, oracle):
    """rm_bounds_l1.h (round 6): at skip 2 the tile bounds are the extrema of the LEVEL-1 footprint, formed by a streaming kernel
    (three tile columns per wave, DPP neighbours, bands of tile rows).  The bounds themselves must equal the oracle's pyrUp of the
    collapsed level bit for bit, on ragged widths / heights, several column chunks and bands; the heatmap must not change by a bit
    against the level-2 bounds and against exhaustive evaluation; and the selection must not keep more pairs than before."""
    rng = np.random.default_rng(61)
    cases = [(4, 40, 200, 4, 2, 0), (3, 70, 450, 5, 2, 2), (5, 33, 131, 4, 2, 1), (2, 130, 70, 6, 2, 4), (3, 17, 64, 4, 2, 0),
             (2, 96, 388, 4, 2, 1), (3, 5, 7, 4, 2, 0)]
    for (T, H, W, L, S, trb) in cases:
        v = rng.random((T, H, W))
        v[:, : H // 2, : W // 3] *= 0.05          # a quiet corner: some pairs can be pruned at all
        emu.debug_set("bounds_l1_rows", trb)
        emu.debug_set("bounds_l1", 0)
        ref, mm = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512)     # (512: one launch per level -- the separate bounds kernel runs)
        kept_l2 = emu.counters()[2]
        emu.debug_set("bounds_l1", 1)                                      # the level-1 values in the chain's own float64 operations: the exact extrema
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512)
        kept_l1 = emu.counters()[2]
        assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (T, H, W, L, S, trb)
        assert kept_l1 <= kept_l2, (T, H, W, kept_l1, kept_l2)
        Th = T // 2 + 1
        h2, w2 = ((H + 1) // 2 + 1) // 2, ((W + 1) // 2 + 1) // 2
        ntiles = ((H + 15) // 16) * ((W + 63) // 64)
        c2 = emu.workspace("cS", (Th, h2, w2))
        lo = emu.workspace("tile_lo", (Th, ntiles)); hi = emu.workspace("tile_hi", (Th, ntiles))
        want_lo, want_hi = _level1_footprint_extrema(oracle, c2, H, W)
        assert np.array_equal(lo, want_lo) and np.array_equal(hi, want_hi), (T, H, W, L, S, trb)
        emu.debug_set("bounds_l1", 2)                                      # (the default) packed float32 + margin: SOUND, and within 2^-19 max|C_2| of the exact extrema
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512)
        assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (T, H, W, L, S, trb, "float32 bounds")
        assert emu.counters()[2] <= kept_l2
        lo = emu.workspace("tile_lo", (Th, ntiles)); hi = emu.workspace("tile_hi", (Th, ntiles))
        slack = np.abs(c2).max() * 2.0 ** -19 + 1e-40
        assert (lo <= want_lo).all() and (hi >= want_hi).all(), (T, H, W, "float32 bounds must contain the exact extrema")
        assert (lo >= want_lo - slack).all() and (hi <= want_hi + slack).all(), (T, H, W, "float32 bounds too loose")
        exhaustive, mm3 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512 | 1)
        assert np.array_equal(got, exhaustive) and tuple(mm) == tuple(mm3), (T, H, W, "no prune")
        for f in (128, 256):                      # the store-less and the store-based sum on the new bounds
            alt, mm4 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512 | f)
            assert np.array_equal(alt, ref) and tuple(mm) == tuple(mm4), (T, H, W, f)
        emu.debug_set("dense_t_low", 1)           # ... and the TileEval sum (what large frames take), which re-tests every pair against the exact top
        alt, mm4 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=512 | 128)
        emu.debug_set("dense_t_low", -1)
        assert np.array_equal(alt, ref) and tuple(mm) == tuple(mm4), (T, H, W, "k_dense_sum_t")
    emu.debug_set("bounds_l1_rows", 0)
    emu.debug_set("bounds_l1", 2)


def test_emu_labelling_rule_counts_not_clocks(emu, oracle):
    """whether the next ROI extraction of a geometry is labelled on the device is decided from what the previous
    one COUNTED (contours met, border steps walked on the host), never from how long it took -- so a stream takes the same path in
    every run.  A frame-spanning comb (two components, ~6 000 border steps) with the step bar lowered to 1 000: extraction 1 follows
    every border on the host, extractions 2 .. are labelled; with the bar at its default (24 000) every extraction stays on the
    host; the sequence of paths is the same when the whole experiment is repeated; the ROI never changes."""
    comb = np.zeros((72, 192), bool); comb[2, :] = True; comb[2:70, ::2] = True; comb[60:70, 1:40] = True
    heat = comb.astype(np.float64)
    want = oracle.roi_from_heatmap_u8(oracle.heatmap_u8(heat[None])[1], 20)
    runs = []
    for rep in range(2):
        seq = []
        for bar in (1000, 0):
            emu.debug_set("label_host_steps", bar)
            emu.heatmap_to_roi(np.zeros((8, 64)), threshold=20)          # another geometry: the rule's memory starts afresh
            for k in range(4):
                roi, _, _ = emu.heatmap_to_roi(heat, threshold=20)
                seq.append((bar, k, roi, emu.contour_stats()[1], emu.roi_path()))
        runs.append(seq)
    emu.debug_set("label_host_steps", 0)
    assert runs[0] == runs[1]
    rois = {s[2] for s in runs[0]}
    assert len(rois) == 1 and None not in rois
    assert rois == {want}
    lab = [(s[0], s[3]) for s in runs[0]]
    assert lab == [(1000, 0), (1000, 1), (1000, 1), (1000, 1), (0, 0), (0, 0), (0, 0), (0, 0)], lab


def test_emu_exception_store(emu, oracle):
    """rm_xstore.h (round 6): a dense selection's masked time sum through the exception store -- every kept pair evaluated once by a
    flat pass (k_xs_eval), its values below `top` parked in a compact record, the time-ordered additions by k_xs_sum with 1 / 2 / 4
    waves per tile -- against the value-store path (flags=256), bit for bit: few exceptions per pair (they travel with the record's
    header) and many (fetched from the record), ragged tiles, skip 1 .. 4, exhaustive evaluation, frame shards, a store that
    overflows (the store-less kernel behind k_xs_sum takes over), and the oracle's heatmap."""
    rng = np.random.default_rng(97)
    cases = [(6, 64, 96, 4, 2, 0.7), (5, 67, 131, 5, 3, 0.7), (4, 48, 64, 3, 1, 0.7), (5, 100, 160, 6, 4, 0.7), (7, 70, 300, 4, 2, 0.05),
             (4, 33, 70, 4, 2, 0.3), (9, 40, 200, 5, 2, 0.02), (3, 5, 7, 4, 2, 0.7)]
    for n, (T, H, W, L, S, thr) in enumerate(cases):
        v = rng.random((T, H, W))
        v[:, H // 3:, W // 4:] *= 0.2                       # a quieter part: fewer exceptions there
        emu.debug_set("xs", 1)
        want, mm = emu.calibrate(v, 10.0, levels=L, skip=S, thr=thr, flags=256)
        for nw in (1, 2, 4):
            emu.debug_set("xs_waves", nw)
            got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, thr=thr, flags=128)
            assert np.array_equal(got, want) and tuple(mm) == tuple(mm2), (T, H, W, L, S, thr, nw)
        emu.debug_set("xs_waves", 0)
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, thr=thr, flags=128 | 1)
        assert np.array_equal(got, want) and tuple(mm) == tuple(mm2), (T, H, W, L, S, thr, "no prune")
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, thr=thr, flags=4)      # (an 8-slot value store overflows: the dense route, decided on the device)
        assert np.array_equal(got, want) and tuple(mm) == tuple(mm2), (T, H, W, L, S, thr, "value store overflow")
        emu.debug_set("xs_budget_words", 3000)             # the exception store itself overflows: the store-less kernel takes the sum
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S, thr=thr, flags=128)
        emu.debug_set("xs_budget_words", 0)
        assert np.array_equal(got, want) and tuple(mm) == tuple(mm2), (T, H, W, L, S, thr, "exception store overflow")
        if n in (0, 4):
            raw = oracle.eulerian_magnification_bandpass(v.copy(), 10.0, 0.1, 1.0, 500.0, pyramid_levels=L, skip_levels_at_top=S, threshold=thr)[0]
            ref = np.average(raw, axis=0)
            assert np.abs(got - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300), (T, H, W, "oracle")
    v = rng.random((11, 70, 150))
    for world in (2, 3):                      # frame shards: partial sums over [t0, t1) with the global extrema
        a = emu.locate_sharded(v, world, levels=4, skip=2, flags=256)
        b = emu.locate_sharded(v, world, levels=4, skip=2, flags=128)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), world


def test_emu_bounds_refined_one_level_down(emu, oracle):
    """rm_bounds_l1.h k_bounds_up1 (round 6): at skip 3 / 4 the tile bounds of a dense stream are taken once more from the
    level-(S - 1) footprint.  The refined bounds lie inside the level-S bounds, still contain every full-resolution value of their
    pair (checked against the materialised raw video), keep no more pairs, and change no bit of the heatmap -- ragged shapes, both
    skips, the store-less and the store-based sum."""
    rng = np.random.default_rng(131)
    for (T, H, W, L, S) in [(4, 67, 131, 5, 3), (5, 100, 160, 6, 4), (3, 48, 200, 6, 4), (4, 33, 70, 5, 3), (6, 135, 240, 7, 4)]:
        v = rng.random((T, H, W))
        v[:, : H // 2] *= 0.1
        emu.debug_set("bounds_up1", 0)
        ref, mm = emu.calibrate(v, 10.0, levels=L, skip=S)
        kept0 = emu.counters()[2]
        Th = T // 2 + 1
        nty, ntx = (H + 15) // 16, (W + 63) // 64
        lo0 = emu.workspace("tile_lo", (Th, nty, ntx)); hi0 = emu.workspace("tile_hi", (Th, nty, ntx))
        emu.debug_set("bounds_up1", 1)
        got, mm2 = emu.calibrate(v, 10.0, levels=L, skip=S)
        kept1 = emu.counters()[2]
        lo1 = emu.workspace("tile_lo", (Th, nty, ntx)); hi1 = emu.workspace("tile_hi", (Th, nty, ntx))
        assert np.array_equal(got, ref) and tuple(mm) == tuple(mm2), (T, H, W, L, S)
        assert kept1 <= kept0, (T, H, W, kept1, kept0)
        tol = 1e-12 * max(abs(mm[0]), abs(mm[1]))
        assert (lo1 >= lo0 - tol).all() and (hi1 <= hi0 + tol).all(), (T, H, W, "refined bounds must lie inside the level-S bounds")
        _, raw, _ = emu.eulerian(v, 10.0, 0.1, 1.0, 500.0, L, S)
        for u in range(Th):
            for ty in range(nty):
                for tx in range(ntx):
                    blk = raw[u, 16 * ty: 16 * ty + 16, 64 * tx: 64 * tx + 64]
                    assert lo1[u, ty, tx] - tol <= blk.min() and blk.max() <= hi1[u, ty, tx] + tol, (T, H, W, u, ty, tx)
        for f in (128, 256, 1):
            alt, mm3 = emu.calibrate(v, 10.0, levels=L, skip=S, flags=f)
            assert np.array_equal(alt, ref) and tuple(mm) == tuple(mm3), (T, H, W, f)
    emu.debug_set("bounds_up1", -1)
