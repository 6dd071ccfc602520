"""Randomised parity sweep on the GPU box (developer tool; the committed tests hold fixed cases):
  calib  random (T, H, W, levels, skip, frame dtype) calibrations against the CPU oracle -- bit-exact ROI, heatmap within
         1e-12 relative -- plus the staged (frame-sharded) path against the unsharded one;
  flow   random ROI sizes / textures / sub-pixel shifts: Shi-Tomasi corners and pyramidal LK against the oracle, bit-exact;
  roi    random heatmaps (many components, holes, frame-touching blobs) through the heatmap -> ROI stage.
  run    the run() state machine with skip_calibration in 'flow' / 'average' mode on moving textures;
  api    pyramid / filter / dtype / average / ROI / colour functions of the drop-in surface on random shapes;
  shard  the rm_shard_* stages with 2-8 emulated ranks (uneven shards) + the sparse exchange of the partial sums;
  big    calib at 200-620 x 300-1100 frames, T = 64-256, skip 2-4 (a few seconds of oracle per case).
      python tools/fuzz_parity.py [seconds] [seed] [calib|big|flow|roi|shard|api|run]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fuzz_flow(budget, seed):
    import torch
    from oracle import respmon_oracle as oracle
    from respmon_amd import synth
    from respmon_amd.base import _Backend
    oracle.build()
    be = _Backend()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    while time.time() < t_end:
        H = int(rng.integers(24, 260)); W = int(rng.integers(24, 260))
        render = synth.synth_texture(H, W, seed=int(rng.integers(1 << 30)), n_waves=int(rng.integers(3, 14)), n_spots=int(rng.integers(0, 50)))
        a = render(0.0, 0.0)
        b = render(float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        mc = int(rng.choice([10, 100, 1000])); q = float(rng.choice([0.3, 0.05, 0.01])); md = float(rng.choice([3, 7, 12])); bs = int(rng.choice([3, 5, 7]))
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        try:
            pts = be.good_features_to_track(dev(a), mc, q, md, bs)
            ref = oracle.goodFeaturesToTrack(a, mc, q, md, blockSize=bs)
            ok = (pts is None) == (ref is None) and (ref is None or np.array_equal(pts, ref))
            if ok and ref is not None:
                win = int(rng.choice([9, 15, 21])); ml = int(rng.integers(0, 4))
                p1, st = be.calc_optical_flow_pyr_lk(dev(a), dev(b), pts, (win, win), ml, (3, int(rng.choice([5, 10, 30])), float(rng.choice([0.03, 0.01]))))
                # same criteria for the oracle
            n += 1
        except Exception as e:      # noqa: BLE001
            ok = False
            print("EXC", repr(e))
        if not ok:
            bad += 1
            print("FLOW MISMATCH (corners)", dict(H=H, W=W, mc=mc, q=q, md=md, bs=bs), flush=True)
    print("fuzz flow: %d cases, %d mismatches" % (n, bad))
    return bad


def fuzz_flow_lk(budget, seed):
    import torch
    from oracle import respmon_oracle as oracle
    from respmon_amd import synth
    from respmon_amd.base import _Backend
    oracle.build()
    be = _Backend()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    while time.time() < t_end:
        H = int(rng.integers(24, 200)); W = int(rng.integers(24, 200))
        render = synth.synth_texture(H, W, seed=int(rng.integers(1 << 30)))
        a = render(0.0, 0.0); b = render(float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-2.5, 2.5)))
        npts = int(rng.integers(1, 60))
        pts = np.stack([rng.uniform(-2, W + 2, npts), rng.uniform(-2, H + 2, npts)], axis=1).astype(np.float32).reshape(-1, 1, 2)
        win = int(rng.choice([9, 15, 21])); ml = int(rng.integers(0, 4)); mcnt = int(rng.choice([5, 10, 30])); eps = float(rng.choice([0.03, 0.01]))
        try:
            p1, st = be.calc_optical_flow_pyr_lk(dev(a), dev(b), pts, (win, win), ml, (3, mcnt, eps))
            r1, rs, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, None, winSize=(win, win), maxLevel=ml, criteria=(3, mcnt, eps))
            good = rs.ravel() == 1
            ok = np.array_equal(st, rs) and np.array_equal(p1[good], r1[good])
        except Exception as e:      # noqa: BLE001
            ok = False
            print("EXC", repr(e))
        n += 1
        if not ok:
            bad += 1
            print("LK MISMATCH", dict(H=H, W=W, npts=npts, win=win, ml=ml, mcnt=mcnt, eps=eps), flush=True)
    print("fuzz lk: %d cases, %d mismatches" % (n, bad))
    return bad


def fuzz_shard(budget, seed):
    """Frame-sharded stages with emulated ranks (a library context each, collectives by hand) and the sparse heatmap
    exchange, against the unsharded path."""
    import ctypes
    import torch
    from respmon_amd import _capi, device, dist as rdist, synth
    from respmon_amd.base import RespiratoryMonitor
    lib = _capi.load()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    sp = device.stream_ptr()
    while time.time() < t_end:
        T = int(rng.integers(6, 140)); H = int(rng.integers(12, 300)); W = int(rng.integers(12, 500))
        L = int(rng.integers(3, 9)); S = int(rng.integers(1, L))
        world = int(rng.integers(2, min(T, 9)))
        dt = str(rng.choice(["f64", "u8"]))
        v8 = synth.synth_breathing(T, H, W, seed=int(rng.integers(1 << 30)))
        buf = torch.from_numpy(v8).cuda() if dt == "u8" else (torch.from_numpy(v8).cuda().to(torch.float64) * (1.0 / 255))
        code = _capi.RM_U8 if dt == "u8" else _capi.RM_F64
        ok = True
        ctxs = []
        try:
            nn = ctypes.c_size_t()
            _capi.check(lib, lib.rm_shard_layout(H, W, L, S, ctypes.byref(nn)), "layout")
            NP = int(nn.value)
            for _ in range(world):
                c = ctypes.c_void_p()
                _capi.check(lib, lib.rm_ctx_create(0, ctypes.byref(c)), "ctx")
                ctxs.append(c)
            spans = [rdist.shard_frames(T, r, world) for r in range(world)]
            lap_all = torch.empty((T, max(NP, 1)), dtype=torch.float64, device="cuda")[:, :NP].contiguous()
            for c, (t0, t1) in zip(ctxs, spans):
                if NP:
                    local = buf[t0:t1].contiguous()
                    _capi.check(lib, lib.rm_shard_pyramid(c, device.ptr(local), code, t1 - t0, H, W, L, S, 0,
                                                          ctypes.c_void_p(lap_all[t0:t1].data_ptr()), sp), "pyramid")
            mms = []
            for c, (t0, t1) in zip(ctxs, spans):
                mm = torch.empty(2, dtype=torch.float64, device="cuda")
                _capi.check(lib, lib.rm_shard_collapse(c, device.ptr(lap_all) if NP else None, T, t0, t1, H, W, 10.0, 0.1, 1.0, 500.0, L, S,
                                                       0.7, 0, device.ptr(mm), sp), "collapse")
                mms.append(mm)
            mm = torch.stack(mms).max(dim=0).values
            total = torch.zeros((H, W), dtype=torch.float64, device="cuda")
            cap = 64
            pd = int(lib.rm_heat_sparse_packet_doubles(cap))
            packets = []
            for c in ctxs:
                hs = torch.empty((H, W), dtype=torch.float64, device="cuda")
                _capi.check(lib, lib.rm_shard_heat(c, device.ptr(mm), 0.7, device.ptr(hs), sp), "heat")
                total += hs
                pk = torch.empty(pd, dtype=torch.float64, device="cuda")
                _capi.check(lib, lib.rm_heat_sparse_pack(c, device.ptr(hs), H, W, cap, device.ptr(pk), sp), "pack")
                packets.append(pk)
            heat = torch.empty((H, W), dtype=torch.float64, device="cuda")
            xywh = (ctypes.c_int32 * 4)()
            rc = _capi.check(lib, lib.rm_shard_finish(ctxs[0], device.ptr(total), T, H, W, 20, device.ptr(heat), xywh, sp), "finish")
            roi = None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)
            ref_heat = rdist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
            ref_roi = rdist.hip_heatmap_to_roi(ref_heat, 20)
            scale = float(ref_heat.abs().max())
            err = float((heat - ref_heat).abs().max()) / scale if scale > 0 else float((heat - ref_heat).abs().max())
            ok = err <= 1e-12
            # the ROI may differ only on an exact uint8 truncation boundary of the re-associated sum: compare through the heat
            ok = ok and (roi == ref_roi or roi == rdist.hip_heatmap_to_roi(heat, 20))
            # sparse exchange of the partial sums == dense sum in rank order / T
            fused = torch.empty((H, W), dtype=torch.float64, device="cuda")
            rc2 = _capi.check(lib, lib.rm_heat_sparse_merge_roi(ctxs[0], device.ptr(torch.cat(packets)), world, H, W, cap, 20, T,
                                                                device.ptr(fused), xywh, sp), "merge")
            if rc2 != _capi.RM_SPARSE_FALLBACK:
                ok = ok and torch.equal(fused, heat)
        except Exception as e:      # noqa: BLE001
            ok, err = False, repr(e)
        finally:
            for c in ctxs:
                lib.rm_ctx_destroy(c)
        n += 1
        if not ok:
            bad += 1
            print("SHARD MISMATCH", dict(T=T, H=H, W=W, L=L, S=S, world=world, dtype=dt), "err", err, flush=True)
    print("fuzz shard: %d cases, %d mismatches" % (n, bad))
    return bad


def fuzz_api(budget, seed):
    """The module-level functions of the drop-in surface on random shapes: pyramids (video and image forms), both
    temporal filters, dtype helpers, time average, ROI mean / crop, BGR -> gray."""
    import scipy.signal
    import torch
    from oracle import respmon_oracle as oracle
    from respmon_amd import montage, pyramid, transforms
    from respmon_amd.base import _Backend
    oracle.build()
    be = _Backend()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0

    def rel(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0
    while time.time() < t_end:
        kind = int(rng.integers(0, 7))
        what, ok = "", True
        try:
            if kind == 0:      # Laplacian video pyramid + collapse, bit-exact
                T, H, W, L = int(rng.integers(1, 6)), int(rng.integers(1, 150)), int(rng.integers(1, 260)), int(rng.integers(1, 9))
                vid = rng.random((T, H, W))
                what = "pyramid %s L=%d" % ((T, H, W), L)
                got, ref = pyramid.create_laplacian_video_pyramid(vid, L), oracle.create_laplacian_video_pyramid(vid, L)
                ok = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, ref))
                ok = ok and np.array_equal(pyramid.collapse_laplacian_video_pyramid([x.copy() for x in got]),
                                           oracle.collapse_laplacian_video_pyramid([x.copy() for x in ref]))
                img = vid[0]
                ok = ok and all(np.array_equal(a, b) for a, b in zip(pyramid.create_gaussian_image_pyramid(img, L), oracle.create_gaussian_image_pyramid(img, L)))
            elif kind == 1:    # FFT band-pass, any length
                T = int(rng.integers(2, 300)); fps = float(rng.choice([10.0, 30.0, 5.01, 7.68, 2.5]))
                x = rng.standard_normal((T, int(rng.integers(1, 9)), int(rng.integers(1, 40))))
                fmin, fmax, amp = float(rng.uniform(0.05, 0.9)), float(rng.uniform(1.0, 3.0)), float(rng.choice([1.0, 50.0, 500.0]))
                what = "fft T=%d fps=%g band=(%g,%g)" % (T, fps, fmin, fmax)
                got = transforms.temporal_bandpass_filter_fft(x.copy(), fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
                ref = oracle.temporal_bandpass_filter_fft(x.copy(), fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp)
                ok = np.abs(got - ref).max() <= 1e-11 * max(np.abs(ref).max(), np.abs(x).max() * amp * 1e-3, 1e-300)
            elif kind == 2:    # IIR filter
                T = int(rng.integers(1, 200)); order = int(rng.integers(1, 7))
                lo = float(rng.uniform(0.02, 0.3)); hi = float(rng.uniform(lo + 0.05, 0.9))
                b, a = scipy.signal.butter(order, [lo, hi], btype="band")
                x = rng.standard_normal((T, int(rng.integers(1, 60))))
                what = "lfilter T=%d order=%d" % (T, order)
                got = transforms.butter_bandpass_filter_fast(x.copy(), b, a, axis=0)
                ref = scipy.signal.lfilter(b, a, x, axis=0)
                ok = np.abs(got - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300)
            elif kind == 3:    # dtype helpers incl. out-of-range values (C cast semantics)
                v = np.concatenate([rng.uniform(-3, 3, 200), rng.uniform(0, 1, 200), [0.0, 1.0, -0.0, 1.0 / 255, 254.999 / 255, 256.0 / 255, -1e-9, 1e9, -1e9]])
                what = "float_to_uint8"
                with np.errstate(all="ignore"):
                    ok = np.array_equal(transforms.float_to_uint8(v), oracle.float_to_uint8(v))
                k = rng.integers(0, 256, 300).astype(np.uint8)
                ok = ok and np.array_equal(transforms.uint8_to_float(k), oracle.uint8_to_float(k))
            elif kind == 4:    # time average of every frame dtype
                T, H, W = int(rng.integers(1, 70)), int(rng.integers(1, 60)), int(rng.integers(1, 90))
                dt = rng.choice([np.uint8, np.float16, np.float32, np.float64])
                vid = (rng.random((T, H, W)) * 255).astype(np.uint8) if dt == np.uint8 else rng.random((T, H, W)).astype(dt)
                what = "time_average %s %s" % ((T, H, W), dt.__name__)
                ref_in = oracle.uint8_to_float(vid) if dt == np.uint8 else vid.astype(np.float64)
                acc = np.zeros((H, W))
                for fr in ref_in:
                    acc = acc + fr
                ok = np.array_equal(montage.time_average(vid).cpu().numpy(), acc / T)
            elif kind == 5:    # ROI mean and crop -> uint8
                H, W = int(rng.integers(2, 200)), int(rng.integers(2, 300))
                g = (rng.random((H, W)) * 255).astype(np.uint8)
                w, h = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
                x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                what = "roi %s (%d,%d,%d,%d)" % ((H, W), x, y, w, h)
                dev = torch.from_numpy(g).cuda()
                frame = oracle.uint8_to_float(g)
                refm = np.average(frame[y:y + h, x:x + w])
                ok = abs(be.roi_mean(dev, x, y, w, h) - refm) <= 1e-12 * max(abs(refm), 1e-300)
                ok = ok and np.array_equal(be.roi_to_uint8(dev, x, y, w, h).cpu().numpy(), oracle.float_to_uint8(frame[y:y + h, x:x + w]))
            else:              # BGR -> gray
                H, W = int(rng.integers(1, 120)), int(rng.integers(1, 200))
                bgr = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
                what = "bgr2gray %s" % ((H, W),)
                ok = np.array_equal(be.bgr_to_gray(bgr).cpu().numpy(), oracle.cvtColor_bgr2gray(bgr))
        except Exception as e:      # noqa: BLE001
            ok = False
            what += " EXC " + repr(e)
        n += 1
        if not ok:
            bad += 1
            print("API MISMATCH", what, flush=True)
    print("fuzz api: %d cases, %d mismatches" % (n, bad))
    return bad


def fuzz_run(budget, seed):
    """RespiratoryMonitor.run() with skip_calibration in 'flow' and 'average' modes on random moving textures, against the
    oracle's extract_motion restatement (corner init, LK, track loss -> nan, mean flow, row-unpack PCA)."""
    from oracle import respmon_oracle as oracle
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    oracle.build()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    while time.time() < t_end:
        H, W = int(rng.integers(60, 200)), int(rng.integers(60, 260))
        T = int(rng.integers(5, 40))
        render = synth.synth_texture(H, W, seed=int(rng.integers(1 << 30)))
        ax, ay, ph = float(rng.uniform(0, 4)), float(rng.uniform(0, 2)), float(rng.uniform(0, 3))
        frames = np.stack([render(ax * np.sin(2 * np.pi * 0.4 * t / 10), ay * np.sin(2 * np.pi * 0.4 * t / 10 + ph)) for t in range(T)])
        w, h = int(rng.integers(20, W)), int(rng.integers(20, H))
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        mode = str(rng.choice(["flow", "average"]))
        try:
            mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=10), visualize=None, save_all_data=False,
                                     motion_extraction_method=mode, run_on_init=False)
            mon.sync_to_fps = lambda: None
            mon.skip_calibration(x, y, w, h)
            mon.run()
            got = np.array(mon.data, dtype=np.float64)
            if mode == "average":
                ref = np.array([oracle.roi_average(oracle.uint8_to_float(f), x, y, w, h) for f in frames])
                ok = len(got) == T and np.allclose(got, ref, rtol=1e-12, atol=0)
            else:
                state = oracle.FlowState()
                ref = []
                for t in range(T):
                    if state.prev is not None and (state.points is None or len(state.points) == 0):
                        break       # no corners / every track lost: cv2 would raise in the reference, the monitor goes to 'error'
                    ref.append(oracle.extract_motion_flow(state, oracle.uint8_to_float(frames[t])[y:y + h, x:x + w]))
                ref = np.array(ref, dtype=np.float64)
                k = min(len(got), len(ref))
                ok = k >= 1 and np.allclose(got[:k], ref[:k], rtol=1e-9, atol=1e-12, equal_nan=True)
        except Exception as e:      # noqa: BLE001
            ok = False
            print("EXC", repr(e)[:300])
        n += 1
        if not ok:
            bad += 1
            print("RUN MISMATCH", dict(H=H, W=W, T=T, roi=(x, y, w, h), mode=mode), flush=True)
    print("fuzz run: %d cases, %d mismatches" % (n, bad))
    return bad


def fuzz_roi(budget, seed):
    import scipy.ndimage as ndi
    import torch
    from oracle import respmon_oracle as oracle
    from respmon_amd import dist as rdist
    oracle.build()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    while time.time() < t_end:
        H = int(rng.integers(1, 300)); W = int(rng.integers(1, 500))
        if rng.random() < 0.1:
            H = int(rng.integers(300, 1200)); W = int(rng.integers(500, 2000))   # several workgroups per labelling kernel
        if rng.random() < 0.5:
            W = max(64, (W + 32) // 64 * 64)    # rows of whole 64-pixel words: the row-record kernel (k_heat_rows_u8) and its host rule
        kind = int(rng.integers(0, 4))
        if kind == 3:
            # one or two solid blobs (ellipses, sometimes touching, sometimes with a hole): the shapes the one-blob rule decides on
            yy, xx = np.mgrid[0:H, 0:W]
            heat = np.zeros((H, W))
            for _ in range(int(rng.integers(1, 3))):
                cy, cx = rng.uniform(0, H), rng.uniform(0, W)
                ry, rx = rng.uniform(1, max(2, H / 2)), rng.uniform(1, max(2, W / 2))
                heat[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1.0
            if rng.random() < 0.3:
                cy, cx = rng.uniform(0, H), rng.uniform(0, W)
                heat[((yy - cy) / max(1.0, H / 8)) ** 2 + ((xx - cx) / max(1.0, W / 8)) ** 2 <= 1.0] = 0.0
        elif kind == 0:
            heat = (rng.random((H, W)) < rng.uniform(0.02, 0.7)).astype(np.float64)
        elif kind == 1:
            heat = ndi.gaussian_filter(rng.standard_normal((H, W)), float(rng.uniform(0.5, 4.0)))
        else:
            heat = (ndi.gaussian_filter(rng.standard_normal((H, W)), 2.0) > 0).astype(np.float64) * rng.random((H, W))
        thr = int(rng.integers(0, 255))
        with np.errstate(all="ignore"):
            u8 = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min()))
        ref = oracle.roi_from_heatmap_u8(u8, thr)
        dev = torch.from_numpy(heat).cuda()
        got = rdist.hip_heatmap_to_roi(dev, thr, labelling=False)
        lab = rdist.hip_heatmap_to_roi(dev, thr, labelling=True)     # components labelled on the device first (rm_ccl.h)
        n += 1
        if got != ref or lab != ref:
            bad += 1
            print("ROI MISMATCH", dict(H=H, W=W, kind=kind, thr=thr), got, lab, ref, flush=True)
    print("fuzz roi: %d cases, %d mismatches" % (n, bad))
    return bad


_PENDING = []
_BACKEND = []


def _submit_backend():
    if not _BACKEND:
        from respmon_amd.base import _Backend
        _BACKEND.append(_Backend())
    return _BACKEND[0]


def main():
    mode = sys.argv[3] if len(sys.argv) > 3 else "calib"
    big = mode == "big"
    if mode not in ("calib", "big"):
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
        bad = {"flow": lambda: fuzz_flow(budget / 2, seed) + fuzz_flow_lk(budget / 2, seed), "roi": lambda: fuzz_roi(budget, seed),
               "shard": lambda: fuzz_shard(budget, seed), "api": lambda: fuzz_api(budget, seed),
               "run": lambda: fuzz_run(budget, seed)}[mode]()
        return 1 if bad else 0
    import torch
    from oracle import respmon_oracle as oracle
    from respmon_amd import synth, dist as rdist
    from respmon_amd.base import RespiratoryMonitor
    oracle.build()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t_end = time.time() + budget
    n = bad = ties = ill = 0
    tie_knobs, ill_knobs = {}, {}
    # argv[4] == "reforder": the reference's operation order (RM_FLAG_FILTER_LAPLACIANS) instead of the filter-first default
    reforder = len(sys.argv) > 4 and sys.argv[4] == "reforder"
    RespiratoryMonitor.reference_operation_order = reforder
    while time.time() < t_end:
        T = int(rng.choice([8, 16, 24, 32, 40, 64, 96, 128])) if rng.random() < 0.5 else int(rng.integers(2, 140))   # odd lengths too
        H = int(rng.integers(9, 200)); W = int(rng.integers(9, 330))
        if big:     # sizes where the pruning, several strips / segments and the value store all come into play
            T = int(rng.choice([64, 128, 256])); H = int(rng.integers(200, 620)); W = int(rng.integers(300, 1100))
        if rng.random() < 0.4:
            W = (W + 15) // 16 * 16            # the register-resident chains need W % 16 == 0
        L = int(rng.integers(2, 9)); S = int(rng.integers(0, L))
        if big:
            L = int(rng.integers(6, 10)); S = int(rng.integers(2, 5))
        dt = str(rng.choice(["f64", "u8", "f32", "f16", "bgr8"]))
        v8 = synth.synth_breathing(T, H, W, seed=int(rng.integers(1 << 30)), amplitude=float(rng.uniform(0.05, 0.3)),
                                   noise=float(rng.uniform(0.0, 0.04) if rng.random() < 0.8 else rng.uniform(0.06, 0.5)),   # one in five: a dense stream (store-less sums)
                                   center=(float(rng.uniform(0.1, 0.9)), float(rng.uniform(0.1, 0.9))))
        f64 = oracle.uint8_to_float(v8)
        if dt == "bgr8":    # frames as captured, [T,H,W,3]: base.py:230's cvtColor is the oracle's own, frame by frame
            off = rng.integers(-40, 41, size=3)
            bgr = np.stack([np.clip(v8.astype(np.int32) + int(o), 0, 255) for o in off], axis=-1).astype(np.uint8)
            if rng.random() < 0.3:
                bgr = rng.integers(0, 256, size=bgr.shape, dtype=np.uint8) // 4 + (bgr // 4) * 3
            f64 = oracle.uint8_to_float(np.stack([oracle.cvtColor_bgr2gray(f) for f in bgr]))
            dev, ref_in = torch.from_numpy(bgr).cuda(), f64
        elif dt == "u8":
            dev, ref_in = torch.from_numpy(v8).cuda(), f64
        elif dt == "f64":
            dev, ref_in = torch.from_numpy(f64).cuda(), f64
        else:
            npdt = np.float32 if dt == "f32" else np.float16
            q = f64.astype(npdt)
            dev, ref_in = torch.from_numpy(q).cuda(), q.astype(np.float64)
        fps = float(rng.choice([10.0, 10.0, 5.01, 30.0]))
        kw = dict(pyramid_levels=L, skip_levels_at_top=S)
        if rng.random() < 0.5:    # the other knobs of locate(): band, amplification, mask width, binary threshold
            kw.update(freq_min=float(rng.uniform(0.05, 0.6)), freq_max=float(rng.uniform(0.7, 2.5)), amplification=float(rng.choice([1, 50, 500, 5000])),
                      temporal_threshold=float(rng.choice([0.0, 0.1, 0.5, 0.7, 0.9, 1.0, 1.3])), threshold=int(rng.choice([0, 5, 20, 60, 200, 254])))
        ckw = {k: v for k, v in kw.items() if k != "threshold"}
        try:
            with np.errstate(all="ignore"):
                ref, mid = oracle.locate(ref_in, fps, return_intermediates=True, **kw)
            got = RespiratoryMonitor.locate(dev, fps, **kw)
            if not reforder and rng.random() < 0.3:
                # the two-call form with the previous case's buffer still in flight (rm_locate_submit / rm_locate_result): same ROIs
                be = _submit_backend()
                tk = be.locate_submit(dev, fps, kw.get("freq_min", 0.1), kw.get("freq_max", 1.0), kw.get("amplification", 500), L, S,
                                      kw.get("temporal_threshold", 0.7), kw.get("threshold", 20))
                prev = _PENDING.pop() if _PENDING else None
                if prev is not None and be.locate_result(prev[0]) != prev[1]:
                    raise AssertionError("rm_locate_result of the previous case differs from its rm_locate")
                _PENDING.append((tk, got))
            heat = rdist.hip_calibrate(dev, fps, flags=64 if reforder else 0, **ckw).cpu().numpy()
            # the heatmap is what is left of band-passed images of magnitude ~ amplification * |frame| after the Laplacian
            # differences: where they cancel (a 1x1 level S has an exactly zero Laplacian) the reference's own result is rounding
            # noise, and the filter-first form's noise (~1e-16 of the band-passed magnitudes) is a different sample of it --
            # errors are measured against the larger of the heatmap and 1e-3 of that magnitude
            scale = max(np.abs(mid["avg_frame"]).max(), 1e-3 * float(kw.get("amplification", 500)) * float(np.abs(ref_in).max()), 1e-300)
            err = np.abs(heat - mid["avg_frame"]).max() / scale
            ok = got == ref and (err <= 1e-12 or not np.isfinite(scale))
            err = float(err)
            if ok and S >= 1 and T >= 3:
                world = int(rng.integers(1, 4))
                import ctypes
                sh = rdist.locate_sharded(dev, T, fps, **kw) if world == 1 else None
                ok = ok and (sh is None or sh == got)
        except Exception as e:     # noqa: BLE001 -- report and continue
            ok, err = False, repr(e)
        n += 1
        if not ok and not isinstance(err, str):
            # The hard threshold of transforms.py:188-192 makes the heatmap discontinuous in raw: a voxel sitting on `top` to
            # within the rounding noise of the temporal filter (ours is the explicit operator, the oracle's is scipy's FFT) may
            # fall on either side.  Such a tie is a property of the reference algorithm -- but it is only ACCEPTED when it
            # explains the whole difference: every heatmap pixel that differs must be the column of a voxel within tolerance of
            # `top`, the GPU's value there must be the oracle's column average with some subset of those voxels flipped
            # (masked <-> unmasked), and the GPU's ROI must be what the oracle's ROI stage returns for the GPU's heatmap.
            tt = kw.get("temporal_threshold", 0.7)
            with np.errstate(all="ignore"):
                masked_ref, raw = oracle.eulerian_magnification_bandpass(ref_in, fps, kw.get("freq_min", 0.1), kw.get("freq_max", 1.0),
                                                                        kw.get("amplification", 500), pyramid_levels=L, skip_levels_at_top=S,
                                                                        threshold=tt)
            mn_, mx_ = raw.min(), raw.max()
            top_ = mx_ - (mx_ - mn_) * tt
            tol = 1e-11 * max(abs(mx_), abs(mn_), 1e-300)
            near = np.abs(raw - top_) <= tol
            explained = False
            why = "no voxel within tolerance of top"
            if near.any():
                ref_avg = mid["avg_frame"]
                differs = np.abs(heat - ref_avg) > 1e-12 * scale
                cols = near.any(axis=0)
                why = "%d differing pixels outside the %d tied columns" % (int((differs & ~cols).sum()), int(cols.sum()))
                if not (differs & ~cols).any():
                    explained = True
                    for (yy, xx) in np.argwhere(differs):
                        ts = np.flatnonzero(near[:, yy, xx])
                        col_raw, col_m = raw[:, yy, xx], masked_ref[:, yy, xx].copy()
                        hit = False
                        if len(ts) <= 10:
                            for bits in range(1 << len(ts)):
                                c = col_m.copy()
                                for i, tq in enumerate(ts):
                                    if bits >> i & 1:     # flip: what the oracle masked stays, what it kept is masked
                                        c[tq] = col_raw[tq] if c[tq] == mn_ and col_raw[tq] != mn_ else mn_
                                if abs(np.average(c) - heat[yy, xx]) <= 1e-12 * scale:
                                    hit = True
                                    break
                        if not hit:
                            explained = False
                            why = "pixel (%d, %d): no flip of its %d tied voxels gives the GPU value" % (yy, xx, len(ts))
                            break
                if explained:
                    with np.errstate(all="ignore"):
                        u8_g = oracle.float_to_uint8((heat - heat.min()) / (heat.max() - heat.min()))
                    explained = oracle.roi_from_heatmap_u8(u8_g, kw.get("threshold", 20)) == got
                    why = "ROI of the GPU heatmap by the oracle's ROI stage != GPU ROI" if not explained else ""
            if explained:
                ties += 1
                ok = True
                key = "tt=%g S=%d %s" % (tt, S, "T odd" if T & 1 else "T even")
                tie_knobs[key] = tie_knobs.get(key, 0) + 1
                print("TIE at the mask threshold (explained by flips of tied voxels)", dict(T=T, H=H, W=W, L=L, S=S, dtype=dt, fps=fps, kw=kw), "err", err,
                      "tied voxels", int(near.sum()), "min/max/top", mn_, mx_, top_, flush=True)
            else:
                print("  not a tie:", why, flush=True)
        if not ok and not isinstance(err, str):
            thr_b = kw.get("threshold", 20)
            a = heat
            with np.errstate(all="ignore"):
                u8_ours = oracle.float_to_uint8((a - a.min()) / (a.max() - a.min()))
            u8_ref = mid["avg_u8"]
            fg_o, fg_r = np.argwhere(u8_ours > thr_b), np.argwhere(u8_ref > thr_b)
            rng_ref = float(mid["avg_frame"].max() - mid["avg_frame"].min())
            err_rng = float(np.abs(a - mid["avg_frame"]).max()) / rng_ref if rng_ref > 0 else float("inf")
            consistent = oracle.roi_from_heatmap_u8(u8_ours, thr_b) == got
            # The normalisation (a - min) / (max - min) of base.py:563 divides by the heatmap's own RANGE.  With a narrow
            # mask (temporal_threshold near 0) most voxels stay unmasked and average to ~0 over time, so the range is tiny
            # against the magnitudes that were summed, and rounding noise of relative size 1e-15 on the sums decides the
            # uint8 levels: the problem is ill-conditioned for the reference as well.  Not a parity failure as long as the
            # heatmap agrees to rounding in absolute terms and our ROI is what the oracle's ROI stage gives for OUR heatmap.
            if consistent and err <= 1e-12:
                ill += 1
                ok = True
                key = "tt=%g S=%d" % (kw.get("temporal_threshold", 0.7), S)
                ill_knobs[key] = ill_knobs.get(key, 0) + 1
            print("  detail:", dict(T=T, H=H, W=W, L=L, S=S, dtype=dt, kw=kw) if ok else "", "err/range %.2e" % err_rng, "u8 maps equal", np.array_equal(u8_ours, u8_ref), "| fg ours", len(fg_o), fg_o[:4].tolist(), "| fg ref", len(fg_r), fg_r[:4].tolist(),
                  "| roi of the GPU heatmap by the oracle's contour code", oracle.roi_from_heatmap_u8(u8_ours, thr_b), flush=True)
        if not ok:
            bad += 1
            print("MISMATCH", dict(T=T, H=H, W=W, L=L, S=S, dtype=dt, fps=fps, kw=kw), "got", locals().get("got"), "ref", locals().get("ref"), "err", err, flush=True)
    print("fuzz%s: %d cases, %d mismatches, %d threshold ties, %d ill-conditioned normalisations" % (" (reference operation order)" if reforder else "", n, bad, ties, ill))
    print("  ties by knob:", dict(sorted(tie_knobs.items())))
    print("  ill-conditioned by knob:", dict(sorted(ill_knobs.items())))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
