"""Developer tool (GPU box): how many (64x16 tile, frame) pairs of the materialised raw video hold a value below `top`, and how
many the convexity bound of level k would keep (min over the tile's level-k footprint < top), per workload."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from respmon_amd import _capi, device, synth, pyramid, transforms

def study(name, v8, L, S, dt=torch.float64):
    T, H, W = v8.shape
    buf = torch.empty((T, H, W), dtype=dt, device="cuda")
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = (torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)).to(dt)
    masked, raw = transforms.eulerian_magnification_bandpass(buf, 10, 0.1, 1.0, 500, pyramid_levels=L, skip_levels_at_top=S)
    del masked
    mn, mx = float(raw.min()), float(raw.max()); top = mx - (mx - mn) * 0.7
    Hc, Wc = H // 16 * 16, W // 64 * 64
    ex = torch.zeros((T, Hc // 16, Wc // 64), dtype=torch.bool, device="cuda")
    frac = 0.0
    for t in range(T):
        e = raw[t, :Hc, :Wc] < top
        frac += float(e.float().mean())
        ex[t] = e.view(Hc // 16, 16, Wc // 64, 64).any(dim=3).any(dim=1)
    print("%s: values below top %.4f, pairs with one %.3f, tiles with one in some frame %.3f  (min %.3g max %.3g top %.3g)" % (
        name, frac / T, float(ex.float().mean()), float(ex.any(dim=0).float().mean()), mn, mx, top))
    # level-k bound, k = 1: min over the level-1 footprint, approximated from raw itself: a level-1 value U1[y, x] is bounded below by ... not
    # available here (the library does not expose the levels of the collapse) -- the even-even samples raw[2y, 2x] are convex
    # combinations of U1, so they cannot go below it: min over raw of the tile + its 2-pixel ring is an UPPER estimate of what a level-1 bound keeps.
    ring = torch.zeros_like(ex)
    for t in range(T):
        e = (raw[t] < top).float()[None, None]
        d = torch.nn.functional.max_pool2d(e, kernel_size=5, stride=1, padding=2)[0, 0, :Hc, :Wc] > 0
        ring[t] = d.view(Hc // 16, 16, Wc // 64, 64).any(dim=3).any(dim=1)
    print("    pairs within 2 pixels of a value below top %.3f" % float(ring.float().mean()))

which = sys.argv[1:] or ["noise", "Q", "R"]
if "noise" in which:
    study("P noise (1080p x 256, L9 S4)", synth.synth_noise_only(256, 1080, 1920), 9, 4)
if "P" in which:
    study("P breathing (headline)", synth.synth_breathing(256, 1080, 1920, seed=1234), 9, 4)
if "blobs16" in which:
    study("P blobs16", synth.synth_breathing_16(256, 1080, 1920), 9, 4)
if "Q" in which:
    study("Q (720p x 128, L4 S2)", synth.synth_breathing(128, 720, 1280, seed=1234), 4, 2)
if "R" in which:
    study("R (4K x 512 f16, L6 S2)", synth.synth_breathing_blocks(512, 2160, 3840, seed=1234), 6, 2, torch.float16)
