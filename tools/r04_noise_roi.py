"""ROI stage of the full-frame-noise worst case: time rm_heatmap_to_roi with and without device labelling, count the borders,
and save the thresholded image (bit-packed) for offline work on the host contour stage.  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from respmon_amd import dist, synth, device

out = "gpurun_out/r04"
os.makedirs(out, exist_ok=True)
vid = synth.synth_noise_only(256, 1080, 1920, seed=777)
buf = torch.from_numpy(vid).cuda()
heat = dist.hip_calibrate(buf, 10)
torch.cuda.synchronize()
h = heat.cpu().numpy()
mn, mx = h.min(), h.max()
u8 = np.clip(np.rint((h - mn) / (mx - mn) * 255), 0, 255).astype(np.uint8) if mx > mn else np.zeros(h.shape, np.uint8)
np.save(out + "/noise_heat_u8.npy", u8)
for mode in (None, False, True):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        roi = dist.hip_heatmap_to_roi(heat, 20, labelling=mode)
        dt = time.perf_counter() - t0
    print("labelling", mode, "roi", roi, "ms %.3f" % (dt * 1e3), "stats", dist.contour_stats())
