"""developer: per-frame cost of rm_flow_step (config 3 texture, 1000 points; and 100 corners on a 351x235 ROI).
    python tools/flow_profile.py            (under rocprofv3 --kernel-trace --stats for the kernel table)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from respmon_amd import synth
    from respmon_amd.base import _Backend
    be = _Backend()
    for (H, W, maxc, q, md, label) in ((256, 256, 1000, 0.01, 3, "F: 256x256, 1000 points"), (235, 351, 100, 0.3, 7, "P ROI: 351x235, 100 corners")):
        render = synth.synth_texture(H, W, seed=4321)
        n = 120
        dev = [torch.from_numpy(render(1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3))).cuda() for t in range(n + 1)]
        fs = be.flow_state()
        pts = be.flow_begin(fs, dev[0], 0, 0, W, H, maxc, q, md, 7)
        for i in range(20):
            be.flow_step(fs, dev[i + 1], 0, 0, W, H, (15, 15), 2, (3, 10, 0.03))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20, n):
            mean, ng = be.flow_step(fs, dev[i + 1], 0, 0, W, H, (15, 15), 2, (3, 10, 0.03))
        torch.cuda.synchronize()
        print("%s: %d corners, %.3f ms per rm_flow_step, %d points left" % (label, 0 if pts is None else len(pts), (time.perf_counter() - t0) / (n - 20) * 1e3, ng), flush=True)


if __name__ == "__main__":
    main()
