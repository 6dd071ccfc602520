"""Developer tool (GPU box): the tile-local labelling on LARGE rows of whole words (4K: eight waves per tile workgroup, several rounds of
workgroups; 1080p / 1440p: sixteen) -- random densities, blobs, stripes; the labelled ROI against the host-only ROI, the component count
against scipy.ndimage.label.   python tools/r06_ccl_stress.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.ndimage as ndi
from respmon_amd import device, dist

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    H, W = [(2160, 3840), (1080, 1920), (1440, 2560), (2161, 3840), (1079, 1984), (4320, 7680)][int(rng.integers(0, 6 if n % 25 == 24 else 5))]
    kind = int(rng.integers(0, 4))
    if kind == 0:
        m = rng.random((H, W)) < rng.uniform(0.01, 0.7)
    elif kind == 1:
        m = rng.random((H, W)) < rng.uniform(0.01, 0.2)
        m |= ndi.gaussian_filter(rng.standard_normal(((H + 7) // 8, (W + 7) // 8)), float(rng.uniform(1, 12))).repeat(8, 0).repeat(8, 1)[:H, :W] > rng.uniform(-0.05, 0.1)
    elif kind == 2:
        m = np.zeros((H, W), bool); m[:: int(rng.integers(2, 9))] = True; m[:, :: int(rng.integers(40, 300))] = True
        m &= rng.random((H, W)) < 0.98
    else:
        m = rng.random((H, W)) < 0.5927   # site percolation threshold of the square lattice (8-connectivity: far beyond it; long snaking components at 4-conn. scale)
        m &= rng.random((H, W)) < rng.uniform(0.6, 1.0)
    heat = torch.from_numpy(m.astype(np.float64)).cuda()
    lab = dist.hip_heatmap_to_roi(heat, 20, labelling=True)
    n_l, used = dist.contour_stats()
    ref = dist.hip_heatmap_to_roi(heat, 20, labelling=False)
    want_n = ndi.label(m, structure=np.ones((3, 3)))[1]
    n += 1
    ok = lab == ref and (not used or n_l == want_n or want_n > (1 << 18))
    if not ok:
        bad += 1
        print("CCL MISMATCH", (H, W), kind, lab, ref, n_l, want_n, used, flush=True)
print("ccl stress: %d cases, %d mismatches" % (n, bad))
