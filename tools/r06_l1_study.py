"""Developer tool (CPU, oracle only): on the 720p x 128 breathing stream (BASELINE config 2), how many (64x16 tile, frame) pairs hold a\nvalue below `top`, and how many the minimum of the level-1 / level-2 footprint keeps (rm_bounds_l1.h).  python tools/r06_l1_study.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import respmon_oracle as o
from respmon_amd import synth
NOISE = r"""
T,H,W,L,S = 64,1080,1920,9,4
v8 = synth.synth_noise_only(T,H,W)
frames = o.uint8_to_float(v8)
pyr = o.create_laplacian_video_pyramid(frames, L)
bp = [np.zeros(p.shape) for p in pyr]
for i in range(S, L-1):
    bp[i] = o.temporal_bandpass_filter_fft(pyr[i], 10, 0.1, 1.0, amplification_factor=500)
# collapse down to level S
sizes = [(p.shape[2], p.shape[1]) for p in pyr]
C = bp[L-1]
lv = {}
cur = bp[L-1]
for l in range(L-2, -1, -1):
    cur = np.stack([o.pyrUp(cur[t], sizes[l]) for t in range(T)]) + bp[l]
    lv[l] = cur
raw = lv[0]
mn, mx = raw.min(), raw.max(); top = mx-(mx-mn)*0.7
print("min max top", mn, mx, top)
ty, tx = (H+15)//16, W//64
Hc = H//16*16
ex = (raw[:, :Hc] < top).reshape(T, Hc//16, 16, tx, 64).any(axis=(2,4))
print("pairs with a value below top", ex.mean())
def foot(a, k, nr, nc):
    # rows (16ty>>k)-1 .. +nr-1, cols (64tx>>k)-1 .. +nc-1, clipped
    T_,h,w = a.shape
    out = np.full((T_, Hc//16, tx), np.inf)
    for iy in range(Hc//16):
        y0 = max(((16*iy)>>k)-1, 0); y1 = min(((16*iy)>>k)-1+nr-1, h-1)
        for ix in range(tx):
            x0 = max(((64*ix)>>k)-1, 0); x1 = min(((64*ix)>>k)-1+nc-1, w-1)
            out[:,iy,ix] = a[:, y0:y1+1, x0:x1+1].min(axis=(1,2))
    return out
for k, nr, nc in ((4,4,7),(3,5,11),(2,7,19),(1,10,34)):
    m = foot(lv[k], k, nr, nc)
    print("level-%d footprint bound keeps" % k, (m < top).mean())
"""
if len(sys.argv) > 1 and sys.argv[1] == "noise":   # 1080p, skip 4, sensor noise in every pixel: what the level-k footprint bounds keep (k_bounds_up1)
    exec(NOISE)
    sys.exit(0)
T,H,W,L,S = 128,720,1280,4,2
v8 = synth.synth_breathing(T,H,W,seed=1234)
frames = o.uint8_to_float(v8)
t0=time.time()
pyr = o.create_laplacian_video_pyramid(frames, L)
print("pyr", time.time()-t0, [p.shape for p in pyr])
C2 = o.temporal_bandpass_filter_fft(pyr[2], 10, 0.1, 1.0, amplification_factor=500)
print("C2", C2.shape, C2.min(), C2.max())
h1,w1 = pyr[1].shape[1:]
l1 = np.stack([o.pyrUp(C2[t], (w1,h1)) for t in range(T)])
raw = np.stack([o.pyrUp(l1[t], (W,H)) for t in range(T)])
mn,mx = raw.min(), raw.max(); top = mx-(mx-mn)*0.7
print("min max top", mn,mx,top)
ty,tx = H//16, W//64
ex = (raw < top).reshape(T,ty,16,tx,64).any(axis=(2,4))
print("pairs with a value below top", ex.mean())
# level-1 footprint min: rows 8ty-1..8ty+8, cols 32tx-1..32tx+32
def foot_min(a, th, tw, n_ty, n_tx):
    T_,h,w = a.shape
    out = np.full((T_,n_ty,n_tx), np.inf)
    for iy in range(n_ty):
        y0=max(th*iy-1,0); y1=min(th*iy+th,h-1)
        for ix in range(n_tx):
            x0=max(tw*ix-1,0); x1=min(tw*ix+tw,w-1)
            out[:,iy,ix]=a[:,y0:y1+1,x0:x1+1].min(axis=(1,2))
    return out
m1 = foot_min(l1, 8, 32, ty, tx)
print("level-1 bound keeps", (m1 < top).mean())
# level-2 footprint: rows (4ty-1 >>..) use tile_region rule: level1 region then /2 -1,+1
def foot2(a):
    T_,h,w = a.shape
    out = np.full((T_,ty,tx), np.inf)
    for iy in range(ty):
        y0=max(8*iy-1,0); y1=min(8*iy+8,h1-1); y0=max((y0>>1)-1,0); y1=min((y1>>1)+1,h-1)
        for ix in range(tx):
            x0=max(32*ix-1,0); x1=min(32*ix+32,w1-1); x0=max((x0>>1)-1,0); x1=min((x1>>1)+1,w-1)
            out[:,iy,ix]=a[:,y0:y1+1,x0:x1+1].min(axis=(1,2))
    return out
m2 = foot2(C2)
print("level-2 bound keeps", (m2 < top).mean())
# fraction of pixels below top inside kept tiles
print("values below top overall", (raw<top).mean())
# half tiles (8 rows)
ex8 = (raw < top).reshape(T,H//8,8,tx,64).any(axis=(2,4))
print("half-tile pairs with a value below", ex8.mean())
m1h = foot_min(l1, 4, 32, H//8, tx)
print("level-1 bound on half tiles keeps", (m1h<top).mean())

