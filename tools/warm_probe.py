"""Developer probe (GPU box, first process on a fresh box): does the placement of the frame buffer decide the frame-buffer kernel's speed?"""
import sys, time, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from respmon_amd import synth, _capi, device
from respmon_amd.base import _Backend
T, H, W = 256, 1080, 1920
v8 = synth.synth_breathing(T, H, W, seed=1234)
be = _Backend(); lib = _capi.load()


def fill(buf):
    for a in range(0, T, 16):
        buf[a:a + 16] = torch.from_numpy(v8[a:a + 16]).cuda().to(torch.float64) * (1.0 / 255)


def measure(buf, tag):
    f = lambda: be.locate(buf, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
    for _ in range(300): f()
    torch.cuda.synchronize()
    _capi.check(lib, lib.rm_profile_enable(device.ctx(), 1), 'e')
    t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 200 * 1e3
    ms = (ctypes.c_double * 4)(); n = ctypes.c_int(); lib.rm_profile_read(device.ctx(), ms, ctypes.byref(n))
    lib.rm_profile_enable(device.ctx(), 0)
    print('%-44s ptr %#x  step %.4f  kernel %.4f' % (tag, buf.data_ptr(), el, ms[0] / max(n.value, 1)), flush=True)


b1 = torch.empty((T, H, W), dtype=torch.float64, device='cuda'); fill(b1)
measure(b1, 'first allocation of the process')
b2 = torch.empty((T, H, W), dtype=torch.float64, device='cuda'); fill(b2)
measure(b2, 'second buffer (first still alive)')
measure(b1, 'first again')
del b1
dummy = torch.empty(3 << 30, dtype=torch.uint8, device='cuda')
b3 = torch.empty((T, H, W), dtype=torch.float64, device='cuda'); fill(b3)
measure(b3, 'third buffer after a 3 GB dummy')
big = torch.empty(T * H * W + (1 << 20), dtype=torch.float64, device='cuda')
for off in (256, 4096 + 512, 65536 + 2048):
    v = big[off:off + T * H * W].view(T, H, W); fill(v)
    measure(v, 'view at +%d doubles' % off)
measure(b2, 'second again')
