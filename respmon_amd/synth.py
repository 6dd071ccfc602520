"""
Seeded synthetic inputs for tests and bench.py.  numpy only; runs on
both the build container and the GPU box.  Also a fake capture object with the duck type
`run()` needs (reference base.py:48-51, 227-233: isOpened/read/get/release).
"""
import numpy as np

CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5


def _lowpass_noise(rng, h, w, cutoff=0.06):
    """Seeded low-pass noise in [-1, 1]."""
    n = rng.standard_normal((h, w))
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    filt = np.exp(-(fy * fy + fx * fx) / (2 * cutoff * cutoff))
    t = np.fft.irfft2(np.fft.rfft2(n) * filt, s=(h, w))
    t -= t.mean()
    m = np.abs(t).max()
    return t / m if m > 0 else t


def synth_breathing(T, H, W, seed=1234, fps=10.0, breath_hz=0.4, amplitude=0.2, noise=0.02,
                    center=(0.6, 0.4), sigma=(0.10, 0.08), block=8):
    """uint8 [T,H,W] gray video: static low-pass texture + a Gaussian blob whose brightness
    oscillates at `breath_hz` (inside the 0.1-1.0 Hz calibration band) + white noise.
    The blob is off-centre so bounding-box bugs show.  Frames enter the path as
    uint8_to_float(g) like reference base.py:231.  Generated `block` frames at a time to
    bound host memory at 1080p x 256."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tex = _lowpass_noise(rng, H, W)
    yy = (np.arange(H)[:, None] - center[0] * H) / (sigma[0] * H)
    xx = (np.arange(W)[None, :] - center[1] * W) / (sigma[1] * W)
    blob = (amplitude * 255.0 * np.exp(-0.5 * (yy * yy + xx * xx))).astype(np.float32)
    base = (255.0 * (0.5 + 0.25 * tex)).astype(np.float32)
    sig = np.float32(255.0 * noise)
    out = np.empty((T, H, W), dtype=np.uint8)
    for t0 in range(0, T, block):
        t1 = min(T, t0 + block)
        s = np.sin(2 * np.pi * breath_hz * np.arange(t0, t1) / fps).astype(np.float32)[:, None, None]
        g = rng.standard_normal((t1 - t0, H, W), dtype=np.float32)
        g *= sig
        g += base[None]
        g += blob[None] * s
        np.rint(g, out=g)
        np.clip(g, 0, 255, out=g)
        out[t0:t1] = g.astype(np.uint8)
    return out


def synth_breathing_blocks(T, H, W, seed=1234, fps=10.0, breath_hz=0.4, amplitude=0.2, noise=0.02,
                           center=(0.6, 0.4), sigma=(0.10, 0.08), block=8, workers=None):
    """The same video model as synth_breathing for the large configurations (4K x 512 is 4.2 G samples): every `block`
    frames draw their noise from their own child generator (SeedSequence(seed).spawn), so blocks are filled by a thread
    pool (numpy releases the GIL) and the result depends only on (T, H, W, seed, block), not on the worker count.
    NOT sample-identical to synth_breathing (one sequential generator there; the goldens use that one)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    rng0 = np.random.Generator(np.random.PCG64(seed))
    tex = _lowpass_noise(rng0, H, W)
    yy = (np.arange(H)[:, None] - center[0] * H) / (sigma[0] * H)
    xx = (np.arange(W)[None, :] - center[1] * W) / (sigma[1] * W)
    blob = (amplitude * 255.0 * np.exp(-0.5 * (yy * yy + xx * xx))).astype(np.float32)
    base = (255.0 * (0.5 + 0.25 * tex)).astype(np.float32)
    sig = np.float32(255.0 * noise)
    out = np.empty((T, H, W), dtype=np.uint8)
    starts = list(range(0, T, block))
    children = np.random.SeedSequence(seed).spawn(len(starts))

    def fill(k):
        t0 = starts[k]
        t1 = min(T, t0 + block)
        rng = np.random.Generator(np.random.PCG64(children[k]))
        s = np.sin(2 * np.pi * breath_hz * np.arange(t0, t1) / fps).astype(np.float32)[:, None, None]
        g = rng.standard_normal((t1 - t0, H, W), dtype=np.float32)
        g *= sig
        g += base[None]
        g += blob[None] * s
        np.rint(g, out=g)
        np.clip(g, 0, 255, out=g)
        out[t0:t1] = g.astype(np.uint8)

    if workers is None:
        workers = max(1, min(32, (os.cpu_count() or 1)))
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(fill, range(len(starts))))
    return out


def synth_breathing_dense(T, H, W, seed=4321, fps=10.0, breath_hz=0.4, amplitude=0.2, noise=0.06, block=8, workers=None,
                          centers=None, sigma=(0.10, 0.08), phase_step=np.pi / 2):
    """A stream that is hard on the pruning of the collapse passes (bench.py `dense_stream`): FOUR breathing blobs spread
    over the frame, a quarter period apart, and three times the sensor noise of synth_breathing -- low-tail voxels
    of the band-passed video are no longer confined to one corner of the image.  Block-parallel like
    synth_breathing_blocks."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    rng0 = np.random.Generator(np.random.PCG64(seed))
    tex = _lowpass_noise(rng0, H, W)
    if centers is None:
        centers = [(0.25, 0.2), (0.3, 0.75), (0.7, 0.35), (0.75, 0.8)]
    sig_y, sig_x = sigma
    blobs = []
    for (cy, cx) in centers:
        yy = (np.arange(H)[:, None] - cy * H) / (sig_y * H)
        xx = (np.arange(W)[None, :] - cx * W) / (sig_x * W)
        blobs.append((amplitude * 255.0 * np.exp(-0.5 * (yy * yy + xx * xx))).astype(np.float32))
    base = (255.0 * (0.5 + 0.25 * tex)).astype(np.float32)
    sig = np.float32(255.0 * noise)
    out = np.empty((T, H, W), dtype=np.uint8)
    starts = list(range(0, T, block))
    children = np.random.SeedSequence(seed).spawn(len(starts))

    def fill(k):
        t0 = starts[k]
        t1 = min(T, t0 + block)
        rng = np.random.Generator(np.random.PCG64(children[k]))
        g = rng.standard_normal((t1 - t0, H, W), dtype=np.float32)
        g *= sig
        g += base[None]
        for q, blob in enumerate(blobs):
            s = np.sin(2 * np.pi * breath_hz * np.arange(t0, t1) / fps + q * phase_step).astype(np.float32)[:, None, None]
            g += blob[None] * s
        np.rint(g, out=g)
        np.clip(g, 0, 255, out=g)
        out[t0:t1] = g.astype(np.uint8)

    if workers is None:
        workers = max(1, min(32, (os.cpu_count() or 1)))
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(fill, range(len(starts))))
    return out


def synth_noise_only(T, H, W, seed=777, noise=0.1, workers=None):
    """Worst case of the collapse passes' pruning (bench.py `worst_case`): the static texture plus sensor noise of sigma 0.1 in
    EVERY pixel and no breathing region at all -- low-tail voxels of the band-passed video turn up in every tile."""
    return synth_breathing_dense(T, H, W, seed=seed, noise=noise, workers=workers, centers=[])


def synth_breathing_16(T, H, W, seed=888, noise=0.04, workers=None):
    """Sixteen small breathing blobs on a 4 x 4 lattice, a sixteenth of a period apart, twice the noise of synth_breathing."""
    centers = [((2 * i + 1) / 8.0, (2 * j + 1) / 8.0) for i in range(4) for j in range(4)]
    return synth_breathing_dense(T, H, W, seed=seed, noise=noise, workers=workers, centers=centers, sigma=(0.05, 0.04), phase_step=np.pi / 8)


def synth_brightness_video(T, H, W, fps=10.0, hz=0.4):
    """Config 1: whole-frame brightness 0.5 + 0.2 sin(2 pi hz t / fps), BGR uint8 [T,H,W,3]."""
    t = np.arange(T)
    lvl = np.clip(np.round(255 * (0.5 + 0.2 * np.sin(2 * np.pi * hz * t / fps))), 0, 255).astype(np.uint8)
    return np.broadcast_to(lvl[:, None, None, None], (T, H, W, 3)).copy()


def synth_texture(H, W, seed=4321, n_waves=12, n_spots=40):
    """Config 3: analytic texture (random 2-D sinusoids + Gaussian spots) as a callable
    f(dx, dy) -> uint8 [H,W] rendered at a sub-pixel shift, so ground-truth flow is known."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kx = rng.uniform(-0.9, 0.9, n_waves)
    ky = rng.uniform(-0.9, 0.9, n_waves)
    ph = rng.uniform(0, 2 * np.pi, n_waves)
    am = rng.uniform(0.3, 1.0, n_waves)
    sx = rng.uniform(0, W, n_spots)
    sy = rng.uniform(0, H, n_spots)
    ss = rng.uniform(1.5, 4.0, n_spots)
    sa = rng.uniform(-1.0, 1.0, n_spots)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

    def render(dx=0.0, dy=0.0):
        x = xx - dx
        y = yy - dy
        v = np.zeros((H, W))
        for i in range(n_waves):
            v += am[i] * np.sin(kx[i] * x + ky[i] * y + ph[i])
        v /= max(1.0, am.sum() / 2)
        for i in range(n_spots):
            v += sa[i] * np.exp(-0.5 * ((x - sx[i]) ** 2 + (y - sy[i]) ** 2) / (ss[i] * ss[i]))
        g = 0.5 + 0.22 * v
        return np.clip(np.round(255 * g), 0, 255).astype(np.uint8)

    return render


class FakeCapture:
    """Stands in for cv2.VideoCapture (reference base.py:48-51, 227-233).
    frames: uint8 [T,H,W,3] BGR (or [T,H,W] gray, returned as 3 equal channels)."""

    def __init__(self, frames, fps=10):
        self._frames = frames
        self._i = 0
        self._fps = fps
        self._open = True

    def isOpened(self):
        return self._open

    def get(self, prop):
        f = self._frames
        if prop == CAP_PROP_FPS:
            return float(self._fps)
        if prop == CAP_PROP_FRAME_WIDTH:
            return float(f.shape[2])
        if prop == CAP_PROP_FRAME_HEIGHT:
            return float(f.shape[1])
        return 0.0

    def read(self):
        if self._i >= len(self._frames):
            return False, None
        f = self._frames[self._i]
        self._i += 1
        if f.ndim == 2:
            f = np.repeat(f[:, :, None], 3, axis=2)
        return True, f

    def release(self):
        self._open = False
