"""Seeded synthetic video: the reference ships no inputs.

Base texture = three octaves of seeded value noise + 200 random rectangles, so every tile has
both |gx| and |gy| maxima.  Frame t = base texture resampled (bilinear) under a similarity
about the frame centre with known per-frame jitter, so the true motion is known.

numpy only (CPU): used by the tests at small sizes.  bench.py has a device-side twin built on
torch ops for the full-size clips; both paths hand identical bytes to the GPU path and to the
CPU oracle (the bench copies the device frames to the host for the CPU leg).
"""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(x):
    """vectorised splitmix64 on uint64 arrays"""
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def _lattice(seed, octave, ny, nx):
    iy, ix = np.meshgrid(np.arange(ny, dtype=np.uint64), np.arange(nx, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(octave) * np.uint64(0x9E3779B1)
               + iy * np.uint64(0x1F123BB5) + ix * np.uint64(0x5BD1E995))
        r = splitmix64(key)
    return (r >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)


def base_texture(width, height, seed, max_value=255):
    """(height, width) float64 texture in [0, max_value]"""
    tex = np.zeros((height, width), np.float64)
    ys = np.arange(height, dtype=np.float64)
    xs = np.arange(width, dtype=np.float64)
    for octave, (period, amp) in enumerate(((64, 96.0), (16, 48.0), (4, 24.0))):
        ny, nx = height // period + 2, width // period + 2
        lat = _lattice(seed, octave, ny, nx)
        fy, fx = ys / period, xs / period
        y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
        wy, wx = (fy - y0)[:, None], (fx - x0)[None, :]
        a = lat[y0][:, x0]
        b = lat[y0][:, x0 + 1]
        c = lat[y0 + 1][:, x0]
        d = lat[y0 + 1][:, x0 + 1]
        tex += amp * ((a * (1 - wx) + b * wx) * (1 - wy) + (c * (1 - wx) + d * wx) * wy)
    tex += 40.0
    rng = np.random.Generator(np.random.PCG64(seed * 7919 + 17))
    for _ in range(200):
        rw, rh = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        rx, ry = int(rng.integers(0, max(1, width - rw))), int(rng.integers(0, max(1, height - rh)))
        tex[ry:ry + rh, rx:rx + rw] = float(rng.integers(0, 256))
    tex = np.clip(tex, 0, 255)
    return tex * (max_value / 255.0)


def sample_bilinear(tex, t, out_w, out_h, margin):
    """Sample tex (which carries `margin` extra pixels on every side) at W(p) about the frame centre."""
    A, B, TX, TY = t
    cx, cy = out_w * 0.5, out_h * 0.5
    y, x = np.meshgrid(np.arange(out_h, dtype=np.float64), np.arange(out_w, dtype=np.float64), indexing="ij")
    px, py = x - cx, y - cy
    sx = (1 + A) * px - B * py + cx + TX + margin
    sy = B * px + (1 + A) * py + cy + TY + margin
    x0 = np.clip(np.floor(sx).astype(np.int64), 0, tex.shape[1] - 2)
    y0 = np.clip(np.floor(sy).astype(np.int64), 0, tex.shape[0] - 2)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)
    v = (tex[y0, x0] * (1 - fx) + tex[y0, x0 + 1] * fx) * (1 - fy) + (tex[y0 + 1, x0] * (1 - fx) + tex[y0 + 1, x0 + 1] * fx) * fy
    return v


def camera_path(n_frames, seed, pan=0.5, jitter_t=4.0, jitter_b=0.002, jitter_a=0.001):
    """per-frame (A,B,TX,TY) sampling transforms: slow pan + i.i.d. jitter"""
    rng = np.random.Generator(np.random.PCG64(seed * 104729 + 3))
    path = []
    for t in range(n_frames):
        path.append((rng.uniform(-jitter_a, jitter_a), rng.uniform(-jitter_b, jitter_b),
                     pan * t + rng.uniform(-jitter_t, jitter_t), rng.uniform(-jitter_t, jitter_t)))
    return path


def make_clip(width, height, n_frames, seed, channels=1, bits=8, path=None, margin=128, **path_kw):
    """returns (frames, path): frames (n, h, w) or (n, h, w, 3), uint8 (bits=8) or uint16 (bits=10)"""
    max_value = 255 if bits == 8 else (1 << bits) - 1
    dtype = np.uint8 if bits == 8 else np.uint16
    if path is None:
        path = camera_path(n_frames, seed, **path_kw)
    texs = [base_texture(width + 2 * margin, height + 2 * margin, seed + c, max_value) for c in range(channels)]
    shape = (n_frames, height, width) if channels == 1 else (n_frames, height, width, channels)
    frames = np.empty(shape, dtype)
    for i, t in enumerate(path):
        for c in range(channels):
            v = np.floor(sample_bilinear(texs[c], t, width, height, margin) + 0.5)
            v = np.clip(v, 0, max_value).astype(dtype)
            if channels == 1:
                frames[i] = v
            else:
                frames[i, :, :, c] = v
    return frames, path


class TorchClipFactory:
    """Device-side twin of make_clip for full-size clips: the base textures (numpy, uploaded once) are shared by
    every clip of the factory, each clip has its own camera path (path_seed); the per-frame bilinear resampling
    runs as torch ops on `device`.  torch is plumbing here: it only produces the input bytes."""

    def __init__(self, width, height, seed, device, channels=3, bits=8, margin=128):
        import torch
        self.torch = torch
        self.w, self.h, self.channels, self.bits, self.margin, self.device = width, height, channels, bits, margin, device
        self.max_value = 255 if bits == 8 else (1 << bits) - 1
        self.texs = [torch.from_numpy(base_texture(width + 2 * margin, height + 2 * margin, seed + c, self.max_value)).to(device)
                     for c in range(channels)]
        self.ys = torch.arange(height, dtype=torch.float64, device=device)[:, None]
        self.xs = torch.arange(width, dtype=torch.float64, device=device)[None, :]

    def make(self, n_frames, path_seed, path=None, out=None, **path_kw):
        torch = self.torch
        if path is None:
            path = camera_path(n_frames, path_seed, **path_kw)
        th, tw = self.texs[0].shape
        dt = torch.uint8 if self.bits == 8 else torch.int16      # int16 carries the u16 bit pattern (values < 32768)
        if out is None:
            out = torch.empty((n_frames, self.h, self.w, self.channels), dtype=dt, device=self.device)
        cx, cy = self.w * 0.5, self.h * 0.5
        px, py = self.xs - cx, self.ys - cy
        for i, (A, B, TX, TY) in enumerate(path):
            sx = (1 + A) * px - B * py + cx + TX + self.margin
            sy = B * px + (1 + A) * py + cy + TY + self.margin
            x0 = torch.clamp(torch.floor(sx).long(), 0, tw - 2)
            y0 = torch.clamp(torch.floor(sy).long(), 0, th - 2)
            fx = torch.clamp(sx - x0, 0, 1)
            fy = torch.clamp(sy - y0, 0, 1)
            i00 = y0 * tw + x0
            for c in range(self.channels):
                t = self.texs[c].reshape(-1)
                v = (t[i00] * (1 - fx) + t[i00 + 1] * fx) * (1 - fy) + (t[i00 + tw] * (1 - fx) + t[i00 + tw + 1] * fx) * fy
                out[i, :, :, c] = torch.clamp(torch.floor(v + 0.5), 0, self.max_value).to(dt)
        return out, path


def make_clip_torch(width, height, n_frames, seed, device, channels=3, bits=8, path=None, margin=128, **path_kw):
    """one clip: (tensor (n,h,w,c) uint8 / int16-viewed u16, path)"""
    return TorchClipFactory(width, height, seed, device, channels, bits, margin).make(n_frames, seed, path=path, **path_kw)
