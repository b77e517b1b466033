"""Synthetic frame generators shared by tests, smoke and bench (SURVEY.md 8d inputs)."""
import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64_bytes(seed: int, n: int) -> np.ndarray:
    """n iid uniform bytes from splitmix64(seed) (vectorised)."""
    m = (n + 7) // 8
    idx = np.arange(1, m + 1, dtype=np.uint64)
    z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def random_frame(seed: int, width: int, height: int, bpp: int = 4, stride: int | None = None) -> np.ndarray:
    """height x stride uint8, every byte (padding included) random."""
    stride = stride if stride is not None else width * bpp
    return splitmix64_bytes(seed, stride * height).reshape(height, stride)


def natural_like(width: int, height: int, seed: int = 0x5EED0004) -> np.ndarray:
    """RGBA frame of smooth 2-D colour gradients + seeded noise of +-3 codes per channel (what bench.py calls "natural-like": neighbouring
    pixels fall into the same or adjacent LUT cells, never into exactly the same colour), alpha = a hash of the position."""
    x = np.linspace(0.0, 1.0, width, dtype=np.float64)[None, :]
    y = np.linspace(0.0, 1.0, height, dtype=np.float64)[:, None]
    ph = 0.37 * (seed & 15)
    f = np.empty((height, width, 4), dtype=np.uint8)
    chans = (0.5 + 0.45 * np.sin(3.0 * x + 2.0 * y + ph), 0.5 + 0.45 * np.sin(5.0 * y - 1.5 * x + 2 * ph), 0.5 + 0.45 * np.cos(4.0 * x * y + ph))
    noise = splitmix64_bytes(seed, width * height * 4).reshape(height, width, 4)
    for c in range(3):
        v = np.broadcast_to(chans[c], (height, width)) * 255.0 + (noise[..., c] % 7).astype(np.float64) - 3.0
        f[..., c] = np.clip(v, 0, 255).astype(np.uint8)
    f[..., 3] = noise[..., 3]
    return f.reshape(height, width * 4)


def exhaustive_rgbx() -> np.ndarray:
    """4096x4096 4-byte frame holding all 2^24 (c0,c1,c2) triples; byte 3 =
    (i*2654435761 mod 2^32)>>24 must come through untouched (SURVEY.md 8d iii)."""
    i = np.arange(1 << 24, dtype=np.uint64)
    px = np.empty((1 << 24, 4), dtype=np.uint8)
    px[:, 0] = (i & np.uint64(255)).astype(np.uint8)
    px[:, 1] = ((i >> np.uint64(8)) & np.uint64(255)).astype(np.uint8)
    px[:, 2] = ((i >> np.uint64(16)) & np.uint64(255)).astype(np.uint8)
    px[:, 3] = (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)).astype(np.uint8)
    return px.reshape(4096, 4096 * 4)


def smpte_like(width: int, height: int, seed: int = 0x5EED0003) -> np.ndarray:
    """RGBA bars (top 2/3) + grey ramp + seeded snow strip, videotestsrc-smpte-like."""
    bars = np.array([[191, 191, 191], [191, 191, 0], [0, 191, 191], [0, 191, 0],
                     [191, 0, 191], [191, 0, 0], [0, 0, 191]], dtype=np.uint8)
    f = np.empty((height, width, 4), dtype=np.uint8)
    f[..., 3] = 255
    xs = (np.arange(width) * 7 // max(width, 1)).clip(0, 6)
    f[:, :, :3] = bars[xs][None, :, :]
    y0 = height * 2 // 3
    ramp = (np.arange(width) * 255 // max(width - 1, 1)).astype(np.uint8)
    f[y0:, :, 0] = ramp
    f[y0:, :, 1] = ramp
    f[y0:, :, 2] = ramp
    y1 = height * 11 // 12
    snow = splitmix64_bytes(seed, (height - y1) * width).reshape(height - y1, width)
    f[y1:, :, 0] = snow
    f[y1:, :, 1] = snow
    f[y1:, :, 2] = snow
    return f.reshape(height, width * 4)


# ---------------------------------------------------------------- videotestsrc pattern=smpte, RGBA
# The frames BASELINE.json's workload names ("synthetic videotestsrc buffers").  Restated from the observable output of
# GStreamer 1.14's videotestsrc (checked byte for byte against the element itself in tests/test_videotestsrc_frames_cpu.py,
# several sizes, several consecutive frames): seven 100 % bars over the top 2/3, the reversed blue/black/magenta/black/cyan/
# black/white strip down to 3/4, then -I / white / +Q in sixths, super-black / black / dark grey in twelfths and, in the last
# quarter of the width, "snow": grey = bits 16..23 of the C library style LCG  s <- s * 1103515245 + 12345 (mod 2^32), whose
# state starts at 0 and runs on from frame to frame.

VTS_LCG_A, VTS_LCG_C = 1103515245, 12345
_VTS_BARS = np.array([[255, 255, 255], [255, 255, 0], [0, 255, 255], [0, 255, 0], [255, 0, 255], [255, 0, 0], [0, 0, 255]], dtype=np.uint8)
_VTS_BLACK = np.array([0, 0, 0], dtype=np.uint8)


def vts_lcg_affine(n: int):
    """(A_i, C_i) for i = 1..n with  s_{k+i} = A_i * s_k + C_i  (mod 2^32), as uint64 arrays (built by doubling)."""
    a = np.empty(max(n, 1), dtype=np.uint64)
    c = np.empty(max(n, 1), dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    a[0], c[0] = VTS_LCG_A, VTS_LCG_C
    have = 1
    while have < n:
        take = min(have, n - have)
        ah, ch = a[have - 1], c[have - 1]  # the `have`-step map
        a[have:have + take] = (a[:take] * ah) & m
        c[have:have + take] = (a[:take] * ch + c[:take]) & m
        have += take
    return a[:n], c[:n]


def vts_snow_geometry(width: int, height: int):
    """(x0, y0): the snow rectangle is columns x0..width of rows y0..height."""
    return width * 3 // 4, height * 3 // 4


def videotestsrc_smpte(width: int, height: int, n_frames: int = 1, state: int = 0):
    """(frames[n, height, width*4] uint8 RGBA, LCG state after the last frame)."""
    base = np.empty((height, width, 4), dtype=np.uint8)
    base[..., 3] = 255
    y1, y2 = 2 * height // 3, 3 * height // 4
    for i in range(7):
        xa, xb = i * width // 7, (i + 1) * width // 7
        base[:y1, xa:xb, :3] = _VTS_BARS[i]
        base[y1:y2, xa:xb, :3] = _VTS_BLACK if i & 1 else _VTS_BARS[6 - i]
    for i, col in enumerate(([0, 0, 128], [255, 255, 255], [0, 128, 255])):
        base[y2:, i * width // 6:(i + 1) * width // 6, :3] = col
    for i, g in enumerate((0, 0, 19)):
        base[y2:, width // 2 + i * width // 12:width // 2 + (i + 1) * width // 12, :3] = g
    x0, y0 = vts_snow_geometry(width, height)
    per_frame = (width - x0) * (height - y0)
    a, c = vts_lcg_affine(per_frame)
    out = np.empty((n_frames, height, width, 4), dtype=np.uint8)
    for f in range(n_frames):
        out[f] = base
        if per_frame:
            s = (a * np.uint64(state) + c) & np.uint64(0xFFFFFFFF)
            grey = ((s >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8).reshape(height - y0, width - x0)
            out[f, y0:, x0:, :3] = grey[..., None]
            state = int(s[-1])
    return out.reshape(n_frames, height, width * 4), state
