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
