"""oracle/ssim_oracle.c (hash-algo=dssim restatement, PARITY UNPINNED against dssim-core 3.4.0):
what CAN be pinned here -- the reference test's property (identical frames -> 0,
tests/videocompare.rs:141-182), symmetry, monotonicity, and agreement with an independent numpy
restatement of the same published structure."""
import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

W5 = [0.028, 0.197, 0.322, 0.298, 0.155]


def np_ssim_distance(a, b, w, h, bpp):
    """Independent numpy f64 restatement (separable blur instead of the 25-tap window)."""
    def lin(f):
        px = f[:, :w * bpp].reshape(h, w, bpp).astype(np.float64) / 255.0
        rgb = px[..., :3]
        out = np.where(rgb <= 0.04045, rgb / 12.92, ((rgb + 0.055) / 1.055) ** 2.4)
        if bpp == 4:
            out = out * px[..., 3:4]
        return out

    def lab(l):
        r, g, bl = l[..., 0], l[..., 1], l[..., 2]
        X = (0.4124 * r + 0.3576 * g + 0.1805 * bl) / 0.9505
        Y = 0.2126 * r + 0.7152 * g + 0.0722 * bl
        Z = (0.0193 * r + 0.1192 * g + 0.9505 * bl) / 1.089
        f = lambda t: np.where(t > 216.0 / 24389.0, np.cbrt(t), (24389.0 / 27.0 * t + 16.0) / 116.0)
        fx, fy, fz = f(X), f(Y), f(Z)
        return np.stack([(116 * fy - 16) / 100, (86.2 + 500 * (fx - fy)) / 220, (107.9 + 200 * (fy - fz)) / 220], -1)

    k = np.array([1, 4, 6, 4, 1], np.float64) / 16

    def blur(p):
        pp = np.pad(p, ((2, 2), (2, 2)), mode="edge")
        t = sum(k[i] * pp[i:i + p.shape[0], :] for i in range(5))
        return sum(k[i] * t[:, i:i + p.shape[1]] for i in range(5))

    la, lb = lin(a), lin(b)
    num = den = 0.0
    for s in range(5):
        if s:
            if la.shape[1] // 2 < 8 or la.shape[0] // 2 < 8:
                break
            hh, ww = la.shape[0] // 2 * 2, la.shape[1] // 2 * 2
            ds = lambda l: (l[0:hh:2, 0:ww:2] + l[0:hh:2, 1:ww:2] + l[1:hh:2, 0:ww:2] + l[1:hh:2, 1:ww:2]) * 0.25
            la, lb = ds(la), ds(lb)
        A, B = lab(la), lab(lb)
        acc = 0.0
        for c in range(3):
            x, y = A[..., c], B[..., c]
            m1, m2 = blur(x), blur(y)
            s11, s22, s12 = blur(x * x) - m1 * m1, blur(y * y) - m2 * m2, blur(x * y) - m1 * m2
            acc = acc + ((2 * m1 * m2 + 1e-4) * (2 * s12 + 9e-4)) / ((m1 * m1 + m2 * m2 + 1e-4) * (s11 + s22 + 9e-4))
        m = acc / 3.0
        score = m.mean() - np.abs(m - m.mean()).mean()
        num += W5[s] * score
        den += W5[s]
    return 1.0 / (num / den) - 1.0


def _pair(seed, w, h, bpp, amp, every=97):
    a = frames.random_frame(seed, w, h, bpp)
    b = a.copy()
    flat = b.reshape(-1)
    idx = np.arange(0, flat.size, every)
    flat[idx] = np.clip(flat[idx].astype(np.int32) + amp, 0, 255).astype(np.uint8)
    return a, b


@pytest.mark.parametrize("fmt,bpp", [("RGBA", 4), ("RGB", 3)])
def test_identical_frames_have_distance_exactly_zero(fmt, bpp):
    a = frames.random_frame(0xD551, 80, 64, bpp)
    rc, d, per = orc.ssim_distance(a, a.copy(), 80, 64, 80 * bpp, 80 * bpp, fmt)
    assert rc == 0 and d == 0.0
    assert per[:4] == [1.0] * 4 and np.isnan(per[4])  # 80x64: scale 4 would be 5x4 < 8 px


def test_symmetry_and_monotone_ladder():
    w, h = 128, 96
    last = 0.0
    for amp in (1, 3, 10, 40, 120):
        a, b = _pair(0xD552, w, h, 4, amp)
        rc, d, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
        rc2, d2, _ = orc.ssim_distance(b, a, w, h, w * 4, w * 4, "RGBA")
        assert rc == 0 and rc2 == 0
        assert d == pytest.approx(d2, rel=1e-12)
        assert d > last
        last = d


@pytest.mark.parametrize("fmt,bpp,w,h", [("RGBA", 4, 64, 48), ("RGB", 3, 131, 77), ("RGBA", 4, 160, 160)])
def test_matches_independent_numpy_restatement(fmt, bpp, w, h):
    a, b = _pair(0xD553 + w, w, h, bpp, 25, every=13)
    rc, d, _ = orc.ssim_distance(a, b, w, h, w * bpp, w * bpp, fmt)
    assert rc == 0
    assert d == pytest.approx(np_ssim_distance(a, b, w, h, bpp), rel=1e-9)


def test_stride_padding_is_ignored_and_bad_inputs_rejected():
    w, h = 40, 32
    a = frames.random_frame(1, w, h, 4, stride=w * 4 + 12)
    b = a.copy()
    b[:, w * 4:] ^= 0xFF  # padding differs only
    rc, d, _ = orc.ssim_distance(a, b, w, h, w * 4 + 12, w * 4 + 12, "RGBA")
    assert rc == 0 and d == 0.0
    assert orc.ssim_distance(a, b, w, h, w * 4 + 12, w * 4 + 12, "BGRA")[0] != 0
    assert orc.ssim_distance(a, b, 4, 4, 16, 16, "RGBA")[0] != 0


def test_band_sums_add_up_to_the_whole():
    w, h = 96, 80
    a, b = _pair(0xD554, w, h, 4, 30, every=11)
    whole_s, whole_c, n = orc.ssim_band(a, b, w, h, w * 4, w * 4, "RGBA", 0, h)
    parts = [orc.ssim_band(a, b, w, h, w * 4, w * 4, "RGBA", r0, r1) for r0, r1 in ((0, 32), (32, 64), (64, h))]
    for s in range(n):
        assert sum(p[1][s] for p in parts) == whole_c[s]
        assert sum(p[0][s] for p in parts) == pytest.approx(whole_s[s], rel=1e-13)
    mean = [whole_s[s] / whole_c[s] if whole_c[s] else 0.0 for s in range(5)]
    dev = orc.ssim_band(a, b, w, h, w * 4, w * 4, "RGBA", 0, h, mean)[0]
    mad = [dev[s] / whole_c[s] if whole_c[s] else 0.0 for s in range(5)]
    rc, d, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    assert orc.ssim_combine(mean, mad, n) == pytest.approx(d, rel=1e-13)
