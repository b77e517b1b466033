"""A second, independent CPU restatement of the in-tree arithmetic (numpy float32, vectorised),
used only to cross-check the C oracle: two restatements written separately must agree bit for
bit.  TEST INFRASTRUCTURE (same status as oracle/).

Every intermediate is a float32 array, numpy never fuses a*b+c, np.fmod is C fmodf.
Follows video/hsv/src/hsvutils.rs, hsvfilter/imp.rs:96-119, hsvdetector/imp.rs:130-159 and
video/colorlut/src/colorlut/imp.rs:399-543.
"""
import numpy as np

F = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


def rust_as_u8(v):
    """`f32 as u8`: saturating truncation, NaN -> 0."""
    v = _f(v)
    out = np.zeros(v.shape, dtype=np.uint8)
    ok = ~np.isnan(v)
    out[ok] = np.clip(np.trunc(v[ok]), 0, 255).astype(np.uint8)
    return out


def nan_ignoring_clamp(v, lo, hi):
    """hsvutils.rs:16-38: self.max(lo).min(hi) with f32::max/min (NaN-ignoring)."""
    return np.fmin(np.fmax(_f(v), F(lo)), F(hi))


def from_rgb(R, G, B):
    """hsvutils.rs:44-84 on uint8 arrays holding the true R, G, B."""
    with np.errstate(all="ignore"):
        r, g, b = _f(R) / F(255), _f(G) / F(255), _f(B) / F(255)
        mx = np.maximum(np.maximum(R, G), B)
        mn = np.minimum(np.minimum(R, G), B)
        value = _f(mx) / F(255)
        chroma = value - _f(mn) / F(255)
        eps = F(0.00001)
        is_r = np.abs(value - r) < eps
        is_g = np.abs(value - g) < eps
        is_b = np.abs(value - b) < eps
        hr = F(60) * ((g - b) / chroma)
        hg = F(60) * (F(2) + (b - r) / chroma)
        hb = F(60) * (F(4) + (r - g) / chroma)
        hue = np.where(chroma == 0, F(0), np.where(is_r, hr, np.where(is_g, hg, np.where(is_b, hb, F(0)))))
        hue = _f(hue)
        hue = np.where(hue < 0, hue + F(360), hue).astype(np.float32)
        sat = np.where(value == 0, F(0), chroma / value).astype(np.float32)
        return (np.fmod(hue, F(360)).astype(np.float32), nan_ignoring_clamp(sat, 0, 1),
                nan_ignoring_clamp(value, 0, 1))


def to_rgb(h, s, v):
    """hsvutils.rs:132-163 -> (R, G, B) uint8 arrays."""
    with np.errstate(all="ignore"):
        h, s, v = _f(h), _f(s), _f(v)
        c = v * s
        hp = h / F(60)
        x = c * (F(1) - np.abs(np.fmod(hp, F(2)) - F(1)))
        z = np.zeros_like(c)
        conds = [hp < 0, hp <= 1, hp <= 2, hp <= 3, hp <= 4, hp <= 5, hp <= 6]
        p0 = np.select(conds, [z, c, x, z, z, x, c], default=z)
        p1 = np.select(conds, [z, x, c, c, x, z, z], default=z)
        p2 = np.select(conds, [z, z, z, x, c, c, x], default=z)
        m = v - c
        out = []
        for p in (p0, p1, p2):
            out.append(rust_as_u8(nan_ignoring_clamp((_f(p) + m) * F(255), 0, 255)))
        return out


def hsvfilter_pixels(R, G, B, settings):
    """hsvfilter/imp.rs:96-119 on arrays of true R,G,B."""
    hs, sm, so, vm, vo = [F(x) for x in settings]
    with np.errstate(all="ignore"):
        h, s, v = from_rgb(R, G, B)
        h = np.fmod(h + hs, F(360)).astype(np.float32)
        h = np.where(h < 0, h + F(360), h).astype(np.float32)
        s = nan_ignoring_clamp(sm * s + so, 0, 1)
        v = nan_ignoring_clamp(vm * v + vo, 0, 1)
        return to_rgb(h, s, v)


def hsvdetector_alpha(R, G, B, settings):
    """hsvdetector/imp.rs:138-158 -> uint8 0/255."""
    hue_ref, hue_var, sat_ref, sat_var, val_ref, val_var = [F(x) for x in settings]
    with np.errstate(all="ignore"):
        h, s, v = from_rgb(R, G, B)
        off = F(180) - hue_ref
        sh = (h + off).astype(np.float32)
        sh = np.where(sh < 0, sh + F(360), sh).astype(np.float32)
        sh = np.fmod(sh, F(360)).astype(np.float32)
        hit = (np.abs(sh - F(180)) <= hue_var) & (np.abs(s - sat_ref) <= sat_var) & (np.abs(v - val_ref) <= val_var)
        return np.where(hit, 255, 0).astype(np.uint8)


# ---------------------------------------------------------------- colorlut

def std_clamp01(v):
    """f32::clamp(0,1): NaN propagates."""
    v = _f(v).copy()
    v[v < 0] = 0
    v[v > 1] = 1
    return v


def round_half_away(v):
    v = _f(v)
    t = np.trunc(v)
    return np.where(np.abs(v - t) >= F(0.5), t + np.sign(v), t).astype(np.float32)


def float_to_unorm(v, maxv):
    with np.errstate(all="ignore"):
        r = round_half_away(std_clamp01(v) * F(maxv))
        out = np.zeros(r.shape, dtype=np.uint32)
        ok = ~np.isnan(r)
        out[ok] = np.clip(r[ok], 0, maxv).astype(np.uint32)
        return out


def lattice(values, maxv, scale, offset, size):
    with np.errstate(all="ignore"):
        n = std_clamp01((_f(values) / F(maxv)) * F(scale) + F(offset))
        x = n * (F(size) - F(1))
        fl = np.floor(x)
        idx = np.where(np.isnan(fl), 0, np.clip(np.nan_to_num(fl, nan=0.0), 0, size - 1)).astype(np.int64)
        i1 = np.minimum(idx + 1, size - 1)
        t = (x - idx.astype(np.float32)).astype(np.float32)
        return idx, i1, t


def lerp(a, b, t):
    with np.errstate(all="ignore"):
        return (a + (b - a) * t).astype(np.float32)


def colorlut_3d(rgb_values, maxv, rgba_nodes, size, scale, offset):
    """apply_3d / apply_3d_u16: rgb_values (N,3) integer array -> (N,3) uint32."""
    x0, x1, tx = lattice(rgb_values[:, 0], maxv, scale[0], offset[0], size)
    y0, y1, ty = lattice(rgb_values[:, 1], maxv, scale[1], offset[1], size)
    z0, z1, tz = lattice(rgb_values[:, 2], maxv, scale[2], offset[2], size)
    nodes = rgba_nodes.reshape(-1, 4)[:, :3]

    def at(x, y, z):
        return nodes[x + y * size + z * size * size]

    tx, ty, tz = tx[:, None], ty[:, None], tz[:, None]
    c00 = lerp(at(x0, y0, z0), at(x1, y0, z0), tx)
    c10 = lerp(at(x0, y1, z0), at(x1, y1, z0), tx)
    c01 = lerp(at(x0, y0, z1), at(x1, y0, z1), tx)
    c11 = lerp(at(x0, y1, z1), at(x1, y1, z1), tx)
    c0 = lerp(c00, c10, ty)
    c1 = lerp(c01, c11, ty)
    return float_to_unorm(lerp(c0, c1, tz), maxv)


def colorlut_1d(rgb_values, maxv, tables, size, scale, offset):
    out = np.empty(rgb_values.shape, dtype=np.uint32)
    for c in range(3):
        i0, i1, t = lattice(rgb_values[:, c], maxv, scale[c], offset[c], size)
        tab = _f(tables[c])
        out[:, c] = float_to_unorm(lerp(tab[i0], tab[i1], t), maxv)
    return out


f32 = np.float32

# ---------------------------------------------------------------- image 0.25 grayscale + Lanczos3 resize, image_hasher bits
# Second, independent restatement (numpy float32, vectorised over the axis that is NOT summed; the tap loop stays
# sequential so that every f32 rounding happens in the crate's order).  Shares only libm's sinf with the C oracle
# (numpy's own float32 sin is a different implementation and need not agree with glibc in the last bit).

def _sinf(x):
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.argtypes = [ctypes.c_float]
    libm.sinf.restype = ctypes.c_float
    return f32(libm.sinf(float(x)))


def _lanczos3(x):
    x = f32(x)
    if not abs(x) < f32(3.0):
        return f32(0.0)

    def sinc(t):
        a = f32(t * f32(np.pi))
        return f32(1.0) if t == 0 else f32(_sinf(a) / a)
    return f32(sinc(x) * sinc(f32(x / f32(3.0))))


def lanczos3_taps(in_size, out_size, out):
    ratio = f32(f32(in_size) / f32(out_size))
    sratio = f32(1.0) if ratio < 1 else ratio
    support = f32(f32(3.0) * sratio)
    centre = f32(f32(f32(out) + f32(0.5)) * ratio)
    left = int(np.floor(f32(centre - support)))
    left = min(max(left, 0), in_size - 1)
    right = int(np.ceil(f32(centre + support)))
    right = min(max(right, left + 1), in_size)
    centre = f32(centre - f32(0.5))
    ws = [_lanczos3(f32(f32(f32(i) - centre) / sratio)) for i in range(left, right)]
    total = f32(0.0)
    for w in ws:
        total = f32(total + w)
    return left, np.array([f32(w / total) for w in ws], dtype=np.float32)


def gray_resize_lanczos3(frame, width, height, bpp, nw, nh):
    """frame: height x stride uint8 -> nh x nw uint8"""
    px = frame[:, :width * bpp].reshape(height, width, bpp).astype(np.uint32)
    gray = ((2126 * px[..., 0] + 7152 * px[..., 1] + 722 * px[..., 2]) // 10000).astype(np.uint8)
    if (nw, nh) == (width, height):
        return gray.copy()
    g = gray.astype(np.float32)
    tmp = np.zeros((nh, width), np.float32)
    for oy in range(nh):
        left, ws = lanczos3_taps(height, nh, oy)
        t = np.zeros(width, np.float32)
        for i, w in enumerate(ws):
            t = t + g[left + i] * w          # float32 arrays: mul rounds, then add rounds
        tmp[oy] = t
    out = np.zeros((nh, nw), np.uint8)
    for ox in range(nw):
        left, ws = lanczos3_taps(width, nw, ox)
        t = np.zeros(nh, np.float32)
        for i, w in enumerate(ws):
            t = t + tmp[:, left + i] * w
        t = np.where(t < 0, f32(0), np.where(t > 255, f32(255), t))
        r = np.sign(t) * np.floor(np.abs(t) + f32(0.5))   # round half away from zero (t >= 0 here)
        out[:, ox] = np.nan_to_num(r, nan=0.0).astype(np.uint8)
    return out


def image_hash_bits(px, algo):
    """px: the resized gray image (rows x cols uint8); returns the list of bools in the crate's iteration order"""
    rows, cols = px.shape
    bits = []
    if algo == "mean":
        mean = int(px.astype(np.uint32).sum()) // px.size
        bits = [int(v) >= mean for v in px.reshape(-1)]
    if algo in ("gradient", "doublegradient"):
        bits += [bool(px[y, x] < px[y, x + 1]) for y in range(rows) for x in range(cols - 1)]
    if algo in ("vertgradient", "doublegradient"):
        bits += [bool(px[y, x] < px[y + 1, x]) for x in range(cols) for y in range(rows - 1)]
    return bits


HASH_RESIZE = {"mean": (8, 8), "gradient": (9, 8), "vertgradient": (8, 9), "doublegradient": (5, 5)}


# ---- image_hasher 3.1.1 blockhash, f32 slow path (sizes not divisible by 8) -------------------------------------
def blockhash_slow_sums(frame, width, height, bpp):
    """Second restatement of `blockhash_slow` (see oracle/videofx_oracle.c for the algorithm): returns float32[64].
    Written differently from the C loop on purpose: pixels are bucketed per block first and every block's chain of f32
    additions is one np.cumsum (sequential float32 accumulation); the four weighted terms of a pixel are interleaved in
    the crate's order.  Terms that are exactly +0.0 are kept (adding them is what the crate does)."""
    px = frame[:height, :width * bpp].reshape(height, width, bpp).astype(np.uint32)
    v = px[..., 0] + px[..., 1] + px[..., 2]
    if bpp == 4:
        v = np.where(px[..., 3] == 0, np.uint32(765), v)
    p = v.astype(np.float32)
    bw, bh = F(width) / F(8), F(height) / F(8)
    xs, ys = np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32)
    mx, my = np.fmod(F(1), bw), np.fmod(F(1), bh)
    xm, ym = xs + mx, ys + my
    wl, wt = xm - np.trunc(xm), ym - np.trunc(ym)
    wr, wb = F(1) - wl, F(1) - wt
    left, top = np.floor(xs / bw).astype(np.int64), np.floor(ys / bh).astype(np.int64)
    right = np.where(np.trunc(xm) == 0, np.ceil(xs / bw).astype(np.int64), left)
    bottom = np.where(np.trunc(ym) == 0, np.ceil(ys / bh).astype(np.int64), top)
    assert right.max() < 8 and bottom.max() < 8
    # term k of pixel (y, x): (block index, value), k = 0..3 in the crate's order
    terms = [(top[:, None] * 8 + left[None, :], (p * wl[None, :]) * wt[:, None]),
             (bottom[:, None] * 8 + left[None, :], (p * wl[None, :]) * wb[:, None]),
             (top[:, None] * 8 + right[None, :], (p * wr[None, :]) * wt[:, None]),
             (bottom[:, None] * 8 + right[None, :], (p * wr[None, :]) * wb[:, None])]
    blk = np.stack([t[0] for t in terms], axis=-1).reshape(-1)       # raster order, 4 terms per pixel
    val = np.stack([t[1] for t in terms], axis=-1).astype(np.float32).reshape(-1)
    out = np.zeros(64, dtype=np.float32)
    for b in range(64):
        chain = val[blk == b]
        if chain.size:
            out[b] = np.cumsum(chain, dtype=np.float32)[-1]
    return out


def blockhash_slow_bits(blocks, width, height):
    """gen_hash! for the f32 path: upper median of each group of 16, FLOAT_EQ_MARGIN 0.001."""
    blocks = np.asarray(blocks, dtype=np.float32)
    area = (F(width) / F(8)) * (F(height) / F(8))
    cmp_factor = F(765) * area / F(2)
    h = 0
    for band in range(4):
        g = blocks[16 * band:16 * band + 16]
        median = np.sort(g)[8]
        for i in range(16):
            if g[i] > median or (abs(F(g[i] - median)) < F(0.001) and median > cmp_factor):
                h |= 1 << (16 * band + i)
    return h
