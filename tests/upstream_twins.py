"""Restatements of the UPSTREAM PUBLISHED algorithms the third-party crates descend from, written from those texts and not from
SURVEY.md Appendix A (which is what oracle/videofx_oracle.c and host/mmcq.cpp were written from): a second line of descent for the
rows the reference itself cannot pin.  TEST INFRASTRUCTURE, CPU only.

  * `mmcq_java_palette`: Modified Median Cut Quantization in the lineage color-thief-rs 0.2.2 names: Leptonica's colorquant2.c
    (Dan Bloomberg) -> quantize.js (Nick Rabinowitz) -> Color Thief's Java port MMCQ.java (Sven Woltmann), plus Color Thief's pixel
    filter (alpha >= 125, not r,g,b > 250).  Functions carry those texts' names (getHisto, vboxFromPixels, medianCutApply / doCut,
    iterate, VBox.avg).  The Java port's iterate() -- which the Rust crate follows -- differs from quantize.js's iter() in two
    published ways that change results, both reproduced here and both found in the oracle by this twin: the colour counter restarts
    at 1 in the second phase whose target is `maxcolors - size`, so the second phase ALWAYS splits once more (max_colors = 2 yields
    three boxes), and the list is reversed and (in the crate) truncated to max_colors.  `quantize_js_palette` keeps the quantize.js
    control flow for comparison: it gives the same boxes whenever that extra split does not reach the first max_colors entries.
  * `blockhash_py_even`: the even-size path of blockhash.io's reference implementation blockhash.py (blockhash_even,
    translate_blocks_to_bits, total_value_rgba) -- the specification image_hasher's Blockhash cites (the reference's README.md:462
    points at it).  KNOWN, DOCUMENTED DIFFERENCE: blockhash.py's median of a band of 16 is the mean of the two middle values and
    its brightness bound is pixels * 256 * 3 / 2; image_hasher 3.1.1 takes the upper median sorted[8] and 765 * pixels / 2.  The
    block sums are the same quantity in both; the bits can differ only for blocks whose sum EQUALS the upper median.

Nothing here is imported by the product; tests/test_upstream_twins_cpu.py compares the oracle with these."""
import math

import numpy as np

SIGBITS = 5
RSHIFT = 8 - SIGBITS
MAX_ITERATIONS = 1000
FRACT_BY_POPULATIONS = 0.75


def _color_index(r, g, b):
    return (r << (2 * SIGBITS)) + (g << SIGBITS) + b


class VBox:
    def __init__(self, r1, r2, g1, g2, b1, b2, histo):
        self.r1, self.r2, self.g1, self.g2, self.b1, self.b2, self.histo = r1, r2, g1, g2, b1, b2, histo

    def copy(self):
        return VBox(self.r1, self.r2, self.g1, self.g2, self.b1, self.b2, self.histo)

    def volume(self):
        return (self.r2 - self.r1 + 1) * (self.g2 - self.g1 + 1) * (self.b2 - self.b1 + 1)

    def count(self):
        h = self.histo.reshape(32, 32, 32)
        return int(h[self.r1:self.r2 + 1, self.g1:self.g2 + 1, self.b1:self.b2 + 1].sum())

    def avg(self):
        mult = 1 << (8 - SIGBITS)
        ntot = 0
        rsum = gsum = bsum = 0.0
        h = self.histo.reshape(32, 32, 32)
        for i in range(self.r1, self.r2 + 1):
            for j in range(self.g1, self.g2 + 1):
                for k in range(self.b1, self.b2 + 1):
                    hval = int(h[i, j, k])
                    ntot += hval
                    rsum += hval * (i + 0.5) * mult
                    gsum += hval * (j + 0.5) * mult
                    bsum += hval * (k + 0.5) * mult
        if ntot:
            return (int(rsum / ntot), int(gsum / ntot), int(bsum / ntot))
        return (int(mult * (self.r1 + self.r2 + 1) / 2), int(mult * (self.g1 + self.g2 + 1) / 2), int(mult * (self.b1 + self.b2 + 1) / 2))


def get_histo_and_vbox(pixels_rgb):
    """getHisto + vboxFromPixels over an (n, 3) uint8 array of the pixels that passed Color Thief's filter"""
    q = (pixels_rgb >> RSHIFT).astype(np.int64)
    histo = np.bincount(_color_index(q[:, 0], q[:, 1], q[:, 2]), minlength=1 << (3 * SIGBITS)).astype(np.int64)
    return histo, VBox(int(q[:, 0].min()), int(q[:, 0].max()), int(q[:, 1].min()), int(q[:, 1].max()), int(q[:, 2].min()), int(q[:, 2].max()), histo)


def median_cut_apply(histo, vbox):
    if not vbox.count():
        return None
    rw, gw, bw = vbox.r2 - vbox.r1 + 1, vbox.g2 - vbox.g1 + 1, vbox.b2 - vbox.b1 + 1
    maxw = max(rw, gw, bw)
    if vbox.count() == 1:
        return [vbox.copy(), None]
    h = histo.reshape(32, 32, 32)[vbox.r1:vbox.r2 + 1, vbox.g1:vbox.g2 + 1, vbox.b1:vbox.b2 + 1]
    if maxw == rw:
        color, lo, hi, sums = "r", vbox.r1, vbox.r2, h.sum(axis=(1, 2))
    elif maxw == gw:
        color, lo, hi, sums = "g", vbox.g1, vbox.g2, h.sum(axis=(0, 2))
    else:
        color, lo, hi, sums = "b", vbox.b1, vbox.b2, h.sum(axis=(0, 1))
    partialsum = {lo + i: int(v) for i, v in enumerate(np.cumsum(sums))}
    total = partialsum[hi]
    lookaheadsum = {i: total - d for i, d in partialsum.items()}
    for i in range(lo, hi + 1):
        if partialsum[i] > total / 2:
            vbox1, vbox2 = vbox.copy(), vbox.copy()
            left, right = i - lo, hi - i
            if left <= right:
                d2 = min(hi - 1, int(i + right / 2))
            else:
                d2 = max(lo, int(i - 1 - left / 2))
            while not partialsum.get(d2, 0):      # avoid 0-count boxes
                d2 += 1
            count2 = lookaheadsum[d2]
            while not count2 and partialsum.get(d2 - 1, 0):
                d2 -= 1
                count2 = lookaheadsum[d2]
            setattr(vbox1, color + "2", d2)
            setattr(vbox2, color + "1", d2 + 1)
            return [vbox1, vbox2]
    return None


def quantize(pixels_rgb, maxcolors):
    """quantize(pixels, maxcolors) -> list of (r, g, b), in CMap.palette() order"""
    if len(pixels_rgb) == 0 or maxcolors < 2 or maxcolors > 256:
        return []
    histo, vbox = get_histo_and_vbox(pixels_rgb)
    pq = [vbox]  # PQueue: sorted ascending by the comparator, pop() takes the last = largest

    def iterate(lh, key, target):
        ncolors, niters = len(lh), 0
        while niters < MAX_ITERATIONS:
            if ncolors >= target:
                return
            niters += 1
            if niters - 1 > MAX_ITERATIONS:
                return
            lh.sort(key=key)
            vb = lh.pop()
            if not vb.count():
                lh.append(vb)
                niters += 1
                continue
            boxes = median_cut_apply(histo, vb)
            if not boxes or boxes[0] is None:
                return
            lh.append(boxes[0])
            if boxes[1] is not None:
                lh.append(boxes[1])
                ncolors += 1

    iterate(pq, lambda b: b.count(), FRACT_BY_POPULATIONS * maxcolors)
    pq2 = list(pq)
    iterate(pq2, lambda b: b.count() * b.volume(), maxcolors)
    pq2.sort(key=lambda b: b.count() * b.volume())
    cmap = []
    while pq2:
        cmap.append(pq2.pop().avg())  # largest count * volume first
    return cmap


def mmcq_java(pixels_rgb, maxcolors):
    """MMCQ.java quantize(): -> list of (r, g, b), most significant box first (Collections.reverse), NOT truncated"""
    if len(pixels_rgb) == 0 or maxcolors < 2 or maxcolors > 256:
        return []
    histo, vbox = get_histo_and_vbox(pixels_rgb)
    pq = [vbox]

    def by_count(b):
        return b.count()

    def product_key(b):           # COMPARATOR_PRODUCT: equal counts -> by volume, else by count * volume (same order as (product, ...))
        return (b.count() * b.volume(), b.volume())

    def iterate(lh, key, target):
        ncolors, niters = 1, 0
        while niters < MAX_ITERATIONS:
            vb = lh[-1]
            if vb.count() == 0:
                lh.sort(key=key)
                niters += 1
                continue
            lh.pop()
            boxes = median_cut_apply(histo, vb)
            if not boxes or boxes[0] is None:
                raise RuntimeError("vbox1 not defined; shouldn't happen!")
            lh.append(boxes[0])
            if boxes[1] is not None:
                lh.append(boxes[1])
                ncolors += 1
            lh.sort(key=key)          # Collections.sort: stable, like list.sort
            if ncolors >= target:
                return
            niters += 1
            if niters - 1 > MAX_ITERATIONS:
                return

    iterate(pq, by_count, math.ceil(FRACT_BY_POPULATIONS * maxcolors))
    pq.sort(key=product_key)
    iterate(pq, product_key, maxcolors - len(pq))
    pq.reverse()
    return [b.avg() for b in pq]


def mmcq_java_palette(plane, fmt, quality, max_colors):
    """color-thief-rs get_palette: MMCQ.java's boxes, truncated to max_colors, packed 0xRRGGBB"""
    return [(r << 16) | (g << 8) | b for r, g, b in mmcq_java(color_thief_pixels(plane, fmt, quality), max_colors)[:max_colors]]


def color_thief_pixels(plane, fmt, quality):
    """color-thief.js getPalette's pixel loop over the flat byte plane: every `quality`-th pixel with alpha >= 125 that is not white"""
    bpp = 4 if fmt in ("RGBA", "BGRA", "ARGB") else 3
    flat = np.ascontiguousarray(plane).reshape(-1)
    n = flat.size // bpp
    px = flat[:n * bpp].reshape(n, bpp)[::quality]
    order = {"RGB": (0, 1, 2, None), "RGBA": (0, 1, 2, 3), "ARGB": (1, 2, 3, 0), "BGR": (2, 1, 0, None), "BGRA": (2, 1, 0, 3)}[fmt]
    r, g, b = px[:, order[0]], px[:, order[1]], px[:, order[2]]
    a = px[:, order[3]] if order[3] is not None else np.full(len(px), 255, np.uint8)
    keep = (a >= 125) & ~((r > 250) & (g > 250) & (b > 250))
    return np.stack([r[keep], g[keep], b[keep]], axis=1)


def quantize_js_palette(plane, fmt, quality, max_colors):
    return [(r << 16) | (g << 8) | b for r, g, b in quantize(color_thief_pixels(plane, fmt, quality), max_colors)]


# ---------------------------------------------------------------------------------------------------------------- blockhash.py

def _median(data):
    data = sorted(data)
    length = len(data)
    if length % 2 == 0:
        return (data[length // 2 - 1] + data[length // 2]) / 2.0
    return data[length // 2]


def blockhash_py_even(rgba, width, height, bits=8):
    """blockhash_even(im, bits): -> (block sums, bit list) for an RGBA frame given as (height, width * 4) uint8"""
    px = rgba.reshape(height, width, 4).astype(np.int64)
    value = np.where(px[..., 3] == 0, 765, px[..., 0] + px[..., 1] + px[..., 2])  # total_value_rgba
    bx, by = width // bits, height // bits
    blocks = [int(value[y * by:(y + 1) * by, x * bx:(x + 1) * bx].sum()) for y in range(bits) for x in range(bits)]
    result = list(blocks)
    half_block_value = bx * by * 256 * 3 / 2
    bandsize = len(result) // 4
    for i in range(4):
        m = _median(result[i * bandsize:(i + 1) * bandsize])
        for j in range(i * bandsize, (i + 1) * bandsize):
            v = result[j]
            result[j] = int(v > m or (abs(v - m) < 1 and m > half_block_value))
    return blocks, result
