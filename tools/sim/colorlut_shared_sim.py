#!/usr/bin/env python3
"""tools/sim/colorlut_shared_sim.py -- like colorlut_window_sim.py, for a window SHARED by the waves of a tile of tw x th pixels,
anchored at the mean of a 16 x 8 lattice of its pixels.  Prints the miss fraction per noise level and shape."""
import argparse
import numpy as np
from colorlut_window_sim import natural, cell_of


def simulate(img, size, RW, NY, NZ, tw, th, even_round=True):
    H, W, _ = img.shape
    cells = cell_of(size)
    Hc, Wc = (H // th) * th, (W // tw) * tw
    blk = img[:Hc, :Wc].reshape(Hc // th, th, Wc // tw, tw, 3).transpose(0, 2, 1, 3, 4)
    nb = blk.shape[0] * blk.shape[1]
    blk = blk.reshape(nb, th, tw, 3).astype(np.int32)
    sy = (np.arange(8) * th) // 8 + th // 16
    sx = (np.arange(16) * tw) // 16 + tw // 32
    smp = blk[:, sy][:, :, sx].reshape(nb, -1, 3)
    a = (smp.sum(1) + smp.shape[1] // 2) // smp.shape[1]
    ar = np.minimum(np.maximum(a[:, 0] - RW // 2, 0) & ~1, 256 - RW)
    v = np.arange(256, dtype=np.float32) / np.float32(255.0) * np.float32(size - 1)
    frac = v - np.floor(v)
    cy, cz = cells[a[:, 1]], cells[a[:, 2]]
    fy, fz = frac[a[:, 1]], frac[a[:, 2]]
    # odd N: anchor cell in the middle; even N: the half of the anchor's cell decides which side gets the extra cell
    oy = (NY - 1) // 2 if NY % 2 else NY // 2 - (fy >= 0.5)
    oz = (NZ - 1) // 2 if NZ % 2 else NZ // 2 - (fz >= 0.5)
    ay = np.clip(cy - oy, 0, size - NY)
    az = np.clip(cz - oz, 0, size - NZ)
    r, g, b = blk[..., 0], blk[..., 1], blk[..., 2]
    iy, iz = cells[g], cells[b]
    e = lambda t: t[:, None, None]
    miss = (r < e(ar)) | (r >= e(ar + RW)) | (iy < e(ay)) | (iy >= e(ay + NY)) | (iz < e(az)) | (iz >= e(az + NZ))
    return float(miss.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", type=int, nargs="*", default=[0, 3, 5, 8, 16])
    ap.add_argument("--configs", nargs="*", default=["24x3x3@64x20", "24x4x4@64x20", "24x4x4@64x40", "24x4x4@128x20", "32x5x5@128x40", "32x5x5@64x80", "32x5x5@256x20",
                                                      "40x6x6@128x40", "48x8x8@256x80"])
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    for amp in args.noise:
        img = natural(3840, args.height, 3, amp, rng)
        out = []
        for c in args.configs:
            shape, tile = c.split("@")
            RW, NY, NZ = (int(t) for t in shape.split("x"))
            tw, th = (int(t) for t in tile.split("x"))
            out.append(f"{c} ({RW * 24 * NY * (NZ + 1) / 1024:.1f} KB): {100 * simulate(img, 33, RW, NY, NZ, tw, th):5.2f} %")
        print(f"noise +-{amp:2d}: " + "   ".join(out), flush=True)


if __name__ == "__main__":
    main()
