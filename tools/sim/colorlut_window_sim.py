#!/usr/bin/env python3
"""tools/sim/colorlut_window_sim.py -- CPU model of colorlut_xtile_kernel's window hit rate (no GPU needed).

For natural-like 4K frames (bench.py natural_frame: smooth gradients + uniform noise of +-A codes) it replays the kernel's anchor rule
(four centre lanes where they agree, the mean of all 64 lane samples elsewhere) per 64 x (4 x ROWS) block and counts, for a window of
RW r bytes x NY y cells x NZ z cells: the pixels outside the window, the (wave, row-of-four, j) slots in which ANY lane misses (each
costs the wave one masked pass over six 8-byte gathers), and the distinct (y, z) cell pairs / distinct 24-byte entries among the
misses of a wave (what a per-distinct-entry service would fetch).  Used to choose the window shape before spending GPU minutes."""
import argparse
import numpy as np


def natural(W, H, k, noise, rng):
    x = np.linspace(0, 1, W, dtype=np.float32).reshape(1, W)
    y = np.linspace(0, 1, H, dtype=np.float32).reshape(H, 1)
    ph = np.float32(0.37 * k)
    img = np.stack([np.broadcast_to(0.5 + 0.45 * np.sin(3.0 * x + 2.0 * y + ph), (H, W)),
                    np.broadcast_to(0.5 + 0.45 * np.sin(5.0 * y - 1.5 * x + 2 * ph), (H, W)),
                    np.broadcast_to(0.5 + 0.45 * np.cos(4.0 * x * y + ph), (H, W))], axis=-1).astype(np.float32) * 255.0
    if noise:
        img = img + rng.integers(-noise, noise + 1, img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def cell_of(size):
    """byte -> lower node index, like the kernel's coordinate table for a [0,1] domain: x = v/255*(size-1), i0 = min(floor(x), size-1)"""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0) * np.float32(size - 1)
    return np.minimum(np.floor(v).astype(np.int32), size - 1)


def simulate(img, size, RW, NY, NZ, rows, anchor="shipped", tile_w=64):
    H, W, _ = img.shape
    th = 4 * rows
    cells = cell_of(size)
    Hc, Wc = (H // th) * th, (W // tile_w) * tile_w
    blk = img[:Hc, :Wc].reshape(Hc // th, th, Wc // tile_w, tile_w, 3).transpose(0, 2, 1, 3, 4)  # [by, bx, th, 64, 3]
    nb = blk.shape[0] * blk.shape[1]
    blk = blk.reshape(nb, th, tile_w, 3).astype(np.int32)
    # lane l: x = (l % 16) * 4, y0 = (l // 16) * rows; sample = pixel (x + 1, y0 + 1)
    lx = (np.arange(64) % 16) * 4 + 1
    ly = (np.arange(64) // 16) * rows + 1
    smp = blk[:, ly, lx, :]  # [nb, 64, 3]
    if anchor == "centre":
        a = blk[:, th // 2, tile_w // 2, :]
    else:
        i = smp[:, [21, 26, 37, 42], :]
        inner = np.abs(i[:, 0] - i[:, 3]).sum(-1) + np.abs(i[:, 1] - i[:, 2]).sum(-1)
        mean4 = (i.sum(1) + 2) >> 2
        mean64 = (smp.sum(1) + 32) >> 6
        q = smp[:, [0, 15, 48, 63], :]
        spread = np.abs(q[:, 0] - q[:, 3]).sum(-1) + np.abs(q[:, 1] - q[:, 2]).sum(-1)
        a = np.where((inner <= 20)[:, None], mean4, np.where((spread <= 120)[:, None], mean64, smp[:, 40, :]))
        if anchor == "mean64":
            a = mean64
    ar = np.minimum(np.maximum(a[:, 0] - RW // 2, 0) & ~1, 256 - RW)
    cy, cz = cells[a[:, 1]], cells[a[:, 2]]
    # NY cells centred: anchor cell - (NY-1)//2 ... ; z: NZ cells need NZ + 1 rows (rows run 0..size)
    ay = np.clip(cy - (NY - 1) // 2, 0, size - NY)
    az = np.clip(cz - (NZ - 1) // 2, 0, size - NZ)
    r, g, b = blk[..., 0], blk[..., 1], blk[..., 2]
    iy, iz = cells[g], cells[b]
    miss = ((r < ar[:, None, None]) | (r >= (ar + RW)[:, None, None]) | (iy < ay[:, None, None]) | (iy >= (ay + NY)[:, None, None]) |
            (iz < az[:, None, None]) | (iz >= (az + NZ)[:, None, None]))
    frac = miss.mean()
    # slots: lane owns 4 consecutive x pixels j = 0..3 in `rows` rows; the wave's pass (row, j) has any miss?
    m = miss.reshape(nb, 4, rows, 16, 4)  # [nb, lane_row_group, row, lane_x, j]
    any_slot = m.transpose(0, 2, 4, 1, 3).reshape(nb, rows, 4, 64).any(-1)  # [nb, row, j]
    any_row = any_slot.any(-1)
    # distinct entries among a wave's misses: (iy, iz, r) pairs  (x2 rows z0, z1 fetched together: count (iy,iz,r))
    key = (iy * 64 + iz) * 256 + r
    distinct_entries, distinct_cells, n_miss_waves = 0, 0, 0
    sample = np.random.default_rng(1).choice(nb, size=min(nb, 400), replace=False)
    for bi in sample:
        mk = key[bi][miss[bi]]
        if mk.size:
            n_miss_waves += 1
            distinct_entries += np.unique(mk).size
            distinct_cells += np.unique(mk >> 8).size
    per = max(len(sample), 1)
    return {"miss_frac": float(frac), "slots_any_miss": float(any_slot.mean()), "rows_any_miss": float(any_row.mean()),
            "misses_per_wave": float(miss.reshape(nb, -1).sum(1).mean()), "distinct_entries_per_wave": distinct_entries / per,
            "distinct_yz_cells_per_wave": distinct_cells / per, "lds_bytes_per_wave": RW * 24 * NY * (NZ + 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=33)
    ap.add_argument("--rows", type=int, default=5)
    ap.add_argument("--noise", type=int, nargs="*", default=[0, 3, 5, 8, 16])
    ap.add_argument("--shapes", nargs="*", default=["24x3x3", "32x3x3", "24x4x4", "32x4x4", "40x4x4", "48x5x5", "64x5x5"])
    ap.add_argument("--anchor", default="shipped")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    for amp in args.noise:
        img = natural(args.width, args.height, 3, amp, rng)
        for shape in args.shapes:
            RW, NY, NZ = (int(t) for t in shape.split("x"))
            s = simulate(img, args.size, RW, NY, NZ, args.rows, args.anchor)
            print(f"noise +-{amp:2d} window {shape:8s} ({s['lds_bytes_per_wave']:6d} B/wave): miss {100 * s['miss_frac']:6.2f} %  slots with a miss "
                  f"{100 * s['slots_any_miss']:6.2f} %  misses/wave {s['misses_per_wave']:7.1f}  distinct entries {s['distinct_entries_per_wave']:7.1f}  "
                  f"distinct (y,z) cells {s['distinct_yz_cells_per_wave']:6.1f}", flush=True)


if __name__ == "__main__":
    main()
