"""Multi-GPU orchestration of the path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

What shards and how (DESIGN.md "Multi-GPU"):
  * hsvfilter / hsvdetector / colorlut / roundedcorners / colordetect on independent streams:
    streams are dealt round-robin to ranks; NO data-path collective (`shard_streams`).
  * videocompare on frames whose rows are already distributed (rank r holds block-row band r of
    every pad's frame): each rank reduces its band to 64 partial block sums per frame, then ONE
    all-reduce(sum) of n_pads x 64 u32 (512 B for a pair) makes the totals visible everywhere and
    every rank derives the hash bits and Hamming distances redundantly (`videocompare_sharded`).
  * videocompare `hash-algo=dssim`: two all-reduces (10 f64: per-scale sums + counts, then 5 f64:
    per-scale absolute deviations) around the two map passes (`ssim_sharded`; inside the library:
    `ssim_sharded_device` = mvfx_videocompare_sharded_dssim).
  * colordetect on one distributed frame: all-reduce(sum) of the 32768-bin histogram plus
    min/max of the six channel bounds, then the host median cut on every rank
    (`colordetect_sharded`).

The compute callables are injected so that the same orchestration runs with the HIP kernels
(through the C ABI) on GPUs and with a CPU stand-in in the world_size-2 gloo tests.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Independent streams: stream k runs on rank k % world.  No collective is involved."""
    return [k for k in range(n_streams) if k % world == rank]


def band_rows(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Row band of `rank`: block-row aligned when height % (8*world) allows it (8K: 8 ranks <-> the 8
    block rows of the 8x8 blockhash grid), otherwise an even split of rows."""
    if height % 8 == 0 and 8 % world == 0:
        rows_per_block_row = height // 8
        per_rank = 8 // world
        return rank * per_rank * rows_per_block_row, (rank + 1) * per_rank * rows_per_block_row
    return height * rank // world, height * (rank + 1) // world


def videocompare_sharded(partial_sums: Callable[[int], torch.Tensor], n_pads: int, width: int, height: int,
                         bits_from_sums: Callable[[Sequence[int], int, int], int], device: torch.device,
                         group=None, all_pads: bool = False) -> List[float]:
    """partial_sums(pad) -> uint32[64] tensor (on `device`) of THIS rank's band of pad's frame
    (pad 0 = the reference pad, videocompare/imp.rs:210-233); with all_pads=True, partial_sums()
    -> [n_pads, 64] from ONE launch over every pad (mvfx_blockhash_sums_pads).
    Returns the distances of pads 1.. to the reference pad (videocompare/imp.rs:349-353)."""
    # whenever a process group exists the sums go through the collective, also with one rank (the world-1 RCCL test on
    # the single-GPU box runs the very same code path as 8 ranks); without a group (plain single-GPU use) nothing is reduced
    sharded = dist.is_initialized()
    if all_pads:
        parts = partial_sums()
        if sharded:  # u32 block sums travel as int64 so that the all-reduce cannot wrap a signed 32-bit lane
            parts = (parts.to(torch.int64) & 0xFFFFFFFF).to(device)
    else:
        parts = (torch.stack([partial_sums(p).to(torch.int64) for p in range(n_pads)]) & 0xFFFFFFFF).to(device)
    if sharded:
        dist.all_reduce(parts, op=dist.ReduceOp.SUM, group=group)  # n_pads x 64 values, latency-bound
    totals = [[int(v) & 0xFFFFFFFF for v in row] for row in parts.cpu().tolist()]  # u32 sums (int32 views wrap back)
    hashes = [bits_from_sums(t, width, height) for t in totals]
    return [float(bin(hashes[0] ^ h).count("1")) for h in hashes[1:]]


def videocompare_sharded_device(vfx, comm, bands, full_height: int, band_first_row: int, stream=None) -> List[float]:
    """The same aggregate through the library's own collective (round 3): band kernel -> ncclAllReduce of n_pads x 64 u32 on the
    launch stream -> hash bits + Hamming distances on the device -> one D2H of n_pads - 1 words
    (mvfx_videocompare_sharded_distances; `comm` is a vfx.Comm or None for one GPU).  No host round trip of the sums, no Python bit
    derivation; `videocompare_sharded` above stays as the torch.distributed shim the gloo CPU tests drive."""
    return vfx.videocompare_sharded_distances(comm, bands, full_height, band_first_row, stream)


def ssim_sharded_device(vfx, comm, frame_a, frame_b, row_begin: int, row_end: int, stream=None) -> float:
    """hash-algo=dssim through the library's own collective (round 3): band maps -> ncclAllReduce of 10 f64 -> band deviations ->
    ncclAllReduce of 5 f64 -> combine, inside mvfx_videocompare_sharded_dssim (`comm` is a vfx.Comm or None for one GPU); `ssim_sharded`
    below stays as the torch.distributed shim the gloo CPU tests drive."""
    return vfx.videocompare_sharded_dssim(comm, frame_a, frame_b, row_begin, row_end, stream)


def make_comm(vfx, rank: int, world: int, group=None):
    """The library's RCCL communicator for this rank, its 128-byte id travelling from rank 0 over the torch.distributed group
    that already exists (any backend)."""
    def bcast(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    return vfx.Comm(rank, world, bcast if world > 1 else (lambda ident: ident))


def ssim_band_rows(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Row band for the SSIM distance: boundaries are multiples of 16 rows so that every one of the
    five pyramid levels partitions exactly (the last rank takes the remainder)."""
    units = (height + 15) // 16
    lo = min(units * rank // world * 16, height)
    hi = height if rank == world - 1 else min(units * (rank + 1) // world * 16, height)
    return lo, hi


def ssim_sharded(partial_sums: Callable[[], Tuple[Sequence[float], Sequence[float], int]],
                 partial_deviation: Callable[[Sequence[float]], Sequence[float]],
                 combine: Callable[[Sequence[float], Sequence[float], int], float],
                 device: torch.device, group=None) -> float:
    """partial_sums() -> (sums[5], counts[5], n_scales) of THIS rank's band; partial_deviation(mean[5])
    -> sums of |map - mean| over the same band.  Two 80-byte all-reduces; every rank returns the
    same distance (videocompare/hashed_image.rs:72-75)."""
    sums, counts, n_scales = partial_sums()
    t = torch.tensor([list(sums), list(counts)], dtype=torch.float64, device=device)
    multi = dist.is_initialized()
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    tot = t.cpu().tolist()
    mean = [tot[0][s] / tot[1][s] if tot[1][s] else 0.0 for s in range(5)]
    d = torch.tensor(list(partial_deviation(mean)), dtype=torch.float64, device=device)
    if multi:
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
    dev = d.cpu().tolist()
    mad = [dev[s] / tot[1][s] if tot[1][s] else 0.0 for s in range(5)]
    return combine(mean, mad, n_scales)


def colordetect_sharded(partial_hist: Callable[[], Tuple[torch.Tensor, torch.Tensor]],
                        palette_from_hist: Callable[[Sequence[int], Sequence[int]], List[int]],
                        device: torch.device, group=None) -> List[int]:
    """partial_hist() -> (uint32[32768] histogram, uint32[6] {rmin,rmax,gmin,gmax,bmin,bmax}) of THIS
    rank's sample range.  All ranks return the same palette."""
    hist, mm = partial_hist()
    hist = hist.to(torch.int64).to(device)
    lo = mm[0::2].to(torch.int64).to(device)
    hi = mm[1::2].to(torch.int64).to(device)
    if dist.is_initialized():
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    minmax = [0] * 6
    minmax[0::2] = lo.cpu().tolist()
    minmax[1::2] = hi.cpu().tolist()
    return palette_from_hist(hist.cpu().tolist(), minmax)
