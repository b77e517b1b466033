"""world_size-2 gloo tests (CPU) of the N>1 orchestration in gst-plugin-rs_amd/distributed.py.
The per-rank compute is the oracle here (no GPU in this container); on the GPU node the same
functions are fed by the HIP kernels (bench.py --workload videocompare, tests/test_videofx_gpu.py
checks the band kernels against the oracle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import frames
from tests import oracle_binding as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _band_sums(frame, w, h, r0, r1):
    px = frame[r0:r1, :w * 4].reshape(r1 - r0, w, 4).astype(np.uint64)
    v = px[..., 0] + px[..., 1] + px[..., 2]
    v[px[..., 3] == 0] = 765
    sums = np.zeros(64, np.uint64)
    bw, bh = w // 8, h // 8
    for y in range(r0, r1):
        row = v[y - r0].reshape(8, bw).sum(axis=1)
        sums[(y // bh) * 8:(y // bh) * 8 + 8] += row
    return sums


def _worker(rank, world, port, q):
    import _pkg  # noqa: F401
    from gst_plugin_rs_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 256, 128
        a = frames.random_frame(0x5EED0001, w, h)
        b = a.copy()
        b[::3, 0:w * 4:16] ^= 0x3C  # perturbed copy
        c = 255 - a                  # inverted
        pads = [a, b, c, a]
        r0, r1 = D.band_rows(h, rank, world)
        dev = torch.device("cpu")

        def partial(p):
            return torch.from_numpy(_band_sums(pads[p], w, h, r0, r1).astype(np.int64))

        d = D.videocompare_sharded(partial, len(pads), w, h, lambda s, ww, hh: orc.blockhash_bits(s, ww, hh), dev)

        # colordetect: each rank histograms half of the samples
        quality = 10
        flat = a.reshape(-1)
        n_px = flat.size // 4
        n_samples = (n_px + quality - 1) // quality
        s0, s1 = n_samples * rank // world, n_samples * (rank + 1) // world
        idx = (np.arange(s0, s1) * quality)
        sub = flat.reshape(-1, 4)[idx].copy()
        rc, hist, mm, _ = orc.colordetect_histogram(sub.reshape(-1), "RGBA", 1)

        def pal(hh, m):
            rc2, p = orc.mmcq_from_histogram(np.array(hh, dtype=np.int32), m, 5)
            return p

        palette = D.colordetect_sharded(lambda: (torch.from_numpy(hist.astype(np.int64)), torch.tensor(mm, dtype=torch.int64)), pal, dev)
        # hash-algo=dssim: two small all-reduces around the two map passes
        sw, sh = 96, 80
        sa = frames.random_frame(0x5EED0002, sw, sh)
        sb = sa.copy()
        sb[5::7, 3:sw * 4:11] ^= 0x15
        y0, y1 = D.ssim_band_rows(sh, rank, world)
        ssim = D.ssim_sharded(lambda: orc.ssim_band(sa, sb, sw, sh, sw * 4, sw * 4, "RGBA", y0, y1),
                              lambda mean: orc.ssim_band(sa, sb, sw, sh, sw * 4, sw * 4, "RGBA", y0, y1, mean)[0],
                              orc.ssim_combine, dev)
        q.put((rank, d, palette, D.shard_streams(5, rank, world), ssim))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_videocompare_and_colordetect_match_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    results.sort()
    # single-process truth from the oracle on the whole frames
    w, h = 256, 128
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    c = 255 - a
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in (a, b, c, a)]
    truth = [float(orc.hamming(hs[0], x)) for x in hs[1:]]
    rc, pal = orc.colordetect_palette(a, "RGBA", 10, 5)
    sw, sh = 96, 80
    sa = frames.random_frame(0x5EED0002, sw, sh)
    sb = sa.copy()
    sb[5::7, 3:sw * 4:11] ^= 0x15
    rc, ssim_truth, _ = orc.ssim_distance(sa, sb, sw, sh, sw * 4, sw * 4, "RGBA")
    assert rc == 0 and ssim_truth > 0
    for rank, d, palette, streams, ssim in results:
        assert d == truth and d[2] == 0.0
        assert palette == pal
        assert streams == [k for k in range(5) if k % world == rank]
        assert ssim == pytest.approx(ssim_truth, rel=1e-12)
    assert results[0][4] == results[1][4]  # every rank derives the same value


def test_band_rows_cover_the_frame():
    import _pkg  # noqa: F401
    from gst_plugin_rs_amd import distributed as D
    for h, world in ((4320, 8), (4320, 2), (1080, 4), (1081, 3), (480, 1)):
        bands = [D.band_rows(h, r, world) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == h
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
    assert D.band_rows(4320, 3, 8) == (3 * 540, 4 * 540)  # one block row per rank at 8K
    for h, world in ((4320, 8), (1080, 8), (100, 3), (17, 2), (8, 4)):
        bands = [D.ssim_band_rows(h, r, world) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == h
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
        assert all(b[0] % 16 == 0 and (b[1] % 16 == 0 or b[1] == h) for b in bands)
