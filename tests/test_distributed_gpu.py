"""The multi-GPU orchestration (gst-plugin-rs_amd/distributed.py) with the REAL pieces on the one-GPU box: per-rank compute
= the HIP band kernels through the C ABI, collective = RCCL (torch.distributed backend "nccl", world_size 1).  The
world-2 logic is covered on CPU by tests/test_distributed_cpu.py (gloo); this test makes sure the RCCL communicator is
really initialised by this code and that HIP kernels and all-reduces share a stream order correctly."""
import ctypes
import socket

import numpy as np
import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl(gpu):
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    one = torch.ones(1, dtype=torch.int32, device="cuda:0")
    dist.all_reduce(one)
    assert int(one[0]) == 1 and dist.get_backend() == "nccl"
    yield dist
    dist.destroy_process_group()


def test_videocompare_sharded_hip_bands_over_rccl(gpu, rccl):
    """band kernels -> ONE all-reduce of n_pads x 64 sums -> bits + Hamming on every rank."""
    import torch
    from gst_plugin_rs_amd import distributed as D
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    c = 255 - a
    pads = [a, b, c, a]
    dev = torch.device("cuda", 0)
    bufs = [torch.from_numpy(p.reshape(-1)).to(dev) for p in pads]
    fr = (gpu.Frame * len(pads))(*[gpu.make_frame(t.data_ptr(), w, h, w * 4, "RGBA") for t in bufs])
    sums = torch.zeros((len(pads), 64), dtype=torch.int32, device=dev)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    r0, r1 = D.band_rows(h, 0, 1)
    assert (r0, r1) == (0, h)

    def partial():
        gpu.check(gpu.lib().mvfx_blockhash_sums_pads(fr, len(pads), h, r0, ctypes.c_void_p(sums.data_ptr()), sptr))
        return sums

    def bits(s, ww, hh):
        arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
        out = ctypes.c_uint64()
        gpu.check(gpu.lib().mvfx_blockhash_bits(arr, ww, hh, ctypes.byref(out)))
        return out.value

    d = D.videocompare_sharded(partial, len(pads), w, h, bits, dev, all_pads=True)
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in pads]
    assert d == [float(orc.hamming(hs[0], x)) for x in hs[1:]] and d[2] == 0.0


def test_ssim_and_colordetect_sharded_over_rccl(gpu, rccl):
    import torch
    from gst_plugin_rs_amd import distributed as D
    dev = torch.device("cuda", 0)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    w, h = 320, 240
    a = frames.random_frame(0x5EED0002, w, h)
    b = a.copy()
    b[5::7, 3:w * 4:11] ^= 0x15
    ta, tb = torch.from_numpy(a.reshape(-1)).to(dev), torch.from_numpy(b.reshape(-1)).to(dev)
    fa, fb = gpu.make_frame(ta.data_ptr(), w, h, w * 4, "RGBA"), gpu.make_frame(tb.data_ptr(), w, h, w * 4, "RGBA")
    y0, y1 = D.ssim_band_rows(h, 0, 1)
    got = D.ssim_sharded(lambda: gpu.ssim_partial_sums(fa, fb, y0, y1, sptr), lambda mean: gpu.ssim_partial_deviation(mean, sptr),
                         gpu.ssim_combine, dev)
    rc, want, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    assert rc == 0 and got == pytest.approx(want, rel=1e-9, abs=1e-12)
    # colordetect: device histogram -> all-reduce(sum) + min/max -> host median cut
    hist = torch.zeros(32768 + 8, dtype=torch.int32, device=dev)

    def partial_hist():
        gpu.check(gpu.lib().mvfx_colordetect_histogram(ctypes.byref(fa), 10, 0, gpu.ALL_SAMPLES, ctypes.c_void_p(hist.data_ptr()),
                                                       ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr))
        torch.cuda.synchronize(dev)
        return hist[:32768].clone(), hist[32768:32774].clone()

    def pal(hh, mm):
        arr = (ctypes.c_int32 * 32768)(*[int(x) for x in hh])
        m = (ctypes.c_uint32 * 6)(*[int(x) for x in mm])
        out = (ctypes.c_uint32 * 5)()
        n = ctypes.c_uint32()
        gpu.check(gpu.lib().mvfx_mmcq_palette_from_histogram(arr, m, 5, out, ctypes.byref(n)))
        return [int(out[i]) for i in range(n.value)]

    palette = D.colordetect_sharded(partial_hist, pal, dev)
    rc, want_pal = orc.colordetect_palette(a, "RGBA", 10, 5)
    assert rc == 0 and palette == [int(x) for x in want_pal]
