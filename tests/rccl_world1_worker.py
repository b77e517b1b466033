"""Worker of tests/test_distributed_gpu.py: a fresh process (torch first, then libmi355vfx -- the order bench.py uses, so
both share one HIP runtime) that initialises RCCL (backend "nccl", world_size 1) and runs the orchestration of
gst-plugin-rs_amd/distributed.py over the HIP band kernels.  Prints one JSON line; compares nothing itself."""
import ctypes
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkg
    from gst_plugin_rs_amd import distributed as D
    from tests import frames
    gpu = _pkg.vfx
    lib = gpu.lib()
    assert torch.cuda.is_available(), "no GPU visible to torch"
    torch.cuda.set_device(0)
    gpu.check(lib.mvfx_set_device(0))
    dev = torch.device("cuda", 0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    one = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(one)
    out = {"backend": dist.get_backend(), "rccl_ranks": int(one[0])}
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    # videocompare: band kernels -> ONE all-reduce of n_pads x 64 sums -> bits + Hamming
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    pads = [a, b, 255 - a, a]
    bufs = [torch.from_numpy(p.reshape(-1)).to(dev) for p in pads]
    fr = (gpu.Frame * len(pads))(*[gpu.make_frame(t.data_ptr(), w, h, w * 4, "RGBA") for t in bufs])
    sums = torch.zeros((len(pads), 64), dtype=torch.int32, device=dev)
    r0, r1 = D.band_rows(h, 0, 1)

    def partial():
        gpu.check(lib.mvfx_blockhash_sums_pads(fr, len(pads), h, r0, ctypes.c_void_p(sums.data_ptr()), sptr))
        return sums

    def bits(s, ww, hh):
        arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
        o = ctypes.c_uint64()
        gpu.check(lib.mvfx_blockhash_bits(arr, ww, hh, ctypes.byref(o)))
        return o.value

    out["videocompare"] = D.videocompare_sharded(partial, len(pads), w, h, bits, dev, all_pads=True)
    # the same aggregate through the library's own RCCL communicator (world 1): all-reduce + bits + Hamming on the device
    comm = D.make_comm(gpu, 0, 1)
    out["videocompare_c_entry"], out["hashes_c_entry"] = gpu.videocompare_sharded_distances(comm, fr, h, 0, sptr, want_hashes=True)
    buf = torch.arange(10, dtype=torch.float64, device=dev)
    comm.allreduce(buf.data_ptr(), 10, gpu.DTYPE_F64, gpu.REDUCE_SUM, sptr)
    torch.cuda.synchronize(dev)
    out["allreduce_f64_world1"] = buf.cpu().tolist()
    comm.destroy()

    # dssim: two all-reduces of 10 f64 around the two map passes
    sw, sh = 320, 240
    sa = frames.random_frame(0x5EED0002, sw, sh)
    sb = sa.copy()
    sb[5::7, 3:sw * 4:11] ^= 0x15
    ta, tb = torch.from_numpy(sa.reshape(-1)).to(dev), torch.from_numpy(sb.reshape(-1)).to(dev)
    fa, fb = gpu.make_frame(ta.data_ptr(), sw, sh, sw * 4, "RGBA"), gpu.make_frame(tb.data_ptr(), sw, sh, sw * 4, "RGBA")
    y0, y1 = D.ssim_band_rows(sh, 0, 1)
    out["ssim"] = D.ssim_sharded(lambda: gpu.ssim_partial_sums(fa, fb, y0, y1, sptr), lambda mean: gpu.ssim_partial_deviation(mean, sptr),
                                 gpu.ssim_combine, dev)

    # the same distance through the library's own communicator (world 1): both all-reduces inside mvfx_videocompare_sharded_dssim
    comm2 = D.make_comm(gpu, 0, 1)
    out["ssim_c_entry"] = gpu.videocompare_sharded_dssim(comm2, fa, fb, y0, y1, sptr)
    out["ssim_c_entry_no_comm"] = gpu.videocompare_sharded_dssim(None, fa, fb, y0, y1, sptr)
    comm2.destroy()

    # colordetect: device histogram -> all-reduce(sum) + min/max -> host median cut
    hist = torch.zeros(32768 + 8, dtype=torch.int32, device=dev)

    def partial_hist():
        gpu.check(lib.mvfx_colordetect_histogram(ctypes.byref(fa), 10, 0, gpu.ALL_SAMPLES, ctypes.c_void_p(hist.data_ptr()),
                                                 ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr))
        torch.cuda.synchronize(dev)
        return hist[:32768].clone(), hist[32768:32774].clone()

    def pal(hh, mm):
        arr = (ctypes.c_uint32 * 32768)(*[int(x) & 0xFFFFFFFF for x in hh])
        m = (ctypes.c_uint32 * 6)(*[int(x) for x in mm])
        o = (ctypes.c_uint32 * 5)()
        n = ctypes.c_uint32()
        gpu.check(lib.mvfx_mmcq_palette_from_histogram(arr, m, 5, o, ctypes.byref(n)))
        return [int(o[i]) for i in range(n.value)]

    out["palette"] = D.colordetect_sharded(partial_hist, pal, dev)
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
