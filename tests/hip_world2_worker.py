"""Worker of tests/test_distributed_gpu.py::world-2 tests: TWO of these processes share the box's one GPU (each with its own HIP
context), hold only their share of the work -- block-row band r of every pad for blockhash, row band r of the SSIM maps, sample
range r of the colordetect histogram -- compute it with the HIP kernels through the C ABI and meet in gst-plugin-rs_amd/distributed.py's
collectives over gloo (RCCL cannot put two ranks on one device).  Prints one JSON line per rank; compares nothing itself."""
import ctypes
import json
import os
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
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert torch.cuda.is_available(), "no GPU visible to torch"
    torch.cuda.set_device(0)
    gpu.check(lib.mvfx_set_device(0))
    dev, cpu = torch.device("cuda", 0), torch.device("cpu")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {"rank": rank, "world": dist.get_world_size()}

    # videocompare / blockhash: this rank holds ONLY rows [r0, r1) of the four pads
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    pads = [a, b, 255 - a, a]
    r0, r1 = D.band_rows(h, rank, world)
    bufs = [torch.from_numpy(np.ascontiguousarray(p[r0:r1]).reshape(-1)).to(dev) for p in pads]
    fr = (gpu.Frame * len(pads))(*[gpu.make_frame(t.data_ptr(), w, r1 - r0, w * 4, "RGBA") for t in bufs])
    sums = torch.zeros((len(pads), 64), dtype=torch.int32, device=dev)

    def partial():
        gpu.check(lib.mvfx_blockhash_sums_pads(fr, len(pads), h, r0, ctypes.c_void_p(sums.data_ptr()), sptr))
        torch.cuda.synchronize(dev)
        return sums

    def bits(s, ww, hh):
        arr = (ctypes.c_uint32 * 64)(*[int(x) for x in s])
        o = ctypes.c_uint64()
        gpu.check(lib.mvfx_blockhash_bits(arr, ww, hh, ctypes.byref(o)))
        return o.value

    out["videocompare"] = D.videocompare_sharded(partial, len(pads), w, h, bits, cpu, all_pads=True)
    out["band"] = [r0, r1]

    # dssim: both frames resident (the 5-level pyramid of a band needs a halo), this rank maps rows [y0, y1)
    sw, sh = 320, 240
    sa = frames.random_frame(0x5EED0002, sw, sh)
    sb = sa.copy()
    sb[5::7, 3:sw * 4:11] ^= 0x15
    ta, tb = torch.from_numpy(sa.reshape(-1)).to(dev), torch.from_numpy(sb.reshape(-1)).to(dev)
    fa, fb = gpu.make_frame(ta.data_ptr(), sw, sh, sw * 4, "RGBA"), gpu.make_frame(tb.data_ptr(), sw, sh, sw * 4, "RGBA")
    y0, y1 = D.ssim_band_rows(sh, rank, world)
    out["ssim"] = D.ssim_sharded(lambda: gpu.ssim_partial_sums(fa, fb, y0, y1, sptr), lambda mean: gpu.ssim_partial_deviation(mean, sptr),
                                 gpu.ssim_combine, cpu)
    out["ssim_band"] = [y0, y1]
    # the same with a height that does not split evenly: 250 rows = 16 units of 16 rows, the last rank takes the remainder (122 rows)
    uh = 250
    ua, ub = frames.random_frame(0x5EED0007, sw, uh), None
    ub = ua.copy()
    ub[3::5, 1:sw * 4:13] ^= 0x21
    tua, tub = torch.from_numpy(ua.reshape(-1)).to(dev), torch.from_numpy(ub.reshape(-1)).to(dev)
    fua, fub = gpu.make_frame(tua.data_ptr(), sw, uh, sw * 4, "RGBA"), gpu.make_frame(tub.data_ptr(), sw, uh, sw * 4, "RGBA")
    u0, u1 = D.ssim_band_rows(uh, rank, world)
    out["ssim_uneven"] = D.ssim_sharded(lambda: gpu.ssim_partial_sums(fua, fub, u0, u1, sptr), lambda mean: gpu.ssim_partial_deviation(mean, sptr),
                                        gpu.ssim_combine, cpu)
    out["ssim_uneven_band"] = [u0, u1]

    # colordetect: this rank bins samples [first, first + n) of the quality-10 sample sequence
    hist = torch.zeros(32768 + 8, dtype=torch.int32, device=dev)
    total = (sw * sh + 9) // 10
    first = total * rank // world
    n = total * (rank + 1) // world - first

    def partial_hist():
        gpu.check(lib.mvfx_colordetect_histogram(ctypes.byref(fa), 10, first, n, ctypes.c_void_p(hist.data_ptr()),
                                                 ctypes.c_void_p(hist.data_ptr() + 32768 * 4), sptr))
        torch.cuda.synchronize(dev)
        return hist[:32768].clone(), hist[32768:32774].clone()

    def pal(hh, mm):
        arr = (ctypes.c_uint32 * 32768)(*[int(x) & 0xFFFFFFFF for x in hh])
        m = (ctypes.c_uint32 * 6)(*[int(x) for x in mm])
        o = (ctypes.c_uint32 * 5)()
        cnt = ctypes.c_uint32()
        gpu.check(lib.mvfx_mmcq_palette_from_histogram(arr, m, 5, o, ctypes.byref(cnt)))
        return [int(o[i]) for i in range(cnt.value)]

    out["palette"] = D.colordetect_sharded(partial_hist, pal, cpu)
    out["samples"] = [first, n]
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
