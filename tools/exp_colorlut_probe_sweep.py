#!/usr/bin/env python3
"""tools/exp_colorlut_probe_sweep.py -- round 6: colorlut 33^3 on 16 x 4K natural-like frames per launch, noise +-0 ... 16 codes: the per-wave window kernel
(MVFX_XWG=0), the workgroup window kernel (MVFX_XWG=1) and the automatic choice of the content probe, with the probe's count of busy blocks.  The
automatic column must follow the faster of the two and fall monotonically with the noise."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AMPS = (0, 2, 3, 4, 5, 6, 8, 12, 16)


def child():
    import torch
    import _pkg
    import bench
    from tests import cubes
    vfx = _pkg.vfx
    lib = vfx.lib()
    dev = torch.device("cuda", 0)
    vfx.check(lib.mvfx_set_device(0))
    W, H, nb, pool = 3840, 2160, 16, 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0300)
    lut = vfx.CubeLut(cubes.analytic_3d(33))
    src = torch.empty((pool * nb, W * H * 4), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    fi = [(vfx.Frame * nb)(*[vfx.make_frame(src[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
    fo = [(vfx.Frame * nb)(*[vfx.make_frame(dst[b * nb + i].data_ptr(), W, H, W * 4, "RGBA") for i in range(nb)]) for b in range(pool)]
    st = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(st.cuda_stream)
    out = []
    for amp in AMPS:
        bench.fill_frames(torch, dev, gen, src, "natural", W, H, noise=amp)
        for i in range(80):
            vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[i % pool], fo[i % pool], nb, sp))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(40):
            vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, fi[i % pool], fo[i % pool], nb, sp))
        e1.record(st)
        torch.cuda.synchronize()
        busy = ctypes.c_uint32()
        verdict = lib.mvfx_cube_lut_content_verdict(lut.h, ctypes.byref(busy))
        out.append(f"{nb * 40 / (e0.elapsed_time(e1) * 1e-3):.0f}:{verdict}:{busy.value}")
    print(" ".join(out))


def main():
    if os.environ.get("MVFX_EXP_CHILD"):
        return child()
    print("# noise:      " + "  ".join(f"+-{a:<8d}" for a in AMPS))
    for rep in range(2):
        for name, env in (("xtile (per wave)", {"MVFX_XWG": "0"}), ("xwg (workgroup)", {"MVFX_XWG": "1"}), ("auto (probe)", {})):
            e = dict(os.environ, MVFX_EXP_CHILD="1")
            e.pop("MVFX_XWG", None)
            e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            if r.returncode != 0:
                print(name, "FAILED", r.stderr[-400:])
                continue
            cells = r.stdout.strip().splitlines()[-1].split()
            if env:
                print(f"{name:>17}: " + "  ".join(f"{int(c.split(':')[0]):<10d}" for c in cells), flush=True)
            else:
                print(f"{name:>17}: " + "  ".join(f"{c.split(':')[0]}/{'calm' if c.split(':')[1] == '1' else 'busy'}/{c.split(':')[2]:<3s}"[:10].ljust(10) for c in cells), flush=True)


if __name__ == "__main__":
    main()
