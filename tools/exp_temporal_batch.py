#!/usr/bin/env python3
"""tools/exp_temporal_batch.py -- round 4: ONE thread, `batch` 4K RGBA frames per launch, launches rotated over 1 / 2 streams: what a
single element could reach by holding a few buffers back and launching them together (temporal batching).  Median of 5 x ~4000 frames."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402
from tests import frames as _frames  # noqa: E402

vfx = _pkg.vfx
lib = vfx.lib()
bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
dev = torch.device("cuda", 0)
vfx.check(lib.mvfx_set_device(0))
W, H = 3840, 2160
fb = W * H * 4
settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
base = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
fpt = 32
pool = base.unsqueeze(0).repeat(fpt, 1).contiguous()
torch.cuda.synchronize()
fr = (vfx.Frame * fpt)(*[vfx.make_frame(pool[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(fpt)])
print("frames per launch x streams -> frames/s (fraction of 8 TB/s)")
for batch in (1, 2, 4, 8, 16):
    row = []
    for streams in (1, 2, 3):
        launches = 4000 // batch
        secs = (ctypes.c_double * 5)()
        per = (ctypes.c_double * 1)()
        rc = bench.mvfxbench_hsvfilter_streams_rot_batched(0, 1, streams, 400 // batch, launches, 5, fr, fpt, batch, ctypes.byref(settings),
                                                           vfx.OPT_NONTEMPORAL, secs, per)
        assert rc == 0, (rc, vfx.last_error())
        fps = launches * batch / sorted(secs)[2]
        row.append(f"{streams} stream(s) {fps:8.0f} ({fps * 2 * fb / 8e12:.3f})")
    print(f"batch {batch:2d}:  " + "   ".join(row), flush=True)
