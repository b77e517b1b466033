#!/usr/bin/env python3
"""tools/exp_cabi_pairs.py -- one host thread, two alternating streams, 4K frames through the bare C ABI: hsvfilter, hsvdetector and colorlut with
one and with two frames per launch -- what the elements are measured against (profiles/r4/element_pairs.txt section 8)."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, _pkg
from tests import cubes, frames as _frames
vfx = _pkg.vfx; lib = vfx.lib(); lib.mvfx_thread_stream_n.restype = ctypes.c_void_p
dev = torch.device("cuda", 0); vfx.check(lib.mvfx_set_device(0))
W, H, N = 3840, 2160, 8
base = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
a = base.unsqueeze(0).repeat(N, 1).contiguous(); b = torch.empty_like(a)
torch.cuda.synchronize()
mk = lambda t, fmt: (vfx.Frame * N)(*[vfx.make_frame(t[i].data_ptr(), W, H, W * 4, fmt) for i in range(N)])
fa, fb, fr = mk(a, "RGBx"), mk(b, "RGBA"), mk(a, "RGBA")
hs = vfx.HsvFilterSettings(45.0, 1.0, 0.0, 1.0, 0.0); ds = vfx.HsvDetectorSettings(120.0, 60.0, 0.6, 0.4, 0.6, 0.4)
lut = vfx.CubeLut(cubes.analytic_3d(33))
st = [ctypes.c_void_p(lib.mvfx_thread_stream_n(k)) for k in range(2)]
F = ctypes.sizeof(vfx.Frame); at = lambda arr, i: ctypes.cast(ctypes.addressof(arr) + i * F, ctypes.POINTER(vfx.Frame))
def run(name, one, two):
    for label, fn in (("1 frame/launch", lambda k: one(k % N, st[k & 1])), ("2 frames/launch", lambda k: two((2 * k) % N, st[k & 1]))):
        per = 1 if label[0] == "1" else 2
        res = []
        for rep in range(3):
            for k in range(400): fn(k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 8000 // per
            for k in range(n): fn(k)
            torch.cuda.synchronize(); res.append(n * per / (time.perf_counter() - t0))
        print(f"{name} {label}: {sorted(res)[1]:8.0f} frames/s", flush=True)
run("hsvfilter", lambda i, s: vfx.check(lib.mvfx_hsvfilter_transform_frame_ip(at(fr, i), ctypes.byref(hs), s)),
    lambda i, s: vfx.check(lib.mvfx_hsvfilter_transform_frames_ip(at(fr, i), 2, ctypes.byref(hs), s)))
run("hsvdetector", lambda i, s: vfx.check(lib.mvfx_hsvdetector_transform_frame(at(fa, i), at(fb, i), ctypes.byref(ds), s)),
    lambda i, s: vfx.check(lib.mvfx_hsvdetector_transform_frames(at(fa, i), at(fb, i), 2, ctypes.byref(ds), s)))
run("colorlut", lambda i, s: vfx.check(lib.mvfx_colorlut_transform_frame(lut.h, at(fr, i), at(fb, i), s)),
    lambda i, s: vfx.check(lib.mvfx_colorlut_transform_frames(lut.h, at(fr, i), at(fb, i), 2, s)))

# what do the fences of the element layer cost on the device?  the pair launches of hsvdetector again, with k event records (and as many
# stream waits on events of the OTHER stream's previous launch) after every launch: the element does four records per pair launch
evs = [[ctypes.c_void_p() for _ in range(4)] for _ in range(2)]
for row in evs:
    for e in row:
        vfx.check(lib.mvfx_event_create(ctypes.byref(e)))
        vfx.check(lib.mvfx_event_record(e, st[0]))
for k_rec, k_wait in ((0, 0), (1, 0), (2, 0), (4, 0), (4, 4), (1, 1)):
    def fn(k):
        s, o = st[k & 1], (k & 1) ^ 1
        for j in range(k_wait):
            vfx.check(lib.mvfx_stream_wait_event(s, evs[o][j]))
        vfx.check(lib.mvfx_hsvdetector_transform_frames(at(fa, (2 * k) % N), at(fb, (2 * k) % N), 2, ctypes.byref(ds), s))
        for j in range(k_rec):
            vfx.check(lib.mvfx_event_record(evs[k & 1][j], s))
    res = []
    for rep in range(3):
        for k in range(200): fn(k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(4000): fn(k)
        torch.cuda.synchronize(); res.append(8000 / (time.perf_counter() - t0))
    print(f"hsvdetector 2 frames/launch + {k_rec} event records + {k_wait} waits on the other stream's events per launch: {sorted(res)[1]:8.0f} frames/s", flush=True)
