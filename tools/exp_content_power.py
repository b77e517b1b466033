#!/usr/bin/env python3
"""How much of the hsvfilter frame rate depends on the BYTES in the frames (not on any data-dependent branch: the kernel has
none)?  1 thread x 16 frames per launch over a pool of fresh frames, every frame filtered exactly once per pass; clocks settled
on a scratch pool first.  Contents: uniform random bytes, videotestsrc-smpte-like bars, smooth gradients + noise."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, _pkg
from tests import frames as _frames
vfx = _pkg.vfx; lib = vfx.lib()
bench = ctypes.CDLL(os.path.join(ROOT, "gst-plugin-rs_amd", "libmvfxbench.so"))
dev = torch.device("cuda", 0); vfx.check(lib.mvfx_set_device(0))
W, H = 3840, 2160; fb = W * H * 4
settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
opts = vfx.OPT_NONTEMPORAL
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 832
def frames_of(t):
    return (vfx.Frame * t.shape[0])(*[vfx.make_frame(t[i].data_ptr(), W, H, W * 4, "RGBA") for i in range(t.shape[0])])
scratch = torch.randint(0, 256, (64, fb), dtype=torch.uint8, device=dev)
big = torch.randint(0, 256, (NB, fb), dtype=torch.uint8, device=dev)
fs, fbig = frames_of(scratch), frames_of(big)
secs = (ctypes.c_double * 1)(); per = (ctypes.c_double * 1)()
def run(fr, n, launches):
    rc = bench.mvfxbench_hsvfilter_streams_batched(0, 1, 0, launches, 1, fr, n, 16, ctypes.byref(settings), opts, secs, per); assert rc == 0
    return 16 * launches / secs[0]
run(fbig, NB, NB // 16)  # first touch of the pool's pages
smpte = torch.from_numpy(_frames.smpte_like(W, H).reshape(-1)).to(dev)
x = torch.linspace(0, 1, W, device=dev).view(1, W); y = torch.linspace(0, 1, H, device=dev).view(H, 1)
def natural(k):
    ph = 0.37 * k
    img = torch.stack([(0.5 + 0.45 * torch.sin(3.0 * x + 2.0 * y + ph)).expand(H, W), (0.5 + 0.45 * torch.sin(5.0 * y - 1.5 * x + 2 * ph)).expand(H, W),
                       (0.5 + 0.45 * torch.cos(4.0 * x * y + ph)).expand(H, W), torch.ones((H, W), device=dev)], dim=-1) * 255.0
    noise = torch.randint(-3, 4, img.shape, device=dev).float(); noise[..., 3] = 0
    return (img + noise).clamp(0, 255).to(torch.uint8).view(-1)
def fill(kind):
    if kind == "random": big.random_(0, 256)
    elif kind == "smpte": big.copy_(smpte.unsqueeze(0).expand(NB, fb))
    elif kind == "zeros": big.zero_()
    else:
        for k in range(0, NB, 16):
            big[k] = natural(k)
            big[k + 1:k + 16] = big[k]
    torch.cuda.synchronize()
for trial in range(2):
    for kind in ("random", "smpte", "natural", "zeros"):
        fill(kind)
        s = run(fs, 64, 2000)
        a = run(fbig, NB, NB // 16); b = run(fbig, NB, NB // 16)
        print(f"{kind:8s}: settle(scratch, converged) {s:7.0f}   fresh pass {a:7.0f} fps = {a*2*fb/8e12:.3f}   second pass {b:7.0f} = {b*2*fb/8e12:.3f}", flush=True)
