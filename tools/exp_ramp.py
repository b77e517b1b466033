"""Launch-duration time series of the headline kernel: does the clock ramp during a long run?"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg
vfx = _pkg.vfx
lib = vfx.lib()
W, H = 3840, 2160
dev = torch.device("cuda", 0)
vfx.check(lib.mvfx_set_device(0))
pool, batch = 24, 16
frames = torch.randint(0, 256, (pool, batch, W * H * 4), dtype=torch.uint8, device=dev)
settings = vfx.HsvFilterSettings(90.0, 1.25, -0.05, 0.9, 0.02)
arrs = [(vfx.Frame * batch)(*[vfx.make_frame(frames[b, i].data_ptr(), W, H, W * 4, "RGBA") for i in range(batch)]) for b in range(pool)]
stream = torch.cuda.current_stream(dev)
sptr = ctypes.c_void_p(stream.cuda_stream)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
torch.cuda.synchronize()
t_start = time.perf_counter()
step = 0
for c in range(n_chunks):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(chunk):
        lib.mvfx_hsvfilter_transform_frames_ip(arrs[step % pool], batch, ctypes.byref(settings), sptr)
        step += 1
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / chunk
    print(f"t={time.perf_counter() - t_start:7.2f}s chunk {c:3d}: {ms * 1e3:7.1f} us/launch  {batch / ms * 1e3:8.0f} fps", flush=True)
    if idle:
        time.sleep(idle)
