import ctypes, time, sys
sys.path.insert(0,'/root/repo')
import _pkg
vfx=_pkg.vfx; lib=vfx.lib()
hip=ctypes.CDLL("libamdhip64.so")
vfx.check(lib.mvfx_set_device(0))
d=ctypes.c_int()
N=200000
t=time.perf_counter()
for _ in range(N): hip.hipGetDevice(ctypes.byref(d))
t1=time.perf_counter()-t
t=time.perf_counter()
for _ in range(N): hip.hipGetLastError()
t2=time.perf_counter()-t
t=time.perf_counter()
for _ in range(N): lib.mvfx_current_device()
t3=time.perf_counter()-t
print(f"hipGetDevice {t1/N*1e6:.2f} us/call, hipGetLastError {t2/N*1e6:.2f}, mvfx_current_device {t3/N*1e6:.2f}")
