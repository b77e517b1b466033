import ctypes, os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
import torch, _pkg
vfx=_pkg.vfx; lib=vfx.lib()
bench=ctypes.CDLL(os.path.join(ROOT,"gst-plugin-rs_amd","libmvfxbench.so"))
dev=torch.device("cuda",0); vfx.check(lib.mvfx_set_device(0))
W,H=3840,2160; fb=W*H*4
settings=vfx.HsvFilterSettings(90.0,1.25,-0.05,0.9,0.02)
opts=vfx.OPT_NONTEMPORAL
def frames_of(t): 
    n=t.shape[0]
    return (vfx.Frame*n)(*[vfx.make_frame(t[i].data_ptr(),W,H,W*4,"RGBA") for i in range(n)])
scratch=torch.randint(0,256,(64,fb),dtype=torch.uint8,device=dev)
NB=int(sys.argv[1]) if len(sys.argv)>1 else 2080
big=torch.randint(0,256,(NB,fb),dtype=torch.uint8,device=dev)
torch.cuda.synchronize()
fs, fbig = frames_of(scratch), frames_of(big)
secs=(ctypes.c_double*1)(); per=(ctypes.c_double*1)()
def run(fr,n,launches,warm=0):
    rc=bench.mvfxbench_hsvfilter_streams_batched(0,1,warm,launches,1,fr,n,16,ctypes.byref(settings),opts,secs,per); assert rc==0
    return 16*launches/secs[0]
for trial in range(2):
    print("settle on scratch:", round(run(fs,64,2500)))
    for k in (26, NB//16):
        pass
    # fresh big pool, each frame once
    for p in range(4):
        print(f" pass {p} over the {NB}-frame pool ({NB//16} launches, frames filtered {p} times before): {run(fbig,NB,NB//16):.0f} fps", flush=True)
    big.random_(0,256); torch.cuda.synchronize()
    print("settle on scratch:", round(run(fs,64,2500)))
    # short timed regions of 26 launches on fresh regions of the pool
    sub=[]
    for j in range(4):
        part=(vfx.Frame*416)(*fbig[j*416:(j+1)*416])
        sub.append(run(part,416,26))
    print(" 26-launch regions on fresh data:", [round(x) for x in sub], flush=True)
    big.random_(0,256); torch.cuda.synchronize()
