import os, sys, tempfile, subprocess, time
sys.path.insert(0, os.getcwd())
from tests import gst_env
tmp = tempfile.mkdtemp()
L = gst_env.tool("gst-launch-1.0")
w, h, n = 3840, 2160, 400000
det = "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
def threads(pid):
    out = {}
    for tid in os.listdir(f"/proc/{pid}/task"):
        try:
            st = open(f"/proc/{pid}/task/{tid}/stat").read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            ut, stt = int(f[11]), int(f[12])
            vol = invol = 0
            for l in open(f"/proc/{pid}/task/{tid}/status"):
                if l.startswith("voluntary_ctxt_switches"): vol = int(l.split()[1])
                if l.startswith("nonvoluntary_ctxt_switches"): invol = int(l.split()[1])
            out[tid] = (comm, ut, stt, vol, invol)
        except Exception:
            pass
    return out
for name, fmt, chain in (("hsvfilter", "RGBA", "hsvfilter hue-shift=90"), ("hsvdetector", "RGBx", det)):
    caps = f"video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h},framerate=30/1"
    cmd = f"hiptestsrc num-buffers={n} refresh=false ! {caps} ! {chain} ! fakesink sync=false"
    e = gst_env.env(tmp); e["MVFX_ELEMENT_PAIR"] = "0"
    p = subprocess.Popen([L, "-q"] + cmd.split(), env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(2.0); a = threads(p.pid); time.sleep(2.0); b = threads(p.pid)
    print(name, "per-thread over 2 s (ticks of 10 ms): comm user sys vol invol")
    for tid, (comm, ut, stt, vol, invol) in sorted(b.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        if tid in a:
            c = a[tid]
            d = (ut - c[1], stt - c[2], vol - c[3], invol - c[4])
            if sum(d[:3]) > 0: print(f"   {comm:20s} user {d[0]:4d} sys {d[1]:4d} vol {d[2]:7d} invol {d[3]:4d}")
    p.wait()
