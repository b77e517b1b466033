import sys, os, numpy as np, tempfile
sys.path.insert(0, os.getcwd())
from tests import gst_env, cubes
from tests import oracle_binding as orc
tmp = tempfile.mkdtemp()
cube = os.path.join(tmp, "look.cube"); open(cube, "w").write(cubes.analytic_3d(17))
L = gst_env.tool("gst-launch-1.0")
for (w, h, n) in ((320, 240, 3000), (1920, 1080, 150)):
    src = f"hiptestsrc num-buffers={{n}} ! video/x-raw,format=RGBx,width={w},height={h},framerate=30/1"
    r = gst_env.run([L, "-q"] + (src.format(n=1) + f" ! filesink location={tmp}/in.raw").split(), tmp); assert r.returncode == 0, r.stdout
    raw = np.fromfile(f"{tmp}/in.raw", np.uint8)
    r = gst_env.run([L, "-q"] + (src.format(n=n) + " ! hipupload ! queue max-size-buffers=3 ! hsvfilter hue-shift=45 ! queue max-size-buffers=3 ! hsvdetector hue-ref=120 hue-var=60 "
        "saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA ! queue max-size-buffers=3 ! "
        f"colorlut location={cube} ! queue max-size-buffers=3 ! hipdownload ! filesink location={tmp}/out.raw").split(), tmp, timeout=900)
    assert r.returncode == 0, r.stdout
    mid = raw.copy().reshape(h, w * 4); orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
    det = np.empty_like(mid); orc.hsvdetector(mid, w * 4, "RGBx", det, w * 4, "RGBA", w, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
    exp = np.empty_like(det); assert orc.CubeLut(open(cube).read()).apply(det, w * 4, exp, w * 4, w, h, "RGBA") == 0
    got = np.memmap(f"{tmp}/out.raw", np.uint8, "r").reshape(n, -1)
    bad = [k for k in range(n) if not np.array_equal(got[k], exp.reshape(-1))]
    print(w, h, n, "frames, mismatching:", len(bad), bad[:10])
    del got; os.remove(f"{tmp}/out.raw")
