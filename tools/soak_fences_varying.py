"""tools/soak_fences_varying.py -- every frame different (videotestsrc pattern=snow), so an ordering mistake between the elements' streams,
the recycled pool blocks and the upload / download copies shows as a wrong frame: CPU source -> hipupload -> three device filters with queues
between them -> hipdownload, every output frame against the oracle applied to ITS input frame."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.getcwd())
from tests import cubes, gst_env
from tests import oracle_binding as orc

tmp = tempfile.mkdtemp()
cube = os.path.join(tmp, "look.cube")
open(cube, "w").write(cubes.analytic_3d(17))
L = gst_env.tool("gst-launch-1.0")
lut = orc.CubeLut(open(cube).read())
for (w, h, n, queues) in ((320, 240, 1500, True), (320, 240, 1500, False), (1280, 720, 200, True)):
    q = " ! queue max-size-buffers=3" if queues else ""
    pipe = (f"videotestsrc pattern=snow num-buffers={n} ! video/x-raw,format=RGBx,width={w},height={h},framerate=30/1 ! tee name=t "
            f"t. ! queue ! filesink location={tmp}/in.raw "
            f"t. ! queue ! hipupload{q} ! hsvfilter hue-shift=45{q} ! hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 "
            f"value-ref=0.6 value-var=0.4 ! video/x-raw(memory:HIPMemory),format=RGBA{q} ! colorlut location={cube}{q} ! hipdownload ! "
            f"filesink location={tmp}/out.raw")
    r = gst_env.run([L, "-q"] + pipe.split(), tmp, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    fin = np.memmap(f"{tmp}/in.raw", np.uint8, "r").reshape(n, h, w * 4)
    fout = np.memmap(f"{tmp}/out.raw", np.uint8, "r").reshape(n, h, w * 4)
    bad = []
    for k in range(n):
        mid = np.array(fin[k])
        orc.hsvfilter(mid, w, w * 4, "RGBx", (45.0, 1.0, 0.0, 1.0, 0.0))
        det = np.empty_like(mid)
        orc.hsvdetector(mid, w * 4, "RGBx", det, w * 4, "RGBA", w, (120.0, 60.0, 0.6, 0.4, 0.6, 0.4))
        exp = np.empty_like(det)
        assert lut.apply(det, w * 4, exp, w * 4, w, h, "RGBA") == 0
        if not np.array_equal(fout[k], exp):
            bad.append(k)
    print(f"{w}x{h} {n} frames, queues={queues}: mismatching {len(bad)} {bad[:10]}", flush=True)
    del fin, fout
    os.remove(f"{tmp}/in.raw"); os.remove(f"{tmp}/out.raw")
