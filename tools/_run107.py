import os, sys, tempfile
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("b", "tools/bench_gst_pipeline.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
tmp = tempfile.mkdtemp()
w = h = 64
det = "hsvdetector hue-ref=120 hue-var=60 saturation-ref=0.6 saturation-var=0.4 value-ref=0.6 value-var=0.4"
for name, fmt, chain in (("source RGBx alone", "RGBx", ""), ("source RGBA alone", "RGBA", ""), ("hsvfilter RGBx", "RGBx", "hsvfilter hue-shift=90 ! "),
                         ("hsvfilter RGBA", "RGBA", "hsvfilter hue-shift=90 ! "), ("hsvdetector RGBx->RGBA", "RGBx", det + " ! "),
                         ("hsvdetector RGBx->ARGB", "RGBx", det + " ! video/x-raw(memory:HIPMemory),format=ARGB ! "),
                         ("hsvdetector defaults", "RGBx", "hsvdetector ! ")):
    caps = f"video/x-raw(memory:HIPMemory),format={fmt},width={w},height={h},framerate=30/1"
    tpl = f"hiptestsrc num-buffers={{n}} refresh=false ! {caps} ! {chain}fakesink sync=false"
    b.run(tpl.format(n=2000), tmp)
    r = [round(b.rate(tpl, tmp, 10000, 160000, {"MVFX_ELEMENT_PAIR": "0"})) for _ in range(2)]
    print(name, r, flush=True)
