#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in layer (one crt_modulate + crt_demodulate per call pair, host struct CRT),
next to the reference on one host core.  usage: tools/time_dropin.py [system] [w h]
Runs the drop-in library three times in sub-processes: strict mirror, CRTHIP_LAZY_MIRROR=1 and =2."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import crtref as R
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "ntsc"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)


def run(label, lib):
    img = R.synth_image(w, h, 4, 1)
    c = lib.new_crt(w, h, R.FMT_BGRA)
    c.set("scanlines", 1)
    c.settings(np.concatenate([img, img[-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=0, frame=0)
    c.time_fieldpasses(24, 5, True)
    reps = 300
    t, tm, td = c.time_fieldpasses(24, reps, True)
    print("%-34s %8.1f field-passes/s  (modulate %.3f ms, demodulate %.3f ms per call)" % (label, reps / t, 1e3 * tm / reps, 1e3 * td / reps))


if "--child" in sys.argv:
    run("drop-in, CRTHIP_LAZY_MIRROR=%s" % os.environ.get("CRTHIP_LAZY_MIRROR", "0"), R.RefLib(name, dropin=True))
else:
    R.build_dropin_probe(name)
    if R.have_ref(name):
        run("reference (1 core)", R.RefLib(name))
    for mode in ("0", "1", "2"):
        subprocess.run([sys.executable, os.path.abspath(__file__), name, str(w), str(h), "--child"], env=dict(os.environ, CRTHIP_LAZY_MIRROR=mode))
