#!/usr/bin/env python3
"""Generates tests/golden/golden.json (+ one small full-buffer .npy) from the REAL reference
(oracle/_ref, compiled unmodified from /root/reference).  The reference ships no golden vectors
(SURVEY.md section 4); these fixtures are the committed stand-in so that machines without
/root/reference (the GPU box) can still pin both the oracle and the HIP path.

    python tests/golden/make_golden.py          # needs oracle/_ref (make -C oracle ref)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import crtref as R  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


# every case: a sequence of field-passes on one struct CRT; inputs are crtref.synth_* (seeded)
CASES = [
    dict(id="cfg1_ntsc_640x480_progressive_noise0_crtmain", sys="ntsc", w=640, h=480, ifmt=R.FMT_BGRA, outw=640, outh=480,
         ofmt=R.FMT_BGRA, noise=0, knobs=dict(blend=1, scanlines=1), settings=dict(as_color=1), steps=4, interlaced=False,
         image=("bars", 1)),
    dict(id="cfg2_ntsc_640x480_interlaced_noise24", sys="ntsc", w=640, h=480, ifmt=R.FMT_BGRA, outw=640, outh=480,
         ofmt=R.FMT_BGRA, noise=24, knobs=dict(scanlines=1), settings=dict(as_color=1, hue=0), steps=4, interlaced=True,
         image=("random", 12345)),
    dict(id="cfg3_ntsc_1920x1080_interlaced_noise0", sys="ntsc", w=1920, h=1080, ifmt=R.FMT_BGRA, outw=1920, outh=1080,
         ofmt=R.FMT_BGRA, noise=0, knobs=dict(scanlines=1), settings=dict(as_color=1), steps=2, interlaced=True,
         image=("random", 777)),
    dict(id="ntsc_rgb24_raw_mono_832x624", sys="ntsc", w=320, h=200, ifmt=R.FMT_RGB, outw=832, outh=624,
         ofmt=R.FMT_RGB, noise=40, knobs=dict(hue=17, saturation=14, brightness=3), settings=dict(as_color=0, raw=1),
         steps=3, interlaced=True, image=("bars", 5)),
    dict(id="cfg4_vhs_832x624_srand_per_step", sys="vhs", w=832, h=624, ifmt=R.FMT_BGRA, outw=832, outh=624,
         ofmt=R.FMT_BGRA, noise=12, knobs=dict(scanlines=1), settings=dict(as_color=1, do_aberration=0), steps=3,
         interlaced=True, image=("random", 4), srand=1000),        # glibc rand(): srand(1000 + step) before each pass
    dict(id="cfg5_nes_pattern0_256x240_to_640x480", sys="nesp0", w=256, h=240, outw=640, outh=480, ofmt=R.FMT_BGRA,
         noise=12, knobs=dict(), settings=dict(hue=0), steps=3, image=("ppu", 99)),
    dict(id="nes_pattern2_256x240_to_640x480", sys="nes", w=256, h=240, outw=640, outh=480, ofmt=R.FMT_BGRA,
         noise=0, knobs=dict(scanlines=1), settings=dict(hue=30), steps=3, image=("ppu", 100)),
    dict(id="small_full_buffers_64x48_to_96x240", sys="ntsc", w=64, h=48, ifmt=R.FMT_BGRA, outw=96, outh=240,
         ofmt=R.FMT_BGRA, noise=20, knobs=dict(), settings=dict(as_color=1), steps=2, interlaced=True,
         image=("random", 3), full="small_64x48_to_96x240.npz"),
]


def make_image(case, step):
    kind, seed = case["image"]
    if kind == "ppu":
        ppu = R.synth_ppu(case["w"], case["h"], seed + step)
        return np.concatenate([ppu, ppu[-1:]], axis=0)
    img = R.synth_image(case["w"], case["h"], R.bpp4fmt(case["ifmt"]), seed, kind)
    return np.concatenate([img, img[-1:]], axis=0)      # spare row: crt_ntsc.c:263


def run_case(lib, case, on_step):
    """Drive `lib` (RefLib / Oracle / anything with the crtref surface) through the case."""
    c = lib.new_crt(case["outw"], case["outh"], case["ofmt"])
    for k, v in case["knobs"].items():
        c.set(k, v)
    nes = case["sys"].startswith("nes")
    for step in range(case["steps"]):
        img = make_image(case, step)
        if nes:
            c.settings(img, w=case["w"], h=case["h"], dot_crawl_offset=step % 3, **case["settings"])
        elif step == 0:
            c.settings(img, format=case["ifmt"], w=case["w"], h=case["h"], field=0, frame=0, **case["settings"])
        if "srand" in case:
            lib.srand(case["srand"] + step)
        c.modulate()
        analog = c.analog.copy()
        c.demodulate(case["noise"])
        on_step(step, c, analog)
        if not nes and case.get("interlaced"):
            c.sset("field", c.sget("field") ^ 1)
            if step % 2 == 0:
                c.sset("frame", c.sget("frame") ^ 1)


def record(c, analog):
    return dict(analog=sha(analog), inp=sha(c.inp), out=sha(c.out), ccf=np.asarray(c.ccf).reshape(-1).tolist(),
                hsync=c.get("hsync"), vsync=c.get("vsync"), rn=c.get("rn"))


def main():
    if not R.have_ref("ntsc"):
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    out = {"generator": "tests/golden/make_golden.py", "source": "LMP88959/NTSC-CRT v2.3.2, unmodified, gcc -O3",
           "cases": {}}
    for case in CASES:
        ref = R.RefLib(case["sys"])
        steps = []
        full = {}

        def on_step(step, c, analog):
            steps.append(record(c, analog))
            if case.get("full"):
                full["analog%d" % step] = analog.copy()
                full["inp%d" % step] = c.inp.copy()
                full["out%d" % step] = c.out.copy()
        run_case(ref, case, on_step)
        out["cases"][case["id"]] = steps
        if case.get("full"):
            np.savez_compressed(os.path.join(HERE, case["full"]), **full)
        print(case["id"], "ok", steps[-1]["out"])
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
