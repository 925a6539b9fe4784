#!/usr/bin/env python3
"""One-off soak: tests/test_gpu_parity.py's random configurations (HIP vs oracle, bit-exact) over a seed range,
also for the VHS / NES / FIR variants.  usage: tools/soak_random.py first_seed count [wide]
(wide: picture widths of 1500-4095 pixels, i.e. the wide-run decoder of crt_decode4.hip where it applies; lane shape only)"""
# seeds alternate systems (NTSC, FIR builds, pattern 0, bloom builds, SNES, template, NES-RGB, PV-1000) and both kernel shapes
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ntsc-crt_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import crtlib
import test_gpu_parity as T
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
if len(sys.argv) > 3 and sys.argv[3] == "wide":
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed ^ 0x71de)
        case = T._random_wide_case(rng)
        try:
            T._run_case(crtlib, case, fused=bool(seed & 1), steps=2, n=2, shape=1)
        except Exception as e:
            bad += 1
            print("WIDE SEED", seed, case[:8], "FAILED:", str(e).splitlines()[0][:200])
    print("soak: %d wide cases, %d failures" % (count, bad))
    sys.exit(1 if bad else 0)
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    case = list(T._random_case(rng))
    variant = seed % 8
    shape = 2 if seed & 16 else 1                     # both kernel shapes
    if variant == 1:
        case[0] = "ntscfir%d" % (4 + (seed >> 3) % 4)
        shape = 1                                     # the FIR decoder exists in the lane-per-scanline shape only
    elif variant == 2:
        case[0] = "ntscp0"
    elif variant == 3:                                # CRT_DO_BLOOM builds, both shapes (shape 1: lines sorted by beam width)
        case[0] = ("ntscbloom", "snesbloom", "pv1kbloom")[(seed >> 5) % 3]
    elif variant in (4, 5, 6, 7):                     # SURVEY 8(f) f4 systems
        case[0] = ("snes", "temp", "nesrgb", "pv1k")[variant - 4]
    try:
        T._run_case(crtlib, tuple(case), fused=bool(seed & 8), steps=2, n=2, shape=shape)
    except Exception as e:
        bad += 1
        print("SEED", seed, case[:8], "FAILED:", str(e).splitlines()[0][:200])
# NES: random output sizes / knobs through the NES parity test (table encoder, 3-line chroma period, dot crawl)
nes_bad = 0
nes_count = max(1, count // 10)
for seed in range(first, first + nes_count):
    rng = np.random.default_rng(seed ^ 0x5e5)
    knobs = dict(hue=int(rng.integers(-360, 720)), brightness=int(rng.integers(-40, 40)), contrast=int(rng.integers(0, 400)),
                 saturation=int(rng.integers(-5, 40)), black_point=int(rng.integers(-10, 10)), white_point=int(rng.integers(50, 150)),
                 scanlines=int(rng.integers(0, 2)), blend=int(rng.integers(0, 2)))
    case = (str(rng.choice(["nes", "nesp0"])), int(rng.choice([1, 33, 256, 640, 768, 1283])), int(rng.choice([240, 241, 480, 500, 720])),
            int(rng.choice([0, 1, 24, 77, 300])), knobs)
    T.NES_CASES.append(case)
    try:
        T.test_nes_parity(crtlib, len(T.NES_CASES) - 1, bool(seed & 1))
    except Exception as e:
        nes_bad += 1
        print("NES SEED", seed, case, "FAILED:", str(e).splitlines()[0][:200])
print("soak: %d NTSC-family cases, %d failures; %d NES cases, %d failures" % (count, bad, nes_count, nes_bad))
