#!/usr/bin/env python3
"""One-off soak: tests/test_gpu_parity.py's random configurations (HIP vs oracle, bit-exact) over a seed range,
also for the VHS / NES / FIR variants.  usage: tools/soak_random.py first_seed count"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ntsc-crt_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import crtlib
import test_gpu_parity as T
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    case = list(T._random_case(rng))
    variant = seed % 4
    if variant == 1:
        case[0] = "ntscfir%d" % (4 + seed % 4)
    elif variant == 2:
        case[0] = "ntscp0"
    try:
        T._run_case(crtlib, tuple(case), fused=bool(seed & 8), steps=2, n=2)
    except Exception as e:
        bad += 1
        print("SEED", seed, case[:8], "FAILED:", str(e).splitlines()[0][:200])
print("soak: %d cases, %d failures" % (count, bad))
