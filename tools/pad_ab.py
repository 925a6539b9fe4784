#!/usr/bin/env python3
"""Padded against flat signal lines (CRTHIP_SIG_PAD=1 / 0) for the workloads beside the headline: bench.py in fresh processes, round
robin, one batch in flight; frames/s and the kernel groups' ms.   python tools/pad_ab.py [--procs 3] > profiles/r06_ab_padded_by_system.txt"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = [("nesp0 x 4096", ["--system", "nesp0", "--noise", "12"]), ("pv1k x 4096", ["--system", "pv1k"]), ("ntscbloom x 4096", ["--system", "ntscbloom"]),
      ("snes x 4096", ["--system", "snes"]), ("ntsc 640x480 x 1024", ["--batch", "1024"]), ("ntsc 640x480 x 256", ["--batch", "256"]),
      ("ntsc 640x480 x 64", ["--batch", "64"]), ("ntsc 1080p x 64", ["--width", "1920", "--height", "1080", "--noise", "0", "--batch", "64"]),
      ("ntsc 1080p x 512", ["--width", "1920", "--height", "1080", "--noise", "0", "--batch", "512"])]
procs = int(sys.argv[sys.argv.index("--procs") + 1]) if "--procs" in sys.argv else 3
res = {}
for r in range(procs):
    for name, args in WL:
        for pad in ("1", "0"):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra", "--streams", "1", "--steps", "20", "--warmup", "3"] + args,
                                 env=dict(os.environ, CRTHIP_SIG_PAD=pad), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
            try:
                j = json.loads(out.strip().splitlines()[-1])
                res.setdefault((name, pad), []).append((j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
            except Exception as ex:                                  # noqa: BLE001
                print("# %s pad %s failed: %s" % (name, pad, ex))
print("%-22s %-7s | %12s %9s | %8s %8s %8s %8s | %s" % ("workload", "lines", "frames/s", "ms/step", "margin", "active", "sync", "decode", "frames/s per process"))
for name, _ in WL:
    for pad in ("1", "0"):
        rs = res.get((name, pad))
        if not rs:
            continue
        med = lambda f: statistics.median(f(x) for x in rs)          # noqa: E731
        print("%-22s %-7s | %12.0f %9.4f | %8.4f %8.4f %8.4f %8.4f | %s" % (name, "padded" if pad == "1" else "flat", med(lambda x: x[0]), med(lambda x: x[1]),
              med(lambda x: x[2].get("template", 0)), med(lambda x: x[2].get("active", 0)), med(lambda x: x[2].get("sync", 0)), med(lambda x: x[2].get("decode", 0)),
              " ".join("%.0f" % x[0] for x in rs)))
