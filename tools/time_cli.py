#!/usr/bin/env python3
"""BASELINE configs[0] as BASELINE states it: the whole-process wall time of

    ./ntsc -op 640 480 0 0 in.ppm out.ppm          (crt_main.c:242-255: 4 field-passes with blend, PPM in, PPM out)

for the reference's own binary (oracle/_ref/ntsc_cli: the unmodified sources, gcc -O3) and for the SAME crt_main.c linked
against the HIP drop-in library (ntsc-crt_amd/lib/ntsc_cli_hip), on the synthetic image SURVEY.md 8(d) names (colour bars,
gradient / XOR rows, uniform random rows; seed 1).  The outputs are compared byte for byte.  A one-shot CLI process pays the
HIP runtime's start-up (device discovery, code-object load, context, first-launch) on every call -- that, not the four
field-passes, is what its wall time consists of; the library-level per-call cost is tools/time_dropin.py's subject.

    tools/time_cli.py [runs]          prints a table, returns the numbers as a dict through measure()"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ntsc_cli")
HIP = os.path.join(ROOT, "ntsc-crt_amd", "lib", "ntsc_cli_hip")
FLOOR = os.path.join(ROOT, "ntsc-crt_amd", "lib", "hip_floor")        # tools/hip_floor.hip: hipInit, hipMalloc, one trivial kernel, exit
PROBE = os.path.join(ROOT, "ntsc-crt_amd", "lib", "startup_probe")    # tools/startup_probe.c: the drop-in API, wall clock around every call


def config1_ppm(path, w=640, h=480, seed=1):
    import numpy as np
    img = np.zeros((h, w, 3), dtype=np.uint8)
    cols = [(255, 255, 255), (255, 255, 0), (0, 255, 255), (0, 255, 0), (255, 0, 255), (255, 0, 0), (0, 0, 255), (0, 0, 0)]
    for k, c in enumerate(cols):
        img[:h * 2 // 3, k * w // 8:(k + 1) * w // 8] = c
    yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
    a, b = h * 2 // 3, h * 5 // 6
    img[a:b] = (np.where(yy[a:b] < (a + b) // 2, xx * 255 // (w - 1), (xx ^ yy[a:b]) & 255)[..., None]).astype(np.uint8)
    x = seed
    rnd = np.empty((h - b) * w * 3, dtype=np.uint8)
    for i in range(rnd.size):                              # the LCG of SURVEY.md 8(c)
        x = (x * 1664525 + 1013904223) & 0xffffffff
        rnd[i] = x >> 24
    img[b:] = rnd.reshape(h - b, w, 3)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


def measure(runs=5, flags="-op", noise=0):
    """{'ref_ms': [min, median], 'hip_ms': [...], 'hip_lazy_ms': [...], 'identical': bool, 'runs': n} or None if a binary is missing"""
    if not (os.path.exists(REF) and os.path.exists(HIP)):
        return None
    d = tempfile.mkdtemp(prefix="cli1_")
    src = os.path.join(d, "in.ppm")
    config1_ppm(src)
    res, outs = {}, {}
    for tag, exe, env in (("ref", REF, {}), ("hip", HIP, {}), ("hip_lazy", HIP, {"CRTHIP_LAZY_MIRROR": "1"})):
        out = os.path.join(d, "out_%s.ppm" % tag)
        ts = []
        for _ in range(runs + 1):                          # (+1: the first run also pages the binaries in)
            t0 = time.perf_counter()
            r = subprocess.run([exe, flags, "640", "480", str(noise), "0", src, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               env=dict(os.environ, **env))
            ts.append(1e3 * (time.perf_counter() - t0))
            if r.returncode != 0:
                raise RuntimeError("%s failed: %s" % (exe, r.stderr.decode()[-300:]))
        ts = sorted(ts[1:])
        res[tag + "_ms"] = [round(ts[0], 1), round(ts[len(ts) // 2], 1)]
        outs[tag] = open(out, "rb").read()
    if os.path.exists(FLOOR):
        # what ANY one-shot HIP process pays on this box before its first kernel has run: the runtime's share of hip_ms
        ts = []
        for _ in range(runs + 1):
            t0 = time.perf_counter()
            subprocess.run([FLOOR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            ts.append(1e3 * (time.perf_counter() - t0))
        ts = sorted(ts[1:])
        res["hip_floor_ms"] = [round(ts[0], 1), round(ts[len(ts) // 2], 1)]
    res["identical"] = outs["ref"] == outs["hip"] == outs["hip_lazy"]
    res["runs"] = runs
    res["cmd"] = "ntsc %s 640 480 %d 0 in.ppm out.ppm" % (flags, noise)
    return res


if __name__ == "__main__":
    r = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    if r is None:
        sys.exit("driver binaries not built (oracle/Makefile builds them where /root/reference exists)")
    print("# %s, whole process, wall clock, %d runs each: min / median ms" % (r["cmd"], r["runs"]))
    print("reference binary (unmodified sources, gcc -O3, 1 core)      %8.1f / %8.1f" % tuple(r["ref_ms"]))
    print("same crt_main.c + libntsccrt_hip_ntsc.so (strict mirror)     %8.1f / %8.1f" % tuple(r["hip_ms"]))
    print("same, CRTHIP_LAZY_MIRROR=1                                   %8.1f / %8.1f" % tuple(r["hip_lazy_ms"]))
    if "hip_floor_ms" in r:
        print("a one-shot HIP process that launches ONE trivial kernel       %8.1f / %8.1f   <- the runtime's floor on this box" % tuple(r["hip_floor_ms"]))
    print("output images byte-identical: %s" % r["identical"])
    if os.path.exists(PROBE):
        print("# the drop-in API called like crt_main.c does (tools/startup_probe.c), wall clock around every call:")
        print(subprocess.run([PROBE], capture_output=True, text=True).stdout.rstrip())
        print("# the same with the HIP runtime initialised by hand first (what of crt_modulate #1 is hipInit):")
        print(subprocess.run([PROBE, "hipinit"], capture_output=True, text=True).stdout.rstrip())
