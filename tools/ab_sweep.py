#!/usr/bin/env python3
"""A/B measurements in fresh processes, round robin: one line of the spec = one variant,

    <workload> | <label> | ENV=VALUE ENV=VALUE ... [| picture stride pad]

workload = 1080p (1920x1080 x 2048, noise 0) or 640 (640x480 x 4096, noise 24) or any of bench.py's geometry as
WxH:BATCH:NOISE.  CRTHIP_LIBDIR=<dir under ntsc-crt_amd/> selects another build of the library.  Per process: one batch in flight,
3 warm-up + 10 timed field-passes (tools/placement_sweep.py --child); printed: the median over the processes of every kernel
group's mean launch duration (HIP events) and of the field-pass on the host clock, and the decoder's / encoder's value per process.

    python tools/ab_sweep.py spec.txt [--procs 3]
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WL = {"1080p": (1920, 1080, 2048, 0), "640": (640, 480, 4096, 24)}


def run(wl, env, pad):
    if wl in WL:
        w, h, batch, noise = WL[wl]
    else:
        geo, batch, noise = wl.split(":")
        w, h = map(int, geo.split("x"))
        batch, noise = int(batch), int(noise)
    e = dict(os.environ)
    for k, v in env.items():
        e[k] = os.path.join(ROOT, "ntsc-crt_amd", v) if k == "CRTHIP_LIBDIR" else v
    out = subprocess.run([sys.executable, os.path.join(HERE, "placement_sweep.py"), "--child", str(pad), "--batch", str(batch),
                          "--w", str(w), "--h", str(h), "--noise", str(noise)], env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    lines = out.stdout.decode().strip().splitlines()
    if not lines:
        raise RuntimeError(out.stderr.decode()[-300:])
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec")
    ap.add_argument("--procs", type=int, default=3)
    args = ap.parse_args()
    variants = []
    for line in open(args.spec):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        parts = [x.strip() for x in line.split("|")]
        env = dict(kv.split("=", 1) for kv in parts[2].split()) if len(parts) > 2 and parts[2] else {}
        variants.append((parts[0], parts[1], env, int(parts[3]) if len(parts) > 3 else 0))
    res = [[] for _ in variants]
    for r in range(args.procs):
        for i, (wl, name, env, pad) in enumerate(variants):
            try:
                res[i].append(run(wl, env, pad))
            except Exception as ex:                               # noqa: BLE001
                print("# %s %s run %d failed: %s" % (wl, name, r, str(ex)[:200]))
    print("%d fresh processes per variant, round robin; medians; ms per launch (HIP events) / per field-pass (host clock)" % args.procs)
    print("%-14s %-44s | %8s %8s %8s %8s | %9s | %-24s | %s" % ("workload", "variant", "margin", "active", "sync", "decode", "fieldpass", "decode per process", "active per process"))
    for (wl, name, env, pad), rs in zip(variants, res):
        if not rs:
            continue
        med = lambda k: statistics.median(x[k] for x in rs)      # noqa: E731
        print("%-14s %-44s | %8.4f %8.4f %8.4f %8.4f | %9.4f | %-24s | %s" % (
            wl, name, med("template_ms"), med("active_ms"), med("sync_ms"), med("decode_ms"), med("fieldpass_ms"),
            " ".join("%.3f" % x["decode_ms"] for x in rs), " ".join("%.3f" % x["active_ms"] for x in rs)))


if __name__ == "__main__":
    main()
