#!/usr/bin/env python3
"""Workgroup order of the three big kernels (crt_dev.h, block_item): k_decode_wide / k_decode / k_active with their work items
dealt out in K strides, K = 0 / 1 (in order), 8 (one contiguous share of the batch per XCD), 64, 512, -1 (one stride per field).
Fresh processes, round robin; per process one batch in flight, 3 warm-up + 10 timed field-passes (tools/placement_sweep.py --child).

    python tools/order_sweep.py [--procs 3] > profiles/r06_block_order.txt
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
WL = {"1080p": dict(w=1920, h=1080, batch=2048, noise=0), "640": dict(w=640, h=480, batch=4096, noise=24)}


def run(wl, env, pad=0):
    a = WL[wl]
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, os.path.join(HERE, "placement_sweep.py"), "--child", str(pad), "--batch", str(a["batch"]),
                          "--w", str(a["w"]), "--h", str(a["h"]), "--noise", str(a["noise"])], env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
    return json.loads(out.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=3)
    args = ap.parse_args()
    variants = []
    for k in (1, 8, 64, 512, -1):
        variants.append(("1080p", "wide order %4d" % k, dict(CRTHIP_WIDE_ORDER=k), 0))
    for k in (8, 64, 512, -1):
        variants.append(("1080p", "wide -1, active order %4d" % k, dict(CRTHIP_WIDE_ORDER=-1, CRTHIP_ACT_ORDER=k), 0))
    variants.append(("1080p", "wide -1, stride + 2 MiB + 256", dict(CRTHIP_WIDE_ORDER=-1), 2097152 + 256))
    for k in (0, 8, 64, 512, -1):
        variants.append(("640", "decode order %4d" % k, dict(CRTHIP_DEC_ORDER=k), 0))
    for k in (8, 64, 512, -1):
        variants.append(("640", "active order %4d" % k, dict(CRTHIP_ACT_ORDER=k), 0))
    variants.append(("640", "decode -1, active -1", dict(CRTHIP_DEC_ORDER=-1, CRTHIP_ACT_ORDER=-1), 0))
    res = {v[:2]: [] for v in variants}
    for r in range(args.procs):
        for wl, name, env, pad in variants:
            try:
                res[(wl, name)].append(run(wl, env, pad))
            except Exception as ex:                               # noqa: BLE001
                print("# %s %s run %d failed: %s" % (wl, name, r, ex))
    print("Workgroup order (crt_dev.h block_item), %d fresh processes per variant, medians; ms per launch (HIP events) / per field-pass (host clock)" % args.procs)
    print("1080p = 1920x1080 x 2048 noise 0;  640 = 640x480 x 4096 noise 24;  one batch in flight")
    print()
    print("%-6s %-34s | %8s %8s %8s %8s | %9s | %s" % ("", "variant", "margin", "active", "sync", "decode", "fieldpass", "decode per process"))
    for wl, name, env, pad in variants:
        rs = res[(wl, name)]
        if not rs:
            continue
        med = lambda k: statistics.median(x[k] for x in rs)      # noqa: E731
        print("%-6s %-34s | %8.4f %8.4f %8.4f %8.4f | %9.4f | %s" % (wl, name, med("template_ms"), med("active_ms"), med("sync_ms"), med("decode_ms"),
                                                                    med("fieldpass_ms"), " ".join("%.3f" % x["decode_ms"] for x in rs)))


if __name__ == "__main__":
    main()
