#!/usr/bin/env python3
"""is k_decode_wide's run-to-run spread at 1920x1080 a matter of where the pictures sit?

Every variant is measured in FRESH processes (the spread in question is between consecutive processes on one box): the batch
ABI's picture stride padded off the tight 8 294 400 bytes (the picture inside a field stays tight, so parity and the algorithmic
bytes are untouched), and the decoder's workgroups field-interleaved (CRTHIP_WIDE_ORDER=-1, since this measurement the default; 1 = in order) so that the waves resident together
write rows of different pictures.  Per process: 1920x1080 x 2048, noise 0, one batch in flight, 3 warm-up + 10 timed field-passes
with an event pair around every launch (crthip_profile_enable); printed: the decoder's and the encoder's mean launch duration
and the field-pass on the host clock.

    python tools/placement_sweep.py [--procs 5] [--batch 2048] > profiles/r06_1080p_placement.txt
    python tools/placement_sweep.py --child PAD        (one measurement; prints one JSON line)
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))


def child(pad, batch, steps, warmup, w=1920, h=1080, noise=0):
    import torch
    import crtlib
    dev = torch.device("cuda", 0)
    tight = w * h * 4
    stride = tight + pad
    buf = torch.zeros(batch * stride + 4096, dtype=torch.uint8, device=dev)
    out = buf[:batch * stride].view(batch, stride)[:, :tight].view(batch, h, w, 4)       # stride(0) = tight + pad bytes
    g = crtlib.CRT(batch, w, h, crtlib.FMT_BGRA, "ntsc", device=0, out=out)
    g.scanlines = 1
    g.reserve(batch)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345)
    base = torch.randint(0, 256, (64, h + 1, w, 4), dtype=torch.uint8, device=dev, generator=gen)
    images = base.repeat(batch // 64, 1, 1, 1)[:, :h]
    s = crtlib.Settings(images, format=crtlib.FMT_BGRA, as_color=1, hue=0, field=[k & 1 for k in range(batch)], frame=0)
    p = g.params(s, noise)
    g._load_field_state(s)

    def step(k):
        g.fieldpass(s, noise, params=p)
        g.state[:, crtlib.ST_FIELD] ^= 1
    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    g.profile(True)
    for k in range(steps):
        step(k)
    pr = g.profile_read()
    g.profile(False)
    rec = {"pad": pad, "order": int(os.environ.get("CRTHIP_WIDE_ORDER", "0")), "fieldpass_ms": round(wall, 4)}
    rec["env"] = {k: v for k, v in os.environ.items() if k.startswith("CRTHIP_")}
    for name, (ms, cnt) in pr.items():
        rec[name + "_ms"] = round(ms / steps, 4)
    rec["out_ptr_mod_64k"] = out.data_ptr() % 65536
    rec["ptrs"] = {"out": hex(out.data_ptr()), "images": hex(images.data_ptr()), "state": hex(g.state.data_ptr())}
    if os.environ.get("SWEEP_SERIES"):
        # per-launch durations (is a slow process slow in every launch?) and the board's state
        series = {"decode": [], "active": []}
        g.profile(True)
        for k in range(steps):
            step(k)
            pr = g.profile_read()
            for nm in series:
                series[nm].append(round(pr[nm][0], 3))
        g.profile(False)
        rec["series"] = series
        # the board's clocks and power WHILE the field-passes run (sysfs, sampled from a thread every 20 ms over ~1.5 s of launches)
        import glob
        import threading
        devs = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(d + "/pp_dpm_mclk")]
        samples, stop = [], threading.Event()

        def active_level(path):
            try:
                for line in open(path):
                    if line.rstrip().endswith("*"):
                        return line.split(":")[1].strip().rstrip("*").strip()
            except OSError:
                return None
            return None

        def sampler():
            d = devs[0]
            hw = (glob.glob(d + "/hwmon/hwmon*") or [None])[0]
            while not stop.is_set():
                row = {k: active_level("%s/pp_dpm_%s" % (d, k)) for k in ("sclk", "mclk", "fclk", "socclk")}
                if hw:
                    for name, f in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("temp_junction_mC", "temp2_input"), ("temp_mem_mC", "temp3_input")):
                        try:
                            row[name] = int(open(hw + "/" + f).read())
                        except (OSError, ValueError):
                            pass
                samples.append(row)
                time.sleep(0.02)
        if devs:
            th = threading.Thread(target=sampler)
            th.start()
            for k in range(400):
                step(k)
            torch.cuda.synchronize()
            stop.set()
            th.join()
            import collections
            summ = {}
            for key in ("sclk", "mclk", "fclk", "socclk"):
                summ[key] = dict(collections.Counter(r_.get(key) for r_ in samples))
            for key in ("power_uW", "power_in_uW", "temp_junction_mC", "temp_mem_mC"):
                vals = [r_[key] for r_ in samples if key in r_]
                if vals:
                    summ[key] = [min(vals), max(vals)]
            rec["under_load"] = summ
        try:
            smi = subprocess.run(["rocm-smi", "--showtemp", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, timeout=30).stdout.decode()
            j = json.loads(smi)
            card = j[sorted(j)[0]]
            rec["smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("temperature", "sclk", "mclk", "fclk", "power"))}
        except Exception as ex:                                  # noqa: BLE001
            rec["smi"] = str(ex)[:80]
    print(json.dumps(rec))
    g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", type=int, default=None)
    ap.add_argument("--procs", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--noise", type=int, default=0)
    args = ap.parse_args()
    if args.child is not None:
        return child(args.child, args.batch, args.steps, args.warmup, args.w, args.h, args.noise)
    # (CRTHIP_WIDE_ORDER: 1 = workgroups in order, the shipped order of rounds 4 / 5; -1 = one stride per field, round 6's default)
    variants = [("tight (shipped)", 0, 1), ("stride + 256", 256, 1), ("stride + 1280", 1280, 1), ("stride + 4096", 4096, 1),
                ("stride + 4352", 4352, 1), ("stride + 65792", 65792, 1), ("stride + 2 MiB + 256", 2097152 + 256, 1),
                ("field-interleaved workgroups", 0, -1), ("field-interleaved, stride + 4352", 4352, -1)]
    print("k_decode_wide<SysNTSC, 16> against picture placement: 1920x1080 x %d, noise 0, one batch in flight, %d fresh processes per variant,"
          % (args.batch, args.procs))
    print("%d timed field-passes each (tools/placement_sweep.py).  decode / active = mean launch duration (HIP events), fieldpass = host clock." % args.steps)
    print()
    print("%-36s | %-44s | %-7s %-7s | %-7s | %s" % ("variant", "decode ms per process", "median", "spread", "active", "fieldpass ms (median)"))
    rows = []
    # round robin over the variants, so that a drift of the box over the minutes hits all of them alike
    results = {v[0]: [] for v in variants}
    for r in range(args.procs):
        for name, pad, order in variants:
            env = dict(os.environ, CRTHIP_WIDE_ORDER=str(order))
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(pad), "--batch", str(args.batch),
                                      "--steps", str(args.steps), "--warmup", str(args.warmup)], env=env, stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
                results[name].append(json.loads(out.strip().splitlines()[-1]))
            except Exception as ex:                               # noqa: BLE001 - a line of the sweep may fail; say so
                print("# %s run %d failed: %s" % (name, r, ex))
    for name, pad, order in variants:
        rs = results[name]
        if not rs:
            continue
        dec = [x["decode_ms"] for x in rs]
        med = statistics.median(dec)
        spread = (max(dec) - min(dec)) / med * 100.0
        print("%-36s | %-44s | %7.3f %6.1f%% | %7.3f | %.3f" % (name, " ".join("%.3f" % d for d in dec), med, spread,
                                                              statistics.median(x["active_ms"] for x in rs),
                                                              statistics.median(x["fieldpass_ms"] for x in rs)))
        rows.append((name, med, spread))
    print()
    base = rows[0][1] if rows else 0.0
    for name, med, spread in rows[1:]:
        print("%-36s %+5.1f %% against the shipped layout's median" % (name, (med / base - 1.0) * 100.0))


if __name__ == "__main__":
    main()
