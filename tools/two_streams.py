#!/usr/bin/env python3
"""Experiment: independent batches on several streams (one context each) against one batch on one stream.
usage: tools/two_streams.py [batch] [width height]
modes: plain = launches enqueued back to back; stagger = streams started 1/S of a step apart (spin kernel);
       hostsync = the host waits for stream (k-1) % S's previous step before enqueueing step k (keeps them apart)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ntsc-crt_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch, crtlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
noise = 24 if w <= 640 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
base = torch.randint(0, 256, (64, h + 1, w, 4), dtype=torch.uint8, generator=g).to(dev)
imgs = base.repeat((n + 63) // 64, 1, 1, 1)[:n]
def make():
    st = torch.cuda.Stream(device=dev)
    c = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
    c.scanlines = 1
    c.use_stream(st)
    s = crtlib.Settings(imgs, format=crtlib.FMT_BGRA, field=[k & 1 for k in range(n)], frame=0)
    p = c.params(s, noise)
    c._load_field_state(s)
    return c, s, p, st
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); torch.cuda._sleep(20000000); ev1.record(); torch.cuda.synchronize()
spin_per_ms = 20000000 / ev0.elapsed_time(ev1)
def timeit(cs, mode, t_step=0.0, steps=24, warm=3):
    S = len(cs)
    for k in range(warm * S):
        c, s, p, st = cs[k % S]; c.fieldpass(s, noise, params=p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "stagger":
        for i in range(1, S):
            with torch.cuda.stream(cs[i][3]):
                torch.cuda._sleep(int(t_step * i / S * spin_per_ms))
    evs = [None] * S
    for k in range(steps):
        c, s, p, st = cs[k % S]
        if mode == "hostsync" and k >= S - 1 and S > 1:
            e = evs[(k + 1) % S]          # the oldest batch still in flight
            if e is not None: e.synchronize()
        if mode == "flips":
            with torch.cuda.stream(st):
                c.fieldpass(s, noise, params=p)
                c.state[:, crtlib.ST_FIELD] ^= 1
                if (k // S) % 2 == 0:
                    c.state[:, crtlib.ST_FRAME] ^= 1
        else:
            c.fieldpass(s, noise, params=p)
        if mode == "hostsync":
            e = torch.cuda.Event(); e.record(st); evs[k % S] = e
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return steps * n / dt, dt / steps * 1e3
ctxs = [make() for _ in range(3)]
for rep in range(2):
    f1, t1 = timeit(ctxs[:1], "plain")
    print("1 stream          : %8.0f fps %.3f ms" % (f1, t1))
    for S in (2, 3):
        for mode in ("plain",):
            print("%d streams %-8s: %8.0f fps %.3f ms" % ((S, mode) + timeit(ctxs[:S], mode, t1)))
