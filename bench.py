#!/usr/bin/env python3
"""bench.py -- throughput of the NTSC-CRT field-pass hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path (crt_modulate + crt_demodulate, fused launch sequence)
over one device-resident batch of synthetic fields.  Workload = BASELINE.json configs[1]:
640x480 BGRA in -> 640x480 BGRA out, CRT_SYSTEM_NTSC, interlaced (field parity alternates per
frame), full colour, noise 24, hue 0, scanlines 1.  "frames/sec" = field-passes/sec (one
field-pass per input frame, as extra/video_convert.c:259-260 does).

Multi-GPU: one process per GPU, frames sharded by rank (weak scaling: fixed batch per GPU);
the only collective on the data path is the RCCL broadcast of the settings blob from rank 0.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel vs the HBM roofline (bytes per SURVEY.md 8(d) / DESIGN.md)
  cpu_baseline  the reference (oracle/_ref, unmodified sources) or the oracle port, 1 core
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290


def algorithmic_bytes(w, h, in_bpp, outw, outh, out_bpp, scanlines, blend, lines=240, desth=236):
    """compulsory HBM bytes of one field-pass."""
    rows_in = min(desth, h)
    rows_out = 0
    for l in range(lines):                       # crt_core.c:428-432, :661-664 (even field, v_fac 0)
        beg, end = l * outh // lines, min((l + 1) * outh // lines, outh)
        if beg < outh:
            rows_out += max(end - scanlines - beg, 1)
    b = rows_in * w * in_bpp + rows_out * outw * out_bpp
    if blend:
        b += lines * outw * out_bpp
    return b


def cpu_baseline(system, w, h, outw, outh, noise, scanlines, budget_s):
    """Time the reference (or the oracle port) on ONE host core on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import crtref as R
    if R.have_ref(system):
        lib, kind = R.RefLib(system), "reference"
    else:
        R.build_oracle()
        lib, kind = R.Oracle(system), "port"
    c = lib.new_crt(outw, outh, R.FMT_BGRA)
    c.set("scanlines", scanlines)
    nes = system.startswith("nes")
    if nes:
        ppu = R.synth_ppu(w, h, 12345)
        img = np.concatenate([ppu, ppu[-1:]])                 # the reference reads one row past the image (sic)
        c.settings(img, w=w, h=h, dot_crawl_offset=0, hue=0)
    else:
        img = R.synth_image(w, h, 4, 12345)
        c.settings(np.concatenate([img, img[-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, hue=0, field=0, frame=0)
    if system == "vhs":
        lib.srand(1)
    t, _, _ = c.time_fieldpasses(noise, 20, not nes)         # warm-up + calibration
    reps = max(50, min(4000, int(budget_s / (t / 20))))
    t, _, _ = c.time_fieldpasses(noise, reps, not nes)
    return {"value": reps / t, "unit": "frames/sec", "cores": 1, "kind": kind,
            "sample": "%d field-passes of the same %s %dx%d -> %dx%d noise-%d workload, 1 thread" % (reps, system, w, h, outw, outh, noise),
            "host_cores_visible": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="fields per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--noise", type=int, default=24)
    ap.add_argument("--scanlines", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--system", default="ntsc", choices=["ntsc", "ntscp0", "vhs", "nes", "nesp0"],
                    help="non-default systems are extra measurements (BASELINE configs[3], [4]), not the headline")
    ap.add_argument("--fir", type=int, default=0, choices=[0, 4, 5, 6, 7],
                    help="decoder of a USE_CONVOLUTION build of the reference (FIR kernel of N taps) instead of the 3-band equaliser")
    ap.add_argument("--outw", type=int, default=0)
    ap.add_argument("--outh", type=int, default=0)
    ap.add_argument("--sequence", action="store_true", help="treat the batch as ONE video (crthip_sequence) instead of independent frames")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--pixel-tile", type=int, default=0, help="decoder output tile: 0 auto, 16, 32")
    ap.add_argument("--overlap", type=int, default=1, help="chunks alternating between two streams")
    args = ap.parse_args()

    import torch
    import crtlib
    import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w, h, n = args.width, args.height, args.batch
    nes = args.system.startswith("nes")
    if nes:
        w, h = 256, 240
    outw, outh = args.outw or (640 if nes else w), args.outh or (480 if nes else h)
    crt = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, args.system, device=local)
    crt.scanlines = args.scanlines
    crt.eq_fir = args.fir
    crt.reserve(n)
    crt.set_overlap(args.overlap)
    crt.set_pixel_tile(args.pixel_tile)

    # synthetic input, generated on the device: uniform random BGRA bytes per frame
    # (at most `--unique` distinct random frames, tiled to the batch: the kernels' work is data-independent,
    #  and generating tens of GB of random bytes would dominate the run)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345 + rank)
    uniq = min(n, args.unique)
    if nes:
        base = torch.randint(0, 512, (uniq, h + 1, w), dtype=torch.int16, device=dev, generator=gen)
        images = base.repeat((n + uniq - 1) // uniq, 1, 1)[:n][:, :h]
        s = crtlib.Settings(images, hue=0, dot_crawl_offset=[(rank * n + k) % 3 for k in range(n)])
    else:
        base = torch.randint(0, 256, (uniq, h + 1, w, 4), dtype=torch.uint8, device=dev, generator=gen)
        images = base.repeat((n + uniq - 1) // uniq, 1, 1, 1)[:n][:, :h]
        # this rank's contiguous block of the global batch: frames [rank*n, (rank+1)*n)
        parity = [shard.field_parity(rank * n + k) for k in range(n)]
        s = crtlib.Settings(images, format=crtlib.FMT_BGRA, as_color=1, hue=0,
                            field=[a for a, _ in parity], frame=[b for _, b in parity])
    if args.system == "vhs":
        crt.srand([1 + rank * n + k for k in range(n)])

    # settings blob: built on rank 0, broadcast over RCCL/xGMI (the path's only collective)
    p = crt.params(s, args.noise)
    if dist is not None:
        shard.broadcast_params(p, dist, dev)
    crt._load_field_state(s)

    def step(k):
        if args.sequence:
            crt.sequence(s, args.noise)
            return
        crt.fieldpass(s, args.noise, params=p)
        # next field of the interlaced sequence (video_convert.c:261-267)
        if not nes:
            crt.state[:, crtlib.ST_FIELD] ^= 1
            if k % 2 == 0:
                crt.state[:, crtlib.ST_FRAME] ^= 1

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = shard.max_over_ranks(elapsed, dist, dev)

    # per-kernel durations with HIP events on the launch stream (separate short run, same workload)
    crt.profile(True)
    for k in range(min(args.steps, 5)):
        step(k)
    prof = crt.profile_read()
    crt.profile(False)
    kern_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
    dom = max(kern_ms, key=lambda k: kern_ms[k])

    if rank == 0:
        total_frames = world * n * args.steps
        fps = total_frames / elapsed
        abytes = algorithmic_bytes(w, h, 2 if nes else 4, outw, outh, 4, args.scanlines, 0, desth=240 if nes else 236)
        achieved = abytes * n / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        headline = (args.system, w, h, outw, outh, args.noise, args.scanlines, args.fir) == ("ntsc", 640, 480, 640, 480, 24, 1, 0)
        if headline and os.path.exists(tpath):               # the PMC passes were made on the headline workload only
            try:
                traffic = json.load(open(tpath)).get("k_" + dom + "_bytes_per_field")
                traffic = traffic * n if traffic else None
            except Exception:
                traffic = None
        # cycle-weighted VALU bound of the dominant kernel at 640x480 (the bound that actually binds there):
        # ISA-inspected inner loops of decoder tier 0, ~60 VALU ~ 170 cycles per sample (16 filter stages) and
        # ~46 VALU ~ 130 cycles per pixel with the measured issue costs (profiles/r01_valu_issue_rates.txt),
        # 3.75 waves per field, 1024 SIMDs
        valu = None
        if dom == "decode" and (args.system, w, h, outw, outh, args.fir) == ("ntsc", 640, 480, 640, 480, 0):
            cycles_per_field = 3.75 * (756 * 170 + 640 * 130)
            clk = 2.34e9                                   # GRBM_GUI_ACTIVE / duration in profiles/r01_final_sq_counters.json
            need_ms = cycles_per_field * n / 1024.0 / clk * 1e3
            valu = {"bound": "int-valu (cycle-weighted)", "simd_cycles_per_field": cycles_per_field, "clock_hz": clk,
                    "min_kernel_ms": need_ms, "frac": need_ms / kern_ms[dom],
                    "source": "DESIGN.md section 5, profiles/r01_valu_issue_rates.txt, profiles/r01_final_sq_counters.json"}
        out = {
            "metric": "frames/sec at 640x480 interlaced, bit-exact vs CPU; % HBM roofline",
            "value": fps, "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s %dx%d -> %dx%d BGRA, %s, full colour, noise %d, hue 0, scanlines %d%s"
                                   % (args.system.upper(), w, h, outw, outh, "progressive" if nes else "interlaced", args.noise,
                                      args.scanlines, " (BASELINE configs[1])" if (args.system, w, h, outw, outh, args.fir) == ("ntsc", 640, 480, 640, 480, 0)
                                      else (", %d-tap FIR decoder (USE_CONVOLUTION build)" % args.fir if args.fir else "")),
                       "fields_per_gpu_per_step": n, "frames_per_step": world * n,
                       "sharding": "frames by rank, RCCL broadcast of settings only",
                       "mode": "one video per GPU (crthip_sequence)" if args.sequence else "independent frames (crthip_fieldpass)"},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_field": abytes,
                         "kernel_ms": kern_ms,
                         "pipeline_achieved": abytes * n * args.steps / elapsed / 1e9,
                         "pipeline_frac": abytes * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "valu_roofline": valu,
                         "note": "640x480 is integer-VALU bound (~37 ops/B, SURVEY.md 8(d)); see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.system + ("fir%d" % args.fir if args.fir else ""), w, h, outw, outh,
                                               args.noise, args.scanlines, args.cpu_seconds)
            out["gpu_over_cpu"] = fps / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
