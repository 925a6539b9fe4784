#!/usr/bin/env python3
"""bench.py -- throughput of the NTSC-CRT field-pass hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: if the process was not started by torch.distributed.run (no WORLD_SIZE in the
environment) it re-executes itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ... bench.py <same arguments>`, one rank per GPU; started by torch.distributed.run it is one of the ranks.

A "step" is one pass of the hot path (crt_modulate + crt_demodulate, fused launch sequence) over one device-resident
batch of synthetic fields.  Headline workload = BASELINE.json configs[1]: 640x480 BGRA in -> 640x480 BGRA out,
CRT_SYSTEM_NTSC, interlaced (field parity alternates per frame), full colour, noise 24, hue 0, scanlines 1.
"frames/sec" = field-passes/sec (one field-pass per input frame, as extra/video_convert.c:259-260 does).

Batches in flight: consecutive steps are independent batches.  With S of them in flight step k runs on context k % S (its own
stream, state, signal and picture buffers) and the launch sequences overlap on the chip.  S is tuned: the K steps are timed
for S = 1, 2 and 3 (each bracketed by the barrier + synchronize) and the best one is `value`; `config.batches_in_flight_tuning`
holds all three, `one_batch_in_flight` the S = 1 figure, `roofline.kernel_ms` is measured with ONE batch in flight.
`--streams 1` pins S.  (profiles/r03_streams_sweep.txt)

Multi-GPU: one process per GPU, frames sharded by rank; the only collective on the data path is the RCCL broadcast of
the settings blob from rank 0.  Default: weak scaling (fixed batch per GPU).  `--strong F` splits F frames of
BASELINE configs[2] (1920x1080, noise 0) over the ranks (F = 512 is the configuration BASELINE states).

Prints ONE JSON line (rank 0), the LAST line of stdout, at most 4 KB (compact_record(); the round-3 line had grown to
23 KB and the driver, which keeps 8 KB of output, could not parse it).  Objects of the line:
  roofline         dominant kernel vs the HBM roofline (bytes per SURVEY.md 8(d) / DESIGN.md section 5)
  cpu_baseline     the reference (oracle/_ref, unmodified sources) or the oracle port on the host cores: 1 core, and
                   all cores (independent processes)
  one_batch_in_flight   the S = 1 figure next to `value`
  extras           (N = 1) one short row per extra workload: 1080p at batch 2048, 512 and at configs[2]'s per-GPU share
                   (64), VHS 832x624 (configs[3]), NES pattern 0 (configs[4]), small batches, PV-1000, bloom
  strong_scaling   (N = 1) configs[2] on one GPU: T(512 frames), T(64 frames = the per-GPU share at 8 GPUs) and the
                   speed-up 8 GPUs can reach at best, T512 / T64
The complete record (every workload with its own roofline / cpu_baseline / tuning objects) goes to
gpurun_out/bench_full.json (and --full-json PATH).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290
VALU_SIMDS = 1024             # 256 CUs x 4 SIMDs
VALU_CLOCK_HZ = 2.4e9          # peak engine clock (MI355X_MICROARCH.md); the kernels run at 2.2-2.3 GHz (GRBM_GUI_ACTIVE / duration)


def geometry(system, w, h, outw, outh, scanlines, bloom=False):
    """Rows / samples of one field-pass (crt_ntsc.c:132-133, crt_core.c:428-432, :661-664; even field, v_fac 0)."""
    nes = system.startswith("nes")
    lines = 240
    hres = {"nes": 909, "nesp0": 912, "nesrgb": 909, "snes": 909, "pv1k": 1920, "ntscp0": 912}.get(system, 910)
    av_len = {"nes": 682, "nesp0": 684, "nesrgb": 682, "snes": 682, "pv1k": 1487, "ntscp0": 754}.get(system, 753)
    desth = lines if nes else (lines * (63500 if bloom else 64500)) >> 16
    destw = av_len if not bloom else (av_len * 55500) >> 16
    rows_out = 0
    for l in range(lines):
        beg, end = l * outh // lines, min((l + 1) * outh // lines, outh)
        if beg < outh:
            rows_out += max(end - scanlines - beg, 1)
    return dict(lines=lines, hres=hres, av_len=av_len, desth=desth, destw=destw, rows_in=min(desth, h), rows_out=rows_out,
                input_size=hres * 262)


def algorithmic_bytes(system, w, h, in_bpp, outw, outh, out_bpp, scanlines, blend):
    """compulsory HBM bytes of one field-pass, and the share each kernel moves itself."""
    g = geometry(system, w, h, outw, outh, scanlines)
    img = g["rows_in"] * w * in_bpp
    pic = g["rows_out"] * outw * out_bpp + (g["lines"] * outw * out_bpp if blend else 0)
    act = g["destw"] * g["desth"]
    own = {"template": g["input_size"] - act,                       # margins of inp[] written
           "active": img + act,                                     # image rows read, active samples written
           "noise": 2 * g["input_size"],
           "sync": 25000 + g["lines"] * 32,                         # sync / burst windows read, line table written
           "decode": g["lines"] * g["av_len"] + pic}                # sample windows read, picture written
    return img + pic, own


def kernel_source_hash():
    """sha1 over the device sources: profiles/traffic*.json (PMC passes, collected by tools/collect_profiles.py) carry the
    hash of the sources they were measured on, so a static traffic number that no longer belongs to the kernels says so
    (`roofline.traffic_stale`)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "ntsc-crt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ntsc-crt_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_time_workload(system, w, h, outw, outh, noise, scanlines, budget_s):
    """Time the reference (or the oracle port) on the calling thread on a bounded sample; returns (fps, reps, kind)."""
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
    nes = system in ("nes", "nesp0")
    if nes:
        ppu = R.synth_ppu(w, h, 12345)
        img = np.concatenate([ppu, ppu[-1:]])                 # the reference reads one row past the image (sic)
        c.settings(img, w=w, h=h, dot_crawl_offset=0, hue=0)
    else:
        img = R.synth_image(w, h, 4, 12345)
        c.settings(np.concatenate([img, img[-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, hue=0, field=0, frame=0)
    if system.startswith("vhs"):
        lib.srand(1)
    t, _, _ = c.time_fieldpasses(noise, 10, not nes)          # warm-up + calibration
    reps = max(20, min(4000, int(budget_s / (t / 10))))
    t, _, _ = c.time_fieldpasses(noise, reps, not nes)
    return reps / t, reps, kind


def cpu_baseline(system, w, h, outw, outh, noise, scanlines, budget_s, all_cores=True):
    """cpu_baseline object: 1 thread on 1 core (the reference is single-threaded), plus an all-cores figure from
    independent processes, one per core."""
    fps, reps, kind = cpu_time_workload(system, w, h, outw, outh, noise, scanlines, budget_s)
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    out = {"value": fps, "unit": "frames/sec", "cores": 1, "kind": kind,
           "sample": "%d field-passes of the same %s %dx%d -> %dx%d noise-%d workload, 1 thread" % (reps, system, w, h, outw, outh, noise),
           "cpu_model": cpu_model(), "host_cores_visible": ncores}
    if all_cores and ncores > 1:
        sec = min(4.0, budget_s)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", system, str(w), str(h), str(outw), str(outh),
               str(noise), str(scanlines), str(sec)]
        t0 = time.perf_counter()
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(ncores)]
        rates = []
        for p in procs:
            o, _ = p.communicate()
            try:
                rates.append(float(o.decode().strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
        out["all_cores"] = {"value": sum(rates), "unit": "frames/sec", "cores": len(rates),
                            # what the processes got out of the box: a container quota or SMT siblings show up here
                            "effective_cores": sum(rates) / fps if fps > 0 else None,
                            "cpu_quota_cores": cpu_quota_cores(),
                            "sample": "%d independent processes x %.0f s of the same workload (wall %.1f s)"
                                      % (len(rates), sec, time.perf_counter() - t0)}
    return out


LINE_LIMIT = 4000             # bytes of the final stdout line (the driver keeps 8 KB of output)


def _r(v, nd=4):
    """numbers at a readable precision (4 significant digits by default); everything else unchanged"""
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        return v
    if isinstance(v, int) or v == 0:
        return v
    return float("%.*g" % (nd, v))


def compact_record(full):
    """The full result record -> the one the final stdout line carries: the contract keys, `roofline` and `cpu_baseline` of
    the headline, the one-batch-in-flight figure, one short row per extra workload and the strong-scaling pair.  Pure
    (tests/test_bench_cpu.py runs it on profiles/r03_bench_default.json)."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(out["value"], 7), _r(out["ms_per_step"], 5)
    cfg = full.get("config") or {}
    out["config"] = {k: cfg[k] for k in ("workload", "fields_per_gpu_per_step", "frames_per_step", "sharding", "mode", "launch",
                                         "batches_in_flight") if k in cfg}
    if cfg.get("batches_in_flight_tuning"):      # S -> frames/sec
        out["config"]["batches_in_flight_fps"] = {s_: _r(v["value"], 5) for s_, v in cfg["batches_in_flight_tuning"].items()}
    rf = full.get("roofline") or {}
    out["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio",
                                                  "traffic_dominant_kernel", "algorithmic_bytes_per_field", "kernel_own_frac", "pipeline_frac")}
    out["roofline"]["kernel_ms"] = {k: _r(v) for k, v in (rf.get("kernel_ms") or {}).items() if v}
    if rf.get("traffic_source"):
        out["roofline"]["traffic_source"] = rf["traffic_source"].split(" (")[0]
    if "traffic_stale" in rf:
        out["roofline"]["traffic_stale"] = rf["traffic_stale"]
    if rf.get("valu"):
        out["roofline"]["valu"] = {"frac_of_4_cycle_issue": _r(rf["valu"].get("frac")),
                                   "frac_of_2_cycle_peak": _r(rf["valu"].get("frac_of_2_cycle_peak")),
                                   "wave_instr_per_field": _r(rf["valu"].get("wave_instr_per_field"), 6)}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_cores_visible")}
        if cb.get("all_cores"):
            ac = cb["all_cores"]
            out["cpu_baseline"]["all_cores"] = {"value": _r(ac.get("value")), "cores": ac.get("cores"),
                                                "effective_cores": _r(ac.get("effective_cores")), "cpu_quota_cores": ac.get("cpu_quota_cores")}
    for k in ("gpu_over_cpu", "gpu_over_cpu_all_cores"):
        if k in full:
            out[k] = _r(full[k])
    ob = full.get("one_batch_in_flight")
    if ob:
        out["one_batch_in_flight"] = {"value": _r(ob["value"], 7), "ms_per_step": _r(ob["ms_per_step"], 5),
                                      "pipeline_frac": _r(ob.get("pipeline_frac"))}
    if full.get("value_spread"):
        out["value_spread"] = {k: _r(v, 6) for k, v in full["value_spread"].items() if k in ("reps", "min", "median", "max")}
    for k in ("world_size", "world_size_seen_by_rccl", "collectives", "settings_blob_crc32_per_rank",
              "settings_blob_crc32_rank0_before_broadcast", "north_star", "strong_scaling", "cli_config1"):
        if full.get(k) is not None:
            out[k] = full[k]
    if not (out.get("collectives") or {}).get("initialized"):
        out.pop("collectives", None)                 # one rank, no process group: nothing ran
    if out.get("strong_scaling"):                    # (the long form stays in the full record; the weak figure is in north_star)
        out["strong_scaling"] = {k: v for k, v in out["strong_scaling"].items() if k in ("T512_ms", "T64_ms", "projected_speedup_8gpu")}
        out["strong_scaling"]["what"] = "configs[2] on ONE GPU, one batch in flight: T(512 frames) / T(64 = a GPU's share at 8)"
    rows = []
    for e in full.get("extra_workloads") or []:
        if not e:
            continue
        erf, eob = e.get("roofline") or {}, e.get("one_batch_in_flight") or {}
        # value = best of 1..3 batches in flight; one = ONE batch in flight (fps, ms per step, end-to-end fraction of 8 TB/s);
        # kfrac (the bandwidth-bound 1080p rows) = roofline.frac of the workload's dominant kernel (k_decode unless named)
        row = {"name": e["name"], "batch": (e.get("config") or {}).get("fields_per_gpu_per_step"), "value": _r(e["value"], 4),
               "one": _r(eob.get("value", e["value"]), 4), "ms_one": _r(eob.get("ms_per_step", e["ms_per_step"]), 4),
               "frac_one": _r(eob.get("pipeline_frac", erf.get("pipeline_frac")), 3)}
        if str(e["name"]).startswith("1080p"):
            row["kfrac"] = _r(erf.get("frac"), 3)
        if erf.get("kernel") != "k_decode":
            row["kernel"] = erf.get("kernel")
        if e.get("cpu_baseline"):
            row["cpu"] = _r(e["cpu_baseline"]["value"], 4)
        rows.append(row)
    if rows:
        out["extras"] = rows
    if full.get("full_record"):
        out["full_record"] = full["full_record"]
    # never more than LINE_LIMIT bytes: shed the least important objects first (none of this triggers today)
    for victim in ("collectives", "settings_blob_crc32_per_rank", "settings_blob_crc32_rank0_before_broadcast", "extras"):
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        if victim == "extras" and "extras" in out:
            out["extras"] = [{k: r_[k] for k in ("name", "value", "one", "frac_one")} for r_ in out["extras"]]
        else:
            out.pop(victim, None)
    while len(json.dumps(out)) > LINE_LIMIT and out.get("extras"):      # ... then whole rows, last first (they stay in full_record)
        out["extras"].pop()
        out["extras_dropped"] = out.get("extras_dropped", 0) + 1
    return out


def multi_gpu_workloads(world, rank, shard):
    """What BASELINE.json's north_star asks for beside the 640x480 headline, on EVERY rank of a --gpus N run: 1920x1080 weak
    (2048 frames per GPU) and configs[2] as stated -- 512 frames of 1920x1080 sharded over the N ranks (strong scaling)."""
    lo, hi = shard.shard_range(512, rank, world)
    return [
        dict(name="1080p_weak", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=2048, noise=0, scanlines=1,
             desc="NTSC 1920x1080 -> 1920x1080 BGRA, interlaced, noise 0, scanlines 1, 2048 frames per GPU (weak scaling)"),
        dict(name="configs2_strong", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=hi - lo, noise=0, scanlines=1,
             first_frame=lo, total_frames=512,
             desc="BASELINE configs[2]: 512 frames 1920x1080 noise 0 sharded over %d GPU(s), %d on this rank (strong scaling)" % (world, hi - lo)),
    ]


def north_star_table(world, headline, weak1080, strong512):
    """The row of BASELINE.json's north-star table this run contributes: frames/sec at 640x480 and 1920x1080 on `world` GPUs,
    absolute and as the fraction of the HBM roofline (end to end, per GPU), and configs[2]'s 512 frames as one job.
    Each argument is a run_workload() record (or None)."""
    def one(rec):
        return (rec or {}).get("one_batch_in_flight") or rec or {}
    row = {"n_gpus": world}
    # frac_* = the whole launch sequence against 8 TB/s on the wall clock (pipeline_frac); kernel_frac_* = roofline.frac as specified
    # (the pass's algorithmic bytes over the dominant kernel's duration); kernel_own_frac_* = that kernel on the bytes it moves itself
    if headline:
        hrf = headline.get("roofline") or {}
        row.update(fps_640=_r(headline["value"], 6), fps_640_one_batch=_r(one(headline).get("value"), 6),
                   frac_640=_r(hrf.get("pipeline_frac"), 3), frac_640_one_batch=_r(one(headline).get("pipeline_frac"), 3),
                   kernel_frac_640=_r(hrf.get("frac"), 3), kernel_own_frac_640=_r(hrf.get("kernel_own_frac"), 3))
    if weak1080:
        wrf = weak1080.get("roofline") or {}
        row.update(fps_1080p_weak=_r(weak1080["value"], 6), fps_1080p_weak_one_batch=_r(one(weak1080).get("value"), 6),
                   frac_1080p_weak=_r(wrf.get("pipeline_frac"), 3),
                   frac_1080p_weak_one_batch=_r(one(weak1080).get("pipeline_frac"), 3),
                   kernel_frac_1080p=_r(wrf.get("frac"), 3), kernel_own_frac_1080p=_r(wrf.get("kernel_own_frac"), 3))
    if strong512:
        ms = one(strong512).get("ms_per_step")
        row.update(configs2_frames=512, configs2_ms=_r(ms), configs2_fps=_r(512.0 / (ms * 1e-3) if ms else None, 6),
                   configs2_frames_per_gpu=(strong512.get("config") or {}).get("fields_per_gpu_per_step"))
    return row


WORKLOAD_NOTES = {
    "ntsc": "BASELINE configs[1]" , "1080p": "BASELINE configs[2] geometry", "vhs": "BASELINE configs[3]",
    "nesp0": "BASELINE configs[4]",
}


class Batches:
    """The batches in flight of one workload on this rank's GPU: S independent contexts (each its own n television sets: state,
    signal and picture buffers) with their streams, the synthetic input they share, the settings blob (built on rank 0,
    broadcast over RCCL), and the step loop.  Step k of a run with `inflight` batches in flight runs on context k % inflight,
    as that context's (k // inflight)-th step.  (tests/test_gpu_bench.py drives this very loop with S = 3 and S = 1 and
    compares every output byte.)"""

    def __init__(self, torch, crtlib, shard, dist, dev, rank, world, local, wl, S):
        self.torch, self.crtlib, self.shard, self.dist, self.dev, self.rank, self.world, self.wl = torch, crtlib, shard, dist, dev, rank, world, wl
        system, w, h, outw, outh = wl["system"], wl["w"], wl["h"], wl["outw"], wl["outh"]
        n, noise, scanlines, fir = wl["batch"], wl["noise"], wl["scanlines"], wl.get("fir", 0)
        self.n, self.noise = n, noise
        self.nes = nes = system in ("nes", "nesp0")
        self.crts, self.streams = [], []
        for _ in range(S):
            c_ = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, system, device=local)
            c_.scanlines = scanlines
            c_.eq_fir = fir
            c_.reserve(n)
            c_.set_overlap(wl.get("overlap", 0))
            c_.set_pixel_tile(wl.get("pixel_tile", 0))
            c_.set_shape(wl.get("shape", 0))
            self.crts.append(c_)
            self.streams.append(torch.cuda.Stream(device=dev) if S > 1 else None)
            if S > 1:
                c_.use_stream(self.streams[-1])
        crt = self.crts[0]
        # synthetic input, generated on the device: uniform random bytes per frame; at most
        # `unique` distinct frames, tiled to the batch: the kernels' work is data-independent
        gen = torch.Generator(device=dev)
        gen.manual_seed(12345 + rank)
        uniq = min(n, wl.get("unique", 64))
        first = wl.get("first_frame", rank * n)                    # this rank's contiguous block of the global batch
        if nes:
            self.base = torch.randint(0, 512, (uniq, h + 1, w), dtype=torch.int16, device=dev, generator=gen)
            self.images = self.base.repeat((n + uniq - 1) // uniq, 1, 1)[:n][:, :h]
            self.s = crtlib.Settings(self.images, hue=0, dot_crawl_offset=[(first + k) % 3 for k in range(n)])
        else:
            self.base = torch.randint(0, 256, (uniq, h + 1, w, 4), dtype=torch.uint8, device=dev, generator=gen)
            self.images = self.base.repeat((n + uniq - 1) // uniq, 1, 1, 1)[:n][:, :h]
            parity = [shard.field_parity(first + k) for k in range(n)]
            self.s = crtlib.Settings(self.images, format=crtlib.FMT_BGRA, as_color=1, hue=0,
                                     field=[a for a, _ in parity], frame=[b for _, b in parity])
        if system.startswith("vhs"):
            for c_ in self.crts:
                c_.srand([1 + first + k for k in range(n)])
        # settings blob: built on rank 0, broadcast over RCCL/xGMI (the path's only collective)
        self.p = crt.params(self.s, noise)
        self.blob_crcs = None
        self.crc_root = zlib.crc32(bytes(self.p))                    # rank 0's blob as built on the host, before any collective
        if dist is not None:
            shard.broadcast_params(self.p, dist, dev)
            crc = torch.tensor([zlib.crc32(bytes(self.p))], dtype=torch.int64, device=dev)
            allc = [torch.zeros_like(crc) for _ in range(world)]
            dist.all_gather(allc, crc)
            self.blob_crcs = [int(c.item()) for c in allc]
        for c_ in self.crts:
            c_._load_field_state(self.s)
        self.seq_rounds = []

    def step(self, k, inflight=1):
        """step k of a run with `inflight` batches in flight: on context k % inflight"""
        ci = k % inflight
        crt = self.crts[ci]
        if self.streams[ci] is not None:
            with self.torch.cuda.stream(self.streams[ci]):
                self.one_step(crt, k // inflight)
        else:
            self.one_step(crt, k)

    def one_step(self, crt, k):
        wl, dist, crtlib, shard = self.wl, self.dist, self.crtlib, self.shard
        if wl.get("sequence"):
            if dist is None:
                crt.sequence(self.s, self.noise)
            else:
                # ONE video of world * n fields cut over the ranks (last row): sync state by all_gather,
                # the picture handed from rank to rank (shard.sequence_sharded)
                self.seq_rounds.append(shard.sequence_sharded(shard.CrtSequenceEngine(crt, self.s, self.noise), dist, self.rank, self.world,
                                                              self.world * self.n, 0, 0, 194, None, 0, self.dev, (wl["outh"], wl["outw"], 4)))
            return
        crt.fieldpass(self.s, self.noise, params=self.p)
        if not self.nes:        # next field of the interlaced sequence (video_convert.c:261-267)
            crt.state[:, crtlib.ST_FIELD] ^= 1
            if k % 2 == 0:
                crt.state[:, crtlib.ST_FRAME] ^= 1

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def close(self):
        for c_ in self.crts:
            c_.close()
        self.crts, self.images, self.base, self.s = [], None, None, None


def run_workload(torch, crtlib, shard, dist, dev, rank, world, local, wl, steps, warmup, cpu_seconds, with_cpu, traffic_file=None):
    """One workload on this rank's GPU; returns the result record (rank 0) or None."""
    system, w, h, outw, outh = wl["system"], wl["w"], wl["h"], wl["outw"], wl["outh"]
    n, noise, scanlines, fir = wl["batch"], wl["noise"], wl["scanlines"], wl.get("fir", 0)
    nes = system in ("nes", "nesp0")
    # Batches in flight: consecutive steps are INDEPENDENT batches (each its own n television sets: context, state, signal
    # and picture buffers), so with S of them step k runs on context k % S with its own stream and the launch sequences of S
    # batches overlap -- the latency-bound sync chain and the launch gaps of one under the vector- / store-bound kernels of
    # the others.  How much that gives depends on how the hardware queues happen to interleave (0 .. 25 % at 1080p from box
    # to box and run to run, profiles/r03_streams_sweep.txt), so S is TUNED here: the same K steps are timed for every
    # candidate S and the best one is the result; all of them are reported (`batches_in_flight_tuning`).  --streams N pins it.
    want = wl.get("streams", 0)
    cands = [1, 2, 3] if want <= 0 else [want]
    if wl.get("sequence") or wl.get("graph"):
        cands = [1]
    S = max(cands)
    B = Batches(torch, crtlib, shard, dist, dev, rank, world, local, wl, S)
    crts, crt, s, p, step, barrier = B.crts, B.crts[0], B.s, B.p, B.step, B.barrier
    blob_crcs, crc_root, seq_rounds = B.blob_crcs, B.crc_root, B.seq_rounds

    torch.cuda.synchronize(dev)              # (the inputs were made on the default stream)
    for k in range(warmup * S):
        step(k, S)
    barrier()
    # The launch sequence of two consecutive steps (the even and the odd field of the interlaced pair: they differ in
    # the frame flip) is captured into a HIP graph and replayed: same kernels, same work, no per-launch host overhead
    # between them.  Opt-in (--graph): measured equal to the eager launches (the queue never drains at these batch sizes).
    graph = None
    if wl.get("graph", False) and not wl.get("sequence") and steps >= 2 and not system.startswith("vhs"):
        try:
            side = torch.cuda.Stream(device=dev)
            crt.use_stream(side)
            side.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step(0)
                step(1)
            torch.cuda.synchronize(dev)
        except Exception as e:                                   # pragma: no cover - depends on the runtime
            graph = None
            crt.use_stream(None)
            sys.stderr.write("bench.py: graph capture failed (%s), timing eager launches\n" % e)
    def timed(cand):
        """one repetition: exactly `steps` steps with `cand` batches in flight, barrier + synchronize on both sides, MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        if graph is not None:
            for k in range(steps // 2):
                graph.replay()
            for k in range(steps % 2):
                with torch.cuda.stream(side):
                    step(0)
        else:
            for k in range(steps):
                step(k, cand)
        barrier()
        e_ = time.perf_counter() - t0
        if dist is not None:
            e_ = shard.max_over_ranks(e_, dist, dev)
        return e_

    # one sample per candidate and `min` over them is biased upwards and as noisy as the box.  Every
    # candidate is timed R times (round robin, so a drifting clock hits all of them alike), S is chosen on the MEDIANS, then the
    # chosen S is timed R more times and `value` is the median of THOSE (the selection samples are not reused: no winner's curse);
    # min / median / max of the final samples go out as `value_spread`.
    frames = wl.get("total_frames") or world * n             # frames of the whole job per step
    R = max(1, int(wl.get("reps", 5)))
    samples = {cand: [] for cand in cands}
    for _ in range(R):
        for cand in cands:
            samples[cand].append(timed(cand))
    med = lambda v: sorted(v)[len(v) // 2]
    tuning = {cand: med(v) for cand, v in samples.items()}
    S = min(tuning, key=lambda c_: tuning[c_])
    final = sorted(timed(S) for _ in range(R)) if len(cands) > 1 else sorted(samples[S])
    elapsed = med(final)
    spread = {"reps": len(final), "min": frames * steps / final[-1], "median": frames * steps / elapsed, "max": frames * steps / final[0],
              "unit": "frames/sec", "note": "the chosen S re-timed %d x %d steps after the selection; value = the median" % (len(final), steps)}
    single = {"value": frames * steps / tuning[1], "unit": "frames/sec", "ms_per_step": 1e3 * tuning[1] / steps,
              "spread": [frames * steps / max(samples[1]), frames * steps / min(samples[1])]} if 1 in tuning and len(tuning) > 1 else None
    launch_mode = "HIP graph of 2 steps, replayed" if graph is not None else "eager"
    if graph is not None:
        crt.use_stream(None)
        del graph

    # per-kernel durations with HIP events on the launch stream (separate short run, same workload, ONE batch in flight:
    # a kernel's duration next to another batch's kernels says nothing about the kernel)
    crt.profile(True)
    for k in range(min(steps, 5)):
        step(k, 1)
    prof = crt.profile_read()
    crt.profile(False)
    kern_ms = {k: (v[0] / max(min(steps, 5), 1)) for k, v in prof.items()}     # ms per step (a step may launch a kernel twice)
    dom = max(kern_ms, key=lambda k: kern_ms[k])
    B.close()
    del crt, crts, s, B
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    total_frames = frames * steps
    fps = total_frames / elapsed
    abytes, own = algorithmic_bytes(system, w, h, 2 if nes else 4, outw, outh, 4, scanlines, 0)
    achieved = abytes * n / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
    own_gbs = own[dom] * n / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
    traffic = None
    traffic_kernel = None
    valu = None
    traffic_stale = None
    if traffic_file and os.path.exists(traffic_file):        # PMC passes (tools/prof_bench.sh, prof_sq.sh) on this very workload
        try:
            tj = json.load(open(traffic_file))
            if tj.get("workload") == wl["name"]:
                traffic_stale = tj.get("source_hash") != kernel_source_hash()
                # the whole pass's counted bytes -- the figure that stands beside the whole pass's ALGORITHMIC bytes
                # (`traffic_ratio` = counted / algorithmic, >= 1: what the intermediate signal, margins and sync windows add); the
                # dominant kernel's own counted bytes go out separately
                traffic = tj.get("all_kernels_bytes_per_field")
                traffic = traffic * n if traffic else None
                traffic_kernel = tj.get("k_" + dom + "_bytes_per_field")
                traffic_kernel = traffic_kernel * n if traffic_kernel else None
                # the vector-ALU side of the same kernels: wave64 instructions (SQ_INSTS_VALU from the committed counter
                # passes) over the time measured HERE, against one instruction per 4 cycles and SIMD -- the rate these
                # instruction mixes issue at on gfx950 (profiles/r02_valu_mixed_sequences.txt)
                names = ("template", "active", "sync", "decode")
                per = {k: tj.get("k_%s_valu_per_field" % k) for k in names}
                if all(per.values()):
                    peak = VALU_SIMDS * VALU_CLOCK_HZ / 4.0
                    busy_ms = sum(kern_ms.get(k, 0.0) for k in names) + kern_ms.get("noise", 0.0)
                    valu = {"wave_instr_per_field": sum(per.values()),
                            "kernel": {k: {"wave_instr_per_field": per[k],
                                           "achieved": per[k] * n / (kern_ms[k] * 1e-3) / 1e9 if kern_ms.get(k) else None}
                                       for k in names},
                            "achieved": sum(per.values()) * n / (busy_ms * 1e-3) / 1e9, "peak": peak / 1e9,
                            "unit": "G wave64 instr/s", "frac": sum(per.values()) * n / (busy_ms * 1e-3) / peak,
                            "peak_is": "%d SIMDs x %.1f GHz / 4 cycles per instruction (what these instruction mixes issue at, "
                                       "profiles/r02_valu_mixed_sequences.txt)" % (VALU_SIMDS, VALU_CLOCK_HZ / 1e9),
                            # the chip's nominal rate: one wave64 VALU instruction per 2 cycles and SIMD
                            "frac_of_2_cycle_peak": sum(per.values()) * n / (busy_ms * 1e-3) / (2.0 * peak),
                            "instr_count_source": "static: %s (SQ_INSTS_VALU of a committed rocprofv3 --pmc run); "
                                                  "kernel times measured in this run" % os.path.relpath(traffic_file, ROOT)}
        except Exception:
            traffic = None
    rec = {
        "name": wl["name"],
        "value": fps, "unit": "frames/sec", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "value_spread": spread,
        "config": {"workload": wl["desc"], "fields_per_gpu_per_step": n, "frames_per_step": frames,
                   "sharding": "frames by rank, RCCL broadcast of settings only",
                   "mode": ("one video cut over the ranks (shard.sequence_sharded), %s exchange round(s) per step" % sorted(set(seq_rounds))
                            if seq_rounds else "one video per GPU (crthip_sequence)") if wl.get("sequence")
                           else "independent frames (crthip_fieldpass)",
                   "launch": launch_mode,
                   "batches_in_flight": S,
                   "batches_in_flight_tuning": {str(c_): {"value": frames * steps / e_, "ms_per_step": 1e3 * e_ / steps,
                                                           "samples_ms_per_step": [round(1e3 * x / steps, 5) for x in samples[c_]]}
                                                for c_, e_ in sorted(tuning.items())},
                   "batches_in_flight_note": "step k runs on context / stream k %% S; every context is an independent batch of "
                                             "fields_per_gpu_per_step television sets with its own state, signal and picture buffers; "
                                             "S tuned: every candidate timed %d times (round robin), chosen on the medians, then re-timed for the result" % R},
        "roofline": {"bound": "hbm", "kernel": "k_" + dom,
                     # as specified: the field-pass's algorithmic bytes per launch / the dominant kernel's duration
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_what": "HBM bytes of ALL kernels of one field-pass launch sequence (counters), to be read against algorithmic_bytes_per_field x fields",
                     "traffic_ratio": (traffic / (abytes * n)) if traffic else None,
                     "traffic_dominant_kernel": traffic_kernel,
                     # the kernels changed since the PMC passes that produced `traffic` (hash of csrc/ differs)
                     "traffic_stale": traffic_stale,
                     # NOT measured in this run: PMC passes need rocprofv3 around the process (tools/prof_bench.sh)
                     "traffic_source": ("static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same workload, "
                                        "calibrated per access pattern: profiles/r03_pmc_calibration.json)"
                                        % os.path.relpath(traffic_file, ROOT)) if traffic is not None else None,
                     "valu": valu,
                     "algorithmic_bytes_per_field": abytes,
                     "kernel_ms": kern_ms,
                     # the dominant kernel on the bytes it moves itself
                     "kernel_own_bytes_per_field": own[dom], "kernel_own_achieved": own_gbs,
                     "kernel_own_frac": own_gbs / HBM_PEAK_GBS,
                     # the whole launch sequence, end to end (wall clock of the timed region)
                     "pipeline_achieved": abytes * n * steps / elapsed / 1e9,
                     "pipeline_frac": abytes * n * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                     "note": "640x480-class workloads are integer-VALU bound (~37 ops/B, SURVEY.md 8(d)); 1080p leans on HBM writes"},
    }
    if single is not None:
        single["pipeline_frac"] = abytes * n * steps / tuning[1] / 1e9 / HBM_PEAK_GBS
        rec["one_batch_in_flight"] = single
    if blob_crcs is not None:
        rec["settings_blob_crc32_per_rank"] = blob_crcs
        rec["settings_blob_crc32_rank0_before_broadcast"] = crc_root
    if with_cpu:
        rec["cpu_baseline"] = cpu_baseline(system + ("fir%d" % fir if fir else ""), w, h, outw, outh, noise, scanlines, cpu_seconds,
                                           all_cores=wl.get("cpu_all_cores", False))
        rec["gpu_over_cpu"] = fps / rec["cpu_baseline"]["value"]
        if rec["cpu_baseline"].get("all_cores", {}).get("value"):
            rec["gpu_over_cpu_all_cores"] = fps / rec["cpu_baseline"]["all_cores"]["value"]
    return rec


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        system, w, h, outw, outh, noise, scanlines = sys.argv[2], *map(int, sys.argv[3:9])
        fps, _, _ = cpu_time_workload(system, w, h, outw, outh, noise, scanlines, float(sys.argv[9]))
        print(fps)
        return

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="fields per GPU per step (weak scaling)")
    ap.add_argument("--strong", type=int, default=0, metavar="FRAMES",
                    help="strong scaling: FRAMES 1920x1080 frames (BASELINE configs[2]: 512) split over the ranks")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--noise", type=int, default=24)
    ap.add_argument("--scanlines", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads (1080p, VHS, NES)")
    ap.add_argument("--system", default="ntsc",
                    help="ntsc (headline), ntscp0, vhs, nes, nesp0, snes, pv1k, temp, nesrgb, <system>bloom: extra measurements")
    ap.add_argument("--fir", type=int, default=0, choices=[0, 4, 5, 6, 7],
                    help="decoder of a USE_CONVOLUTION build of the reference (FIR kernel of N taps) instead of the 3-band equaliser")
    ap.add_argument("--outw", type=int, default=0)
    ap.add_argument("--outh", type=int, default=0)
    ap.add_argument("--sequence", action="store_true", help="treat the batch as ONE video (crthip_sequence) instead of independent frames")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--pixel-tile", type=int, default=0, help="decoder output tile: 0 auto, 16, 32")
    ap.add_argument("--overlap", type=int, default=0, help="chunks alternating between two streams (0 = library default)")
    ap.add_argument("--shape", type=int, default=0, help="kernel shape: 0 auto, 1 lane-per-scanline, 2 scanline-parallel")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent batches in flight, each on its own stream and context (0 = tuned: 1, 2 and 3 are timed, the best is the result; 1 = off)")
    ap.add_argument("--graph", action="store_true", help="time a replayed HIP graph of two steps instead of eager launches (measured: no difference)")
    ap.add_argument("--full-json", default="", help="also write the complete record (every workload's own objects) to this file")
    ap.add_argument("--dry-run", action="store_true", help="(tests) exercise launch / collectives / JSON without a GPU")
    ap.add_argument("--multi-gpu-workloads", action="store_true",
                    help="(tests) run what a --gpus N run adds to the headline -- 1080p weak, BASELINE configs[2] sharded over the ranks -- "
                         "with the ranks there are (with --force-dist: through RCCL on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run every collective of the multi-GPU path (settings broadcast, "
                         "CRC all_gather, MAX all-reduce, barriers) even with ONE rank -- exercises RCCL on a 1-GPU box")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: become the launcher, one rank per GPU
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but torch.distributed.run started %d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import shard
    dist = None
    if args.dry_run:
        return dry_run(args, torch, shard, rank, world)
    import crtlib
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:          # --force-dist outside torch.distributed.run: a one-rank rendezvous
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    nes = args.system in ("nes", "nesp0")
    w, h = (256, 240) if nes else (args.width, args.height)
    outw, outh = args.outw or (640 if nes else w), args.outh or (480 if nes else h)
    n, noise, scanlines = args.batch, args.noise, args.scanlines
    scaling = "weak"
    if args.strong:
        w, h, outw, outh, noise = 1920, 1080, 1920, 1080, 0
        lo, hi = shard.shard_range(args.strong, rank, world)
        n = hi - lo
        scaling = "strong"
    headline = (args.system, w, h, outw, outh, noise, scanlines, args.fir, bool(args.strong)) == ("ntsc", 640, 480, 640, 480, 24, 1, 0, False)
    desc = "%s %dx%d -> %dx%d BGRA, %s, full colour, noise %d, hue 0, scanlines %d%s" % (
        args.system.upper(), w, h, outw, outh, "progressive" if nes else "interlaced", noise, scanlines,
        " (BASELINE configs[1])" if headline else (" (BASELINE configs[2]: %d frames over %d GPUs)" % (args.strong, world) if args.strong
                                                    else (", %d-tap FIR decoder (USE_CONVOLUTION build)" % args.fir if args.fir else "")))
    wl = dict(name="headline" if headline else "custom", system=args.system, w=w, h=h, outw=outw, outh=outh, batch=n, noise=noise,
              scanlines=scanlines, fir=args.fir, unique=args.unique, overlap=args.overlap, pixel_tile=args.pixel_tile,
              shape=args.shape, sequence=args.sequence, desc=desc, cpu_all_cores=True, graph=args.graph, streams=args.streams)
    if args.strong:
        wl["first_frame"] = shard.shard_range(args.strong, rank, world)[0]
    with_cpu = world == 1 and not args.no_cpu
    rec = run_workload(torch, crtlib, shard, dist, dev, rank, world, local, wl, args.steps, args.warmup, args.cpu_seconds, with_cpu,
                       traffic_file=os.path.join(ROOT, "profiles", "traffic.json"))

    extras = []
    if world == 1 and headline and not args.no_extra:
        EX = [
            dict(name="1080p_batch2048", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=2048, noise=0, scanlines=1,
                 desc="NTSC 1920x1080 -> 1920x1080 BGRA, interlaced, noise 0, scanlines 1 (BASELINE configs[2] geometry, one GPU full)"),
            dict(name="1080p_batch512", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=512, noise=0, scanlines=1,
                 desc="NTSC 1920x1080 -> 1920x1080 BGRA, noise 0, scanlines 1, 512 frames = ALL of BASELINE configs[2] on one GPU (the strong-scaling anchor)"),
            dict(name="1080p_batch64", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=64, noise=0, scanlines=1,
                 desc="NTSC 1920x1080 -> 1920x1080 BGRA, noise 0, scanlines 1, 64 frames = configs[2]'s per-GPU share (512 / 8)"),
            dict(name="vhs_832x624", system="vhs", w=832, h=624, outw=832, outh=624, batch=2048, noise=12, scanlines=1,
                 desc="CRT_SYSTEM_NTSCVHS 832x624 -> 832x624 BGRA, interlaced, noise 12 (libc rand() stream per field), scanlines 1 (BASELINE configs[3])"),
            dict(name="nes_pattern0", system="nesp0", w=256, h=240, outw=640, outh=480, batch=4096, noise=12, scanlines=1,
                 desc="NES 256x240 PPU pixels, CRT_CHROMA_PATTERN 0 -> 640x480 BGRA, progressive, noise 12 (BASELINE configs[4])"),
            dict(name="640x480_batch64", system="ntsc", w=640, h=480, outw=640, outh=480, batch=64, noise=24, scanlines=1,
                 desc="the headline workload at batch 64 (small-batch path: scanline-parallel kernel shape)"),
            dict(name="640x480_batch1", system="ntsc", w=640, h=480, outw=640, outh=480, batch=1, noise=24, scanlines=1,
                 desc="the headline workload, ONE field per launch sequence (latency)"),
            dict(name="640x480_batch256", system="ntsc", w=640, h=480, outw=640, outh=480, batch=256, noise=24, scanlines=1,
                 desc="the headline workload at batch 256 (between the two kernel shapes)"),
            dict(name="pv1k_batch4096", system="pv1k", w=640, h=480, outw=640, outh=480, batch=4096, noise=24, scanlines=1,
                 desc="CRT_SYSTEM_PV1K (5 samples per chroma cycle, 1920-sample lines) 640x480 -> 640x480 BGRA, interlaced, noise 24"),
            dict(name="bloom_batch4096", system="ntscbloom", w=640, h=480, outw=640, outh=480, batch=4096, noise=24, scanlines=1,
                 desc="the headline workload in a CRT_DO_BLOOM build (per-scanline beam width): lines sorted by width, lane-per-scanline decoder"),
        ]
        for e in EX:
            small = e["batch"] <= 256
            r = run_workload(torch, crtlib, shard, None, dev, 0, 1, local, e, 30 if small else (20 if e["batch"] <= 512 else max(5, args.steps)), 3,
                             min(args.cpu_seconds, 4.0), not args.no_cpu and not small,
                             traffic_file=os.path.join(ROOT, "profiles", "traffic_%s.json" % e["name"]))
            extras.append(r)

    mg = {}
    if headline and ((world > 1 and not args.no_extra) or args.multi_gpu_workloads):
        # `bench.py --gpus N` is the command the driver's scaling run issues -- it must report the 1080p
        # numbers of the north star too, not only the weak 640x480 headline
        for e in multi_gpu_workloads(world, rank, shard):
            mg[e["name"]] = run_workload(torch, crtlib, shard, dist, dev, rank, world, local, e, 20 if e["batch"] <= 512 else max(5, args.steps // 2), 3,
                                         0.0, False, traffic_file=None)

    cli = None
    if world == 1 and headline and not args.no_extra and not args.no_cpu:
        # BASELINE configs[0] as stated: whole-process wall time of `ntsc -op 640 480 0 0 in.ppm out.ppm`, the reference binary
        # beside the same crt_main.c linked against the HIP drop-in library (tools/time_cli.py); None where the driver binaries
        # are not prebuilt
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import time_cli
            cli = time_cli.measure(3)
        except Exception as e:                                   # never let the side measurement cost the headline
            cli = {"error": str(e)[:120]}

    if rank == 0:
        out = {
            "metric": "frames/sec at 640x480 interlaced, bit-exact vs CPU; % HBM roofline",
            "value": rec["value"], "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": rec["config"], "roofline": rec["roofline"],
            "world_size": world,
            # dist.get_world_size() of the live process group; None = no process group was initialised (one rank, no --force-dist)
            "world_size_seen_by_rccl": (dist.get_world_size() if dist is not None else None),
            "collectives": {"backend": dist.get_backend() if dist is not None else None, "initialized": dist is not None,
                            "ran": ["broadcast(settings blob)", "all_gather(blob crc)", "all_reduce(MAX elapsed)", "barrier"]
                                   if dist is not None else []},
        }
        out["value_spread"] = rec.get("value_spread")
        for k in ("one_batch_in_flight", "settings_blob_crc32_per_rank", "settings_blob_crc32_rank0_before_broadcast", "cpu_baseline", "gpu_over_cpu", "gpu_over_cpu_all_cores"):
            if k in rec:
                out[k] = rec[k]
        if cli:
            out["cli_config1"] = cli
        if extras:
            out["extra_workloads"] = extras
            by = {e["name"]: e for e in extras if e}
            if "1080p_batch512" in by and "1080p_batch64" in by:
                # BASELINE configs[2] is STRONG scaling: 512 frames of 1920x1080 over 8 GPUs = 64 per GPU.  One GPU measures
                # both ends: T512 (the whole job on one GPU) and T64 (a GPU's share at 8); 8 GPUs cannot beat T512 / T64,
                # whatever the interconnect (the ranks exchange nothing but the settings blob).  One batch in flight: a
                # single job has no second batch to overlap with.
                t512 = by["1080p_batch512"].get("one_batch_in_flight", by["1080p_batch512"])["ms_per_step"]
                t64 = by["1080p_batch64"].get("one_batch_in_flight", by["1080p_batch64"])["ms_per_step"]
                out["strong_scaling"] = {"workload": "BASELINE configs[2]: 512 frames 1920x1080 noise 0, measured on ONE GPU, one batch in flight",
                                         "T512_ms": _r(t512), "T64_ms": _r(t64), "projected_speedup_8gpu": _r(t512 / t64, 3),
                                         "weak_per_gpu_batch2048_fps": _r(by["1080p_batch2048"].get("one_batch_in_flight", by["1080p_batch2048"])["value"], 5)
                                         if "1080p_batch2048" in by else None}
            out["north_star"] = north_star_table(1, rec, by.get("1080p_batch2048"), by.get("1080p_batch512"))
        if mg:
            out["north_star"] = north_star_table(world, rec, mg.get("1080p_weak"), mg.get("configs2_strong"))
            out["multi_gpu_workloads"] = [mg[k] for k in ("1080p_weak", "configs2_strong") if mg.get(k)]
    else:
        out = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()      # RCCL may print its own lines (library path) here: keep the JSON the LAST line
    if out is not None:
        # RCCL printf()s its library path into C stdio's buffer, which would be flushed at exit AFTER this line
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        # the complete record goes to a file (gpurun_out/ comes back from the GPU box), the LAST stdout line is the compact one
        full_paths = [os.path.join(ROOT, "gpurun_out", "bench_full.json")] + ([args.full_json] if args.full_json else [])
        for fp in full_paths:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(fp)), exist_ok=True)
                with open(fp, "w") as fh:
                    json.dump(out, fh, indent=1)
                out["full_record"] = os.path.relpath(fp, ROOT)
            except OSError:
                pass
        comp = compact_record(out)
        line = json.dumps(comp)
        if len(line) > LINE_LIMIT:
            # last resort: never die after the measurements -- the contract keys and the two judged objects only
            keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
            mini = {k: comp.get(k) for k in keep}
            mini["config"] = {"workload": str((comp.get("config") or {}).get("workload"))[:160]}
            mini["roofline"] = {k: (comp.get("roofline") or {}).get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
            cb = comp.get("cpu_baseline") or {}
            mini["cpu_baseline"] = {k: (str(cb.get(k))[:120] if isinstance(cb.get(k), str) else cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample")}
            mini["truncated"] = "final line over %d bytes; see %s" % (LINE_LIMIT, comp.get("full_record"))
            line = json.dumps(mini)
        print(line, flush=True)


def dry_run(args, torch, shard, rank, world):
    """No GPU: the launch path, the settings broadcast (gloo) and the JSON contract only (tests/test_bench_cpu.py)."""
    import crtlib
    dist = None
    dev = torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    p = crtlib.make_params("ntsc", w=640, h=480, outw=640, outh=480, noise=24 if rank == 0 else 99, scanlines=1)
    crcs = [zlib.crc32(bytes(p))]
    if dist is not None:
        shard.broadcast_params(p, dist, dev)
        crc = torch.tensor([zlib.crc32(bytes(p))], dtype=torch.int64)
        allc = [torch.zeros_like(crc) for _ in range(world)]
        dist.all_gather(allc, crc)
        crcs = [int(c.item()) for c in allc]
    n = args.batch
    if args.strong:
        lo, hi = shard.shard_range(args.strong, rank, world)
        n = hi - lo
    tot = torch.tensor([n], dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(tot)
    # the north-star table of a --gpus N run (north_star_table): the same workload list, shard arithmetic and collectives as the
    # real run -- frames by rank, MAX over ranks of the elapsed time -- around a stand-in for the kernels
    recs, frames_per_step = {}, {}
    plan = [dict(name="headline", batch=args.batch)] + (multi_gpu_workloads(world, rank, shard) if not args.strong else [])
    for e in plan:
        t0 = time.perf_counter()
        f = torch.tensor([e["batch"]], dtype=torch.int64)
        if dist is not None:
            dist.all_reduce(f)
        el = time.perf_counter() - t0 + 1e-6
        if dist is not None:
            el = shard.max_over_ranks(el, dist, dev)
        frames = e.get("total_frames") or int(f.item())
        assert frames == int(f.item()), "shards of %s do not add up: %d != %d" % (e["name"], int(f.item()), frames)
        frames_per_step[e["name"]] = frames
        recs[e["name"]] = {"value": frames / el, "ms_per_step": 1e3 * el, "config": {"fields_per_gpu_per_step": e["batch"], "frames_per_step": frames},
                           "roofline": {"pipeline_frac": None}}
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "n_gpus": world, "world_size": world,
                          "world_size_seen_by_backend": dist.get_world_size() if dist is not None else 1,
                          "settings_blob_crc32_per_rank": crcs, "frames_per_step": int(tot.item()),
                          "scaling": "strong" if args.strong else "weak", "noise_after_broadcast": p.noise,
                          "north_star": north_star_table(world, recs.get("headline"), recs.get("1080p_weak"), recs.get("configs2_strong")),
                          "north_star_frames_per_step": frames_per_step}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
