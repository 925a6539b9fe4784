#!/usr/bin/env python3
"""Copy the judged summaries of a tools/prof_bench.sh (+ tools/prof_sq.sh) run from gpurun_out/prof_<tag>
(gpurun_out/sq_<tag>) into profiles/ (tracked), next to the exact command line they came from.

    tools/collect_profiles.py <tag> <name> <fields per launch> [<bench workload name>]

Writes profiles/<name>_kernel_stats.csv, <name>_kernel_stats.args.txt, <name>_pmc.json and, with a workload name,
profiles/traffic.json (headline) or profiles/traffic_<workload>.json (what bench.py reads for roofline.traffic: it
only trusts a file made for the same workload).
HBM bytes = (f x FETCH_SIZE + WRITE_SIZE) x 1024.  rocprofv3's *_SIZE counters are in KB; on gfx950 FETCH_SIZE tallies
every read REQUEST of the L2's memory side as 64 bytes, whether it asked for 64 or for 128 (MI355X_MICROARCH.md, HBM
section; calibrated per access pattern in profiles/r03_pmc_calibration.json with tools/ubench_hbm.bin calib: 16 B / lane
wave-contiguous reads count half, 64-byte row pieces and scattered 16-byte windows count in full).  So f is a property
of the kernel's load pattern -- FETCH_FACTOR below; WRITE_SIZE counts 32-byte granules and needs no factor."""
import collections, csv, glob, json, os, shutil, sys

tag, name, fields = sys.argv[1], sys.argv[2], int(sys.argv[3])
workload = sys.argv[4] if len(sys.argv) > 4 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
bench_cmd = open(os.path.join(src, "args.txt")).read().strip()
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
with open(os.path.join(dst, name + "_kernel_stats.args.txt"), "w") as f:
    f.write(bench_cmd + "\n(fields per launch: %d; rocprofv3 --kernel-trace --stats)\n" % fields)

def _narrow_factor():
    """bytes asked per counted 64 for k_active's narrow image tile, from the calibration of its own pattern (2.0 if absent)"""
    try:
        cal = json.load(open(os.path.join(dst, "r05_pmc_calibration.json")))["patterns"]["void k_rows64<64, 64>"]["factor_vs_1024"]
        return round(float(cal), 3)
    except (OSError, KeyError, ValueError, TypeError):
        return 2.0


NARROW_TILE_FACTOR = _narrow_factor()


def fetch_factor(kernel):
    """bytes per counted 64: what the kernel's dominant read pattern asks the memory side for"""
    k = kernel
    if k.startswith("void k_active"):
        # image rows in 16-byte pieces: 4 lanes = 64 B per row (16-dword image tile) or 8 lanes = 128 B (32-dword tile); the image
        # tile is the template argument before the last (k_active<S, NOISE, FAST, IN4, CLAMP, ACT, OT>).  Both patterns ask the
        # memory side for whole 128-byte lines -- a 64-byte piece is HALF a line whose other half is fetched a tile later and
        # hits in L2 -- so both count a line as one request: calibrated with the encoder's own read pattern
        # (tools/ubench_enc.hip calib; profiles/r05_pmc_calibration.json: k_rows64<64, 64> and <64, 128>).  Round 3's factor 1.0
        # for the narrow tile came from ISOLATED 64-byte pieces and put the encoder's fetch below its own image rows.
        args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
        wide = (args[5] if len(args) >= 7 else args[-1]) == "32"
        return (NARROW_TILE_FACTOR if not wide else 2.0, "128-byte image lines" if wide else "64-byte half lines (the other half a tile later)")
    if k.startswith("void k_decode"):
        return 1.0, "64-byte sample pieces (4 lanes x 16 B per scanline)"
    if k.startswith("void k_hsync") or k.startswith("void k_vsync"):
        return 1.0, "per-lane 16-byte windows of different scanlines (the wave-wide vsync candidates, 1 KB runs, are the smaller part: lower bound)"
    return 2.0, "wave-contiguous 16 B / lane"


out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(os.path.join(src, "pmc_%s.csv" % c))):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void k_"):
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = v / n
for k, d in out.items():
    f, why = fetch_factor(k)
    d["fetch_factor"], d["fetch_pattern"] = f, why
    d["hbm_bytes_per_field"] = (f * d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024 / fields
    d["hbm_bytes_per_field_if_every_request_were_128B"] = (2 * d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024 / fields


def per_field(*prefixes):
    return sum(d["hbm_bytes_per_field"] for k, d in out.items() if any(k.startswith("void " + p) for p in prefixes))


total = sum(d["hbm_bytes_per_field"] for d in out.values())

# SQ counter passes (tools/prof_sq.sh), if they were run for this tag
sq = os.path.join(root, "gpurun_out", "sq_" + tag)
sqagg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
if os.path.isdir(sq):
    for f in glob.glob(sq + "/p*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void k_"):
                a = sqagg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1


def valu_per_field(*prefixes):
    v = sum(d["SQ_INSTS_VALU"][0] / d["SQ_INSTS_VALU"][1] for k, d in sqagg.items()
            if any(k.startswith("void " + p) for p in prefixes) and "SQ_INSTS_VALU" in d)
    return v / fields if v else None
json.dump({"bench": bench_cmd, "fields_per_launch": fields,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (raw counter = KB); HBM bytes per field = "
                     "(f x FETCH_SIZE + WRITE_SIZE) x 1024 / fields with the per-kernel fetch_factor f: gfx950 tallies a read "
                     "request as 64 B whether it asked for 64 or 128 (profiles/r03_pmc_calibration.json)",
           "hbm_bytes_per_field_all_kernels": total, "kernels": out},
          open(os.path.join(dst, name + "_pmc.json"), "w"), indent=1)
if workload:
    sys.path.insert(0, root)
    import bench
    json.dump({"workload": workload, "source": "profiles/%s_pmc.json" % name, "bench": bench_cmd, "fields_per_launch": fields,
               # the device sources these counters were measured on (bench.py: roofline.traffic_stale)
               "source_hash": bench.kernel_source_hash(),
               "k_decode_bytes_per_field": per_field("k_decode"), "k_active_bytes_per_field": per_field("k_active"),
               "k_template_bytes_per_field": per_field("k_margin", "k_skeleton", "k_template"),
               "k_sync_bytes_per_field": per_field("k_hsync", "k_vsync", "k_bloom"),
               "all_kernels_bytes_per_field": total,
               # wave64 vector instructions per field (SQ_INSTS_VALU), for bench.py's roofline.valu
               "k_decode_valu_per_field": valu_per_field("k_decode"), "k_active_valu_per_field": valu_per_field("k_active"),
               "k_template_valu_per_field": valu_per_field("k_margin", "k_skeleton", "k_template"),
               "k_sync_valu_per_field": valu_per_field("k_hsync", "k_vsync", "k_bloom")},
              open(os.path.join(dst, "traffic.json" if workload == "headline" else "traffic_%s.json" % workload), "w"), indent=1)

if sqagg:
    json.dump({"bench": bench_cmd, "fields_per_launch": fields,
               "note": "rocprofv3 --pmc SQ counters per launch, 3 separate passes (tools/prof_sq.sh); SQ_*_CYCLES / "
                       "SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
               "kernels": {k: {c: v / n_ for c, (v, n_) in d.items()} for k, d in sqagg.items()}},
              open(os.path.join(dst, name + "_sq_counters.json"), "w"), indent=1)
print("HBM bytes per field, all kernels: %.0f" % total)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_field"])[:6]:
    print("  %-70s %12.0f" % (k[:70], d["hbm_bytes_per_field"]))
