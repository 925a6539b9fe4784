#!/usr/bin/env python3
"""Copy the judged summaries of a tools/prof_bench.sh + tools/prof_sq.sh run (gpurun_out/prof_<tag>,
gpurun_out/sq_<tag>) into profiles/ (tracked).  usage: tools/collect_profiles.py <tag> <name>"""
import collections, csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(os.path.join(src, "pmc_" + c, "bench_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void k_"):
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = v / n
json.dump({"note": "bench default workload (4096 fields 640x480 noise 24 per launch); rocprofv3 --pmc, separate passes for "
                   "FETCH_SIZE and WRITE_SIZE; raw counter values in KB", "kernels": out},
          open(os.path.join(dst, name + "_pmc.json"), "w"), indent=1)
n = 4096
def per_field(prefix):
    tot = 0.0
    for k, d in out.items():
        if k.startswith("void " + prefix) and "false, false" not in k:
            tot += (2 * d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024 / n
    return tot
json.dump({"source": "profiles/%s_pmc.json" % name,
           "method": "HBM bytes per field = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 4096 fields; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE calibrated on k_margin "
                     "(273 858 KB measured for 4096 x 60 712 B algorithmic = 1.10x)",
           "k_decode_bytes_per_field": per_field("k_decode"), "k_active_bytes_per_field": per_field("k_active"),
           "k_template_bytes_per_field": per_field("k_margin"),
           "k_sync_bytes_per_field": per_field("k_hsync") + per_field("k_vsync")},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
sq = os.path.join(root, "gpurun_out", "sq_" + tag)
if os.path.isdir(sq):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(sq + "/p*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void k_"):
                a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    json.dump({"note": "rocprofv3 --pmc SQ counters per launch (4096 fields, 640x480 noise 24), 3 separate passes (tools/prof_sq.sh); "
                       "SQ_*_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
               "kernels": {k: {c: v / n_ for c, (v, n_) in d.items()} for k, d in agg.items()}},
              open(os.path.join(dst, name + "_sq_counters.json"), "w"), indent=1)
print(open(os.path.join(dst, "traffic.json")).read())
