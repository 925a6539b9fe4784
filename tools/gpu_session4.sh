#!/bin/bash
# gpurun session: parity tests, default bench, kernel traces (batch 1 / 64), 1080p overlap timeline
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
for b in 1 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_b$b -o t -- python bench.py --batch $b --no-cpu --no-extra --steps 30 > /dev/null 2>&1
  f=$(find gpurun_out/trace_b$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/trace_b${b}_kernel_stats.csv
  rm -rf gpurun_out/trace_b$b
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_ov -o t -- python bench.py --width 1920 --height 1080 --noise 0 --batch 2048 --no-cpu --no-extra --overlap 4 --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/trace_ov -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/overlap_timeline_1080p.txt", "w")
for r in rows[-60:]:
    line = "%-28s q%-3s start %9.1f us  dur %8.1f us" % (r["Kernel_Name"][5:30], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out.write(line + "\n")
print(open("gpurun_out/overlap_timeline_1080p.txt").read()[-3500:])
PY
rm -rf gpurun_out/trace_ov
python - <<'PY'
import json, glob, csv
for f in ["gpurun_out/bench_default.json"]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
        for e in j.get("extra_workloads", []):
            print("   ", e["name"], "fps=%.0f ms=%.4f" % (e["value"], e["ms_per_step"]), {k: round(v, 4) for k, v in e["roofline"]["kernel_ms"].items()}, "pipe=%.3f own=%.3f" % (e["roofline"]["pipeline_frac"], e["roofline"]["kernel_own_frac"]))
        print("   cpu:", j.get("cpu_baseline"))
    except Exception as e:
        print(f, "FAILED", e)
for f in sorted(glob.glob("gpurun_out/trace_*_kernel_stats.csv")):
    print(f)
    for row in list(csv.DictReader(open(f)))[:14]:
        if row["Name"].startswith("void k_"):
            print("   %-70s calls %5s avg %10.1f us" % (row["Name"][5:75], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
