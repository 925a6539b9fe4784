#!/bin/bash
# gpurun session: parity tests, small-batch sweeps with both sync kernels, kernel traces of batch 1 / 64 / 4096
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for b in 1 64 256 512 1024 4096; do
  for sk in 1 2; do
    CRTHIP_SYNC_KERNEL=$sk timeout 120 python bench.py --batch $b --no-cpu --no-extra --shape 0 --steps 30 > gpurun_out/sync_b${b}_k${sk}.json 2>/dev/null
  done
done
for b in 64 256 512 1024; do
  for sh in 1 2; do
    timeout 120 python bench.py --batch $b --no-cpu --no-extra --shape $sh --steps 30 > gpurun_out/sweep_b${b}_s${sh}.json 2>/dev/null
  done
done
for b in 1 64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$b -o b$b -- python $OLDPWD/bench.py --batch $b --no-cpu --no-extra --steps 30 > /dev/null 2>&1)
  f=$(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/trace_b${b}_kernel_stats.csv
done
python - <<'PY'
import json, glob, csv
for f in sorted(glob.glob("gpurun_out/sync_*.json") + glob.glob("gpurun_out/sweep_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e)
for f in sorted(glob.glob("gpurun_out/trace_*_kernel_stats.csv")):
    print(f)
    for row in list(csv.DictReader(open(f)))[:12]:
        print("   %-60s calls %5s avg %10.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
