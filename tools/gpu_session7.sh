#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra --no-graph > gpurun_out/bench_nograph.json 2>/dev/null
for b in 1 64 128; do
  for ts in 32 16; do
    CRTHIP_ROW_TILE=$ts timeout 120 python bench.py --batch $b --no-cpu --no-extra --steps 30 > gpurun_out/ts_b${b}_t${ts}.json 2>/dev/null
  done
done
timeout 120 python bench.py --batch 4096 --shape 2 --no-cpu --no-extra --steps 6 > gpurun_out/rows_b4096.json 2>/dev/null
./tools/ubench_hbm.bin > gpurun_out/ubench_hbm.txt 2>&1; cat gpurun_out/ubench_hbm.txt
python - <<'PY'
import json, glob
for f in ["gpurun_out/bench_default.json", "gpurun_out/bench_nograph.json", "gpurun_out/rows_b4096.json"] + sorted(glob.glob("gpurun_out/ts_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), j["config"].get("launch"), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
        for e in j.get("extra_workloads", []):
            print("   ", e["name"], "fps=%.0f ms=%.4f" % (e["value"], e["ms_per_step"]), {k: round(v, 4) for k, v in e["roofline"]["kernel_ms"].items()}, "pipe=%.3f own=%.3f" % (e["roofline"]["pipeline_frac"], e["roofline"]["kernel_own_frac"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
