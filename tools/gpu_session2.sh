#!/bin/bash
# gpurun session: parity tests, default bench line, batch sweep for both kernel shapes, legacy-sync A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
CRTHIP_LEGACY_SYNC=1 timeout 200 python bench.py --no-cpu --no-extra --steps 10 > gpurun_out/legacy_sync_480.json 2>/dev/null
for ov in 1 2 4; do
  timeout 200 python bench.py --width 1920 --height 1080 --noise 0 --batch 2048 --no-cpu --no-extra --overlap $ov --steps 10 > gpurun_out/ov1080_$ov.json 2>/dev/null
done
for b in 1 8 64 256 512 1024; do
  for sh in 1 2; do
    timeout 120 python bench.py --batch $b --no-cpu --no-extra --shape $sh --steps 30 > gpurun_out/sweep_b${b}_s${sh}.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in ["gpurun_out/bench_default.json", "gpurun_out/legacy_sync_480.json"] + sorted(glob.glob("gpurun_out/ov*.json") + glob.glob("gpurun_out/sweep_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
        for e in j.get("extra_workloads", []):
            print("   ", e["name"], "fps=%.0f ms=%.4f" % (e["value"], e["ms_per_step"]), {k: round(v, 4) for k, v in e["roofline"]["kernel_ms"].items()}, "pipe=%.3f own=%.3f" % (e["roofline"]["pipeline_frac"], e["roofline"]["kernel_own_frac"]), "cpu=%s" % (e.get("cpu_baseline", {}).get("value")))
    except Exception as e:
        print(f, "FAILED", e)
PY
