#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python tools/time_dropin.py > gpurun_out/time_dropin.txt 2>&1; cat gpurun_out/time_dropin.txt
for b in 1 8 64 128 256; do
  timeout 120 python bench.py --batch $b --no-cpu --no-extra --steps 30 > gpurun_out/small_b${b}.json 2>/dev/null
done
for pt in 32 64; do
  timeout 200 python bench.py --width 1920 --height 1080 --noise 0 --batch 2048 --no-cpu --no-extra --pixel-tile $pt --steps 10 > gpurun_out/pt1080_$pt.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/small_*.json") + glob.glob("gpurun_out/pt1080_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
    except Exception as e:
        print(f, "FAILED", e)
PY
bash tools/prof_bench.sh 1080p --width 1920 --height 1080 --noise 0 --batch 2048
bash tools/prof_bench.sh b1 --batch 1 --steps 30
