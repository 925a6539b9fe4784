#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python tools/time_dropin.py > gpurun_out/time_dropin.txt 2>&1; cat gpurun_out/time_dropin.txt
for ov in 1 2 4 8; do
  timeout 200 python bench.py --width 1920 --height 1080 --noise 0 --batch 2048 --no-cpu --no-extra --overlap $ov --steps 10 > gpurun_out/ov1080_$ov.json 2>/dev/null
done
for ov in 2 4; do
  timeout 200 python bench.py --no-cpu --no-extra --overlap $ov --steps 10 > gpurun_out/ov480_$ov.json 2>/dev/null
  timeout 200 python bench.py --system vhs --width 832 --height 624 --noise 12 --batch 2048 --no-cpu --no-extra --overlap $ov --steps 10 > gpurun_out/ovvhs_$ov.json 2>/dev/null
done
timeout 200 python bench.py --system vhs --width 832 --height 624 --noise 12 --batch 2048 --no-cpu --no-extra --overlap 1 --steps 10 > gpurun_out/ovvhs_1.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ov*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
    except Exception as e:
        print(f, "FAILED", e)
PY
