#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for rep in 1 2 3 4; do
  timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/ot_r$rep.json 2>/dev/null
done
timeout 200 python bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu --no-extra --system vhs --width 832 --height 624 --noise 12 > gpurun_out/ot_vhs.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ot_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
PY
