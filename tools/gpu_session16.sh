#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for b in 1 64 256 4096; do
  timeout 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu --no-extra > gpurun_out/vs_b$b.json 2>/dev/null
  CRTHIP_SYNC_KERNEL=1 timeout 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu --no-extra > gpurun_out/vs_legacy_b$b.json 2>/dev/null
done
timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/vs_b4096_2.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/vs_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
PY
