#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for b in 4096 8192; do
  for ov in 1 2 4; do
    timeout 200 python bench.py --batch $b --overlap $ov --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/ov_b${b}_$ov.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ov_b*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]))
PY
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "dropin or rectangle or refused" 2>&1 | tail -2
