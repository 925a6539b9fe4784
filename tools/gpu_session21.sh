#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for rep in 1 2 3; do
  for sm in 0 1; do
    CRTHIP_SIDE_MARGIN=$sm timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/sm${sm}_r$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sm?_r?.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]))
PY
