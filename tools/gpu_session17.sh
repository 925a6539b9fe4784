#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for rep in 1 2 3; do
  for t in 32 64; do
    timeout 200 python bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu --no-extra --width 1920 --height 1080 --noise 0 --pixel-tile $t > gpurun_out/pt${t}_r$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pt*_r*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
PY
