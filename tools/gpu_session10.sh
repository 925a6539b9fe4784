#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in 16 32; do
  CRTHIP_AC_TILE=$t timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/ac${t}_b4096.json 2>/dev/null
  CRTHIP_AC_TILE=$t timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra --noise 0 > gpurun_out/ac${t}_b4096_n0.json 2>/dev/null
  CRTHIP_AC_TILE=$t timeout 200 python bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu --no-extra --width 1920 --height 1080 --noise 0 > gpurun_out/ac${t}_1080.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ac*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
    except Exception as e:
        print(f, "FAILED", e)
PY
