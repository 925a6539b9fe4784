#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -k "vhs or VHS or video_convert or sequence" > gpurun_out/pytest_vhs.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_vhs.log
for rep in 1 2 3; do
  timeout 200 python bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu --no-extra --system vhs --width 832 --height 624 --noise 12 > gpurun_out/vhs_r$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/vhs_r*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
PY
