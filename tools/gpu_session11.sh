#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
mkdir -p gpurun_out
for rep in 1 2 3; do
  for t in 16 32; do
    CRTHIP_AC_TILE=$t timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/ac${t}_r$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ac*_r*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
PY
CRTHIP_AC_TILE=32 ./tools/prof_bench.sh ac32 --batch 4096 2>&1 | grep -E "SIZE +void k_active"
