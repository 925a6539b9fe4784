#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for rep in 1 2 3; do
  timeout 200 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/pf_r$rep.json 2>/dev/null
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/bench_extra.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pf_r*.json")) + ["gpurun_out/bench_extra.json"]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"])
        for e in j.get("extra_workloads", []):
            print("   ", e["name"], "fps=%.0f ms=%.4f" % (e["value"], e["ms_per_step"]), {k: round(v, 4) for k, v in e["roofline"]["kernel_ms"].items()}, "pipe=%.3f own=%.3f" % (e["roofline"]["pipeline_frac"], e["roofline"]["kernel_own_frac"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
