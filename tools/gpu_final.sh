#!/bin/bash
# end-of-round validation: GPU tests, smoke, the default bench line, profiles of the headline and 1080p workloads
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"; tail -2 gpurun_out/bench_default.err
./tools/prof_bench.sh headline 2>&1 | tail -3
./tools/prof_sq.sh headline --no-extra > gpurun_out/sq_headline.log 2>&1
./tools/prof_bench.sh 1080p --width 1920 --height 1080 --noise 0 --batch 2048 2>&1 | tail -2
timeout 300 python tools/time_dropin.py > gpurun_out/time_dropin.txt 2>&1; cat gpurun_out/time_dropin.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()}, "pipe=%.3f" % j["roofline"]["pipeline_frac"], "traffic", j["roofline"]["traffic"], "valu", (j["roofline"].get("valu") or {}).get("frac"))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("all_cores"))
for e in j.get("extra_workloads", []):
    print("   ", e["name"], "fps=%.0f ms=%.4f" % (e["value"], e["ms_per_step"]), {k: round(v, 4) for k, v in e["roofline"]["kernel_ms"].items()}, "pipe=%.3f own=%.3f" % (e["roofline"]["pipeline_frac"], e["roofline"]["kernel_own_frac"]), "cpu", (e.get("cpu_baseline") or {}).get("value"))
PY
