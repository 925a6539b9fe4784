#!/bin/bash
# round 5, session 18 (the tree as shipped after the VHS changes): whole GPU suite, VHS soak, default bench, VHS profile
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s18
O=gpurun_out/r5s18
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --full-json $O/bench_full.json > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out > $O/bench_default_line.json
bash tools/prof_bench.sh r05vhs --system vhs --width 832 --height 624 --noise 12 --batch 2048 > $O/prof_vhs.log 2>&1
bash tools/prof_sq.sh r05vhs --no-extra --system vhs --width 832 --height 624 --noise 12 --batch 2048 > gpurun_out/sq_r05vhs.txt 2>&1
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r5s18/bench_default_line.json').read())
print(j['value'], j['ms_per_step'], j['one_batch_in_flight'], j['roofline']['kernel_ms'], j.get('value_spread'))
for e in j['extras']: print(e)
print(j.get('strong_scaling')); print(j.get('north_star')); print(j.get('cli_config1')); print(len(json.dumps(j)))
PY
