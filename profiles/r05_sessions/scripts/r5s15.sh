#!/bin/bash
# round 5, session 15 (the tree as shipped): whole GPU suite, random soaks (ordinary, wide, large signal tiles forced), default bench, CLI times
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s15
O=gpurun_out/r5s15
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python tools/soak_random.py 91000 400 > $O/soak.log 2>&1; tail -1 $O/soak.log
timeout 600 python tools/soak_random.py 92000 200 wide > $O/soak_wide.log 2>&1; tail -1 $O/soak_wide.log
CRTHIP_SIG_TILE=64 CRTHIP_WIDE_LPW=8 timeout 600 python tools/soak_random.py 93000 200 wide > $O/soak_wide_forced.log 2>&1; tail -1 $O/soak_wide_forced.log
CRTHIP_SIG_TILE=32 timeout 600 python tools/soak_random.py 94000 300 > $O/soak_sig32.log 2>&1; tail -1 $O/soak_sig32.log
timeout 600 python bench.py --full-json $O/bench_full.json > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out > $O/bench_default_line.json
python tools/time_cli.py 7 > $O/time_cli.txt 2>&1
cat $O/time_cli.txt
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r5s15/bench_default_line.json').read())
print(j['value'], j['ms_per_step'], j['one_batch_in_flight'], j['roofline']['kernel_ms'], j.get('value_spread'))
for e in j['extras']: print(e)
print(j.get('strong_scaling')); print(j.get('north_star')); print(j.get('cli_config1')); print(len(json.dumps(j)))
PY
