#!/bin/bash
# round 5, session 29: the default bench and the profiles once more (another box of the pool), the tree as shipped
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s29
O=gpurun_out/r5s29
export TMPDIR=/tmp
timeout 600 python bench.py --full-json $O/bench_full.json > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out > $O/bench_default_line.json
bash tools/refresh_profiles.sh r05 > $O/refresh.log 2>&1
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r5s29/bench_default_line.json').read())
print(j['value'], j['ms_per_step'], j['one_batch_in_flight'], j['roofline']['kernel_ms'], j.get('value_spread'), j['roofline'].get('traffic_stale'))
for e in j['extras'][:5]: print(e)
print(j.get('strong_scaling'))
PY
