#!/bin/bash
# round 5, session 6: large signal tiles against the oracle; whole parity file with the large tile forced; threshold check; default bench
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s6
O=gpurun_out/r5s6
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "large_signal_tiles" > $O/pytest_tiles.log 2>&1
echo "pytest(large tiles) rc=$?"; tail -2 $O/pytest_tiles.log
CRTHIP_SIG_TILE=32 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sig32.log 2>&1
echo "pytest(SIG_TILE=32, whole parity file) rc=$?"; tail -2 $O/pytest_sig32.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --steps 10 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 32; do
one "640x2048 sig$t" CRTHIP_SIG_TILE=$t --batch 2048
one "640x1792 sig$t" CRTHIP_SIG_TILE=$t --batch 1792
one "nes sig$t" CRTHIP_SIG_TILE=$t --system nesp0 --noise 12
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python bench.py --full-json $O/bench_full.json > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out > $O/bench_default_line.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r5s6/bench_default_line.json').read())
print(j['value'], j['ms_per_step'], j['one_batch_in_flight'], j['roofline']['kernel_ms'], j.get('value_spread'))
for e in j['extras']: print(e)
print(j.get('strong_scaling')); print(j.get('north_star')); print(len(json.dumps(j)))
PY
