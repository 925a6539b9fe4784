#!/bin/bash
# round 4 session 22: NES / VHS through the wide-run decoder; the sync chain beside the encoder at 1080p once more; default bench on fresh traffic files
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s22; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nes_parity or vhs_wide or graph_capturable" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -4 $O/pytest.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2 3; do
run "1080p2048 sync chain after the encoder" CRTHIP_SPEC_SYNC=0 -- $W --batch 2048
run "1080p2048 sync chain beside the encoder" CRTHIP_SPEC_SYNC=1 -- $W --batch 2048
done
run "1080p512 sync chain after the encoder" CRTHIP_SPEC_SYNC=0 -- $W --batch 512
run "1080p512 sync chain beside the encoder" CRTHIP_SPEC_SYNC=1 -- $W --batch 512
cat $O/ab.txt
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -c 300 $O/bench_default.log; echo
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
