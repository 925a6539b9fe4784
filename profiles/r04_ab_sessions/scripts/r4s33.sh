#!/bin/bash
# round 4 session 33: is the CRTHIP_WIDE_DECODE switch alive?  + the final default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s33; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do
CRTHIP_WIDE_DECODE=$v timeout 120 python bench.py --streams 1 --no-cpu --no-extra --steps 20 --warmup 5 --width 1920 --height 1080 --noise 0 --batch 2048 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('CRTHIP_WIDE_DECODE=$v:', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
cd /tmp && CRTHIP_WIDE_DECODE=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_off -o off -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --no-cpu --no-extra --steps 3 --warmup 1 --width 1920 --height 1080 --noise 0 --batch 512 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
grep -h "k_decode" $O/prof_off/*/*kernel_stats.csv 2>/dev/null | cut -c1-60,200-260 | head -5
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -c 200 $O/bench_default.log; echo
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
