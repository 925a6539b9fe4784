#!/bin/bash
# round 4 session 13: what bounds k_decode_wide?  decomposition builds (no stores / stores only / filter only) + SQ counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s13; mkdir -p $O
export TMPDIR=/tmp
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2; do
run "1080p2048 wide-run v2" CRTHIP_WIDE_DECODE=1 -- $W --batch 2048
run "1080p2048 v2, no picture stores" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_dbg1 -- $W --batch 2048
run "1080p2048 v2, stores only (no filter, no pixel arithmetic)" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_dbg2 -- $W --batch 2048
run "1080p2048 v2, filter stage only" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_dbg3 -- $W --batch 2048
done
cat $O/ab.txt
bash tools/prof_sq.sh r4s13_1080p --no-extra --width 1920 --height 1080 --noise 0 --batch 2048 > $O/sq_1080p.txt 2>&1
grep -A26 "k_decode_wide" $O/sq_1080p.txt | head -60
