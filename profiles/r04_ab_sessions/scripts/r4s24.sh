#!/bin/bash
# round 4 session 24: small wide batches: scanline-parallel decoder vs wide-run decoder
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s24; mkdir -p $O
export TMPDIR=/tmp
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 50 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for b in 8 16 32 64 128; do
run "1080p x $b scanline-parallel decoder" X=1 -- $W --batch $b
run "1080p x $b wide-run decoder" CRTHIP_WIDE_MIN_FIELDS=1 -- $W --batch $b
done
run "1080p x 64 scanline-parallel decoder" X=1 -- $W --batch 64
run "1080p x 64 wide-run decoder" CRTHIP_WIDE_MIN_FIELDS=1 -- $W --batch 64
cat $O/ab.txt
