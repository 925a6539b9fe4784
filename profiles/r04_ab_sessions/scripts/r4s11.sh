#!/bin/bash
# round 4 session 11: the wide-run decoder (crt_decode4.hip): parity, then A/B at 1080p
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s11; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_decoder or stagewise_parity or fused_parity or full_size_batch_properties" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -12 $O/pytest.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2 3; do
run "1080p2048 lane-per-scanline decoder" CRTHIP_WIDE_DECODE=0 -- $W --batch 2048
run "1080p2048 wide-run decoder" CRTHIP_WIDE_DECODE=1 -- $W --batch 2048
done
run "1080p512 lane-per-scanline decoder" CRTHIP_WIDE_DECODE=0 -- $W --batch 512
run "1080p512 wide-run decoder" CRTHIP_WIDE_DECODE=1 -- $W --batch 512
run "1080p256 lane-per-scanline decoder" CRTHIP_WIDE_DECODE=0 -- $W --batch 256 --shape 1
run "1080p256 wide-run decoder" CRTHIP_WIDE_DECODE=1 -- $W --batch 256 --shape 1
run "1080p2048 3 in flight, lane-per-scanline" CRTHIP_WIDE_DECODE=0 -- $W --batch 2048 --streams 3
run "1080p2048 3 in flight, wide-run" CRTHIP_WIDE_DECODE=1 -- $W --batch 2048 --streams 3
cat $O/ab.txt
