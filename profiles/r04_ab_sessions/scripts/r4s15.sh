#!/bin/bash
# round 4 session 15: encoder with two image tiles in flight (accumulation-register staging) vs one (lib_prev): parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s15; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -6 $O/pytest.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2; do
run "1080p2048 one tile in flight" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- $W --batch 2048
run "1080p2048 two tiles in flight" X=1 -- $W --batch 2048
run "headline one tile in flight" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev --
run "headline two tiles in flight" X=1 --
done
run "vhs one tile in flight" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- --system vhs --width 832 --height 624 --noise 12 --batch 2048
run "vhs two tiles in flight" X=1 -- --system vhs --width 832 --height 624 --noise 12 --batch 2048
run "nes one tile in flight" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- --system nesp0 --noise 12
run "nes two tiles in flight" X=1 -- --system nesp0 --noise 12
run "1080p2048 3 in flight, two tiles" X=1 -- $W --batch 2048 --streams 3
cat $O/ab.txt
