#!/bin/bash
# round 4 session 25: auto-shape test for wide pictures; kernel breakdown of mid-size 640x480 batches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s25; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_pictures_leave or wide_decoder or graph_capturable" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -4 $O/pytest.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 50 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
for b in 64 128 256 512 1024; do
run "640x480 x $b default shape" X=1 -- --batch $b
run "640x480 x $b lane-per-scanline" X=1 -- --batch $b --shape 1
run "640x480 x $b scanline-parallel" X=1 -- --batch $b --shape 2
done
W="--width 1920 --height 1080 --noise 0"
run "1080p x 64 default" X=1 -- $W --batch 64
run "1080p x 128 default" X=1 -- $W --batch 128
run "1080p x 256 default" X=1 -- $W --batch 256
cat $O/ab.txt
