#!/bin/bash
# round 5, session 7: after the removal of the speculative sync chain, the ccf preset inside k_hsync_wave and the decoder's tier groups:
# the whole GPU suite, then the step times that were to move (64 fields, headline, 1080p)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s7
O=gpurun_out/r5s7
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()}, j['value_spread']['min'], j['value_spread']['max'])"
}
{
for r in 1 2; do
one "640x4096" A=1 --steps 10
one "640x64" A=1 --batch 64 --steps 30
one "640x256" A=1 --batch 256 --steps 30
one "640x1" A=1 --batch 1 --steps 30
one "1080x64" A=1 --width 1920 --height 1080 --noise 0 --batch 64 --steps 30
one "1080x512" A=1 --width 1920 --height 1080 --noise 0 --batch 512 --steps 20
one "1080x2048" A=1 --width 1920 --height 1080 --noise 0 --batch 2048 --steps 10
one "nes" A=1 --system nesp0 --noise 12 --steps 10
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
