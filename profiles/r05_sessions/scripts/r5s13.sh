#!/bin/bash
# round 5, session 13: the sync kernel's chain wave rotated over the workgroup's waves; 4 batches in flight
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s13
O=gpurun_out/r5s13
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()}, j['config'].get('batches_in_flight_fps'))"
}
{
for r in 1 2 3; do
for t in 0 1; do
one "640x4096 rotate$t" CRTHIP_CHAIN_ROTATE=$t --streams 1
one "1080x2048 rotate$t" CRTHIP_CHAIN_ROTATE=$t --streams 1 --width 1920 --height 1080 --noise 0 --batch 2048
one "640x1024 rotate$t" CRTHIP_CHAIN_ROTATE=$t --streams 1 --batch 1024
done
done
one "640x4096 S=4" A=1 --streams 4
one "640x4096 tuned" A=1
one "1080x2048 S=4" A=1 --streams 4 --width 1920 --height 1080 --noise 0 --batch 2048
one "1080x2048 tuned" A=1 --width 1920 --height 1080 --noise 0 --batch 2048
} > $O/ab.txt 2>&1
cat $O/ab.txt
