#!/bin/bash
# round 5, session 22: one batch as two independent halves on two streams (CRTHIP_SPLIT=2)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s22
O=gpurun_out/r5s22
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), j['value_spread']['min'], j['value_spread']['max'])"
}
{
for r in 1 2 3; do
for t in 0 2; do
one "1080x2048 split$t" CRTHIP_SPLIT=$t --width 1920 --height 1080 --noise 0 --batch 2048
one "640x4096 split$t" CRTHIP_SPLIT=$t
one "vhs split$t" CRTHIP_SPLIT=$t --system vhs --width 832 --height 624 --noise 12 --batch 2048
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
