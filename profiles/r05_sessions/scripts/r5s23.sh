#!/bin/bash
# round 5, session 23: fields per workgroup of the sync kernel (CRTHIP_SYNC_KERNEL=2: one, =3: four) over batch sizes
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s23
O=gpurun_out/r5s23
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 2 3; do
one "1080x2048 fpb$t" CRTHIP_SYNC_KERNEL=$t --width 1920 --height 1080 --noise 0 --batch 2048
one "640x4096 fpb$t" CRTHIP_SYNC_KERNEL=$t
one "640x1024 fpb$t" CRTHIP_SYNC_KERNEL=$t --batch 1024
one "640x512 fpb$t" CRTHIP_SYNC_KERNEL=$t --batch 512
one "640x256 fpb$t" CRTHIP_SYNC_KERNEL=$t --batch 256 --shape 1
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
