#!/bin/bash
# round 5, session 5: signal tile 32 against 16 over batch sizes and systems
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s5
O=gpurun_out/r5s5
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --steps 10 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 32; do
one "1080x1024 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 1024
one "1080x512 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 512
one "640x1024 sig$t" CRTHIP_SIG_TILE=$t --batch 1024
one "640x512 sig$t" CRTHIP_SIG_TILE=$t --batch 512
one "vhs sig$t" CRTHIP_SIG_TILE=$t --system vhs --width 832 --height 624 --noise 12 --batch 2048
one "pv1k sig$t" CRTHIP_SIG_TILE=$t --system pv1k --batch 4096
one "bloom sig$t" CRTHIP_SIG_TILE=$t --system ntscbloom --batch 4096
one "1280x720x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1280 --height 720 --noise 24 --batch 2048
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
