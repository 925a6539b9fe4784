#!/bin/bash
# round 5, session 11: the aligned drain for wide inputs over batch sizes (sig 16 / 40 / 64)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s11
O=gpurun_out/r5s11
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 40; do
for b in 320 512 768 1024 1536; do
one "1080x$b sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch $b
done
one "720px2048 sig$t" CRTHIP_SIG_TILE=$t --width 1280 --height 720 --batch 2048
one "720px512 sig$t" CRTHIP_SIG_TILE=$t --width 1280 --height 720 --batch 512
one "1080n24x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 24 --batch 2048
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
