#!/bin/bash
# round 5, session 4: signal tile 16 / 32 / 64; k_decode without its picture stores at 640x480; pixel tile 32 at 640x480
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s4
O=gpurun_out/r5s4
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --steps 10 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()}, j.get('value_spread'))"
}
{
for r in 1 2; do
for t in 16 32 64; do
one "640x4096 sig$t" CRTHIP_SIG_TILE=$t
one "1080x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 2048
done
one "640x4096 k_decode no stores" CRTHIP_LIBDIR=ntsc-crt_amd/lib_dbg3
one "640x4096 pixel tile 32" CRTHIP_SIG_TILE=16 --pixel-tile 32
one "nes" CRTHIP_SIG_TILE=16 --system nesp0 --noise 12
one "nes k_decode no stores" CRTHIP_LIBDIR=ntsc-crt_amd/lib_dbg3 --system nesp0 --noise 12
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
