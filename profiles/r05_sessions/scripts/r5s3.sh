#!/bin/bash
# round 5, session 3: k_active with 256-byte signal pieces (CRTHIP_SIG_TILE=64): parity, then A/B against 64-byte pieces; --overlap once more
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s3
O=gpurun_out/r5s3
export TMPDIR=/tmp
CRTHIP_SIG_TILE=64 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sig64.log 2>&1
echo "pytest(SIG_TILE=64) rc=$?"; tail -3 $O/pytest_sig64.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 64; do
one "640x4096 sig$t" CRTHIP_SIG_TILE=$t
one "1080x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 2048
one "1080x1024 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 1024
one "1080x512 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 512
one "640x2048 sig$t" CRTHIP_SIG_TILE=$t --batch 2048
one "640x1024 sig$t" CRTHIP_SIG_TILE=$t --batch 1024
done
done
one "vhs sig16" CRTHIP_SIG_TILE=16 --system vhs --width 832 --height 624 --noise 12 --batch 2048
one "vhs sig64" CRTHIP_SIG_TILE=64 --system vhs --width 832 --height 624 --noise 12 --batch 2048
one "pv1k sig16" CRTHIP_SIG_TILE=16 --system pv1k --batch 4096
one "pv1k sig64" CRTHIP_SIG_TILE=64 --system pv1k --batch 4096
one "1080x2048 auto overlap2" A=1 --width 1920 --height 1080 --noise 0 --batch 2048 --overlap 2
one "1080x2048 auto overlap3" A=1 --width 1920 --height 1080 --noise 0 --batch 2048 --overlap 3
one "1080x2048 auto overlap4" A=1 --width 1920 --height 1080 --noise 0 --batch 2048 --overlap 4
one "640x4096 auto overlap2" A=1 --overlap 2
} > $O/ab.txt 2>&1
cat $O/ab.txt
