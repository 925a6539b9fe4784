#!/bin/bash
# round 5, session 10: k_active with the ALIGNED drain (CRTHIP_SIG_TILE=40): parity (whole parity file), then A/B against 16 / 32 / 64
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s10
O=gpurun_out/r5s10
export TMPDIR=/tmp
CRTHIP_SIG_TILE=40 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sig40.log 2>&1
echo "pytest(SIG_TILE=40, whole parity file) rc=$?"; tail -5 $O/pytest_sig40.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 32 40; do
one "640x4096 sig$t" CRTHIP_SIG_TILE=$t
one "640x1024 sig$t" CRTHIP_SIG_TILE=$t --batch 1024
one "640x512 sig$t" CRTHIP_SIG_TILE=$t --batch 512
done
for t in 16 40 64; do
one "1080x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 2048
one "1080x512 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 512
done
for t in 16 32 40; do
one "vhs sig$t" CRTHIP_SIG_TILE=$t --system vhs --width 832 --height 624 --noise 12 --batch 2048
one "bloom sig$t" CRTHIP_SIG_TILE=$t --system ntscbloom
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
