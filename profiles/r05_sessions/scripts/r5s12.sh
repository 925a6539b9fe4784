#!/bin/bash
# round 5, session 12: k_active with the DEFERRED drain (CRTHIP_SIG_TILE=48): fused parity cases, then A/B against 16 / 32
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s12
O=gpurun_out/r5s12
export TMPDIR=/tmp
CRTHIP_SIG_TILE=48 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "test_parity or random or vhs or bloom or f4 or full_batch or build_time" > $O/pytest_sig48.log 2>&1
echo "pytest(SIG_TILE=48) rc=$?"; tail -4 $O/pytest_sig48.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for t in 16 32 48; do
one "640x4096 sig$t" CRTHIP_SIG_TILE=$t
one "640x1024 sig$t" CRTHIP_SIG_TILE=$t --batch 1024
one "1080x2048 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 2048
one "1080x512 sig$t" CRTHIP_SIG_TILE=$t --width 1920 --height 1080 --noise 0 --batch 512
one "vhs sig$t" CRTHIP_SIG_TILE=$t --system vhs --width 832 --height 624 --noise 12 --batch 2048
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
