#!/bin/bash
# round 5, session 24: k_active_nes with 128-byte signal pieces: NES parity with the large tile forced, A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s24
O=gpurun_out/r5s24
export TMPDIR=/tmp
CRTHIP_SIG_TILE=32 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "nes" > $O/pytest.log 2>&1
echo "pytest(nes, SIG_TILE=32) rc=$?"; tail -2 $O/pytest.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2 3; do
for t in 16 32; do
one "nes x4096 sig$t" CRTHIP_SIG_TILE=$t --system nesp0 --noise 12
done
done
one "nes x2048 sig16" CRTHIP_SIG_TILE=16 --system nesp0 --noise 12 --batch 2048
one "nes x2048 sig32" CRTHIP_SIG_TILE=32 --system nesp0 --noise 12 --batch 2048
one "nes(pattern 2) x4096 sig16" CRTHIP_SIG_TILE=16 --system nes --noise 12
one "nes(pattern 2) x4096 sig32" CRTHIP_SIG_TILE=32 --system nes --noise 12
} > $O/ab.txt 2>&1
cat $O/ab.txt
