#!/bin/bash
# round 5, session 9: k_decode_wide with 8 scanlines per wave (parity of the wide cases, then A/B by batch size); the encoder's memory
# pattern at 640x480 (tools/ubench_enc.hip -DGEO640)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s9
O=gpurun_out/r5s9
export TMPDIR=/tmp
CRTHIP_WIDE_LPW=8 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wide or 1080 or case4 or case5 or random or graph" > $O/pytest_lpw8.log 2>&1
echo "pytest(LPW=8 forced, wide cases) rc=$?"; tail -3 $O/pytest_lpw8.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wide_pictures or test_parity" > $O/pytest_auto.log 2>&1
echo "pytest(auto) rc=$?"; tail -2 $O/pytest_auto.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
for l in 16 8; do
for b in 32 64 128 256; do
one "1080x$b lpw$l" CRTHIP_WIDE_LPW=$l --width 1920 --height 1080 --noise 0 --batch $b --steps 30
done
done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 200 tools/ubench_enc640.bin 4096 > $O/ubench_enc640_4096.txt 2>&1
head -40 $O/ubench_enc640_4096.txt
