#!/bin/bash
# round 5, session 27: k_margin with the field as grid y and a multiply-high instead of two integer divisions: parity file, timings
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s27
O=gpurun_out/r5s27
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest(parity file) rc=$?"; tail -3 $O/pytest.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
one "640x4096" A=1
one "1080x2048" A=1 --width 1920 --height 1080 --noise 0 --batch 2048
one "nes x4096" A=1 --system nesp0 --noise 12
one "pv1k" A=1 --system pv1k
one "vhs" A=1 --system vhs --width 832 --height 624 --noise 12 --batch 2048
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
