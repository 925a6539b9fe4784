#!/bin/bash
# round 5, session 19: full-rate LCG / noise products in k_active_nes and k_active_row: NES + small-batch parity, timings
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s19
O=gpurun_out/r5s19
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "nes or scanline_parallel or random or shape" > $O/pytest.log 2>&1
echo "pytest(nes, row shape) rc=$?"; tail -3 $O/pytest.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
one "nes x4096" A=1 --system nesp0 --noise 12 --steps 10
one "640x64" A=1 --batch 64 --steps 30
one "640x256" A=1 --batch 256 --steps 30
one "640x1" A=1 --batch 1 --steps 30
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
