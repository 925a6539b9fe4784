#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "f4_systems or bloom or pv1k or dropin_f4 or crt_main" > gpurun_out/pytest_pv1k.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_pv1k.log
timeout 300 python tools/soak_random.py 70007 8 2>&1 | tail -2
timeout 300 python tools/soak_random.py 70023 8 2>&1 | tail -2
for b in 1024 4096; do
  timeout 300 python bench.py --system pv1k --batch $b --steps 5 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pv1k', j['config']['fields_per_gpu_per_step'], round(j['value']), round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
done
