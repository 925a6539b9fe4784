cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s8
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s8/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s8/pytest_all.log
tail -15 gpurun_out/r6s8/pytest_all.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s8/ab.txt
timeout 900 python tools/ab_sweep.py tools/specs_r6s8.txt --procs 3 >> gpurun_out/r6s8/ab.txt 2> gpurun_out/r6s8/ab.err
cat gpurun_out/r6s8/ab.txt
for sys in nesp0 pv1k ntscbloom; do python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --streams 1 --system $sys --noise 12 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sys', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"; done
