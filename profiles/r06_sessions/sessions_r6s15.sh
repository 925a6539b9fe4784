cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s15
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s15/ab.txt
timeout 900 python tools/ab_sweep.py tools/specs_r6s15.txt --procs 3 >> gpurun_out/r6s15/ab.txt 2> gpurun_out/r6s15/ab.err
cat gpurun_out/r6s15/ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "full_size or graph or tuning_switches or bench or vhs_fieldpass" > gpurun_out/r6s15/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s15/pytest_sel.log
tail -4 gpurun_out/r6s15/pytest_sel.log
for s in 1 2; do for m in 1 0; do CRTHIP_MARGIN_SIDE=$m python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams $s 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s margin_side $m', j['value'], j['ms_per_step'])"; done; done
