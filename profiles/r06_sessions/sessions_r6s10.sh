cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s10
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s10/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s10/pytest_all.log
tail -4 gpurun_out/r6s10/pytest_all.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s10/ab.txt
timeout 900 python tools/ab_sweep.py tools/specs_r6s10.txt --procs 3 >> gpurun_out/r6s10/ab.txt 2> gpurun_out/r6s10/ab.err
cat gpurun_out/r6s10/ab.txt
