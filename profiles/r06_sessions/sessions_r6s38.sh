cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s38
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s38/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s38/pytest_all.log
tail -4 gpurun_out/r6s38/pytest_all.log
timeout 900 python tools/soak_random.py 910000 1500 > gpurun_out/r6s38/soak.log 2>&1; tail -1 gpurun_out/r6s38/soak.log
timeout 600 python tools/soak_random.py 920000 600 wide > gpurun_out/r6s38/soak_wide.log 2>&1; tail -1 gpurun_out/r6s38/soak_wide.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s38/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s38.txt --procs 5 >> gpurun_out/r6s38/ab.txt 2> gpurun_out/r6s38/ab.err
cat gpurun_out/r6s38/ab.txt
