cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s23
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide or 1080 or full_size" > gpurun_out/r6s23/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s23/pytest_sel.log
tail -3 gpurun_out/r6s23/pytest_sel.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s23/ab.txt
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s23.txt --procs 4 >> gpurun_out/r6s23/ab.txt 2> gpurun_out/r6s23/ab.err
cat gpurun_out/r6s23/ab.txt
