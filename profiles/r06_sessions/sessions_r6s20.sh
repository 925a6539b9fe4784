cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s20
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s20/ab.txt
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s20.txt --procs 3 >> gpurun_out/r6s20/ab.txt 2> gpurun_out/r6s20/ab.err
cat gpurun_out/r6s20/ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wild_sync or full_size or full_batch" > gpurun_out/r6s20/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s20/pytest_sel.log
tail -3 gpurun_out/r6s20/pytest_sel.log
