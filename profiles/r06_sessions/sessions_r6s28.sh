cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s28
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_rearm timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not full_size and not 1080" > gpurun_out/r6s28/pytest_rearm.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s28/pytest_rearm.log
tail -3 gpurun_out/r6s28/pytest_rearm.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s28/ab.txt
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s28.txt --procs 5 >> gpurun_out/r6s28/ab.txt 2> gpurun_out/r6s28/ab.err
cat gpurun_out/r6s28/ab.txt
