cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s29
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_nonop_all timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not full_size" > gpurun_out/r6s29/pytest_nonop_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s29/pytest_nonop_all.log
tail -3 gpurun_out/r6s29/pytest_nonop_all.log
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_rearm_nonop timeout 600 python tools/soak_random.py 400000 600 > gpurun_out/r6s29/soak_rearm_nonop.log 2>&1; tail -2 gpurun_out/r6s29/soak_rearm_nonop.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s29/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s29.txt --procs 5 >> gpurun_out/r6s29/ab.txt 2> gpurun_out/r6s29/ab.err
cat gpurun_out/r6s29/ab.txt
