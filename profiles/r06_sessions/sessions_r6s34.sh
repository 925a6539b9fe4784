cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s34
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_wskew timeout 900 python tools/soak_random.py 700000 1200 wide > gpurun_out/r6s34/soak_wskew.log 2>&1; tail -1 gpurun_out/r6s34/soak_wskew.log
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_wskew timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide or 1080" > gpurun_out/r6s34/pytest_wskew.log 2>&1; tail -2 gpurun_out/r6s34/pytest_wskew.log
timeout 600 python tools/soak_random.py 710000 600 > gpurun_out/r6s34/soak_default.log 2>&1; tail -1 gpurun_out/r6s34/soak_default.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s34/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s34.txt --procs 5 >> gpurun_out/r6s34/ab.txt 2> gpurun_out/r6s34/ab.err
cat gpurun_out/r6s34/ab.txt
