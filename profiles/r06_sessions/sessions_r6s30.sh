cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s30
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_il_rearm timeout 600 python tools/soak_random.py 500000 1500 > gpurun_out/r6s30/soak_il_rearm.log 2>&1; tail -2 gpurun_out/r6s30/soak_il_rearm.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s30/ab.txt
timeout 1800 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s30.txt --procs 5 >> gpurun_out/r6s30/ab.txt 2> gpurun_out/r6s30/ab.err
cat gpurun_out/r6s30/ab.txt
