cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s33
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_enc_il timeout 600 python tools/soak_random.py 600000 600 > gpurun_out/r6s33/soak_enc_il.log 2>&1; tail -1 gpurun_out/r6s33/soak_enc_il.log
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_px timeout 600 python tools/soak_random.py 610000 600 > gpurun_out/r6s33/soak_px.log 2>&1; tail -1 gpurun_out/r6s33/soak_px.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s33/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s33.txt --procs 5 >> gpurun_out/r6s33/ab.txt 2> gpurun_out/r6s33/ab.err
cat gpurun_out/r6s33/ab.txt
