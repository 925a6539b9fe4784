cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s37
timeout 900 python tools/soak_random.py 900000 1500 > gpurun_out/r6s37/soak.log 2>&1; tail -1 gpurun_out/r6s37/soak.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s37/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s37.txt --procs 5 >> gpurun_out/r6s37/ab.txt 2> gpurun_out/r6s37/ab.err
cat gpurun_out/r6s37/ab.txt
