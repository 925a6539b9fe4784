cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s48
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s48/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s48.txt --procs 3 >> gpurun_out/r6s48/ab.txt 2> gpurun_out/r6s48/ab.err
cat gpurun_out/r6s48/ab.txt
