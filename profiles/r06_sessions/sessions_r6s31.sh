cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s31
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s31/ab.txt
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s31.txt --procs 5 >> gpurun_out/r6s31/ab.txt 2> gpurun_out/r6s31/ab.err
cat gpurun_out/r6s31/ab.txt
