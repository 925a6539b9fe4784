cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s26
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s26/ab.txt
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s26.txt --procs 5 >> gpurun_out/r6s26/ab.txt 2> gpurun_out/r6s26/ab.err
cat gpurun_out/r6s26/ab.txt
