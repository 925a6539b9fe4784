cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s47
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s47/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s47.txt --procs 3 >> gpurun_out/r6s47/ab.txt 2> gpurun_out/r6s47/ab.err
cat gpurun_out/r6s47/ab.txt
