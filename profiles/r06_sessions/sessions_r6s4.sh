cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s4
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s4/ab.txt
timeout 1500 python tools/ab_sweep.py tools/specs_r6s4.txt --procs 3 >> gpurun_out/r6s4/ab.txt 2> gpurun_out/r6s4/ab.err
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s4/ab.txt
cat gpurun_out/r6s4/ab.txt
