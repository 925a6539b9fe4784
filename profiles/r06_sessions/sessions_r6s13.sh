cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s13
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s13/pad_ab.txt
timeout 1500 python tools/pad_ab.py --procs 3 >> gpurun_out/r6s13/pad_ab.txt 2> gpurun_out/r6s13/pad_ab.err
cat gpurun_out/r6s13/pad_ab.txt
