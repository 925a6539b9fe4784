cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s9
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s9/ab.txt
timeout 900 python tools/ab_sweep.py tools/specs_r6s9.txt --procs 3 >> gpurun_out/r6s9/ab.txt 2> gpurun_out/r6s9/ab.err
cat gpurun_out/r6s9/ab.txt
timeout 1200 python bench.py > gpurun_out/r6s9/bench_default.json 2> gpurun_out/r6s9/bench_default.err
tail -c 4000 gpurun_out/r6s9/bench_default.json
cp gpurun_out/bench_full.json gpurun_out/r6s9/bench_full.json
