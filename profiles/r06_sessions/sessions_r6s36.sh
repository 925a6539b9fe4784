cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s36
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s36/box.txt
timeout 1200 python bench.py --full-json gpurun_out/r6s36/bench_full.json > gpurun_out/r6s36/bench_default.json 2> gpurun_out/r6s36/bench_default.err
tail -c 1500 gpurun_out/r6s36/bench_default.json
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s36/box.txt
cat gpurun_out/r6s36/box.txt
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_maxilp timeout 600 python tools/soak_random.py 800000 300 > gpurun_out/r6s36/soak_maxilp.log 2>&1; tail -1 gpurun_out/r6s36/soak_maxilp.log
timeout 900 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s36.txt --procs 5 > gpurun_out/r6s36/ab.txt 2> gpurun_out/r6s36/ab.err
cat gpurun_out/r6s36/ab.txt
