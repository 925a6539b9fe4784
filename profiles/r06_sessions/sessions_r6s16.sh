cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s16
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s16/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s16/pytest_all.log
tail -4 gpurun_out/r6s16/pytest_all.log
(time timeout 1500 python tools/soak_random.py 90000 600) > gpurun_out/r6s16/soak.log 2>&1; tail -5 gpurun_out/r6s16/soak.log
(time timeout 900 python tools/soak_random.py 91000 200 wide) > gpurun_out/r6s16/soak_wide.log 2>&1; tail -5 gpurun_out/r6s16/soak_wide.log
bash tools/refresh_profiles.sh r6c > gpurun_out/r6s16/refresh.log 2>&1
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s16/box.txt
timeout 1200 python bench.py --full-json gpurun_out/r6s16/bench_full.json > gpurun_out/r6s16/bench_default.json 2> gpurun_out/r6s16/bench_default.err
tail -c 3000 gpurun_out/r6s16/bench_default.json
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s16/box.txt
cat gpurun_out/r6s16/box.txt
