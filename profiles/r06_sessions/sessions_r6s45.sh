cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s45
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s45/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s45/pytest_all.log
tail -4 gpurun_out/r6s45/pytest_all.log
bash tools/refresh_profiles.sh r6h > gpurun_out/r6s45/refresh.log 2>&1
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s45/box.txt
timeout 1200 python bench.py --full-json gpurun_out/r6s45/bench_full.json > gpurun_out/r6s45/bench_default.json 2> gpurun_out/r6s45/bench_default.err
tail -c 2500 gpurun_out/r6s45/bench_default.json
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s45/box.txt
cat gpurun_out/r6s45/box.txt
