cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s50
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s50/box.txt
timeout 1200 python bench.py --full-json gpurun_out/r6s50/bench_full.json > gpurun_out/r6s50/bench_default.json 2> gpurun_out/r6s50/bench_default.err
tail -c 600 gpurun_out/r6s50/bench_default.json
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s50/box.txt
cat gpurun_out/r6s50/box.txt
python tools/placement_sweep.py --child 0 --batch 4096 --w 640 --h 480 --noise 24 2>/dev/null | cut -c1-400
