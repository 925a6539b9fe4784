cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s3
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s3/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s3/pytest_all.log
tail -30 gpurun_out/r6s3/pytest_all.log
for pad in 1 0; do
  CRTHIP_SIG_PAD=$pad python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams 1 > gpurun_out/r6s3/bench640_pad$pad.json 2> gpurun_out/r6s3/bench640_pad$pad.err
  tail -c 1500 gpurun_out/r6s3/bench640_pad$pad.json | head -c 1500; echo
  CRTHIP_SIG_PAD=$pad python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams 1 --width 1920 --height 1080 --noise 0 --batch 2048 > gpurun_out/r6s3/bench1080_pad$pad.json 2> gpurun_out/r6s3/bench1080_pad$pad.err
  tail -c 1500 gpurun_out/r6s3/bench1080_pad$pad.json | head -c 1500; echo
done
timeout 900 python tools/mode_probe.py --procs 4 > gpurun_out/r6s3/modes.txt 2> gpurun_out/r6s3/modes.err
cat gpurun_out/r6s3/modes.txt | cut -c1-330
