cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s44
(time timeout 1500 python tools/soak_random.py 1000000 8000) > gpurun_out/r6s44/soak.log 2>&1; tail -5 gpurun_out/r6s44/soak.log
(time timeout 900 python tools/soak_random.py 1100000 2500 wide) > gpurun_out/r6s44/soak_wide.log 2>&1; tail -5 gpurun_out/r6s44/soak_wide.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
