cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s1
timeout 900 python -m pytest tests -m gpu -x -q -k "wide or kept or wild or full_size" > gpurun_out/r6s1/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s1/pytest_sel.log
tail -5 gpurun_out/r6s1/pytest_sel.log
timeout 1200 python tools/placement_sweep.py --procs 5 > gpurun_out/r6s1/placement.txt 2> gpurun_out/r6s1/placement.err
cat gpurun_out/r6s1/placement.txt
