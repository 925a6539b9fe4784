cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s14
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s14/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s14/pytest_all.log
tail -6 gpurun_out/r6s14/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
