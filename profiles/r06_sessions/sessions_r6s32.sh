cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s32
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s32/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s32/pytest_all.log
tail -4 gpurun_out/r6s32/pytest_all.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s32/ab.txt
timeout 1500 python tools/ab_sweep.py profiles/r06_sessions/specs_r6s32.txt --procs 5 >> gpurun_out/r6s32/ab.txt 2> gpurun_out/r6s32/ab.err
cat gpurun_out/r6s32/ab.txt
