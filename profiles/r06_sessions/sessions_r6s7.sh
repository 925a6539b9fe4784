cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s7
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s7/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s7/pytest_all.log
tail -5 gpurun_out/r6s7/pytest_all.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s7/ab.txt
timeout 900 python tools/ab_sweep.py tools/specs_r6s7.txt --procs 3 >> gpurun_out/r6s7/ab.txt 2> gpurun_out/r6s7/ab.err
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s7/ab.txt
cat gpurun_out/r6s7/ab.txt
CRTHIP_LIBDIR=$GRAFT_REPO_ROOT/ntsc-crt_amd/lib_dbgA bash tools/prof_sq.sh r6occ5 --no-extra > gpurun_out/sq_r6occ5.txt 2>&1
bash tools/prof_sq.sh r6occ4 --no-extra > gpurun_out/sq_r6occ4.txt 2>&1
grep -A 26 "void k_decode<SysNTSC, 0" gpurun_out/sq_r6occ5.txt | head -30
