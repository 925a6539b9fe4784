cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s49
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6s49/pytest_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s49/pytest_all.log
tail -4 gpurun_out/r6s49/pytest_all.log
bash tools/refresh_profiles.sh r6i > gpurun_out/r6s49/refresh.log 2>&1
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s49/box.txt
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s49/box.txt
cat gpurun_out/r6s49/box.txt
