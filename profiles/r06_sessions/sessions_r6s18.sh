cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s18
(time timeout 1500 python tools/soak_random.py 100000 4000) > gpurun_out/r6s18/soak.log 2>&1; tail -5 gpurun_out/r6s18/soak.log
(time timeout 900 python tools/soak_random.py 110000 1500 wide) > gpurun_out/r6s18/soak_wide.log 2>&1; tail -5 gpurun_out/r6s18/soak_wide.log
(time CRTHIP_SIG_PAD=0 timeout 900 python tools/soak_random.py 120000 800) > gpurun_out/r6s18/soak_flat.log 2>&1; tail -5 gpurun_out/r6s18/soak_flat.log
(time CRTHIP_SIG_TILE=64 timeout 900 python tools/soak_random.py 130000 800) > gpurun_out/r6s18/soak_tile64.log 2>&1; tail -5 gpurun_out/r6s18/soak_tile64.log
(time CRTHIP_WIDE_LPW=16 CRTHIP_SIG_TILE=64 timeout 900 python tools/soak_random.py 140000 500 wide) > gpurun_out/r6s18/soak_wide16.log 2>&1; tail -5 gpurun_out/r6s18/soak_wide16.log
