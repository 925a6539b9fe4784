cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s51
(time timeout 600 python tools/soak_random.py 1200000 2500) > gpurun_out/r6s51/soak.log 2>&1; tail -4 gpurun_out/r6s51/soak.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or shape_by_batch or large_signal" > gpurun_out/r6s51/pytest_sel.log 2>&1; tail -2 gpurun_out/r6s51/pytest_sel.log
timeout 400 python bench.py --no-cpu --no-extra > gpurun_out/r6s51/bench_nocpu.json 2>/dev/null; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6s51/bench_nocpu.json").read().strip().splitlines()[-1]); print("bench", j["value"], j["one_batch_in_flight"]["value"], j["roofline"]["kernel_ms"], j["roofline"]["traffic_stale"])
PY
