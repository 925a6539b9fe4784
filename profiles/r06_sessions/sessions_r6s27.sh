cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s27
(time timeout 2400 python tools/soak_random.py 200000 16000) > gpurun_out/r6s27/soak.log 2>&1; tail -5 gpurun_out/r6s27/soak.log
(time timeout 1500 python tools/soak_random.py 300000 5000 wide) > gpurun_out/r6s27/soak_wide.log 2>&1; tail -5 gpurun_out/r6s27/soak_wide.log
timeout 600 python bench.py --no-cpu > gpurun_out/r6s27/bench_nocpu.json 2>/dev/null; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6s27/bench_nocpu.json").read().strip().splitlines()[-1]); print("bench", j["value"], j["one_batch_in_flight"], j["config"]["batches_in_flight_fps"], j["roofline"]["kernel_ms"], j["roofline"]["traffic_stale"]); print([(e["name"], e["value"], e["one"]) for e in j["extras"]])
PY
