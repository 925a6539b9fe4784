cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s5
timeout 900 python -m pytest tests -m gpu -q -x -k "fused_fieldpass or stagewise or nes_parity or large_signal or random_configurations or f4_systems or wide_decoder_parity or full_size" > gpurun_out/r6s5/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s5/pytest_sel.log
tail -4 gpurun_out/r6s5/pytest_sel.log
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s5/modes.txt
MODE_ORDERS=-1,1 timeout 900 python tools/mode_probe.py --procs 8 >> gpurun_out/r6s5/modes.txt 2> gpurun_out/r6s5/modes.err
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s5/modes.txt
cut -c1-260 gpurun_out/r6s5/modes.txt
python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams 1 > gpurun_out/r6s5/bench640.json 2>/dev/null; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6s5/bench640.json").read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"])
PY
