#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python bench.py --strong 512 --width 1920 --height 1080 --noise 0 --steps 10 --warmup 3 --no-cpu --no-extra > gpurun_out/strong512.json 2>gpurun_out/strong512.err; echo "strong rc=$?"
timeout 300 python bench.py --sequence --batch 1024 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/seq1024.json 2>gpurun_out/seq.err; echo "seq rc=$?"
timeout 300 python bench.py --system ntscbloom --batch 1024 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/bloom1024.json 2>gpurun_out/bloom.err; echo "bloom rc=$?"
timeout 300 python bench.py --system pv1k --batch 1024 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/pv1k1024.json 2>gpurun_out/pv1k.err; echo "pv1k rc=$?"
timeout 300 python bench.py --system snes --batch 4096 --steps 10 --warmup 2 --no-cpu --no-extra > gpurun_out/snes4096.json 2>gpurun_out/snes.err; echo "snes rc=$?"
python - <<'PY'
import json
for f in ("strong512", "seq1024", "bloom1024", "pv1k1024", "snes4096"):
    try:
        j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "fps=%.0f ms=%.4f" % (j["value"], j["ms_per_step"]), j["scaling"], j["config"]["workload"][:60], {k: round(v, 4) for k, v in j["roofline"]["kernel_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e, open("gpurun_out/%s.err" % f.rstrip("0123456789")).read()[-300:] if False else "")
PY
tail -3 gpurun_out/*.err 2>/dev/null | tail -20
