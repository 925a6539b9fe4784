#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
mkdir -p gpurun_out/vhstrace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vhstrace -o t -- python bench.py --batch 2048 --steps 3 --warmup 1 --no-cpu --no-extra --system vhs --width 832 --height 624 --noise 12 > gpurun_out/vhstrace.log 2>&1
f=$(find gpurun_out/vhstrace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-14:]:
    print("%-40s start %10.1f us  dur %8.1f us  queue %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY
