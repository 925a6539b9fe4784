#!/bin/bash
# rocprofv3 kernel trace of the default bench workload; summaries land in gpurun_out/prof_*
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/prof_$1.log 2>&1
tail -2 gpurun_out/prof_$1.log
find gpurun_out/prof_$1 -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
