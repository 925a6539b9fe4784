#!/bin/bash
# final profiles of the round: headline (trace + PMC + SQ counters) and 1080p batch 2048 (trace + PMC)
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
./tools/prof_bench.sh headline 2>&1 | tail -12
./tools/prof_sq.sh headline --no-extra > gpurun_out/sq_headline.log 2>&1; tail -3 gpurun_out/sq_headline.log
./tools/prof_bench.sh 1080p --width 1920 --height 1080 --noise 0 --batch 2048 2>&1 | tail -8
