#!/bin/bash
# the round's committed profiles: headline (kernel trace + PMC + SQ counters), 1080p batch 2048 and the VHS workload (trace)
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
./tools/prof_bench.sh headline 2>&1 | tail -2
./tools/prof_sq.sh headline --no-extra > gpurun_out/sq_headline.log 2>&1
./tools/prof_bench.sh 1080p --width 1920 --height 1080 --noise 0 --batch 2048 2>&1 | tail -1
./tools/prof_bench.sh vhs --system vhs --width 832 --height 624 --noise 12 --batch 2048 2>&1 | tail -1
./tools/prof_bench.sh nes --system nesp0 --width 256 --height 240 --outw 640 --outh 480 --noise 12 --batch 4096 2>&1 | tail -1
