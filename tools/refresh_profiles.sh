#!/bin/bash
# GPU side of "make the committed profiles match the tree" (roofline.traffic is read from
# profiles/traffic*.json and must not go stale):
#     gpurun -- 'bash tools/refresh_profiles.sh r05'        (about 2.5 minutes of GPU time)
# then, back here, where gpurun_out/ has been merged:
#     tools/collect_all_profiles.sh r05                      (writes profiles/r05_* and profiles/traffic*.json)
# Per workload: rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in PMC passes of their own (tools/prof_bench.sh);
# SQ counters for the three main workloads (tools/prof_sq.sh).  bench.py compares the hash of ntsc-crt_amd/csrc/ stored in the
# traffic files with the tree it runs on and says `traffic_stale` when they differ.
R=${1:-rXX}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/prof_bench.sh ${R}headline
bash tools/prof_bench.sh ${R}1080p --width 1920 --height 1080 --noise 0 --batch 2048
bash tools/prof_bench.sh ${R}vhs --system vhs --width 832 --height 624 --noise 12 --batch 2048
bash tools/prof_bench.sh ${R}nes --system nesp0 --noise 12
bash tools/prof_bench.sh ${R}pv1k --system pv1k
bash tools/prof_bench.sh ${R}bloom --system ntscbloom
bash tools/prof_bench.sh ${R}batch1 --batch 1
bash tools/prof_sq.sh ${R}headline --no-extra > gpurun_out/sq_${R}headline.txt 2>&1
bash tools/prof_sq.sh ${R}1080p --no-extra --width 1920 --height 1080 --noise 0 --batch 2048 > gpurun_out/sq_${R}1080p.txt 2>&1
bash tools/prof_sq.sh ${R}vhs --no-extra --system vhs --width 832 --height 624 --noise 12 --batch 2048 > gpurun_out/sq_${R}vhs.txt 2>&1
