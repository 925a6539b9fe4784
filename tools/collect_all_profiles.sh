#!/bin/bash
# local side of tools/refresh_profiles.sh: gpurun_out/prof_<round><workload> -> profiles/<round>_* + profiles/traffic*.json
R=${1:-rXX}; P=${2:-$R}          # P: name prefix under profiles/ (e.g. r05)
cd "$(dirname "$0")/.."
python tools/collect_profiles.py ${R}headline ${P}_headline 4096 headline
python tools/collect_profiles.py ${R}1080p ${P}_1080p 2048 1080p_batch2048
python tools/collect_profiles.py ${R}vhs ${P}_vhs 2048 vhs_832x624
python tools/collect_profiles.py ${R}nes ${P}_nes 4096 nes_pattern0
python tools/collect_profiles.py ${R}pv1k ${P}_pv1k 4096 pv1k_batch4096
python tools/collect_profiles.py ${R}bloom ${P}_bloom 4096 bloom_batch4096
python tools/collect_profiles.py ${R}batch1 ${P}_batch1 1 640x480_batch1
