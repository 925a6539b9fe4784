#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s23; mkdir -p $O
timeout 300 tools/ubench_hbm.bin spread > $O/spread.txt 2>&1; cat $O/spread.txt
