#!/bin/bash
# round 5, session 21 (final): PMC / rocprofv3 profiles of every workload on the tree as shipped
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s21
export TMPDIR=/tmp
bash tools/refresh_profiles.sh r05 > gpurun_out/r5s21/refresh.log 2>&1
tail -3 gpurun_out/r5s21/refresh.log
