#!/bin/bash
# round 5, session 30 (comments only changed since r5s29): smoke + a parity subset, profiles of every workload for the traffic files' source hash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s30
O=gpurun_out/r5s30
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "smoke or test_parity or kept_graph or vhs_rand" > $O/pytest.log 2>&1
echo "pytest(subset) rc=$?"; tail -2 $O/pytest.log
bash tools/refresh_profiles.sh r05 > $O/refresh.log 2>&1
tail -2 $O/refresh.log
