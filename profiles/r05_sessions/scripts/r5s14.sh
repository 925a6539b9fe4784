#!/bin/bash
# round 5, session 14: the kept-graph test; PMC calibration (incl. the encoder's own read pattern); rocprofv3 profiles of every workload
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s14
O=gpurun_out/r5s14
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "kept_graph or graph_capturable or wild_sync" > $O/pytest_graph.log 2>&1
echo "pytest(graphs, wild sync) rc=$?"; tail -3 $O/pytest_graph.log
C=gpurun_out/r5calib; rm -rf $C; mkdir -p $C
( cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$C/F -- $GRAFT_REPO_ROOT/tools/ubench_hbm.bin calib > $GRAFT_REPO_ROOT/$C/F.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$C/W -- $GRAFT_REPO_ROOT/tools/ubench_hbm.bin calib > $GRAFT_REPO_ROOT/$C/W.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$C/E -- $GRAFT_REPO_ROOT/tools/ubench_enc640.bin 4096 calib > $GRAFT_REPO_ROOT/$C/E.log 2>&1
)
tail -2 $C/E.log
find $C -name "*counter_collection.csv" | head
bash tools/refresh_profiles.sh r05 > $O/refresh.log 2>&1
tail -5 $O/refresh.log
