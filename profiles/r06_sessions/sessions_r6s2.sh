cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s2
timeout 600 python -m pytest tests -m gpu -x -q -k "wide_decoder_parity or fused_fieldpass_parity or full_size" > gpurun_out/r6s2/pytest_sel.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6s2/pytest_sel.log
tail -3 gpurun_out/r6s2/pytest_sel.log
timeout 1500 python tools/order_sweep.py --procs 3 > gpurun_out/r6s2/order.txt 2> gpurun_out/r6s2/order.err
cat gpurun_out/r6s2/order.txt
