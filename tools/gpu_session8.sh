#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/shape_sweep.py > gpurun_out/shape_sweep.txt 2>gpurun_out/shape_sweep.err
cat gpurun_out/shape_sweep.txt
