#!/bin/bash
# round 4 session 39: the 1080p encoder taken apart (no signal stores / no image loads) and with larger signal pieces (128 / 256 bytes instead of 64)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s39; mkdir -p $O
export TMPDIR=/tmp
W="--width 1920 --height 1080 --noise 0 --batch 2048"
for i in 1 2; do
for v in lib lib_ot32 lib_ot64 lib_edbg1 lib_edbg2; do
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/$v timeout 200 python bench.py --streams 1 --no-cpu --no-extra --steps 20 --warmup 5 $W 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s field-pass %.4f ms  kernel_ms %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/ab.txt
done
done
