#!/bin/bash
# round 4 session 42: 64-byte vs 256-byte signal pieces at 512 / 1024 fields of 1080p
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s42; mkdir -p $O
export TMPDIR=/tmp
for b in 512 1024 512; do
for v in lib_ot16 lib; do
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/$v timeout 60 python bench.py --streams 1 --no-cpu --no-extra --steps 20 --warmup 5 --width 1920 --height 1080 --noise 0 --batch $b 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-9s x $b  field-pass %.4f ms  kernel_ms %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/ab.txt
done
done
