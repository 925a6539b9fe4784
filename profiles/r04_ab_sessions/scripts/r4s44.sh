#!/bin/bash
# round 4 session 44: does the DRAFT wide encoder (next/wire_in.patch, built into lib_we) produce the reference's bytes, and how fast is it?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s44; mkdir -p $O
export TMPDIR=/tmp
( CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_we timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide_decoder_parity" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -6 $O/pytest.log | cut -c1-250
W="--width 1920 --height 1080 --noise 0"
for v in lib lib_we; do
for b in 2048 512; do
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/$v timeout 40 python bench.py --streams 1 --no-cpu --no-extra --steps 10 --warmup 3 $W --batch $b 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-7s x $b field-pass %.4f ms  kernel_ms %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/ab.txt
done
done
