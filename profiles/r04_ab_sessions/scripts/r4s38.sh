#!/bin/bash
# round 4 session 38: is k_active at 1080p bound by how many waves a CU holds?  dynamic LDS padding lowers its occupancy (11 waves per CU as built)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s38; mkdir -p $O
export TMPDIR=/tmp
W="--width 1920 --height 1080 --noise 0 --batch 2048"
for pad in 0 2400 6600 13500 18900 40000; do
CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_pad CRTHIP_ACTIVE_PAD_LDS=$pad timeout 200 python bench.py --streams 1 --no-cpu --no-extra --steps 20 --warmup 5 $W 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); lds=13824+$pad; print('pad $pad  LDS %d  waves/CU %d  field-pass %.4f ms  kernel_ms %s' % (lds, 163840//lds, d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $O/ab.txt
done
