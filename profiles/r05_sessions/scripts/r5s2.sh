#!/bin/bash
# round 5, session 2: start-up breakdown (fixed), encoder memory shapes v2 (alignment / store policy)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s2
O=gpurun_out/r5s2
w() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "wall $(( (e-s)/1000000 )) ms"; }
{
echo "## hip_floor (one-shot HIP process: runtime floor)"
for i in 1 2 3; do w tools/hip_floor.bin; done
echo "## startup_probe (drop-in API, as crt_main.c calls it)"
for i in 1 2 3; do w ntsc-crt_amd/lib/startup_probe; done
echo "## startup_probe hipinit (runtime initialised by hand first)"
for i in 1 2; do w ntsc-crt_amd/lib/startup_probe hipinit; done
echo "## startup_probe, AMD_LOG_LEVEL=0 HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=1 w ntsc-crt_amd/lib/startup_probe
} > $O/startup.txt 2>&1
timeout 300 tools/ubench_enc.bin 2048 > $O/ubench_enc_2048.txt 2>&1
cat $O/startup.txt $O/ubench_enc_2048.txt
