#!/bin/bash
# round 5, session 1: today's box -- start-up breakdown, encoder memory shapes, baseline bench, 640x480 / 1080p encoder knock-outs
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s1
export TMPDIR=/tmp
O=gpurun_out/r5s1
{
echo "## hip_floor (one-shot HIP process: runtime floor)"
for i in 1 2 3; do /usr/bin/time -f "wall %e s" tools/hip_floor.bin; done
echo "## startup_probe (drop-in API, as crt_main.c calls it)"
for i in 1 2 3; do /usr/bin/time -f "wall %e s" ntsc-crt_amd/lib/startup_probe; done
echo "## startup_probe hipinit (runtime initialised by hand first)"
for i in 1 2; do /usr/bin/time -f "wall %e s" ntsc-crt_amd/lib/startup_probe hipinit; done
echo "## time_cli"
python tools/time_cli.py 5
} > $O/startup.txt 2>&1
timeout 300 tools/ubench_enc.bin 2048 > $O/ubench_enc_2048.txt 2>&1
timeout 300 tools/ubench_enc.bin 512 > $O/ubench_enc_512.txt 2>&1
timeout 600 python bench.py --full-json $O/bench_full.json > $O/bench_default.out 2> $O/bench_default.err
tail -1 $O/bench_default.out > $O/bench_default_line.json
one() { # tag libdir args...
  tag=$1; dir=$2; shift 2
  CRTHIP_LIBDIR=$dir timeout 300 python bench.py --no-cpu --no-extra --streams 1 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2; do
one "640 base   " ntsc-crt_amd/lib
one "640 nostore" ntsc-crt_amd/lib_dbg1
one "640 noload " ntsc-crt_amd/lib_dbg2
one "1080 base   " ntsc-crt_amd/lib --width 1920 --height 1080 --noise 0 --batch 2048
one "1080 nostore" ntsc-crt_amd/lib_dbg1 --width 1920 --height 1080 --noise 0 --batch 2048
one "1080 noload " ntsc-crt_amd/lib_dbg2 --width 1920 --height 1080 --noise 0 --batch 2048
done
} > $O/knockouts.txt 2>&1
cat $O/startup.txt $O/ubench_enc_2048.txt $O/knockouts.txt
