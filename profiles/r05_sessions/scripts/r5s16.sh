#!/bin/bash
# round 5, session 17: k_vhs_tail with the band's cosine in a per-segment table: VHS parity, then the noise pair
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r5s17
O=gpurun_out/r5s17
export TMPDIR=/tmp
tools/probe_lds128.bin
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "vhs or VHS" > $O/pytest_vhs.log 2>&1
echo "pytest(vhs) rc=$?"; tail -3 $O/pytest_vhs.log
one() { # tag env args...
  tag=$1; envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu --no-extra --streams 1 --warmup 2 --steps 10 "$@" 2>>$O/err.txt | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$tag', round(j['ms_per_step'],4), round(j['value']), {k: round(v,4) for k,v in j['roofline']['kernel_ms'].items()})"
}
{
for r in 1 2 3; do
one "vhs x2048" A=1 --system vhs --width 832 --height 624 --noise 12 --batch 2048
done
one "vhs x512" A=1 --system vhs --width 832 --height 624 --noise 12 --batch 512
one "vhs x64" A=1 --system vhs --width 832 --height 624 --noise 12 --batch 64 --steps 30
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extra --streams 1 --system vhs --width 832 --height 624 --noise 12 --batch 2048 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/vhs_kernel_stats.csv; rm -rf $O/trace; grep "k_vhs" $O/vhs_kernel_stats.csv | cut -c1-40,200-330
