#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace (+ separate PMC passes for HBM bytes: --pmc never together with a trace domain) of one bench
# workload.  Summaries land in gpurun_out/prof_<tag>/ together with args.txt (the exact bench command line); copy what
# should be judged into profiles/ (tools/collect_profiles.py).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
out=gpurun_out/prof_$tag
mkdir -p $out
args="--steps 20 --warmup 3 --no-cpu --no-extra --streams 1 $*"   # one batch in flight: a kernel's duration beside another batch's kernels says nothing about the kernel
# (r6) CRTHIP_MARGIN_SIDE=0: the margin kernel in sequence with the active-video kernel, as bench.py's own per-kernel timing runs them --
# beside each other (the default from 512 fields on) both kernels' durations stretch over the same interval and their sum says nothing
echo "CRTHIP_MARGIN_SIDE=0 python bench.py $args" > $out/args.txt
CRTHIP_MARGIN_SIDE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py $args > $out/trace.log 2>&1
tail -1 $out/trace.log | cut -c1-300
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/kernel_stats.csv && head -8 $out/kernel_stats.csv | cut -c1-160
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu --no-extra --streams 1 $* > $out/pmc_$c.log 2>&1
  g=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && cp "$g" $out/pmc_$c.csv && python3 - "$g" $c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = r.get("Kernel_Name", "?").split("(")[0][:60]
    agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
    print("%-12s %-60s launches %4d  avg/launch %.1f (counter units; KB for *_SIZE)" % (sys.argv[2], k, n, v / n))
PY
  rm -rf $out/pmc_$c
done
