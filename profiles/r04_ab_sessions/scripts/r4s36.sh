#!/bin/bash
# round 4 session 36: instruction-cache and latency counters of the encoder / decoder kernels (lead 5 of DESIGN section 9)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s36; mkdir -p $O
export TMPDIR=/tmp
i=0
for wl in "h" "w"; do
  if [ $wl = h ]; then A=""; else A="--width 1920 --height 1080 --noise 0 --batch 2048"; fi
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES"; do
    i=$((i+1))
    ( cd /tmp && timeout 200 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --streams 1 $A > $GRAFT_REPO_ROOT/$O/p$i.log 2>&1 )
  done
done
python3 - $O <<'PY'
import csv, sys, glob, collections
for wl, ps in (("640x480 x 4096", ("p1", "p2")), ("1080p x 2048", ("p3", "p4"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for p in ps:
        for f in glob.glob(sys.argv[1] + "/" + p + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if not k.startswith("void k_"): continue
                a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print("==", wl)
    for k, d in agg.items():
        if not ("k_active" in k or "k_decode" in k): continue
        print(k[:70])
        print("   " + "  ".join("%s %.4g" % (c, v / n) for c, (v, n) in sorted(d.items())))
PY
