#!/bin/bash
# round 4 session 37: memory-side stall counters of the 1080p kernels (TLB, DRAM credit stalls, queue levels); PMC only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s37; mkdir -p $O
export TMPDIR=/tmp
i=0
A="--width 1920 --height 1080 --noise 0 --batch 2048"
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum" \
           "TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --streams 1 $A > $GRAFT_REPO_ROOT/$O/p$i.log 2>&1; tail -2 $GRAFT_REPO_ROOT/$O/p$i.log | cut -c1-200 )
done
python3 - $O <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    if not ("k_active<" in k or "k_decode_wide<SysNTSC, 0" in k): continue
    print(k[:70])
    for c, (v, n) in sorted(d.items()): print("   %-40s %.5g" % (c, v / n))
PY
