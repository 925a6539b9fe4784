#!/bin/bash
# usage: tools/prof_sq.sh <tag>  -- SQ occupancy / issue / stall counters per kernel (PMC only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu --streams 1 $* > $out/p$i.log 2>&1
done
python3 - $out <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("void k_"): continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %16.0f per launch" % (c, v / n))
PY
