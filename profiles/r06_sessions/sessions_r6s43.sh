cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s43
# bench-level A/B on one box, alternating: the library of commit a894067 (row-pointer tables in LDS) against the shipped one
for r in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev; else unset CRTHIP_LIBDIR; fi
    timeout 600 python bench.py --no-cpu --no-extra > gpurun_out/r6s43/b_${v}_$r.json 2>/dev/null
    python - $v $r <<'PY'
import json, sys
j=json.loads(open("gpurun_out/r6s43/b_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "value", j["value"], "one", j["one_batch_in_flight"]["value"], j["config"]["batches_in_flight_fps"], j["roofline"]["kernel_ms"])
PY
  done
done | tee gpurun_out/r6s43/ab_bench.txt
