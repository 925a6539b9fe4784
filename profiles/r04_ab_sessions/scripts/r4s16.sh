#!/bin/bash
# round 4 session 16: the tests the subset runs left out, with the wide-run decoder in; cache policy of its picture stores
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s16; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py -k "not soak" > $O/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.log )
tail -4 $O/pytest_rest.log | cut -c1-220
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "graph or smoke or capture" > $O/pytest_graph.log 2>&1; echo "pytest rc=$?" >> $O/pytest_graph.log )
tail -4 $O/pytest_graph.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2; do
run "1080p2048 stores nt (as committed)" X=1 -- $W --batch 2048
run "1080p2048 stores plain" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_st1 -- $W --batch 2048
run "1080p2048 stores sc1" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_st2 -- $W --batch 2048
run "1080p2048 stores sc0 sc1" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_st3 -- $W --batch 2048
run "1080p2048 stores sc0 sc1 nt" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_st4 -- $W --batch 2048
run "1080p2048 stores sc0 nt" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_st5 -- $W --batch 2048
done
cat $O/ab.txt
