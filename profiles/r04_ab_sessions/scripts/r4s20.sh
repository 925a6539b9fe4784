#!/bin/bash
# round 4 session 20: bloom graph capture with the clear kernel; full GPU suite; encoder fast path (four samples of a tile without a branch) vs lib_prev
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r4s20; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 200 python tools/debug/graph_bloom.py ntscbloom 1 graph >> $O/graph.txt 2>&1; echo "rc=$?" >> $O/graph.txt; done
grep -v "^  File\|^$\|Extension modules\|amdgpu.ids" $O/graph.txt | tail -24 | cut -c1-200
( timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log | cut -c1-220
run() { lab=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --streams 1 --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lab  %.4f ms/step  %.0f fps  frac %.4f kernel_ms %s' % (d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['roofline']['kernel_ms']))
except Exception as e: print('$lab  FAILED', e)" >> $O/ab.txt
}
W="--width 1920 --height 1080 --noise 0"
for i in 1 2; do
run "1080p2048 sample-at-a-time encoder" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- $W --batch 2048
run "1080p2048 four-sample fast path" X=1 -- $W --batch 2048
run "headline sample-at-a-time encoder" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev --
run "headline four-sample fast path" X=1 --
done
run "vhs sample-at-a-time" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- --system vhs --width 832 --height 624 --noise 12 --batch 2048
run "vhs fast path" X=1 -- --system vhs --width 832 --height 624 --noise 12 --batch 2048
run "pv1k sample-at-a-time" CRTHIP_LIBDIR=$PWD/ntsc-crt_amd/lib_prev -- --system pv1k
run "pv1k fast path" X=1 -- --system pv1k
cat $O/ab.txt
