cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s22
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s22/overlap.txt
for r in 1 2 3; do for ov in 0 2 3 4; do python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams 1 --overlap $ov 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('640x480 x 4096 overlap $ov', j['value'], j['ms_per_step'])"; done; done >> gpurun_out/r6s22/overlap.txt
for r in 1 2; do for ov in 0 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --streams 1 --overlap $ov --width 1920 --height 1080 --noise 0 --batch 2048 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1080p x 2048 overlap $ov', j['value'], j['ms_per_step'])"; done; done >> gpurun_out/r6s22/overlap.txt
cat gpurun_out/r6s22/overlap.txt
