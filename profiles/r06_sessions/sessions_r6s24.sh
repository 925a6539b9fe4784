cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s24
for k in 8 16; do CRTHIP_SYNC_KERNEL=$k timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_fieldpass_parity or wild_sync or full_size" > gpurun_out/r6s24/pytest_fpb$k.log 2>&1; echo "fpb $k pytest rc $?"; tail -2 gpurun_out/r6s24/pytest_fpb$k.log; done
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s24/fpb.txt
for r in 1 2 3; do for k in 0 8 16; do CRTHIP_SYNC_KERNEL=$k python tools/placement_sweep.py --child 0 --batch 4096 --w 640 --h 480 --noise 24 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('640x480 x 4096 sync kernel $k: sync %.4f  fieldpass %.4f  active %.4f decode %.4f' % (j['sync_ms'], j['fieldpass_ms'], j['active_ms'], j['decode_ms']))"; done; done >> gpurun_out/r6s24/fpb.txt
for r in 1 2; do for k in 0 8 16; do CRTHIP_SYNC_KERNEL=$k python tools/placement_sweep.py --child 0 --batch 2048 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1080p x 2048 sync kernel $k: sync %.4f  fieldpass %.4f' % (j['sync_ms'], j['fieldpass_ms']))"; done; done >> gpurun_out/r6s24/fpb.txt
cat gpurun_out/r6s24/fpb.txt
