"""RCCL on the hardware there is: gpurun boxes have ONE MI355X, so the multi-GPU path's
collectives -- init_process_group("nccl"), the device-tensor broadcast of the settings blob, the all_gather of its CRC, the
MAX all-reduce of the elapsed time, the barriers -- are run with a one-rank process group; and the C-level node entry
(include/crt_hip_node.h) with as many shards as the box allows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("{"), "the JSON record must be the last line of stdout: %r" % last[:200]
    return json.loads(last)


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_force_dist_runs_every_collective_over_rccl(launcher):
    args = ["bench.py", "--gpus", "1", "--force-dist", "--no-cpu", "--no-extra", "--steps", "3", "--warmup", "1", "--batch", "512"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29611"] + args
    else:
        cmd = [sys.executable] + args
    j = _bench(cmd)
    assert j["n_gpus"] == 1 and j["world_size_seen_by_rccl"] == 1
    assert j["collectives"]["backend"] == "nccl" and j["collectives"]["initialized"] is True
    crcs = j["settings_blob_crc32_per_rank"]
    assert len(crcs) == 1
    # the blob every rank ended up with (device tensor through RCCL and back) is the one rank 0 built on the host
    assert crcs[0] == j["settings_blob_crc32_rank0_before_broadcast"]
    assert j["value"] > 1e5


@pytest.mark.parametrize("n_fields", [11, 4])
def test_node_entry_matches_the_single_context_path(n_fields):
    """include/crt_hip_node.h through its C test program (tests/node_probe.c, built by ntsc-crt_amd/Makefile):
    crthip_node_fieldpass / crthip_node_sequence (blend 0 and 1) over 1 shard per device, and over 2 and 3 shards that
    share device 0, byte-identical with crthip_fieldpass / crthip_sequence on one context.  Every layout broadcasts the
    settings blob with ncclBroadcast over its communicator."""
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(ROOT, "ntsc-crt_amd", "lib", "node_probe")
    r = subprocess.run([exe, str(n_fields)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "node_probe ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("0 pictures, 0 states differ") == 9, r.stdout
    assert "several exchange rounds" in r.stdout, "the heavy-noise video should need more than one exchange round somewhere:\n" + r.stdout
