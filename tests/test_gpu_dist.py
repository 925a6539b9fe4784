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


def test_bench_multi_gpu_workloads_over_rccl():
    """what `bench.py --gpus N` adds to the headline on every rank -- 1080p weak and BASELINE configs[2] sharded over the ranks -- run
    here with the one rank there is, every collective through RCCL: the final line carries the north star's row"""
    j = _bench([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--multi-gpu-workloads", "--no-cpu", "--no-extra", "--steps", "4",
                "--warmup", "1"])
    ns = j["north_star"]
    assert ns["n_gpus"] == 1 and j["world_size_seen_by_rccl"] == 1
    assert ns["fps_640"] > 1e5 and ns["fps_1080p_weak"] > 1e5 and 0.3 < ns["frac_1080p_weak"] < 1.0
    assert ns["configs2_frames"] == 512 and ns["configs2_frames_per_gpu"] == 512 and 0.3 < ns["configs2_ms"] < 5.0
    assert abs(ns["configs2_fps"] - 512e3 / ns["configs2_ms"]) < 1e-2 * ns["configs2_fps"]


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
    assert r.stdout.count("0 pictures, 0 states differ") == 15, r.stdout      # 5 modes (NTSC x 3, VHS x 2) x 3 shard layouts
    assert "several exchange rounds" in r.stdout, "the heavy-noise video should need more than one exchange round somewhere:\n" + r.stdout


class _LocalComm:
    """The subset of torch.distributed that shard.sequence_sharded uses, between THREADS of one process: lets several
    crtlib.CRT objects on the one GPU of a gpurun box play the ranks of a multi-GPU run."""

    class ReduceOp:
        MAX = "max"

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.mail = {}
        self.cv = threading.Condition()

    class _View:
        def __init__(self, comm, rank):
            self.c, self.rank, self.ReduceOp = comm, rank, _LocalComm.ReduceOp

        def all_gather(self, out, t):
            self.c.slots[self.rank] = t.clone()
            self.c.barrier.wait()
            for r in range(self.c.world):
                out[r].copy_(self.c.slots[r])
            self.c.barrier.wait()

        def all_reduce(self, t, op=None):
            self.c.slots[self.rank] = t.clone()
            self.c.barrier.wait()
            m = max(int(x.item()) for x in self.c.slots)
            self.c.barrier.wait()
            t.fill_(m)

        def send(self, t, dst):
            with self.c.cv:
                self.c.mail[(self.rank, dst)] = t.clone()
                self.c.cv.notify_all()

        def recv(self, t, src):
            with self.c.cv:
                self.c.cv.wait_for(lambda: (src, self.rank) in self.c.mail, timeout=120)
                t.copy_(self.c.mail.pop((src, self.rank)))

    def view(self, rank):
        return _LocalComm._View(self, rank)


@pytest.mark.parametrize("world,total,blend,scanlines", [(2, 9, 0, 1), (3, 10, 0, 0), (2, 7, 1, 0), (4, 5, 0, 1)])
def test_sequence_sharded_over_several_contexts_on_one_gpu(world, total, blend, scanlines):
    """shard.sequence_sharded driving the HIP phases (crthip_seq_*): one video cut over `world` crtlib.CRT objects -- the
    ranks of a multi-GPU run, here threads on one GPU -- must give exactly what crthip_sequence gives on one context
    (which tests/test_gpu_parity.py holds against the oracle): pictures, hsync / vsync / rn of every field.
    Heavy noise + an odd starting state: the sync state really travels across the seams."""
    import threading
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import crtlib
    import crtref as R
    import shard
    w, h, outw, outh, noise = 320, 240, 400, 300, 120
    frames = np.stack([R.synth_image(w, h, 4, 900 + k, "random" if k % 3 else "bars") for k in range(total)])
    init = torch.from_numpy(R.lcg_bytes(outh * outw * 4, 5).reshape(outh, outw, 4).copy()).to("cuda:0")

    def settings(lo, hi):
        full = torch.zeros((hi - lo, h + 1, w, 4), dtype=torch.uint8, device="cuda:0")
        full[:, :h] = torch.from_numpy(frames[lo:hi]).to("cuda:0")
        par = [shard.field_parity(k) for k in range(lo, hi)]
        return crtlib.Settings(full[:, :h], format=crtlib.FMT_BGRA, field=[a for a, _ in par], frame=[b for _, b in par])

    one = crtlib.CRT(total, outw, outh, crtlib.FMT_BGRA, "ntsc", device=0)
    one.scanlines, one.blend = scanlines, blend
    one.state[0, crtlib.ST_HSYNC] = 7
    one.state[0, crtlib.ST_VSYNC] = 2
    one.sequence(settings(0, total), noise, out_init=init)
    one.synchronize()
    want = one.out.cpu().numpy()
    want_state = [one.get(f) for f in ("hsync", "vsync", "rn")]

    comm = _LocalComm(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            lo, hi = shard.shard_range(total, rank, world)
            crt = crtlib.CRT(max(hi - lo, 1), outw, outh, crtlib.FMT_BGRA, "ntsc", device=0)
            crt.scanlines, crt.blend = scanlines, blend
            eng = shard.CrtSequenceEngine(crt, settings(lo, max(hi, lo + 1)) if hi > lo else None, noise)
            rounds = shard.sequence_sharded(eng, comm.view(rank), rank, world, total, 7, 2, 194,
                                            init if rank == 0 else None, blend, torch.device("cuda:0"), (outh, outw, 4))
            crt.synchronize()
            results[rank] = (lo, hi, crt.out.cpu().numpy()[:hi - lo], [crt.get(f)[:hi - lo] for f in ("hsync", "vsync", "rn")], rounds)
            crt.close()
        except Exception as e:                                   # pragma: no cover
            errors.append((rank, repr(e)))
            try:
                comm.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for lo, hi, out, st, rounds in results:
        assert 1 <= rounds <= world + 1
        np.testing.assert_array_equal(out, want[lo:hi], err_msg="fields %d..%d" % (lo, hi))
        for got, full in zip(st, want_state):
            assert got == full[lo:hi]
    one.close()


def test_bench_sequence_mode_over_rccl():
    j = _bench([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--sequence", "--no-cpu", "--no-extra", "--steps", "3",
                "--warmup", "1", "--batch", "256"])
    assert "shard.sequence_sharded" in j["config"]["mode"] and j["collectives"]["initialized"] is True
    assert j["value"] > 1e4


@pytest.mark.gpu
def test_node_bench_runs_batches_in_flight_through_the_c_abi():
    """tools/node_bench.c: S shards on one device = S independent batches in flight, plain C over libcrthip_node.so
    (DESIGN.md 6a).  Here only: it builds, runs and reports a rate for 1, 2 and 3 shards."""
    exe = os.path.join(ROOT, "ntsc-crt_amd", "lib", "node_bench")
    assert os.path.exists(exe), "ntsc-crt_amd/lib/node_bench missing: run make -C ntsc-crt_amd"
    for shards in (1, 2, 3):
        r = subprocess.run([exe, str(shards), "96", "6"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "frames/sec" in r.stdout and "%d shard(s)" % shards in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_tunes_the_batches_in_flight_and_reports_every_candidate():
    """bench.py: the same K steps timed with 1, 2 and 3 independent batches in flight, five times each; S is chosen on the medians
    (all of them in the JSON, the one-batch figure separately), then the chosen S is timed five more times and
    `value` is the median of THOSE, with min / median / max as `value_spread`"""
    import tempfile
    full = os.path.join(tempfile.mkdtemp(prefix="benchfull_"), "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "96", "--steps", "6", "--warmup", "1", "--no-cpu", "--no-extra",
                        "--full-json", full], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 4096                                   # the driver keeps 8 KB of output
    c = json.loads(line)                                      # the compact record on stdout ...
    j = json.load(open(full))                                 # ... and the complete one in the file
    tuning = j["config"]["batches_in_flight_tuning"]
    assert sorted(tuning) == ["1", "2", "3"] and j["config"]["batches_in_flight"] in (1, 2, 3)
    best = max(tuning, key=lambda s_: tuning[s_]["value"])
    assert j["config"]["batches_in_flight"] == int(best), "S is chosen on the medians of the candidates' samples"
    assert all(len(t_["samples_ms_per_step"]) == 5 for t_ in tuning.values())
    sp = j["value_spread"]
    assert sp["reps"] == 5 and sp["min"] <= sp["median"] <= sp["max"] and abs(j["value"] - sp["median"]) < 1e-6 * sp["median"]
    assert abs(j["ms_per_step"] * j["value"] - 96 * 1e3) < 1e-3 * 96 * 1e3          # ms per step x frames per second = frames per step
    assert c["value_spread"]["reps"] == 5 and abs(c["value_spread"]["median"] - j["value"]) < 1e-5 * j["value"]
    assert abs(j["one_batch_in_flight"]["value"] - tuning["1"]["value"]) < 1e-6 * tuning["1"]["value"]
    assert j["steps"] == 6 and j["roofline"]["kernel_ms"]["decode"] > 0
    # the compact line says the same at its precision
    assert abs(c["value"] - j["value"]) < 1e-5 * j["value"] and sorted(c["config"]["batches_in_flight_fps"]) == ["1", "2", "3"]
    assert abs(c["one_batch_in_flight"]["value"] - tuning["1"]["value"]) < 1e-5 * tuning["1"]["value"]
    assert c["roofline"]["kernel_ms"]["decode"] > 0 and c["roofline"]["frac"] > 0
    # pinned
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "96", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-extra", "--streams", "1",
                        "--full-json", full], capture_output=True, text=True, timeout=600, cwd=ROOT)
    j = json.load(open(full))
    assert j["config"]["batches_in_flight"] == 1 and sorted(j["config"]["batches_in_flight_tuning"]) == ["1"] and "one_batch_in_flight" not in j
