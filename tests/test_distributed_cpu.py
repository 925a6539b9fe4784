"""world_size-2 gloo test of the multi-GPU host path (ntsc-crt_amd/shard.py + what bench.py does):
settings-blob broadcast, contiguous frame sharding, max-over-ranks timing, and -- with the oracle as
the stand-in compute -- that sharded processing reproduces the unsharded result (frames are
independent given their own state)."""
import os
import socket

import numpy as np
import pytest

import crtref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _process_frames(lo, hi, w, h, outw, outh, noise):
    """the oracle over absolute frame indices [lo, hi): one independent CRT per frame"""
    orc = R.Oracle("ntsc")
    import shard
    outs = []
    for k in range(lo, hi):
        c = orc.new_crt(outw, outh, R.FMT_BGRA)
        c.set("scanlines", 1)
        field, frame = shard.field_parity(k)
        c.settings(R.synth_image(w, h, 4, 12345 + k), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=field, frame=frame)
        c.modulate()
        c.demodulate(noise)
        outs.append(R.fnv1a32(c.out[::97]))
    return outs


def _worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, os.path.join(R.ROOT, "ntsc-crt_amd"))
    import torch
    import torch.distributed as dist
    import crtlib
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # rank 1 starts with DIFFERENT settings; after the broadcast both must hold rank 0's blob
    p = crtlib.make_params("ntsc", w=64, h=48, outw=64, outh=240, noise=24 if rank == 0 else 3,
                           scanlines=1 if rank == 0 else 0, hue=0 if rank == 0 else 77)
    shard.broadcast_params(p, dist, dev)
    lo, hi = shard.shard_range(total, rank, world)
    hashes = _process_frames(lo, hi, 64, 48, 64, 240, p.noise)
    tmax = shard.max_over_ranks(1.0 + rank, dist, dev)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, hashes, bytes(p), tmax))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition_the_batch():
    import sys
    sys.path.insert(0, os.path.join(R.ROOT, "ntsc-crt_amd"))
    import shard
    for total in (0, 1, 7, 8, 9, 512, 513):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = shard.shard_range(total, r, world)
                assert 0 <= lo <= hi <= total
                covered += list(range(lo, hi))
            assert covered == list(range(total))
    assert [shard.field_parity(k) for k in range(5)] == [(0, 0), (1, 1), (0, 1), (1, 0), (0, 0)]


def test_two_rank_gloo_run_matches_single_process():
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    g.build()
    total, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # identical blob on both ranks = rank 0's
    assert gathered[0][3] == gathered[1][3]
    import sys
    sys.path.insert(0, os.path.join(R.ROOT, "ntsc-crt_amd"))
    import crtlib
    p0 = crtlib.make_params("ntsc", w=64, h=48, outw=64, outh=240, noise=24, scanlines=1, hue=0)
    assert gathered[1][3] == bytes(p0)
    # max-over-ranks timing
    assert gathered[0][4] == gathered[1][4] == 2.0
    # shards are contiguous, disjoint, complete and reproduce the unsharded result
    assert (gathered[0][0], gathered[1][1]) == (0, total) and gathered[0][1] == gathered[1][0]
    assert gathered[0][2] + gathered[1][2] == _process_frames(0, total, 64, 48, 64, 240, 24)
