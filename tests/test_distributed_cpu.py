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


# ---------------------------------------------------------------------------------------------------------------------
# SEQUENCE mode across ranks (missing #1): shard.sequence_sharded with the oracle as the compute
# ---------------------------------------------------------------------------------------------------------------------
SEQ = dict(w=64, h=48, outw=96, outh=240, noise=120, hsync0=7, vsync0=2, rn0=194)


def _seq_frames(lo, hi):
    return [R.synth_image(SEQ["w"], SEQ["h"], 4, 4000 + k, "random" if k % 3 else "bars") for k in range(lo, hi)]


def _lcg_jump(rn, steps):
    """rn after `steps` steps of crt_core.c:359 (affine map, square and multiply)"""
    m, a, pm, pa = 1, 0, 214019, 140327895
    while steps:
        if steps & 1:
            m, a = (pm * m) & 0xffffffff, (pm * a + pa) & 0xffffffff
        pa = (pm * pa + pa) & 0xffffffff
        pm = (pm * pm) & 0xffffffff
        steps >>= 1
    v = (m * (rn & 0xffffffff) + a) & 0xffffffff
    return v - (1 << 32) if v >= (1 << 31) else v


class OracleSequenceEngine:
    """Stand-in for crtlib.CRT's sequence phases in the CPU tests: this rank's block of the video processed by the
    oracle, one field after the other, from whatever incoming state / picture the protocol hands it."""

    def __init__(self, lo, hi, blend, scanlines):
        import shard
        self.shard = shard
        self.lo, self.hi, self.blend, self.scanlines = lo, hi, blend, scanlines
        self.orc = R.Oracle("ntsc")
        self.frames = _seq_frames(lo, hi)
        self.images = None
        self.sync_runs = 0

    def _run(self, h_in, v_in, init):
        c = self.orc.new_crt(SEQ["outw"], SEQ["outh"], R.FMT_BGRA)
        c.set("scanlines", self.scanlines)
        c.set("blend", self.blend)
        c.set("hsync", h_in)
        c.set("vsync", v_in)
        c.set("rn", self.rn_first)
        if init is not None:
            c.out[:] = init.reshape(-1)
        imgs = []
        for k, img in zip(range(self.lo, self.hi), self.frames):
            field, frame = self.shard.field_parity(k)
            c.settings(np.concatenate([img, img[-1:]]), format=R.FMT_BGRA, w=SEQ["w"], h=SEQ["h"], as_color=1, field=field, frame=frame)
            c.modulate()
            c.demodulate(SEQ["noise"])
            imgs.append(c.out.copy())
        return imgs, c.get("hsync"), c.get("vsync")

    def seq_encode(self, first_index, rn0):
        assert first_index == self.lo
        self.rn_first = _lcg_jump(rn0, first_index * self.orc.input_size)

    def seq_sync(self, h_in, v_in):
        self.sync_runs += 1
        self.h_in, self.v_in = h_in, v_in
        _, h, v = self._run(h_in, v_in, None)
        return h, v

    def seq_decode(self):
        pass

    def seq_weave(self, out_init, patch_only):
        init = None if out_init is None else out_init.numpy()
        self.images, _, _ = self._run(self.h_in, self.v_in, init)

    def last_picture(self):
        import torch
        return torch.from_numpy(self.images[-1].reshape(SEQ["outh"], SEQ["outw"], 4).copy())


def _seq_worker(rank, world, port, total, blend, scanlines, q):
    import sys
    sys.path.insert(0, os.path.join(R.ROOT, "ntsc-crt_amd"))
    import torch
    import torch.distributed as dist
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(total, rank, world)
    eng = OracleSequenceEngine(lo, hi, blend, scanlines)
    init = torch.from_numpy(R.lcg_bytes(SEQ["outh"] * SEQ["outw"] * 4, 5).reshape(SEQ["outh"], SEQ["outw"], 4).copy())
    rounds = shard.sequence_sharded(eng, dist, rank, world, total, SEQ["hsync0"], SEQ["vsync0"], SEQ["rn0"],
                                    init if rank == 0 else None, blend, torch.device("cpu"), (SEQ["outh"], SEQ["outw"], 4))
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, [R.fnv1a32(im[::53]) for im in (eng.images or [])], rounds, eng.sync_runs))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,blend,scanlines", [(2, 7, 0, 1), (3, 8, 0, 0), (2, 5, 1, 0), (3, 2, 0, 1)])
def test_sequence_mode_across_ranks_equals_one_sequential_run(world, total, blend, scanlines):
    """One video cut over `world` gloo ranks: every rank's pictures must be what ONE oracle produces running all the
    fields one after the other (hsync / vsync / rn and the output buffer carried across the seams).  Heavy noise and an
    odd starting state make the sync state travel; (3, 2): one rank has an empty block."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, total, blend, scanlines, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the sequential truth
    one = OracleSequenceEngine(0, total, blend, scanlines)
    one.seq_encode(0, SEQ["rn0"])
    init = R.lcg_bytes(SEQ["outh"] * SEQ["outw"] * 4, 5)
    want, _, _ = one._run(SEQ["hsync0"], SEQ["vsync0"], init)
    got = []
    for lo, hi, hashes, rounds, sync_runs in gathered:
        assert len(hashes) == hi - lo
        got += hashes
        assert 1 <= rounds <= world + 1
    assert got == [R.fnv1a32(im[::53]) for im in want]
    # ranks behind the first one had to re-run their chain at least once (their first guess was the set's initial state)
    assert any(g[4] >= 2 for g in gathered[1:] if g[1] > g[0])
