"""Multi-GPU host logic of the field-pass path: frames shard by rank, nothing else moves.

The path partitions by field: a batch of independent frames is split into
contiguous blocks, one per rank (= one process per GPU); every rank runs the same launch sequence
on its block.  The ONLY collective on the data path is the broadcast of the settings blob
(`crthip_params`, a few hundred bytes) from rank 0 -- RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests.  Timing uses a barrier and a MAX all-reduce of the elapsed time.

SEQUENCE mode across ranks (last row; extra/video_convert.c:246-277 over one long video): see
`sequence_sharded` -- the one place where the path has a real exchange step: 8 bytes of sync state per rank and round
(all_gather), and one picture per seam (send / recv)."""
import ctypes as C


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` frames owned by `rank` (blocks of ceil(total/world))."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def field_parity(frame_index):
    """Interlaced sequence of extra/video_convert.c:261-267 for absolute frame index k:
    field = k & 1, frame toggles after every even k."""
    return frame_index & 1, ((frame_index + 1) >> 1) & 1


def broadcast_params(p, dist, device):
    """Rank 0's parameter blob wins: every rank ends up with a byte-identical crthip_params."""
    import torch
    raw = bytearray(bytes(p))
    blob = torch.frombuffer(raw, dtype=torch.uint8).clone().to(device)
    dist.broadcast(blob, src=0)
    data = bytes(blob.cpu().numpy().tobytes())
    C.memmove(C.byref(p), data, C.sizeof(p))
    return p


def max_over_ranks(value, dist, device):
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sequence_sharded(engine, dist, rank, world, n_total, hsync0, vsync0, rn0, out_init, blend, device, pic_shape):
    """One video of n_total consecutive fields cut into contiguous blocks over the ranks, sequential semantics kept.

    `engine` holds this rank's block [first, first + n) (first, n = shard_range) and offers the phases of
    crthip_sequence (crtlib.CRT has them; the CPU tests plug in an oracle-backed stand-in):
        seq_encode(first_index, rn0)            rn in closed form, all fields encoded
        seq_sync(hsync_in, vsync_in) -> (h, v)  the sync chain from the incoming pair; may be called again
        seq_decode()                            all fields decoded
        seq_weave(out_init, patch_only)         pictures made sequential given the buffer before the block's first field
        last_picture() -> tensor                the output buffer after the block's last field
    What crosses the seams (crt_core.c:379-396, 437-450; :431, :608, :662):
      * hsync / vsync: every rank runs its chain from a guess (the set's state before field 0), the finals are
        all_gathered, a rank whose incoming pair changed re-runs; after round j the first j ranks are final, so at most
        `world` rounds -- two in practice, because a field's final state hardly depends on its initial one;
      * the output picture: rank r's last picture goes to rank r + 1 (send / recv), which takes from it only the rows none
        of its own fields wrote (blend: the recurrence over the fields continues down this chain, rank after rank).
    Returns the number of exchange rounds.  Ranks with an empty block take part in the collectives and pass state on."""
    import torch
    if getattr(getattr(engine, "crt", None), "base_system", "") == "vhs" and world > 1:
        raise NotImplementedError("a VHS video shares one rand() stream: cut it over GPUs with crthip_node_sequence (libcrthip_node.so)")
    first, hi = shard_range(n_total, rank, world)
    n = hi - first
    if n > 0:
        engine.seq_encode(first, rn0)
    h_in, v_in = hsync0, vsync0
    h_out, v_out = h_in, v_in
    rounds, dirty = 0, True
    while True:
        rounds += 1
        if dirty and n > 0:
            h_out, v_out = engine.seq_sync(h_in, v_in)
        elif n == 0:
            h_out, v_out = h_in, v_in
        mine = torch.tensor([h_out, v_out], dtype=torch.int32, device=device)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        if dist is not None:
            dist.all_gather(allv, mine)
        else:
            allv = [mine]
        dirty = False
        if rank > 0:
            ph, pv = (int(x) for x in allv[rank - 1].cpu().tolist())
            if (ph, pv) != (h_in, v_in):
                h_in, v_in, dirty = ph, pv, True
        flag = torch.tensor([1 if dirty else 0], dtype=torch.int32, device=device)
        if dist is not None:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 0 or rounds > world + 1:
            break
    if n > 0:
        engine.seq_decode()
        if not blend:
            engine.seq_weave(out_init if rank == 0 else None, False)      # placeholder init: patched below
    # the picture chain: rank r receives its predecessor's last picture, finishes, passes its own last picture on
    prev_pic = out_init
    if rank > 0:
        prev_pic = torch.zeros(tuple(pic_shape), dtype=torch.uint8, device=device)      # (outh, outw, bpp)
        dist.recv(prev_pic, src=rank - 1)
    if n > 0:
        if blend:
            engine.seq_weave(prev_pic, False)
        elif rank > 0:
            engine.seq_weave(prev_pic, True)
    if rank + 1 < world:
        pic = engine.last_picture() if n > 0 else prev_pic
        if pic is None:                                    # rank 0 without an initial picture and without fields
            raise ValueError("an empty block needs out_init to pass on")
        dist.send(pic.contiguous(), dst=rank + 1)
    return rounds


class CrtSequenceEngine:
    """The phases of crthip_sequence of one crtlib.CRT (this rank's block of the video) behind the interface
    sequence_sharded drives."""

    def __init__(self, crt, settings, noise):
        self.crt, self.s, self.noise = crt, settings, noise

    def seq_encode(self, first_index, rn0):
        self.crt.seq_encode(self.s, self.noise, first_index, rn0)

    def seq_sync(self, hsync_in, vsync_in):
        return self.crt.seq_sync(hsync_in, vsync_in)

    def seq_decode(self):
        self.crt.seq_decode()

    def seq_weave(self, out_init, patch_only):
        self.crt.seq_weave(out_init, patch_only)

    def last_picture(self):
        return self.crt.last_picture()
