"""Multi-GPU host logic of the field-pass path: frames shard by rank, nothing else moves.

The path partitions by field: a batch of independent frames is split into
contiguous blocks, one per rank (= one process per GPU); every rank runs the same launch sequence
on its block.  The ONLY collective on the data path is the broadcast of the settings blob
(`crthip_params`, a few hundred bytes) from rank 0 -- RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests.  Timing uses a barrier and a MAX all-reduce of the elapsed time."""
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
