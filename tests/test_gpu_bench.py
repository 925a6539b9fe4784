"""bench.py's own step loop on the GPU (parity hole (b)): `value` is the best of 1, 2 or 3 batches in
flight -- consecutive steps on S contexts with their own streams.  Nothing asserted so far that S = 3 produces the same
bytes as S = 1; this drives bench.Batches exactly like run_workload does and compares every output byte and every
field's state, and one field of the result with the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))

import crtref as R

pytestmark = pytest.mark.gpu

WORKLOADS = {
    "ntsc": dict(name="t", system="ntsc", w=640, h=480, outw=640, outh=480, batch=300, noise=24, scanlines=1, unique=5),
    "1080p": dict(name="t", system="ntsc", w=1920, h=1080, outw=1920, outh=1080, batch=40, noise=0, scanlines=1, unique=3),
    "vhs": dict(name="t", system="vhs", w=832, h=624, outw=832, outh=624, batch=70, noise=12, scanlines=1, unique=4),
    "nesp0": dict(name="t", system="nesp0", w=256, h=240, outw=640, outh=480, batch=130, noise=12, scanlines=1, unique=6),
}


@pytest.mark.parametrize("wl", sorted(WORKLOADS))
def test_batches_in_flight_give_the_same_bytes_as_one(wl):
    import torch
    import bench
    import crtlib
    import shard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = WORKLOADS[wl]
    per_ctx = 2                                            # steps per context (the second one starts from carried state)

    def run(S):
        B = bench.Batches(torch, crtlib, shard, None, dev, 0, 1, 0, dict(w), S)
        torch.cuda.synchronize(dev)
        for k in range(per_ctx * S):
            B.step(k, S)
        B.barrier()
        res = [(c.out.clone(), c.state.clone()) for c in B.crts]
        imgs = B.images[:2].cpu().numpy()
        B.close()
        return res, imgs
    one, imgs = run(1)
    three, imgs3 = run(3)
    assert np.array_equal(imgs, imgs3)                     # same synthetic input (seeded generator)
    for ci, (out, st) in enumerate(three):
        assert torch.equal(out, one[0][0]), "%s: context %d of 3 in flight differs from one batch in flight" % (wl, ci)
        assert torch.equal(st, one[0][1]), "%s: state of context %d" % (wl, ci)
    # ... and what they all agree on is the reference's result: field 1 of the batch after its two steps, against the oracle
    name = w["system"]
    orc = R.Oracle(name)
    c = orc.new_crt(w["outw"], w["outh"], R.FMT_BGRA)
    c.set("scanlines", 1)
    k = 1
    if name == "nesp0":
        pad = np.concatenate([imgs[k], imgs[k][-1:]]).astype(np.uint16)
        c.settings(pad, w=w["w"], h=w["h"], dot_crawl_offset=k % 3, hue=0)
    else:
        fld, frm = shard.field_parity(k)
        c.settings(np.concatenate([imgs[k], imgs[k][-1:]]), format=R.FMT_BGRA, w=w["w"], h=w["h"], as_color=1, hue=0, field=fld, frame=frm,
                   **({"do_aberration": 0} if name == "vhs" else {}))
    if name == "vhs":
        import ctypes as C
        C.CDLL(None).srand(1 + k)
    for step in range(per_ctx):
        c.analog[:] = 0                                    # batch semantics: every field-pass starts from a clean analog[]
        if name == "nesp0":
            c.sset("field_initialized", 0)
        c.modulate()
        c.demodulate(w["noise"])
        if name != "nesp0":                                # bench.Batches.one_step: the interlaced sequence of video_convert.c:261-267
            c.sset("field", c.sget("field") ^ 1)
            if step % 2 == 0:
                c.sset("frame", c.sget("frame") ^ 1)
    np.testing.assert_array_equal(one[0][0][k].cpu().numpy().reshape(-1), c.out, err_msg="%s: field %d vs the oracle" % (wl, k))
