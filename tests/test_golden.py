"""The committed golden fixtures (tests/golden/, generated from the real reference by
tests/golden/make_golden.py) against (a) the oracle -- CPU, runs everywhere -- and (b) the HIP path
through the crthip C ABI -- GPU.  This is the pin that survives on machines without /root/reference."""
import json
import os
import sys

import numpy as np
import pytest

import crtref as R

sys.path.insert(0, os.path.join(R.ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402

GOLD = json.load(open(os.path.join(R.ROOT, "tests", "golden", "golden.json")))
IDS = [c["id"] for c in G.CASES]


@pytest.mark.parametrize("cid", IDS)
def test_oracle_reproduces_golden(cid):
    case = [c for c in G.CASES if c["id"] == cid][0]
    want = GOLD["cases"][cid]
    got = []
    full = {}

    def on_step(step, c, analog):
        got.append(G.record(c, analog))
        full[step] = (analog.copy(), c.inp.copy(), c.out.copy())
    G.run_case(R.Oracle(case["sys"]), case, on_step)
    assert got == want
    if case.get("full"):
        z = np.load(os.path.join(R.ROOT, "tests", "golden", case["full"]))
        for step, (a, i, o) in full.items():
            np.testing.assert_array_equal(a, z["analog%d" % step])
            np.testing.assert_array_equal(i, z["inp%d" % step])
            np.testing.assert_array_equal(o, z["out%d" % step])


class _HipLib:
    """crtref-shaped adapter over the batch ABI (n = 1), so G.run_case can drive the HIP path."""

    def __init__(self, name):
        import crtlib
        self.crtlib, self.name = crtlib, name

    def new_crt(self, outw, outh, fmt):
        self.last = _HipCRT(self, outw, outh, fmt)
        return self.last

    def srand(self, seed):
        self.last.g.srand([seed])


class _HipCRT:
    def __init__(self, lib, outw, outh, fmt):
        import torch
        self.torch, self.L = torch, lib.crtlib
        self.g = lib.crtlib.CRT(1, outw, outh, fmt, lib.name, device=0)
        self.kw = {}
        self.s = None
        self.nes = lib.name.startswith("nes")

    def set(self, k, v):
        setattr(self.g, k, v)

    def get(self, k):
        return self.g.get(k)[0]

    def settings(self, img, **kw):
        torch = self.torch
        self.kw.update(kw)
        h = self.kw["h"]
        arr = np.ascontiguousarray(img)
        if self.nes:
            arr = arr.astype(np.int16)
        full = torch.from_numpy(arr).to("cuda:0")[None]
        init = self.s.initialized if self.s is not None else 0
        k = self.kw
        if self.nes:
            self.s = self.L.Settings(full[:, :h], hue=k.get("hue", 0), dot_crawl_offset=k.get("dot_crawl_offset", 0),
                                     xoffset=k.get("xoffset", 0), yoffset=k.get("yoffset", 0))
        else:
            self.s = self.L.Settings(full[:, :h], format=k["format"], raw=k.get("raw", 0), as_color=k.get("as_color", 0),
                                     field=k.get("field", 0), frame=k.get("frame", 0), hue=k.get("hue", 0))
            assert not k.get("do_aberration", 0)
        self.s.initialized = init

    def sget(self, k):
        return getattr(self.s, k)

    def sset(self, k, v):
        setattr(self.s, k, v)

    def modulate(self):
        self.g.modulate(self.s)

    def demodulate(self, noise):
        self.g.demodulate(noise)
        self.g.synchronize()

    @property
    def analog(self):
        return self.g.analog[0, :self.g.input_size].cpu().numpy()

    @property
    def inp(self):
        return self.g.inp[0, :self.g.input_size].cpu().numpy()

    @property
    def out(self):
        return self.g.out[0].reshape(-1).cpu().numpy()

    @property
    def ccf(self):
        return self.g.ccf[0, :(3 if self.nes else 1), :4]


@pytest.mark.gpu
@pytest.mark.parametrize("cid", IDS)
def test_hip_reproduces_golden(cid):
    import __graft_entry__ as g
    g.build()
    case = [c for c in G.CASES if c["id"] == cid][0]
    got = []
    G.run_case(_HipLib(case["sys"]), case, lambda step, c, analog: got.append(G.record(c, analog)))
    assert got == GOLD["cases"][cid]
