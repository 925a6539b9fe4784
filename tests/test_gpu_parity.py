"""GPU parity tests proper: the HIP path (through the crthip_* C ABI) against the CPU oracle on
the same seeded inputs.  Bit-exact: this is integer/byte work, tolerance = 0.  Every stage the
reference exposes through struct CRT is compared (analog, inp, ccf, hsync, vsync, rn, out) plus
the internal per-line sync chain."""
import numpy as np
import pytest

import crtref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crtlib():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import __graft_entry__ as g
    g.build()
    import crtlib
    return crtlib


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _padded(imgs):
    """[n,h,w,c] -> device tensor view [n,h,w,c] inside an [n,h+1,w,c] allocation (row h is
    addressable: crt_ntsc.c:263 quirk)."""
    import torch
    n, h = imgs.shape[0], imgs.shape[1]
    full = torch.zeros((n, h + 1) + tuple(imgs.shape[2:]), dtype=torch.uint8, device="cuda:0")
    full[:, :h] = _to_dev(imgs)
    full[:, h] = full[:, h - 1]
    return full[:, :h]


def _oracle_batch(name, n, outw, outh, ofmt, knobs):
    orc = R.Oracle(name)
    crts = []
    for _ in range(n):
        c = orc.new_crt(outw, outh, ofmt)
        for k, v in knobs.items():
            c.set(k, v)
        crts.append(c)
    return orc, crts


CASES = [
    # name, outw, outh, ofmt, w, h, ifmt, noise, settings, knobs
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 0, dict(as_color=1), {}),
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(scanlines=1)),
    ("ntsc", 832, 624, R.FMT_ARGB, 640, 480, R.FMT_ABGR, 40, dict(as_color=1, hue=350), dict(hue=17, saturation=14)),
    ("ntsc", 640, 480, R.FMT_RGB, 320, 200, R.FMT_RGB, 12, dict(as_color=1, hue=30), dict(blend=1)),
    ("ntsc", 1920, 1080, R.FMT_RGBA, 1920, 1080, R.FMT_BGR, 0, dict(as_color=1), dict(scanlines=1)),
    ("ntsc", 1920, 1080, R.FMT_BGRA, 1920, 1080, R.FMT_BGRA, 0, dict(as_color=1), dict(scanlines=0, blend=1)),
    ("ntsc", 753, 240, R.FMT_ABGR, 753, 236, R.FMT_RGBA, 5, dict(as_color=0), dict(brightness=9, contrast=200)),
    ("ntsc", 500, 300, R.FMT_BGR, 200, 100, R.FMT_ARGB, 60, dict(as_color=1, raw=1), dict(black_point=3, white_point=90)),
    ("ntsc", 333, 481, R.FMT_RGB, 100, 300, R.FMT_RGB, 100, dict(as_color=1, raw=1, xoffset=8, yoffset=2), dict(v_fac=10)),
    ("ntsc", 257, 243, R.FMT_BGRA, 800, 600, R.FMT_BGRA, 24, dict(as_color=1, hue=180), dict(scanlines=1, blend=1)),
    # between the envelopes: saturation 40 puts the carrier above tier 0's bound (lines flagged NOT64, tier 1),
    # brightness 5000 lifts the whole batch to tier 1
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(saturation=40)),
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(brightness=5000, contrast=20)),
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 60, dict(as_color=1), dict(brightness=-2600, saturation=24)),
    # either side of tier 0's chroma bound (|wave| <= 65532: I/Q low cascades dropped), with heavy noise
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 110, dict(as_color=1, hue=77), dict(saturation=13)),
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 110, dict(as_color=1, hue=77), dict(saturation=14)),
    # outside the 24-bit multiply envelope: huge saturation (lines flagged CRTHIP_LINE_EXACT by k_sync) ...
    ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 30, dict(as_color=1), dict(saturation=900, contrast=300)),
    # ... and huge brightness / contrast / white point (host-side check picks the exact kernels)
    ("ntsc", 640, 480, R.FMT_RGBA, 640, 480, R.FMT_RGBA, 10, dict(as_color=1), dict(brightness=200000, contrast=9000000, white_point=9000000)),
    ("ntsc", 320, 240, R.FMT_BGRA, 320, 240, R.FMT_BGRA, 50000000, dict(as_color=1), dict(saturation=-70)),
    # standard NTSC built with CRT_CHROMA_PATTERN 0 (HRES 912, vertical chroma, no phase flip)
    ("ntscp0", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1, hue=40), dict(scanlines=1)),
    ("ntscp0", 832, 624, R.FMT_RGB, 320, 240, R.FMT_BGR, 0, dict(as_color=1, raw=1), dict(blend=1)),
    # outh < CRT_LINES: several CRT lines land on one output row; sequential semantics incl. blend chains
    ("ntsc", 640, 200, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(blend=1)),
    ("ntsc", 101, 77, R.FMT_RGB, 64, 48, R.FMT_RGB, 10, dict(as_color=1), dict(scanlines=1)),
    ("ntsc", 64, 120, R.FMT_ABGR, 64, 48, R.FMT_BGRA, 0, dict(as_color=1), dict(blend=1, v_fac=30)),
]


def _run_case(crtlib, case, fused, steps=4, n=3, exact=False, shape=1, sig_tile=0, wide_lpw=0, sig_pad=1, want_padded=None):
    """shape: 1 = lane-per-scanline kernels (the throughput shape; forced, because small batches would otherwise
    pick the other one), 2 = scanline-parallel kernels, 0 = the library's own choice;  wide_lpw: 8 / 16 pins the wide-run
    decoder's scanlines per wavefront (0: by batch size, i.e. 8 for the small batches of these tests)"""
    name, outw, outh, ofmt, w, h, ifmt, noise, skw, knobs = case
    bpp = R.bpp4fmt(ifmt)
    imgs = np.stack([R.synth_image(w, h, bpp, 777 + 13 * k, "random" if k % 2 == 0 else "bars") for k in range(n)])
    dimgs = _padded(imgs)
    orc, ocrts = _oracle_batch(name, n, outw, outh, ofmt, knobs)
    g = crtlib.CRT(n, outw, outh, ofmt, name[:4] if name.startswith("ntscfir") else name, device=0)
    g.eq_fir = R.EQ_KERNEL.get(name, 0)              # "ntscfir7": the FIR decoder of a USE_CONVOLUTION build
    g.set_exact(exact)
    g.set_shape(shape)
    g.set_signal_tile(sig_tile)
    g.set_wide_lpw(wide_lpw)
    g.set_signal_layout(sig_pad)                     # fused path: padded signal lines (round 6) or the reference's flat ones
    for k, v in knobs.items():
        setattr(g, k, v)
    fields = [k & 1 for k in range(n)]
    frames = [(k >> 1) & 1 for k in range(n)]
    dot_crawl = orc.system in R.DOT_CRAWL_SYSTEMS      # SNES, template, PV-1000: NTSC_SETTINGS.dot_crawl_offset
    dcos = [(2 * k + 1) % 6 if name.startswith(("pv1k", "temp")) else (k + 1) % 3 for k in range(n)]
    s = crtlib.Settings(dimgs, format=ifmt, field=list(fields), frame=list(frames),
                        dot_crawl_offset=list(dcos) if dot_crawl else 0, **skw)
    for k, c in enumerate(ocrts):
        pad = np.concatenate([imgs[k], imgs[k][-1:]], axis=0)
        c.settings(pad, format=ifmt, w=w, h=h, field=fields[k], frame=frames[k], **skw)
        if dot_crawl:
            c.sset("dot_crawl_offset", dcos[k])
    undefined = [False] * n
    for step in range(steps):
        if fused:
            g.fieldpass(s, noise)
        else:
            g.modulate(s)
            analog = g.analog.cpu().numpy()
            ccf_mod = g.ccf.copy()
            g.demodulate(noise)
        g.synchronize()
        gout = g.out.cpu().numpy()
        gst = {f: g.get(f) for f in ("hsync", "vsync", "rn")}
        gccf = g.ccf
        if not fused:
            ginp = g.inp.cpu().numpy()
            glines = g.line_table.cpu().numpy()
        elif R.bpp4fmt(ifmt) and R.bpp4fmt(ofmt):
            # the fused path's own signal, repacked into the reference's layout (crthip_fieldpass_signal): inp[] is compared for
            # the fused path too since round 6 -- whichever layout the workspace keeps it in
            fsig, was_padded = g.fieldpass_signal()
            fsig = fsig.cpu().numpy()
            if want_padded is not None:
                assert was_padded == want_padded, "signal layout of the fused path: padded=%s" % was_padded
        else:
            fsig = None
        for k, c in enumerate(ocrts):
            what = "%s fused=%s step %d field %d" % (name, fused, step, k)
            c.modulate()
            if not fused and not undefined[k]:
                np.testing.assert_array_equal(analog[k, :orc.input_size], c.analog, err_msg=what + " analog")
                np.testing.assert_array_equal(ccf_mod[k, :orc.vper, :orc.ccs], c.ccf, err_msg=what + " ccf after modulate")
            hs_before = c.get("hsync")
            c.demodulate(noise, trace=True)
            # a sync state far from lock on the field's last analog line makes the REFERENCE read past inp[] + 16 (heavy noise):
            # undefined there, different bytes here -- the field is out of the comparison from then on
            undefined[k] = undefined[k] or R.reads_past_inp(orc, c.trace, c.get("vsync"), hs_before)
            if undefined[k]:
                continue
            if fused and fsig is not None:
                np.testing.assert_array_equal(fsig[k, :orc.input_size], c.inp, err_msg=what + " inp of the fused path")
            if not fused:
                np.testing.assert_array_equal(ginp[k, :orc.input_size], c.inp, err_msg=what + " inp")
                tr = c.trace
                valid = tr[:, 0] == 1
                np.testing.assert_array_equal((glines[k][:, 4] & 0xffff) > 0, valid, err_msg=what + " valid lines")
                np.testing.assert_array_equal(glines[k][valid][:, [0, 1, 2, 3, 5, 6, 7]], tr[valid][:, [1, 2, 3, 4, 6, 7, 8]],
                                              err_msg=what + " line table (pos, wave0, wave1, beg, hsync, dx, scanl)")
            for f in ("hsync", "vsync", "rn"):
                assert gst[f][k] == c.get(f), "%s %s: gpu %d oracle %d" % (what, f, gst[f][k], c.get(f))
            np.testing.assert_array_equal(gccf[k, :orc.vper, :orc.ccs], c.ccf, err_msg=what + " ccf")
            np.testing.assert_array_equal(gout[k].reshape(-1), c.out, err_msg=what + " out")
        # next field: interlaced sequence (video_convert.c:259-267)
        fields = [f ^ 1 for f in fields]
        if step % 2 == 0:
            frames = [f ^ 1 for f in frames]
        s.field, s.frame = list(fields), list(frames)
        for k, c in enumerate(ocrts):
            c.sset("field", fields[k])
            c.sset("frame", frames[k])
    g.close()
    assert not all(undefined), "every field fell into the reference's undefined behaviour: the case checks nothing"


@pytest.mark.parametrize("case", range(len(CASES)))
def test_stagewise_parity(crtlib, case):
    _run_case(crtlib, CASES[case], fused=False)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused_fieldpass_parity(crtlib, case):
    _run_case(crtlib, CASES[case], fused=True)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused_fieldpass_flat_signal_layout_parity(crtlib, case):
    """Round 6: the fused path keeps its signal in PADDED lines by default (crt_dev.h, sig_layout; every other fused test runs that).
    The reference's flat layout is still what geometries outside the padded layout's reach, sequence mode and every stage-level
    call use: forced here (crthip_set_signal_layout(ctx, 0)) for the whole table, inp[] compared as well."""
    _run_case(crtlib, CASES[case], fused=True, steps=2, sig_pad=0, want_padded=False)


@pytest.mark.parametrize("sig_tile", [0, 32])
@pytest.mark.parametrize("name,case", [("ntsc", 0), ("ntsc", 1), ("ntsc", 2), ("ntsc", 5), ("ntsc", 9), ("ntsc", 13), ("ntscp0", 18), ("snes", 1), ("temp", 1),
                                       ("pv1k", 1), ("nesrgb", 1), ("ntscbloom", 1), ("pv1kbloom", 1), ("vhslcg", 1), ("ntscnovsync", 1)])
def test_padded_signal_layout_is_what_the_fused_path_runs(crtlib, name, case, sig_tile):
    """... and the default really is the padded layout for every system at its standard geometry (crthip_fieldpass_signal reports
    which layout the workspace held), with inp[] -- repacked to the reference's layout -- equal to the oracle's byte for byte, all
    shapes of the sync chain's windows included (four steps: the sync state moves).  The rand()-noise VHS build and CRT_DO_VSYNC 0
    produce their signal by other kernels and stay flat; so does the VHS build with the LCG noise (one system is one layout)."""
    c = CASES[case] if name.startswith("ntsc") and not name.endswith(("bloom", "novsync")) or name == "ntscp0" else (name,) + F4_CASES[case]
    c = (name,) + tuple(c[1:])
    _run_case(crtlib, c, fused=True, steps=4, sig_tile=sig_tile, want_padded=name not in ("vhslcg", "ntscnovsync"))


def test_padded_signal_layout_falls_back_for_rows_it_cannot_hold(crtlib):
    """An x offset that makes more than 16 samples of the active row run over the line end is legal in the reference (flat index,
    crt_ntsc.c:322: the row continues in the next line's front porch) and outside the padded layout: the flat one takes it.  Up to
    16 samples (xoffset 4: 3 samples, xoffset 16: 15) the padded layout stores the overhang twice -- behind its line and at the
    head of the next (RowTiles::wrap_home)."""
    for xoff, padded in ((24, False), (16, True), (4, True)):
        case = ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 30, dict(as_color=1, xoffset=xoff, yoffset=1), dict(scanlines=1))
        _run_case(crtlib, case, fused=True, steps=2, want_padded=padded)
        _run_case(crtlib, case, fused=True, steps=2, want_padded=padded, shape=2)      # the scanline-parallel encoder's stores


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("taps", [7, 6, 5, 4])
@pytest.mark.parametrize("case", [1, 2, 4, 8])
def test_fir_decoder_parity(crtlib, case, taps, fused):
    """the decoder of a USE_CONVOLUTION build of the reference (crt_core.c:85-147, symmetric FIR
    kernels of 7/6/5/4 taps instead of the 3-band equaliser), selected per launch with CRTHIP_F_EQ_FIR(taps).
    The oracle's FIR mode is pinned against those builds of the reference in tests/test_oracle_vs_ref.py."""
    c = CASES[case]
    _run_case(crtlib, ("ntscfir%d" % taps,) + tuple(c[1:]), fused=fused, steps=2)


@pytest.mark.parametrize("knobs", [dict(saturation=900, contrast=300), dict(brightness=200000, contrast=9000000, white_point=9000000)])
def test_fir_decoder_outside_the_24bit_envelope(crtlib, knobs):
    """huge carrier amplitude (lines flagged by k_hsync) / huge brightness (host-side floor): the exact FIR tier"""
    _run_case(crtlib, ("ntscfir7", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 30, dict(as_color=1), knobs), fused=True, steps=2)


WIDE_CASES = [
    # name, outw, outh, ofmt, w, h, ifmt, noise, settings, knobs -- pictures the 16-scanlines-per-wave decoder takes (crt_decode4.hip)
    ("ntsc", 1920, 1080, R.FMT_BGRA, 1920, 1080, R.FMT_BGRA, 0, dict(as_color=1), dict(scanlines=1)),
    ("ntsc", 1920, 1080, R.FMT_ARGB, 640, 480, R.FMT_RGB, 24, dict(as_color=1, hue=40), dict(scanlines=0, hue=-20, saturation=14)),
    ("ntsc", 2047, 777, R.FMT_ABGR, 333, 100, R.FMT_ABGR, 60, dict(as_color=0), dict(scanlines=1, v_fac=100, brightness=12, contrast=250)),
    ("ntsc", 1700, 300, R.FMT_RGBA, 1281, 601, R.FMT_BGRA, 12, dict(as_color=1, raw=1), dict(scanlines=1, black_point=-5, white_point=120)),
    ("snes", 1920, 1080, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(scanlines=1)),
    ("temp", 1664, 1248, R.FMT_BGRA, 832, 624, R.FMT_BGRA, 0, dict(as_color=1), dict(scanlines=1)),
    ("ntscp0", 1920, 1200, R.FMT_BGRA, 1920, 1080, R.FMT_RGBA, 24, dict(as_color=1), dict(scanlines=1)),
    # strong carriers: some 64-scanline groups leave tiers 0 / 1 and stay with the lane-per-scanline kernel, their neighbours do not
    ("ntsc", 1920, 1080, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(scanlines=1, saturation=25)),
    ("ntsc", 1920, 1080, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 40, dict(as_color=1), dict(scanlines=1, saturation=60, contrast=300)),
]


@pytest.mark.parametrize("lpw", [8, 16])
@pytest.mark.parametrize("case", range(len(WIDE_CASES)))
def test_wide_decoder_parity(crtlib, case, lpw):
    """crt_decode4.hip (round 4): 16 scanlines per wave, a scanline's four filter cascades on four lanes, pixels lane-per-pixel in
    1 KB row runs -- fused, three steps, every field against the oracle (out, state, ccf).  Both instantiations pinned
    (crthip_set_wide_lpw): k_decode_wide<S, 16> is what the 1080p bench batches run, <S, 8> what small batches pick by themselves."""
    _run_case(crtlib, WIDE_CASES[case], fused=True, steps=3, n=5, wide_lpw=lpw)


@pytest.mark.parametrize("lpw", [8, 16])
@pytest.mark.parametrize("case", [0, 1, 7])
def test_wide_decoder_stagewise_and_tiers(crtlib, case, lpw):
    """... stage by stage (line table compared too), and with the decoder tiers forced: tier 1 runs in the wide kernel,
    tiers 2 / 3 hand the whole batch back to the lane-per-scanline kernel"""
    _run_case(crtlib, WIDE_CASES[case], fused=False, steps=2, n=3, wide_lpw=lpw)
    for mode in (1, 2, 3):
        _run_case(crtlib, WIDE_CASES[case], fused=True, exact=mode, steps=1, n=2, wide_lpw=lpw)


def test_wide_decoder_against_the_lane_per_scanline_decoder(crtlib):
    """the two decoders on the same batch (100 fields of 1920x1080, both parities): byte-identical pictures.  crthip_set_pixel_tile(16)
    keeps the batch on k_decode (a narrow pixel tile is no wide picture to the dispatcher)."""
    import torch
    n, w, h = 100, 1920, 1080
    base = np.stack([R.synth_image(w, h, 4, 8300 + k, "bars" if k & 1 else "random") for k in range(4)])
    imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")
    data = imgs.repeat(n // 4, 1, 1, 1)[:, :h]
    outs = []
    for tile in (0, 16):
        g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
        g.scanlines = 1
        g.set_shape(1)
        g.set_pixel_tile(tile)
        s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=[(k // 4) & 1 for k in range(n)], frame=0)
        for step in range(2):
            g.fieldpass(s, 7)
        g.synchronize()
        outs.append((g.out.clone(), g.state.clone()))
        g.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(outs[0][0].to(torch.int64).sum().item()) > 0


def test_wide_pictures_leave_the_scanline_parallel_decoder_early(crtlib):
    """48 fields of 1920x1080: the library's own choice (shape 0) is the scanline-parallel ENCODER with the wide-run DECODER
    (WIDE_SHAPE_MIN_FIELDS, crt_decode.hip); the same batch through the forced shapes 1 (lane-per-row encoder + wide-run decoder)
    and 2 (scanline-parallel kernels, checked against the oracle elsewhere) gives the same bytes and the same state, two passes."""
    import torch
    n, w, h = 48, 1920, 1080
    base = np.stack([R.synth_image(w, h, 4, 9100 + k, "bars" if k & 1 else "random") for k in range(4)])
    imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")
    data = imgs.repeat(n // 4, 1, 1, 1)[:, :h]
    outs = []
    for shape in (0, 1, 2):
        g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
        g.scanlines = 1
        g.set_shape(shape)
        s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=[(k // 4) & 1 for k in range(n)], frame=0)
        for step in range(2):
            g.fieldpass(s, 9)
        g.synchronize()
        outs.append((g.out.clone(), g.state.clone()))
        g.close()
    for k in (1, 2):
        assert torch.equal(outs[0][0], outs[k][0]), "picture: shape 0 vs shape %d" % k
        assert torch.equal(outs[0][1], outs[k][1]), "state: shape 0 vs shape %d" % k
    assert int(outs[0][0].to(torch.int64).sum().item()) > 0


@pytest.mark.parametrize("case", range(len(CASES)))
def test_large_signal_tiles_parity(crtlib, case):
    """the fused encoder with its LARGE signal tile (128-byte pieces beside the narrow image tile, 256-byte pieces beside the wide
    one: what batches of thousands of fields take, crt_encode.hip launch_active) forced onto every case of the table, against the
    oracle; the library's own choice at these batch sizes is the 64-byte tile of every other test"""
    _run_case(crtlib, CASES[case], fused=True, steps=2, sig_tile=64)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("case", [1, 2, 3, 5])
def test_slower_decoder_tiers_parity(crtlib, case, mode):
    """ordinary inputs normally run decoder tier 0 (64-bit mads, no I/Q low cascades); force tier 3 (mode 1:
    exact 32-bit multiplies everywhere), tier 2 (mode 2: 24-bit mads) and tier 1 (mode 3: 64-bit mads, all
    cascades) onto them"""
    _run_case(crtlib, CASES[case], fused=True, exact=mode, steps=2)


# ---------------------------------------------------------------------------------------------------------------
# the scanline-parallel kernel shape (crt_decode2.hip): a DPP row of lanes per scanline
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_scanline_parallel_shape_parity(crtlib, case, fused):
    """every case of the table, decoded by the scanline-parallel kernels instead of the lane-per-scanline ones"""
    _run_case(crtlib, CASES[case], fused=fused, shape=2, steps=2)


@pytest.mark.parametrize("mode", [1, 3])
@pytest.mark.parametrize("case", [1, 2, 5])
def test_scanline_parallel_wide_variant_on_ordinary_inputs(crtlib, case, mode):
    """ordinary inputs take the 16-lane (NARROW) variant; set_exact 1 / 3 force the 32-lane one with all 6 cascades"""
    _run_case(crtlib, CASES[case], fused=True, exact=mode, shape=2, steps=2)


def test_library_picks_the_shape_by_batch_size(crtlib):
    """crthip_set_shape(0): small batches -> scanline-parallel, large ones -> lane-per-scanline; same pictures.  The signal layout
    follows the encoder (round 6, crt_fused_layout): flat lines with the scanline-parallel encoder the library picks for up to 256
    fields, padded lines with the lane-per-row one."""
    _run_case(crtlib, CASES[1], fused=True, shape=0, steps=2, n=2, want_padded=False)
    _run_case(crtlib, ("ntsc", 96, 240, R.FMT_BGRA, 64, 48, R.FMT_BGRA, 24, dict(as_color=1), dict(scanlines=1)),
              fused=True, shape=0, steps=1, n=300, want_padded=True)


def test_the_nes_keeps_flat_signal_lines(crtlib):
    """... and the NES's PPU-pixel encoder (a table look-up per sample, never store-bound) keeps the reference's flat lines: the copies
    behind padded lines cost its margin kernel more than the alignment gives (profiles/r06_ab_padded_by_system.txt); NES-RGB, an RGB
    encoder with the NES's timing, takes the padded ones."""
    import torch
    n = 3
    ppu = np.stack([R.synth_ppu(256, 240, 900 + k) for k in range(n)])
    full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
    full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")
    g = crtlib.CRT(n, 640, 480, crtlib.FMT_BGRA, "nes", device=0)
    g.set_shape(1)
    g.fieldpass(crtlib.Settings(full[:, :240], hue=0, dot_crawl_offset=[k % 3 for k in range(n)]), 12)
    g.synchronize()
    sig, padded = g.fieldpass_signal()
    assert not padded
    orc = R.Oracle("nes")
    for k in range(n):
        c = orc.new_crt(640, 480, R.FMT_BGRA)
        c.settings(np.concatenate([ppu[k], ppu[k][-1:]], axis=0), w=256, h=240, dot_crawl_offset=k % 3, hue=0)
        c.modulate()
        c.demodulate(12)
        np.testing.assert_array_equal(sig[k, :orc.input_size].cpu().numpy(), c.inp, err_msg="NES field %d: inp of the fused path" % k)
        np.testing.assert_array_equal(g.out[k].cpu().numpy().reshape(-1), c.out)
    g.close()
    _run_case(crtlib, ("nesrgb",) + F4_CASES[1], fused=True, steps=2, want_padded=True)


# ---------------------------------------------------------------------------------------------------------------
# SNES, template, PV-1000 (5 samples per chroma cycle), NES-RGB;  8(f3): CRT_DO_BLOOM builds
# ---------------------------------------------------------------------------------------------------------------
F4_CASES = [
    # outw, outh, ofmt, w, h, ifmt, noise, settings, knobs
    (640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 0, dict(as_color=1), {}),
    (640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1, hue=25), dict(scanlines=1)),
    (832, 624, R.FMT_ARGB, 640, 480, R.FMT_ABGR, 40, dict(as_color=1, hue=350), dict(hue=17, saturation=14)),
    (640, 480, R.FMT_RGB, 320, 200, R.FMT_RGB, 12, dict(as_color=1, hue=-57), dict(blend=1)),
    (500, 300, R.FMT_BGR, 200, 100, R.FMT_ARGB, 60, dict(as_color=1, raw=1, xoffset=8, yoffset=2), dict(black_point=3, white_point=90)),
    (753, 240, R.FMT_ABGR, 753, 236, R.FMT_RGBA, 5, dict(as_color=0), dict(brightness=9, contrast=200)),
    (640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 30, dict(as_color=1), dict(saturation=900, contrast=300)),
]


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", range(len(F4_CASES)))
@pytest.mark.parametrize("name", ["snes", "temp", "pv1k"])
def test_f4_systems_parity(crtlib, name, case, fused):
    """crt_snes.c / crt_template.c / crt_pv1k.c + the shared decoder (PV-1000: the 5-sample branch,
    crt_core.c:480-510,544-549).  Even cases run the lane-per-scanline decoder, odd ones the scanline-parallel one."""
    shape = 2 if case % 2 else 1
    _run_case(crtlib, (name,) + F4_CASES[case], fused=fused, shape=shape, steps=3)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("knobs", [dict(saturation=3), dict(saturation=7, contrast=250), dict(saturation=12, brightness=40),
                                   dict(saturation=18), dict(saturation=30, contrast=20000), dict(saturation=-9), dict(saturation=200)])
def test_pv1k_decoder_tiers(crtlib, knobs, fused):
    """round 3: the 5-sample system in every decoder tier -- its five carriers are bounded by (|dci| + |dcq| + 1) * |saturation|
    (crt_core.c:497-505), which opens the 64-bit-mad and 24-bit tiers for it; the saturation walks through all of them"""
    case = ("pv1k", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1, hue=11), dict(knobs, scanlines=1))
    _run_case(crtlib, case, fused=fused, shape=1, steps=3)


@pytest.mark.parametrize("shape", [1, 2])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", [1, 3, 7, 8])
@pytest.mark.parametrize("name", ["vhslcg", "ntscnovsync", "ntscnohsync", "ntschipass"])
def test_build_time_variants_parity(crtlib, name, case, fused, shape):
    """VERDICT round 2, missing #3: the reference's remaining #define switches as crthip_params.flags -- CRT_VHS_NOISE 0
    (crt_ntscvhs.h:29), CRT_DO_VSYNC 0 / CRT_DO_HSYNC 0 (crt_core.h:71-72; crt_core.c:323-341, 446-450), HIPASS 1
    (crt_ntsc.c:115-126) -- against the oracle's switches, which tests/test_oracle_vs_ref.py pins to the rebuilt reference"""
    _run_case(crtlib, (name,) + CASES[case][1:], fused=fused, shape=shape, steps=3)


@pytest.mark.parametrize("sat", [11, 15, 17, 19, 22])
def test_ntsc_between_the_envelopes(crtlib, sat):
    """saturation 10 gives |wave| ~ 48 000: 15 ... 22 crosses 65 532 (products), the signal-range bound of the fused path
    (no low cascades) and 120 000 (64-bit-mad tiers) -- fused and stage-level launches take different tiers here and must
    agree with the oracle all the same"""
    case = ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1), dict(saturation=sat, scanlines=1))
    _run_case(crtlib, case, fused=True, shape=1, steps=3)
    _run_case(crtlib, case, fused=False, shape=1, steps=2)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("name", ["ntscbloom", "snesbloom", "pv1kbloom"])
def test_bloom_build_parity(crtlib, name, case, fused):
    """CRT_DO_BLOOM (crt_core.h:70 set to 1): encoder geometry 55500 / 63500 (crt_ntsc.c:148-161), per-line beam
    width -> per-line resampler step and start (crt_core.c:399-402, 512-526).  The oracle's bloom mode is pinned
    against the reference compiled with the patched header (tests/test_oracle_vs_ref.py)."""
    _run_case(crtlib, (name,) + F4_CASES[case], fused=fused, shape=0, steps=3)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("name", ["ntscbloom", "snesbloom", "pv1kbloom"])
def test_bloom_lane_per_scanline_parity(crtlib, name, case, fused):
    """the bloom build in the lane-per-scanline decoder (crt_decode3.hip) -- the lines of the batch are
    counting-sorted by their beam width (crt_core.c:518-522: dx and scanL follow from line_w alone), a wave decodes 64
    lines of equal geometry.  Forced here (shape 1); 5 fields = 1200 lines over a few dozen widths: full and partial
    waves, padding slots"""
    _run_case(crtlib, (name,) + F4_CASES[case], fused=fused, shape=1, steps=2, n=5)


@pytest.mark.parametrize("knobs", [dict(saturation=14), dict(saturation=19), dict(saturation=40), dict(brightness=5000, contrast=20),
                                   dict(saturation=900, contrast=300), dict(brightness=200000, contrast=9000000, white_point=9000000),
                                   dict(blend=1, v_fac=30), dict(scanlines=1, blend=1)])
@pytest.mark.parametrize("geom", [(640, 480, R.FMT_BGRA), (64, 48, R.FMT_RGB), (1920, 1080, R.FMT_RGBA), (100, 300, R.FMT_BGR)])
def test_bloom_lane_per_scanline_tiers_and_geometries(crtlib, knobs, geom):
    """the bloom kernel's tiers (0: 64-bit mads, 2: 24-bit mads -- also for the lines tier 1 would take --, 3: exact), the
    wide pixel tile (outw >= 1280), 3-byte formats, blend, and pictures shorter than the raster (several scanlines per
    output row, one decoder pass per collision rank)"""
    outw, outh, ofmt = geom
    case = ("ntscbloom", outw, outh, ofmt, 320, 240, R.FMT_BGRA, 24, dict(as_color=1, hue=33), knobs)
    _run_case(crtlib, case, fused=True, shape=1, steps=2, n=3)


def test_bloom_decoder_takes_any_line_table(crtlib):
    """crthip_decode with a line table that did NOT come from k_bloom: steps and starts edited per line, so that lines of
    equal sort key carry different resampler geometries.  The lane-per-scanline decoder then decodes a wave in rounds, one
    geometry at a time; the scanline-parallel decoder, which takes every line's geometry as it comes, is the reference."""
    import ctypes as C
    import torch
    n, w, h = 6, 640, 480
    imgs = np.stack([R.synth_image(w, h, 4, 40 + k, "random" if k % 2 else "bars") for k in range(n)])
    g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntscbloom", device=0)
    s = crtlib.Settings(_padded(imgs), format=R.FMT_BGRA, as_color=1, field=[k & 1 for k in range(n)])
    g.modulate(s)
    g.demodulate(24)                                     # fills inp[] and the line table
    g.synchronize()
    lt = g.line_table                                    # [n, lines, 8]: ..., dx = [6], scanl = [7]
    rng = np.random.default_rng(7)
    dx_add = torch.from_numpy(rng.integers(0, 4, size=(n, lt.shape[1])).astype(np.int32)).to(lt.device)
    sl_add = torch.from_numpy((rng.integers(0, 3, size=(n, lt.shape[1])) * 1365).astype(np.int32)).to(lt.device)
    lt[:, :, 6] += dx_add * 3
    lt[:, :, 7] += sl_add                                # also fractional starts, which k_bloom never produces
    p = g.params(s, 24)
    outs = []
    for shape in (1, 2):
        g.set_shape(shape)
        g.out.zero_()
        g._check(g.L.crthip_decode(g.ctx, C.byref(p), n, C.c_void_p(g.inp.data_ptr()), C.c_void_p(lt.data_ptr()),
                                   C.c_void_p(g.out.data_ptr()), g.out.stride(0)), "crthip_decode")
        g.synchronize()
        outs.append(g.out.cpu().numpy().copy())
    g.close()
    assert outs[1].any()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_bloom_large_batch_takes_the_lane_per_scanline_decoder(crtlib):
    """crthip_set_shape(0) with a bloom build: more than ROWS_SHAPE_MAX_FIELDS fields go through the sort + lane-per-scanline
    decoder, and decode the same pictures as the scanline-parallel shape"""
    n, outw, outh = 200, 320, 240
    imgs = np.stack([R.synth_image(96, 80, 4, 5 + k, "random" if k % 3 else "bars") for k in range(n)])
    outs = []
    for shape in (0, 2):
        g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "ntscbloom", device=0)
        g.set_shape(shape)
        s = crtlib.Settings(_padded(imgs), format=R.FMT_BGRA, as_color=1, field=[k & 1 for k in range(n)])
        g.fieldpass(s, 24)
        g.synchronize()
        outs.append(g.out.cpu().numpy().copy())
        g.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    # ... and a few of them against the oracle
    orc, ocrts = _oracle_batch("ntscbloom", 4, outw, outh, R.FMT_BGRA, {})
    for j, k in enumerate((0, 1, 77, 199)):
        c = ocrts[j]
        c.settings(np.concatenate([imgs[k], imgs[k][-1:]], axis=0), format=R.FMT_BGRA, w=96, h=80, field=k & 1, frame=0, as_color=1)
        c.modulate()
        c.demodulate(24)
        np.testing.assert_array_equal(outs[0][k].reshape(-1), c.out, err_msg="field %d" % k)


@pytest.mark.parametrize("shape", [1, 2])
@pytest.mark.parametrize("fused", [False, True])
def test_nesrgb_parity(crtlib, fused, shape):
    """crt_nesrgb.c: an RGB image encoded with the NES's line timing (progressive, no band limit, 3 line classes)"""
    n, w, h, outw, outh = 3, 256, 240, 640, 480
    orc = R.Oracle("nesrgb")
    g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "nesrgb", device=0)
    g.set_shape(shape)
    g.scanlines = 1
    ocrts = [orc.new_crt(outw, outh, R.FMT_BGRA) for _ in range(n)]
    for c in ocrts:
        c.set("scanlines", 1)
    s = None
    for step in range(4):
        ifmt = [R.FMT_BGRA, R.FMT_RGB, R.FMT_ARGB, R.FMT_ABGR][step]
        imgs = np.stack([R.synth_image(w, h, R.bpp4fmt(ifmt), 91 * step + k, "random" if k % 2 else "bars") for k in range(n)])
        dco = [(step + k) % 3 for k in range(n)]
        hue = [0, 50, -79, 200][step]
        init = s.initialized if s is not None else 0
        s = crtlib.Settings(_padded(imgs), format=ifmt, hue=hue, dot_crawl_offset=dco)
        s.initialized = init
        noise = [0, 12, 24, 50][step]
        if fused:
            g.fieldpass(s, noise)
        else:
            g.modulate(s)
            analog = g.analog.cpu().numpy()
            g.demodulate(noise)
        g.synchronize()
        gout = g.out.cpu().numpy()
        for k, c in enumerate(ocrts):
            what = "nesrgb fused=%s shape %d step %d field %d" % (fused, shape, step, k)
            c.settings(np.concatenate([imgs[k], imgs[k][-1:]]), format=ifmt, w=w, h=h, dot_crawl_offset=dco[k], hue=hue)
            if fused:
                c.analog[:] = 0
                c.sset("field_initialized", 0)
            c.modulate()
            if not fused:
                np.testing.assert_array_equal(analog[k, :orc.input_size], c.analog, err_msg=what + " analog")
            c.demodulate(noise)
            for f in ("hsync", "vsync", "rn"):
                assert g.get(f)[k] == c.get(f), "%s %s" % (what, f)
            np.testing.assert_array_equal(g.ccf[k, :orc.vper, :orc.ccs], c.ccf, err_msg=what + " ccf")
            np.testing.assert_array_equal(gout[k].reshape(-1), c.out, err_msg=what + " out")
    g.close()


@pytest.mark.parametrize("name", ["nes", "nesp0"])
def test_nes_negative_hue_and_dot_crawl(crtlib, name):
    """the burst angle (hue + x*90 + (y + dot_crawl_offset)*120 + 33) % 360 goes negative for negative
    hues and C's truncating % / then differ from the row-reduced form by one 14-bit step (hue -42/-57/-79)."""
    import torch
    n = 3
    orc = R.Oracle(name)
    g = crtlib.CRT(n, 320, 240, crtlib.FMT_BGRA, name, device=0)
    ocrts = [orc.new_crt(320, 240, R.FMT_BGRA) for _ in range(n)]
    for step, hue in enumerate([-42, -57, -79, -200, -359]):
        ppu = np.stack([R.synth_ppu(256, 240, 5 + step + k) for k in range(n)])
        full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
        full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")
        dco = [(step + k) % 3 for k in range(n)]
        s = crtlib.Settings(full[:, :240], hue=hue, dot_crawl_offset=dco)
        g.modulate(s)
        analog = g.analog.cpu().numpy()
        g.demodulate(8)
        g.synchronize()
        gout = g.out.cpu().numpy()
        for k, c in enumerate(ocrts):
            c.settings(np.concatenate([ppu[k], ppu[k][-1:]]), w=256, h=240, dot_crawl_offset=dco[k], hue=hue)
            c.modulate()
            np.testing.assert_array_equal(analog[k, :orc.input_size], c.analog, err_msg="%s hue %d analog %d" % (name, hue, k))
            c.demodulate(8)
            np.testing.assert_array_equal(g.ccf[k, :3, :4], c.ccf)
            np.testing.assert_array_equal(gout[k].reshape(-1), c.out, err_msg="%s hue %d out %d" % (name, hue, k))
    g.close()


def test_rectangle_running_over_the_line_end(crtlib):
    """xoffset 4 in non-raw NTSC puts the rectangle at 160 + 753 = 913 > 910: the reference's flat index
    (crt_ntsc.c:322) continues in the next line's front porch; so do the kernels (both shapes, stagewise and fused).
    xoffset 150 runs 147 samples into the next line -- over its sync pulse and burst: the ENCODER is still exact
    (analog[] compared); what the decoder makes of such a field involves windows far behind inp[] (reference UB)."""
    for fused in (False, True):
        _run_case(crtlib, ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 24, dict(as_color=1, xoffset=4), {}),
                  fused=fused, steps=2)
        _run_case(crtlib, ("ntsc", 640, 480, R.FMT_BGRA, 640, 480, R.FMT_BGRA, 0, dict(as_color=1, xoffset=16, yoffset=1), dict(scanlines=1)),
                  fused=fused, steps=2, shape=2)
    n, w, h = 2, 640, 480
    imgs = np.stack([R.synth_image(w, h, 4, 55 + k) for k in range(n)])
    g = crtlib.CRT(n, 640, 480, crtlib.FMT_BGRA, "ntsc", device=0)
    s = crtlib.Settings(_padded(imgs), format=crtlib.FMT_BGRA, field=[0, 1], frame=[0, 0], xoffset=150, yoffset=2)
    g.modulate(s)
    g.synchronize()
    an = g.analog.cpu().numpy()
    orc = R.Oracle("ntsc")
    for k in range(n):
        c = orc.new_crt(640, 480, R.FMT_BGRA)
        c.settings(np.concatenate([imgs[k], imgs[k][-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=k, frame=0, xoffset=150, yoffset=2)
        c.modulate()
        np.testing.assert_array_equal(an[k, :orc.input_size], c.analog, err_msg="xoffset 150: analog of field %d" % k)
    g.close()


def test_tight_images_are_never_read_past_their_end(crtlib):
    """the reference's `if (sy >= h) sy = h` reads image row h (crt_ntsc.c:263) on odd fields of raw images
    with h <= desth.  Without CRTHIP_F_IMAGE_SPARE_ROW the kernels read row h - 1 instead: a TIGHT [n,h,w,c] tensor
    gives the same picture as the padded one whose spare row repeats row h - 1."""
    import torch
    n, w, h = 3, 100, 200
    imgs = np.stack([R.synth_image(w, h, 4, 321 + k) for k in range(n)])
    outs = []
    for tight in (False, True):
        g = crtlib.CRT(n, 320, 240, crtlib.FMT_BGRA, "ntsc", device=0)
        data = _to_dev(imgs) if tight else _padded(imgs)
        s = crtlib.Settings(data, format=crtlib.FMT_BGRA, raw=1, field=1, frame=0)
        assert g._has_spare_row(s) == (not tight)
        g.fieldpass(s, 0)
        g.synchronize()
        outs.append(g.out.cpu().numpy())
        g.close()
    np.testing.assert_array_equal(outs[0], outs[1])


NES_CASES = [
    # name, outw, outh, noise, knobs
    ("nes", 640, 480, 0, {}),
    ("nes", 640, 480, 24, dict(scanlines=1)),
    ("nesp0", 640, 480, 12, dict(blend=1, hue=15)),       # BASELINE configs[4]: CRT_CHROMA_PATTERN 0
    ("nesp0", 256, 240, 0, dict(saturation=12)),
    ("nes", 768, 720, 40, dict(scanlines=1, black_point=2, white_point=95)),
    # round 3: either side of the decoder envelopes that follow from the NES's own signal range (carrier amplitude grows
    # with the saturation: tier 0 -> tier 1 without / with the I/Q low cascades -> 24-bit tier)
    ("nesp0", 640, 480, 12, dict(saturation=8)),
    ("nesp0", 640, 480, 60, dict(saturation=13, scanlines=1)),
    ("nesp0", 640, 480, 100, dict(saturation=16)),
    ("nes", 640, 480, 30, dict(saturation=19, white_point=120)),
    ("nes", 640, 480, 12, dict(saturation=26, contrast=40000)),
    # NES_BORDER 1 (crt_nes.c:69,138-160): the border colour right of the picture
    ("nesborder", 640, 480, 12, dict(scanlines=1)),
    ("nesborder", 512, 480, 0, dict(black_point=4, white_point=90)),
    # wide pictures: the wide-run decoder (crt_decode4.hip), which the NES enters in tier 1 (its carriers do not fit 24 bits)
    ("nesp0", 1920, 1080, 12, dict(scanlines=1)),
    ("nes", 1700, 600, 24, dict(saturation=16)),
]


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", range(len(NES_CASES)))
def test_nes_parity(crtlib, case, fused):
    """NES encoder (crt_nes.c): 256x240 9-bit PPU pixels, 3-line chroma period, dot crawl."""
    import torch
    name, outw, outh, noise, knobs = NES_CASES[case]
    n = 3
    orc = R.Oracle(name)
    g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, name, device=0)
    if outw >= 1664:
        g.set_shape(1)                       # three fields would otherwise take the scanline-parallel kernels
    ocrts = [orc.new_crt(outw, outh, R.FMT_BGRA) for _ in range(n)]
    for k, v in knobs.items():
        setattr(g, k, v)
        for c in ocrts:
            c.set(k, v)
    s = None
    for step in range(4):
        ppu = np.stack([R.synth_ppu(256, 240, 31 * step + k) for k in range(n)])
        full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
        full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")
        dco = [(step + k) % 3 for k in range(n)]
        init = s.initialized if s is not None else 0
        border = [0x21, 0x16, 0x1c0 | 0x2a, 0x0d][step]
        s = crtlib.Settings(full[:, :240], hue=(step * 50) % 360, dot_crawl_offset=dco, border_color=border)
        s.initialized = init
        for k, c in enumerate(ocrts):
            pad = np.concatenate([ppu[k], ppu[k][-1:]], axis=0)
            c.settings(pad, w=256, h=240, dot_crawl_offset=dco[k], hue=(step * 50) % 360, border_color=border)
        if fused:
            g.fieldpass(s, noise)
        else:
            g.modulate(s)
            analog = g.analog.cpu().numpy()
            g.demodulate(noise)
        g.synchronize()
        gout = g.out.cpu().numpy()
        for k, c in enumerate(ocrts):
            what = "%s fused=%s step %d field %d" % (name, fused, step, k)
            if fused:
                # the fused path starts every field from a crt_init-clean analog[] (documented batch
                # semantics): mirror that in the oracle by re-running setup_field on a zeroed signal
                c.analog[:] = 0
                c.sset("field_initialized", 0)
            c.modulate()
            if not fused:
                np.testing.assert_array_equal(analog[k, :orc.input_size], c.analog, err_msg=what + " analog")
            c.demodulate(noise)
            for f in ("hsync", "vsync", "rn"):
                assert g.get(f)[k] == c.get(f), "%s %s" % (what, f)
            np.testing.assert_array_equal(g.ccf[k, :orc.vper, :orc.ccs], c.ccf, err_msg=what + " ccf")
            np.testing.assert_array_equal(gout[k].reshape(-1), c.out, err_msg=what + " out")
    g.close()


@pytest.mark.parametrize("name", ["nes", "nesrgb"])
def test_nes_context_reused_with_another_yoffset(crtlib, name):
    """with NES timing the burst sits on the active lines only (crt_nes.c:173-178), so the cached clean
    skeleton of the fused path depends on yoffset as well as on the burst table.  One context, same hue, several yoffsets."""
    import torch
    n, outw, outh = 2, 640, 480
    orc = R.Oracle(name)
    g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, name, device=0)
    ocrts = [orc.new_crt(outw, outh, R.FMT_BGRA) for _ in range(n)]
    for step, yoff in enumerate([0, 3, -2, 3]):
        if name == "nes":
            ppu = np.stack([R.synth_ppu(256, 240, 5 * step + k) for k in range(n)])
            full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
            full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")
            s = crtlib.Settings(full[:, :240], hue=20, dot_crawl_offset=[1, 2], yoffset=yoff)
            pads = [np.concatenate([ppu[k], ppu[k][-1:]], axis=0) for k in range(n)]
            skw = dict(w=256, h=240, hue=20, yoffset=yoff)
        else:
            imgs = np.stack([R.synth_image(256, 240, 4, 40 + 5 * step + k) for k in range(n)])
            s = crtlib.Settings(_padded(imgs), format=R.FMT_BGRA, hue=20, dot_crawl_offset=[1, 2], yoffset=yoff)
            pads = [np.concatenate([imgs[k], imgs[k][-1:]], axis=0) for k in range(n)]
            skw = dict(format=R.FMT_BGRA, w=256, h=240, hue=20, yoffset=yoff)
        g.fieldpass(s, 12)
        g.synchronize()
        gout = g.out.cpu().numpy()
        for k, c in enumerate(ocrts):
            c.settings(pads[k], dot_crawl_offset=[1, 2][k], **skw)
            c.analog[:] = 0                      # batch semantics: every field starts from a crt_init-clean analog[]
            c.sset("field_initialized", 0)
            c.modulate()
            c.demodulate(12)
            np.testing.assert_array_equal(gout[k].reshape(-1), c.out, err_msg="%s yoffset %d field %d" % (name, yoff, k))
            for f in ("hsync", "vsync", "rn"):
                assert g.get(f)[k] == c.get(f), "%s yoffset %d %s" % (name, yoff, f)
    g.close()


@pytest.mark.parametrize("aberration", [0, 9, 17])
def test_vhs_encoder_parity(crtlib, aberration):
    """crt_ntscvhs.c: VHS band limits, aberration band without sync pulses, hsync/ccf reset.  Only the
    encoder: the VHS decoder's rand()-driven noise is not in the batch ABI yet (fails loudly)."""
    import torch
    n, w, h = 2, 832, 624
    imgs = np.stack([R.synth_image(w, h, 4, 50 + k, "bars" if k else "random") for k in range(n)])
    g = crtlib.CRT(n, 832, 624, crtlib.FMT_BGRA, "vhs", device=0)
    s = crtlib.Settings(_padded(imgs), format=crtlib.FMT_BGRA, field=[0, 1], frame=[1, 1], hue=12, aberration=aberration)
    g.state[:, crtlib.ST_HSYNC] = 5
    g.modulate(s)
    g.synchronize()
    an = g.analog.cpu().numpy()
    orc = R.Oracle("vhs")
    for k in range(n):
        c = orc.new_crt(832, 624, R.FMT_BGRA)
        c.set("hsync", 5)
        c.settings(imgs[k], format=R.FMT_BGRA, w=w, h=h, as_color=1, hue=12, field=[0, 1][k], frame=1,
                   do_aberration=1 if aberration else 0)
        if aberration:
            # the oracle draws the band height from rand(); find a seed that yields this one
            import ctypes as C
            libc = C.CDLL(None)
            seed = next(sd for sd in range(1, 5000) if (libc.srand(sd), ((libc.rand() % 12) - 8) + 14)[1] == aberration)
            libc.srand(seed)
        c.modulate()
        np.testing.assert_array_equal(an[k, :orc.input_size], c.analog, err_msg="vhs analog %d" % k)
        np.testing.assert_array_equal(g.ccf[k, :1, :4], c.ccf)
        assert g.get("hsync")[k] == c.get("hsync") == 0
    g.close()


@pytest.mark.parametrize("sysname", ["vhs", "vhsbloom", "vhslp", "vhsep"])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("noise", [0, 12, 40])
def test_vhs_fieldpass_parity(crtlib, fused, noise, sysname):
    _vhs_fieldpass(crtlib, fused, noise, sysname, 0)


@pytest.mark.parametrize("fused", [False, True])
def test_vhs_bloom_lane_per_scanline_parity(crtlib, fused):
    """the VHS bloom build through the sort + lane-per-scanline decoder (crt_decode3.hip)"""
    _vhs_fieldpass(crtlib, fused, 12, "vhsbloom", 1)


def test_vhs_wide_run_decoder_parity(crtlib):
    """the VHS build at 1920x1080 through the wide-run decoder (crt_decode4.hip)"""
    _vhs_fieldpass(crtlib, True, 12, "vhs", 1, size=(1920, 1080))


def _vhs_fieldpass(crtlib, fused, noise, sysname, shape, size=(832, 624)):
    """BASELINE configs[3]: CRT_SYSTEM_NTSCVHS, 832x624 (and its CRT_DO_BLOOM build).  The decoder's noise is the C library's rand()
    stream (crt_core.c:344-351): field k's generator starts at srand(seed_k) and carries over from
    step to step, so the oracle processes each field's whole sequence under its own libc stream.
    (The aberration band draws from the same stream in crt_modulate; that interplay is covered by the
    drop-in test, where the host draws it exactly like the reference.)"""
    import ctypes as C
    libc = C.CDLL(None)
    n, (w, h), steps = 3, size, 3
    seeds = [1, 77, 20260924]
    imgs = np.stack([R.synth_image(w, h, 4, 60 + k, "random" if k != 1 else "bars") for k in range(n)])
    orc = R.Oracle(sysname)
    want = []
    for k in range(n):
        c = orc.new_crt(w, h, R.FMT_BGRA)
        c.set("scanlines", 1)
        c.settings(np.concatenate([imgs[k], imgs[k][-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=k & 1, frame=0,
                   do_aberration=0)
        libc.srand(seeds[k])
        per = []
        for step in range(steps):
            if fused:
                c.analog[:] = 0                        # batch semantics: clean analog[] per field-pass
            c.modulate()
            c.demodulate(noise)
            per.append((c.inp.copy(), c.out.copy(), c.get("hsync"), c.get("vsync"), c.get("rn"), c.ccf.copy()))
            c.sset("field", c.sget("field") ^ 1)
        want.append(per)
    g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, sysname, device=0)
    g.scanlines = 1
    g.set_shape(shape)
    g.srand(seeds)
    fields = [k & 1 for k in range(n)]
    s = crtlib.Settings(_padded(imgs), format=crtlib.FMT_BGRA, field=list(fields), frame=0)
    for step in range(steps):
        if fused:
            g.fieldpass(s, noise)
        else:
            g.modulate(s)
            g.demodulate(noise)
        g.synchronize()
        gout = g.out.cpu().numpy()
        for k in range(n):
            inp, out, hs, vs, rn, ccf = want[k][step]
            what = "%s fused=%s noise %d step %d field %d" % (sysname, fused, noise, step, k)
            if not fused:
                np.testing.assert_array_equal(g.inp[k, :orc.input_size].cpu().numpy(), inp, err_msg=what + " inp")
            assert (g.get("hsync")[k], g.get("vsync")[k], g.get("rn")[k]) == (hs, vs, rn), what
            np.testing.assert_array_equal(g.ccf[k, :1, :4], ccf, err_msg=what + " ccf")
            np.testing.assert_array_equal(gout[k].reshape(-1), out, err_msg=what + " out")
        fields = [f ^ 1 for f in fields]
        s.field = list(fields)
    g.close()


@pytest.mark.parametrize("noise", [7, 100])
def test_vhs_rand_noise_many_seeds(crtlib, noise):
    """The data-dependent part of the VHS rand() walk (last 25 lines: 2 or 3 calls per sample) under many
    generator seeds, two crt_demodulate calls in a row so that the second one
    starts from the history the first one hands back.  Checked: inp[], rn, hsync/vsync and the picture."""
    import ctypes as C
    import torch
    libc = C.CDLL(None)
    n, w, h = 40, 96, 240
    seeds = [1 + 7919 * k for k in range(n)]
    orc = R.Oracle("vhs")
    # a properly encoded field (the decoder must find its sync pulses, or it reads past inp[] like the reference
    # would), the same for every seed
    img = R.synth_image(w, h, 4, 4242)
    c0 = orc.new_crt(w, h, R.FMT_BGRA)
    c0.settings(np.concatenate([img, img[-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=0, frame=0, do_aberration=0)
    c0.modulate()
    analog = np.repeat(c0.analog.copy()[None, :], n, axis=0)
    g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "vhs", device=0)
    g.srand(seeds)
    g.analog[:, :orc.input_size].copy_(torch.from_numpy(analog).to(g.dev))
    want = []
    for k in range(n):
        c = orc.new_crt(w, h, R.FMT_BGRA)
        c.analog[:] = analog[k]
        libc.srand(seeds[k])
        per = []
        for step in range(2):
            c.demodulate(noise)
            per.append((c.inp.copy(), c.out.copy(), c.get("hsync"), c.get("vsync"), c.get("rn")))
        want.append(per)
    for step in range(2):
        g.demodulate(noise)
        g.synchronize()
        ginp = g.inp[:, :orc.input_size].cpu().numpy()
        gout = g.out.cpu().numpy()
        for k in range(n):
            inp, out, hs, vs, rn = want[k][step]
            what = "seed %d noise %d call %d" % (seeds[k], noise, step)
            bad = np.nonzero(ginp[k] != inp)[0]
            assert bad.size == 0, "%s: inp differs first at sample %d (of %d)" % (what, bad[0], orc.input_size)
            assert (g.get("hsync")[k], g.get("vsync")[k], g.get("rn")[k]) == (hs, vs, rn), what
            np.testing.assert_array_equal(gout[k].reshape(-1), out, err_msg=what)
    g.close()


@pytest.mark.parametrize("name,noise,scanlines,outsz", [("ntsc", 24, 1, (640, 480)), ("ntsc", 0, 0, (832, 624)),
                                                         ("ntsc", 120, 1, (320, 240)), ("nes", 12, 1, (640, 480)),
                                                         # the other systems and a bloom build
                                                         ("snes", 24, 1, (640, 480)), ("pv1k", 30, 0, (640, 480)),
                                                         ("temp", 60, 1, (400, 300)), ("nesrgb", 24, 1, (640, 480)),
                                                         ("ntscbloom", 24, 1, (640, 480)), ("snesbloom", 40, 0, (512, 384)),
                                                         ("ntscnohsync", 24, 1, (640, 480)), ("ntschipass", 24, 1, (640, 480))])
def test_sequence_mode_equals_sequential_processing(crtlib, name, noise, scanlines, outsz):
    """n consecutive fields of ONE set (video_convert.c:246-277 semantics: hsync/vsync/rn and the
    output buffer carried over) through crthip_sequence, against the oracle doing them one after the other."""
    import torch
    import shard
    n = 9
    outw, outh = outsz
    nes = name in ("nes", "nesp0")
    orc = R.Oracle(name)
    nesrgb = orc.system == R.SYS_NESRGB
    dot_crawl = orc.system in R.DOT_CRAWL_SYSTEMS
    c = orc.new_crt(outw, outh, R.FMT_BGRA)
    c.set("scanlines", scanlines)
    c.out[:] = R.lcg_bytes(c.out.size, 5)                 # the output buffer's content before field 0
    init = c.out.copy()
    c.set("hsync", 7)
    c.set("vsync", 2)
    want = []
    iw, ih = (256, 240) if nes or nesrgb else (640, 480)
    if nes:
        frames = np.stack([R.synth_ppu(256, 240, 300 + k) for k in range(n)])
    else:
        frames = np.stack([R.synth_image(iw, ih, 4, 300 + k, "random" if k % 3 else "bars") for k in range(n)])
    for k in range(n):
        field, frame = shard.field_parity(k)
        pad = np.concatenate([frames[k], frames[k][-1:]], axis=0)
        if nes:
            c.settings(pad, w=256, h=240, dot_crawl_offset=k % 3, hue=0)
        elif nesrgb:
            c.settings(pad, format=R.FMT_BGRA, w=iw, h=ih, dot_crawl_offset=k % 3, hue=0)
        else:
            c.settings(pad, format=R.FMT_BGRA, w=iw, h=ih, as_color=1, field=field, frame=frame)
            if dot_crawl:
                c.sset("dot_crawl_offset", k % 3)
        c.modulate()
        c.demodulate(noise)
        want.append((c.out.copy(), c.get("hsync"), c.get("vsync"), c.get("rn")))
    # bloom builds: also through the beam-width sort + lane-per-scanline decoder (shape 1; small batches default to the other)
    for shape in ((0, 1) if name.endswith("bloom") else (0,)):
        g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, name, device=0)
        g.scanlines = scanlines
        g.set_shape(shape)
        g.state[0, crtlib.ST_HSYNC] = 7
        g.state[0, crtlib.ST_VSYNC] = 2
        if nes:
            full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
            full[:, :240] = torch.from_numpy(frames.astype(np.int16)).to("cuda:0")
            s = crtlib.Settings(full[:, :240], hue=0, dot_crawl_offset=[k % 3 for k in range(n)])
        else:
            par = [shard.field_parity(k) for k in range(n)]
            s = crtlib.Settings(_padded(frames), format=crtlib.FMT_BGRA, field=[a for a, _ in par], frame=[b for _, b in par],
                                dot_crawl_offset=[k % 3 for k in range(n)] if dot_crawl else 0)
        passes = g.sequence(s, noise, out_init=_to_dev(init))
        g.synchronize()
        assert 1 <= passes <= n + 1
        out = g.out.cpu().numpy()
        for k in range(n):
            o, hs, vs, rn = want[k]
            assert (g.get("hsync")[k], g.get("vsync")[k], g.get("rn")[k]) == (hs, vs, rn), "field %d state" % k
            np.testing.assert_array_equal(out[k].reshape(-1), o, err_msg="sequence field %d (passes %d)" % (k, passes))
        print("sequence %s noise %d shape %d: %d sync passes" % (name, noise, shape, passes))
        g.close()


@pytest.mark.parametrize("ofmt,scanlines,outsz", [(R.FMT_BGRA, 1, (640, 480)), (R.FMT_RGB, 0, (832, 624)), (R.FMT_ARGB, 1, (320, 240))])
def test_sequence_mode_with_blend(crtlib, ofmt, scanlines, outsz):
    """VERDICT r1 missing #5: blend = 1 (crt_main.c:235) makes the picture a recurrence over the fields; crthip_sequence
    decodes the fields in parallel and folds them into each other afterwards -- against the oracle running the
    crt_main.c loop (modulate / demodulate with the output buffer carried over) one field after the other."""
    import shard
    n, noise = 8, 24
    outw, outh = outsz
    bpp = R.bpp4fmt(ofmt)
    orc = R.Oracle("ntsc")
    c = orc.new_crt(outw, outh, ofmt)
    c.set("scanlines", scanlines)
    c.set("blend", 1)
    c.out[:] = R.lcg_bytes(c.out.size, 9)
    init = c.out.copy()
    frames = np.stack([R.synth_image(640, 480, 4, 800 + k, "random" if k % 3 else "bars") for k in range(n)])
    want = []
    for k in range(n):
        field, frame = shard.field_parity(k)
        c.settings(np.concatenate([frames[k], frames[k][-1:]]), format=R.FMT_BGRA, w=640, h=480, as_color=1, field=field, frame=frame)
        c.modulate()
        c.demodulate(noise)
        want.append((c.out.copy(), c.get("hsync"), c.get("vsync"), c.get("rn")))
    g = crtlib.CRT(n, outw, outh, ofmt, "ntsc", device=0)
    g.scanlines = scanlines
    g.blend = 1
    par = [shard.field_parity(k) for k in range(n)]
    s = crtlib.Settings(_padded(frames), format=crtlib.FMT_BGRA, field=[a for a, _ in par], frame=[b for _, b in par])
    g.sequence(s, noise, out_init=_to_dev(init.reshape(outh, outw, bpp)))
    g.synchronize()
    out = g.out.cpu().numpy()
    for k in range(n):
        o, hs, vs, rn = want[k]
        assert (g.get("hsync")[k], g.get("vsync")[k], g.get("rn")[k]) == (hs, vs, rn), "field %d state" % k
        np.testing.assert_array_equal(out[k].reshape(-1), o, err_msg="blend sequence field %d" % k)
    g.close()


@pytest.mark.parametrize("aberration", [0, 1])
@pytest.mark.parametrize("noise", [0, 12])
def test_vhs_sequence_mode_equals_sequential_processing(crtlib, noise, aberration):
    """extra/video_convert.c built for CRT_SYSTEM_NTSCVHS: besides hsync/vsync and the output buffer the fields
    share the process's rand() stream (aberration height in crt_modulate, noise in crt_demodulate).
    crthip_sequence walks the stream's data-dependent part ahead of time (k_vhs_chain) and then treats the
    fields in parallel; checked against the oracle doing one field after the other under one srand()."""
    import ctypes as C
    import shard
    libc = C.CDLL(None)
    n, w, h, outw, outh, seed = 7, 400, 300, 416, 312, 20260924
    orc = R.Oracle("vhs")
    c = orc.new_crt(outw, outh, R.FMT_BGRA)
    c.set("scanlines", 0)
    frames = np.stack([R.synth_image(w, h, 4, 500 + k, "random" if k % 3 else "bars") for k in range(n)])
    libc.srand(seed)
    want = []
    for k in range(n):
        field, frame = shard.field_parity(k)
        pad = np.concatenate([frames[k], frames[k][-1:]], axis=0)
        c.settings(pad, format=R.FMT_BGRA, w=w, h=h, as_color=1, field=field, frame=frame, do_aberration=aberration)
        c.modulate()
        c.demodulate(noise)
        want.append((c.out.copy(), c.get("hsync"), c.get("vsync"), c.get("rn")))
    next_rand = libc.rand()
    g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "vhs", device=0)
    g.scanlines = 0
    g.srand([seed] * n)                                   # entry 0 is the one that counts
    par = [shard.field_parity(k) for k in range(n)]
    s = crtlib.Settings(_padded(frames), format=crtlib.FMT_BGRA, field=[a for a, _ in par], frame=[b for _, b in par])
    s.draw_aberration = aberration
    passes = g.sequence(s, noise)
    g.synchronize()
    out = g.out.cpu().numpy()
    # with the aberration band the last lines lose their sync pulse and their filter windows run past inp[] (UB in
    # the reference): their rows are not part of the contract
    keep = outh - 12 if aberration else outh
    for k in range(n):
        o, hs, vs, rn = want[k]
        assert (g.get("hsync")[k], g.get("vsync")[k], g.get("rn")[k]) == (hs, vs, rn), "field %d state" % k
        np.testing.assert_array_equal(out[k][:keep], o.reshape(outh, outw, 4)[:keep], err_msg="vhs sequence field %d" % k)
    hist = g.vhs_hist.cpu().numpy().view(np.uint32)
    got_next = ((int(hist[n - 1, 0]) + int(hist[n - 1, 28])) & 0xffffffff) >> 1
    assert got_next == next_rand, "the generator after the last field is not where libc's is"
    g.close()


@pytest.mark.parametrize("overlap,tile", [(2, 0), (4, 32), (3, 16)])
def test_tuning_switches_do_not_change_results(crtlib, overlap, tile):
    """crthip_set_overlap (two-stream chunking) and crthip_set_pixel_tile only re-arrange the work"""
    import torch
    n, w, h = 1030, 64, 48                       # > 256 * chunks so that chunking really happens for overlap <= 4
    outw, outh = 96, 240
    imgs = np.stack([R.synth_image(w, h, 4, 900 + (k % 7)) for k in range(n)])
    fields = [k & 1 for k in range(n)]
    outs = []
    for ov, tl in ((1, 0), (overlap, tile)):
        g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "ntsc", device=0)
        g.scanlines = 1
        g.set_overlap(ov)
        g.set_pixel_tile(tl)
        s = crtlib.Settings(_padded(imgs), format=crtlib.FMT_BGRA, field=list(fields), frame=0)
        g.fieldpass(s, 24)
        g.fieldpass(s, 24)
        g.synchronize()
        outs.append((g.out.cpu().numpy(), g.state.cpu().numpy()))
        g.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    # and the reference configuration itself is right for a few of the fields
    orc = R.Oracle("ntsc")
    for k in (0, 1, 513, 1029):
        c = orc.new_crt(outw, outh, R.FMT_BGRA)
        c.set("scanlines", 1)
        c.settings(imgs[k], format=R.FMT_BGRA, w=w, h=h, as_color=1, field=fields[k], frame=0)
        for _ in range(2):
            c.analog[:] = 0
            c.modulate()
            c.demodulate(24)
        np.testing.assert_array_equal(outs[1][0][k].reshape(-1), c.out, err_msg="field %d" % k)


def _random_case(rng):
    w = int(rng.choice([1, 2, 3, 5, 17, 64, 100, 333, 640, 753, 800, 1281]))
    h = int(rng.choice([1, 2, 7, 48, 100, 236, 237, 480, 601]))
    outw = int(rng.choice([1, 3, 4, 5, 31, 32, 33, 250, 640, 1283]))
    outh = int(rng.choice([1, 17, 120, 239, 240, 241, 480, 500, 777]))
    ifmt, ofmt = int(rng.integers(0, 6)), int(rng.integers(0, 6))
    raw = int(rng.integers(0, 2))
    skw = dict(as_color=int(rng.integers(0, 2)), raw=raw, hue=int(rng.integers(-400, 800)))
    if raw and w <= 640 and h <= 200:
        # offsets only where the active rectangle still fits the raster: beyond that crt_modulate scribbles
        # over neighbouring lines (crt_ntsc.c:322) -- outside the contract, the library refuses it
        skw.update(xoffset=int(rng.integers(0, 3)) * 4, yoffset=int(rng.integers(0, 3)))
    knobs = dict(hue=int(rng.integers(-360, 720)), brightness=int(rng.integers(-40, 40)), contrast=int(rng.integers(0, 400)),
                 saturation=int(rng.integers(-5, 40)), black_point=int(rng.integers(-10, 10)),
                 white_point=int(rng.integers(50, 150)), scanlines=int(rng.integers(0, 2)), blend=int(rng.integers(0, 2)),
                 v_fac=int(rng.choice([0, 0, 0, 7, 100])))
    noise = int(rng.choice([0, 1, 24, 77, 300]))
    return ("ntsc", outw, outh, ofmt, w, h, ifmt, noise, skw, knobs)


def _random_wide_case(rng):
    """_random_case with picture widths the wide-run decoder takes (crt_decode4.hip: outw from ~1650 up, 4-byte pixels, no blend);
    three-byte formats, blend and the narrower widths in the list fall back to the lane-per-scanline kernel -- also a case"""
    case = list(_random_case(rng))
    case[0] = str(rng.choice(["ntsc", "ntscp0", "snes", "temp", "nesrgb"]))
    case[1] = int(rng.choice([1500, 1650, 1664, 1700, 1919, 1920, 1921, 2047, 2560, 3001, 3840, 4095]))
    case[2] = int(rng.choice([1, 120, 240, 241, 480, 777, 1080, 1200, 2160]))
    case[3] = int(rng.choice([R.FMT_RGBA, R.FMT_BGRA, R.FMT_ARGB, R.FMT_ABGR, R.FMT_RGBA, R.FMT_BGRA, R.FMT_RGB]))
    case[9] = dict(case[9], blend=int(rng.integers(0, 8) == 0))
    return tuple(case)


@pytest.mark.parametrize("lpw", [8, 16])
@pytest.mark.parametrize("seed", range(16))
def test_random_wide_configurations(crtlib, seed, lpw):
    """random wide pictures, knobs, formats and systems through the lane shape (wide-run decoder where it applies), with each of
    its two instantiations pinned"""
    rng = np.random.default_rng(7000 + seed)
    _run_case(crtlib, _random_wide_case(rng), fused=bool(seed & 1), steps=2, n=2, shape=1, wide_lpw=lpw)


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations(crtlib, seed):
    """odd / tiny / large geometries, all format pairs, random knobs: stagewise AND fused, 2 steps each"""
    rng = np.random.default_rng(1000 + seed)
    case = _random_case(rng)
    _run_case(crtlib, case, fused=bool(seed & 1), steps=2, n=2, shape=1)


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_scanline_parallel_shape(crtlib, seed):
    """the same random configurations forced onto the scanline-parallel kernels"""
    rng = np.random.default_rng(1000 + seed)
    case = _random_case(rng)
    _run_case(crtlib, case, fused=bool(seed & 1), steps=2, n=2, shape=2)


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_bloom_lane_per_scanline(crtlib, seed):
    """random configurations of the bloom builds through the beam-width sort + lane-per-scanline decoder"""
    rng = np.random.default_rng(5000 + seed)
    case = ("ntscbloom", "snesbloom", "pv1kbloom")[seed % 3], *_random_case(rng)[1:]
    _run_case(crtlib, case, fused=bool(seed & 1), steps=2, n=3, shape=1)


@pytest.mark.parametrize("name,n,w,h,noise", [("ntsc", 4096, 640, 480, 24), ("ntsc", 64, 1920, 1080, 0), ("ntsc", 512, 1920, 1080, 0),
                                              ("ntscbloom", 4096, 640, 480, 24)])
def test_full_size_batch_properties(crtlib, name, n, w, h, noise):
    """BASELINE configs[1] at the bench's full batch (4096 fields of 640x480, noise 24), configs[2]'s per-GPU
    share (512 frames of 1920x1080 over 8 GPUs = 64, noise 0) and ALL of configs[2] on one GPU (512 frames: the wide-run decoder
    with 16 scanlines per wave and the large signal tile chosen by wave count, as in the bench's `1080p_batch512`): (a) replication -- fields that carry the same image,
    parity and state produce the same picture and state wherever they sit in the batch; (b) a checksum over all
    pictures is reproducible from run to run; (c) one field of EVERY (image, parity) class equals the oracle -- with (a) that
    covers every field of the batch."""
    import torch
    uniq = 8 if w <= 640 else 2                           # (synthesising 1080p images on the host is slow)
    base = np.stack([R.synth_image(w, h, 4, 7000 + k) for k in range(uniq)])
    if name.endswith("bloom"):
        # a different brightness ramp down every image: the beam energy, hence the beam width, changes from line to line
        # and from image to image -- a million scanlines over a few dozen widths through the sort of crt_decode3.hip
        for k in range(uniq):
            ramp = (np.arange(h)[:, None, None] * (k + 1) * 37 // h) % 256
            base[k] = (base[k].astype(np.int32) * ramp // 255).astype(np.uint8)
    imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")       # + the spare row
    data = imgs.repeat(n // uniq, 1, 1, 1)[:, :h]
    fields = [(k // uniq) & 1 for k in range(n)]                  # parity changes every `uniq` fields
    sums = []
    for run in range(2):
        g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, name, device=0)
        g.scanlines = 1
        s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=list(fields), frame=0)
        g.fieldpass(s, noise)
        g.synchronize()
        out, st = g.out, g.state
        # (a) field k and field k + 2 * uniq share image (k % uniq) and parity ((k // uniq) & 1)
        assert torch.equal(out[:-2 * uniq], out[2 * uniq:]), "replicated fields differ"
        assert torch.equal(st[:-2 * uniq], st[2 * uniq:])
        sums.append(int(out.to(torch.int64).sum().item()) ^ int(st.to(torch.int64).sum().item()))
        # the classes: field k is (image k % uniq, parity (k // uniq) & 1); one representative each, spread over the batch
        reps = [k + 2 * uniq * ((k * 7) % (n // (2 * uniq))) for k in range(2 * uniq)]
        if run == 0:
            host = out[reps].cpu().numpy()
        g.close()
    assert sums[0] == sums[1]
    orc = R.Oracle(name)
    for j, k in enumerate(reps):
        c = orc.new_crt(w, h, R.FMT_BGRA)
        c.set("scanlines", 1)
        c.settings(np.concatenate([base[k % uniq], base[k % uniq][-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1,
                   field=fields[k], frame=0)
        c.modulate()
        c.demodulate(noise)
        np.testing.assert_array_equal(host[j].reshape(-1), c.out, err_msg="field %d of the full batch" % k)


@pytest.mark.parametrize("name,n,noise", [("vhs", 2048, 12), ("nesp0", 4096, 12)])
def test_full_size_batch_properties_vhs_nes(crtlib, name, n, noise):
    """The bench's full batches of BASELINE configs[3] (VHS 832x624, 2048 fields, noise 12: the libc rand() stream per field)
    and configs[4] (NES, CRT_CHROMA_PATTERN 0, 4096 fields of 256x240 PPU pixels -> 640x480) -- VERDICT round 3, parity
    hole (a).  Same three properties as test_full_size_batch_properties: (a) fields of the same class (image, parity or dot
    crawl offset, generator seed) give the same picture and state wherever they sit in the batch; (b) the checksum over
    everything is reproducible; (c) one field of every class equals the oracle."""
    import ctypes as C
    import torch
    libc = C.CDLL(None)
    nes = name.startswith("nes")
    uniq = 6 if nes else 4                                 # NES: classes = image x dot crawl offset (k % 3)
    period = uniq if nes else 2 * uniq                     # VHS: classes = image x field parity
    if nes:
        w, h, outw, outh = 256, 240, 640, 480
        base = np.stack([R.synth_ppu(w, h, 7100 + k) for k in range(uniq)]).astype(np.int16)
        imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")
        data = imgs.repeat((n + uniq - 1) // uniq, 1, 1)[:n, :h]
        dco = [k % 3 for k in range(n)]                    # uniq is a multiple of 3: field k and k + uniq share image and offset
    else:
        w, h, outw, outh = 832, 624, 832, 624
        base = np.stack([R.synth_image(w, h, 4, 7200 + k, "random" if k & 1 else "bars") for k in range(uniq)])
        imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")
        data = imgs.repeat(n // uniq, 1, 1, 1)[:, :h]
        fields = [(k // uniq) & 1 for k in range(n)]
        seeds = [1 + 31 * (k % period) for k in range(n)]  # one generator seed per class
    sums = []
    for run in range(2):
        g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, name, device=0)
        g.scanlines = 1
        if nes:
            s = crtlib.Settings(data, hue=0, dot_crawl_offset=list(dco))
        else:
            g.srand(seeds)
            s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=list(fields), frame=0)
        g.fieldpass(s, noise)
        g.synchronize()
        out, st = g.out, g.state
        assert torch.equal(out[:-period], out[period:]), "replicated fields differ"
        assert torch.equal(st[:-period], st[period:])
        sums.append(int(out.to(torch.int64).sum().item()) ^ int(st.to(torch.int64).sum().item()))
        reps = [k + period * ((k * 7) % (n // period - 1)) for k in range(period)]
        if run == 0:
            host, hst = out[reps].cpu().numpy(), st[reps].cpu().numpy()
        g.close()
    assert sums[0] == sums[1]
    orc = R.Oracle(name)
    for j, k in enumerate(reps):
        c = orc.new_crt(outw, outh, R.FMT_BGRA)
        c.set("scanlines", 1)
        if nes:
            c.settings(np.concatenate([base[k % uniq], base[k % uniq][-1:]]).astype(np.uint16), w=w, h=h, dot_crawl_offset=dco[k], hue=0)
        else:
            c.settings(np.concatenate([base[k % uniq], base[k % uniq][-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1,
                       field=fields[k], frame=0, do_aberration=0)
            libc.srand(seeds[k])
        c.modulate()
        c.demodulate(noise)
        np.testing.assert_array_equal(host[j].reshape(-1), c.out, err_msg="%s: field %d of the full batch" % (name, k))
        assert (int(hst[j, crtlib.ST_HSYNC]), int(hst[j, crtlib.ST_VSYNC]), int(hst[j, crtlib.ST_RN])) == \
               (c.get("hsync"), c.get("vsync"), c.get("rn")), "%s: state of field %d" % (name, k)


def test_a_kept_graph_keeps_its_tables(crtlib):
    """the encoder's cached tables (skeleton fields: a function of the burst table, i.e. of the hue) used to swap
    between two buffer sets, so a later capture or an eager call with other settings rewrote the set an EARLIER graph was reading
    and that graph silently encoded with the wrong burst.  Now a set a graph may be reading is never written again: graph A
    (hue 0), an eager pass with hue 40, graph B (hue 80), another eager pass with hue 120 -- then A and B are replayed and must
    give what eager passes with THEIR settings give; crthip_table_generation counts the rebuilds."""
    import torch
    n, w, h = 4, 640, 480
    imgs = _padded(np.stack([R.synth_image(w, h, 4, 60 + k) for k in range(n)]))

    def settings(hue):
        return crtlib.Settings(imgs, format=crtlib.FMT_BGRA, field=[k & 1 for k in range(n)], frame=0, hue=hue)

    def eager_result(hue):
        g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
        g.scanlines = 1
        s = settings(hue)
        g.fieldpass(s, 24)
        g.synchronize()
        out = g.out.clone()
        g.close()
        return out
    want = {hue: eager_result(hue) for hue in (0, 80)}
    assert not torch.equal(want[0], want[80])
    g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
    g.scanlines = 1
    g.reserve(n)
    side = torch.cuda.Stream()
    g.use_stream(side)
    s0 = settings(0)
    g._load_field_state(s0)
    torch.cuda.synchronize()
    state0 = g.state.clone()
    graphs, gens = {}, {}
    for hue in (0, 40, 80, 120):
        s = settings(hue)
        p = g.params(s, 24)
        g.state.copy_(state0)
        torch.cuda.synchronize()
        if hue in (0, 80):
            graphs[hue] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graphs[hue], stream=side):
                g.fieldpass(s, 24, params=p)
            gens[hue] = g.table_generation()
        else:
            g.fieldpass(s, 24, params=p)
            g.synchronize()
    assert gens[80] > gens[0] and g.table_generation() > gens[80]
    for hue in (0, 80, 0):
        g.state.copy_(state0)
        g.out.zero_()
        torch.cuda.synchronize()
        graphs[hue].replay()
        torch.cuda.synchronize()
        assert torch.equal(g.out, want[hue]), "the graph captured with hue %d no longer encodes with its own tables" % hue
    del graphs
    g.close()


def test_a_kept_nes_graph_keeps_both_of_its_tables(crtlib):
    """a NES context caches TWO tables that are rebuilt independently -- the skeleton fields (burst table, i.e. hue)
    and the PPU sample table (black / white point) -- and one shared "a graph reads the current set" flag let this sequence through:
    capture graph A; an eager pass with another black point moves the sample table to a fresh buffer and clears the flag; an eager
    pass with another hue then rebuilds the skeleton IN PLACE -- the buffer A still reads.  The flags are per table now: A, replayed
    after both eager passes, must give what an eager pass with A's settings gives."""
    import torch
    n, outw, outh = 3, 640, 480
    ppu = np.stack([R.synth_ppu(256, 240, 500 + k) for k in range(n)])
    full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
    full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")

    def settings(hue):
        return crtlib.Settings(full[:, :240], hue=hue, dot_crawl_offset=[k % 3 for k in range(n)])

    def eager_result(hue, black):
        g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "nes", device=0)
        g.scanlines = 1
        g.black_point = black
        g.fieldpass(settings(hue), 12)
        g.synchronize()
        out = g.out.clone()
        g.close()
        return out
    want = eager_result(0, 0)
    assert not torch.equal(want, eager_result(90, 0)) and not torch.equal(want, eager_result(0, 6))
    g = crtlib.CRT(n, outw, outh, crtlib.FMT_BGRA, "nes", device=0)
    g.scanlines = 1
    g.reserve(n)
    side = torch.cuda.Stream()
    g.use_stream(side)
    s0 = settings(0)
    g._load_field_state(s0)
    torch.cuda.synchronize()
    state0 = g.state.clone()
    p0 = g.params(s0, 12)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g.fieldpass(s0, 12, params=p0)
    gen0 = g.table_generation()
    for hue, black in ((0, 6), (90, 6)):                  # the sample table moves; then the skeleton is rebuilt
        g.black_point = black
        s = settings(hue)
        g.state.copy_(state0)
        torch.cuda.synchronize()
        g.fieldpass(s, 12, params=g.params(s, 12))
        g.synchronize()
    assert g.table_generation() >= gen0 + 2
    for _ in range(2):
        g.state.copy_(state0)
        g.out.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g.out, want), "the NES graph no longer encodes with the tables it was captured with"
    del graph
    g.close()


@pytest.mark.parametrize("name,n,noise,fpb4", [("ntsc", 24, 0, 0), ("ntsc", 24, 150, 0), ("ntsc", 520, 60, 1), ("snes", 24, 100, 0), ("pv1k", 12, 90, 0),
                                               ("nes", 24, 80, 0), ("ntscbloom", 24, 40, 0), ("ntsc", 24, 150, 1), ("snes", 24, 100, 1), ("nes", 24, 80, 1)])
def test_wild_sync_states_against_the_oracle(crtlib, name, n, noise, fpb4):
    """Fields started from sync states far from lock -- vertical sync candidates in the middle of the picture under heavy noise,
    hsync values that put the search window or the burst window into the picture, values near the line end (windows that wrap)
    -- next to ordinary ones, in one batch.  Every field, three consecutive field-passes, against the oracle: hsync, vsync, rn,
    the burst integrators the chain starts from (the ccf preset of crt_modulate is applied by the sync kernel's own lanes in a
    fused pass, k_hsync_wave preset_ccf) and every picture byte.  fpb4: the sync kernel pinned to FOUR fields per workgroup
    (CRTHIP_SYNC_KERNEL=3, read when the context is created; batches below 768 fields take one field per workgroup by themselves)
    -- the shape in which wave 0's chain lanes preset and integrate the other three waves' fields, for the systems with line
    classes (snes, nes: class -> ccf row through CCF_SHIFT) as well (the PV-1000's 25 chain lanes per field do not
    fit four fields into a wave).  Round 6: these are also the fields whose decoder windows do not fit the padded signal lines of
    the fused path and are decoded from the per-field scratch copy (crt_sync.hip).
    (Rounds 4's speculative sync chain was tested with these inputs; the chain is gone, the inputs stayed.)"""
    import torch
    nes = name == "nes"
    w, h = (256, 240) if nes else (640, 480)
    hs0 = [0, 3, 30, 60, 140, 300, 500, 700, 880, 892, 905, 909]
    vs0 = [0, 4, 9, 100, 180, 250, 255, 258, 261]
    uniq = min(n, 24)
    if nes:
        base = np.stack([R.synth_ppu(w, h, 8100 + k) for k in range(uniq)]).astype(np.int16)
    else:
        base = np.stack([R.synth_image(w, h, 4, 8100 + k, "bars" if k % 3 == 1 else "random") for k in range(uniq)])
        base[2::3] //= 16                                  # dark pictures: a noisy candidate line crosses the threshold more easily
    imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")
    reps = (n + uniq - 1) // uniq
    data = (imgs.repeat(reps, 1, 1) if nes else imgs.repeat(reps, 1, 1, 1))[:n, :h]
    hs = [hs0[(k // uniq + k) % len(hs0)] for k in range(n)]
    vs = [vs0[(k // uniq * 5 + k) % len(vs0)] for k in range(n)]
    fields = [k & 1 for k in range(n)]
    outs = {}
    for spec in (1,):      # (one pass; the dict keeps the shape the comparisons below were written for)
        import os
        saved = os.environ.get("CRTHIP_SYNC_KERNEL")
        if fpb4:
            os.environ["CRTHIP_SYNC_KERNEL"] = "3"
        try:
            g = crtlib.CRT(n, 640, 480, crtlib.FMT_BGRA, name, device=0)
        finally:
            if fpb4:
                if saved is None:
                    del os.environ["CRTHIP_SYNC_KERNEL"]
                else:
                    os.environ["CRTHIP_SYNC_KERNEL"] = saved
        g.scanlines = 1
        g.state[:, crtlib.ST_HSYNC] = torch.tensor(hs, dtype=torch.int32, device="cuda:0")
        g.state[:, crtlib.ST_VSYNC] = torch.tensor(vs, dtype=torch.int32, device="cuda:0")
        if nes:
            s = crtlib.Settings(data, hue=0, dot_crawl_offset=[k % 3 for k in range(n)])
        elif name in ("snes", "pv1k"):
            s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=list(fields), frame=0, dot_crawl_offset=[k % 3 for k in range(n)])
        else:
            s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=list(fields), frame=0)
        per = []
        for step in range(3):
            g.fieldpass(s, noise)
            g.synchronize()
            per.append((g.out.cpu().numpy().copy(), g.state.cpu().numpy().copy()))
        outs[spec] = per
        g.close()
    orc = R.Oracle(name)
    check = range(n) if n <= 64 else list(range(0, n, 7)) + [n - 1]
    checked = 0
    for k in check:
        c = orc.new_crt(640, 480, R.FMT_BGRA)
        c.set("scanlines", 1)
        c.set("hsync", hs[k])
        c.set("vsync", vs[k])
        pad = np.concatenate([base[k % uniq], base[k % uniq][-1:]])
        if nes:
            c.settings(pad.astype(np.uint16), w=w, h=h, dot_crawl_offset=k % 3, hue=0)
        else:
            c.settings(pad, format=R.FMT_BGRA, w=w, h=h, as_color=1, field=fields[k], frame=0)
            if name in ("snes", "pv1k"):
                c.sset("dot_crawl_offset", k % 3)
        for step in range(3):
            c.analog[:] = 0                                # batch semantics: every field-pass starts from a clean analog[]
            if nes:
                c.sset("field_initialized", 0)
            c.modulate()
            hs_before = c.get("hsync")
            c.demodulate(noise, trace=True)
            if R.reads_past_inp(orc, c.trace, c.get("vsync"), hs_before):
                break                                      # the reference itself reads past inp[] + 16 here (sync far out on the field's last analog line): UB, not compared
            checked += 1
            st = outs[1][step][1][k]
            what = "%s noise %d field %d (hsync0 %d vsync0 %d) step %d" % (name, noise, k, hs[k], vs[k], step)
            assert (int(st[crtlib.ST_HSYNC]), int(st[crtlib.ST_VSYNC]), int(st[crtlib.ST_RN])) == (c.get("hsync"), c.get("vsync"), c.get("rn")), what
            np.testing.assert_array_equal(st[crtlib.ST_CCF:crtlib.ST_CCF + 25].reshape(5, 5)[:orc.vper, :orc.ccs], c.ccf, err_msg=what + " ccf")
            np.testing.assert_array_equal(outs[1][step][0][k].reshape(-1), c.out, err_msg=what + " out")
    assert checked >= 2 * len(list(check)), "too many fields fell into the reference's undefined behaviour to mean anything"


def test_full_batch_every_field_checked_against_the_oracle(crtlib):
    """the 4096-field batch of BASELINE configs[1] with 64 DISTINCT images x 2 field parities; a per-field
    checksum (plain byte sum + position-weighted sum, computed on the device) of EVERY one of the 4096 pictures is
    compared with the checksum of the oracle's picture for that (image, parity), and so is every field's state."""
    import torch
    n, w, h, uniq, noise = 4096, 640, 480, 64, 24
    base = np.stack([R.synth_image(w, h, 4, 9000 + k, "random" if k % 4 else "bars") for k in range(uniq)])
    imgs = torch.from_numpy(np.concatenate([base, base[:, -1:]], axis=1)).to("cuda:0")       # + the spare row
    data = imgs.repeat(n // uniq, 1, 1, 1)[:, :h]
    fields = [(k // uniq) & 1 for k in range(n)]                   # image k % 64, parity (k // 64) & 1
    g = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
    g.scanlines = 1
    s = crtlib.Settings(data, format=crtlib.FMT_BGRA, field=list(fields), frame=0)
    g.fieldpass(s, noise)
    g.synchronize()
    flat = g.out.reshape(n, -1).to(torch.int64)
    wts = (torch.arange(flat.shape[1], device=flat.device, dtype=torch.int64) % 65521) + 1
    sums = torch.stack([flat.sum(dim=1), (flat * wts).sum(dim=1)], dim=1).cpu().numpy()
    st = g.state.cpu().numpy()
    g.close()
    orc = R.Oracle("ntsc")
    wn = (np.arange(w * h * 4, dtype=np.int64) % 65521) + 1
    want = {}
    for img in range(uniq):
        for par in (0, 1):
            c = orc.new_crt(w, h, R.FMT_BGRA)
            c.set("scanlines", 1)
            c.settings(np.concatenate([base[img], base[img][-1:]]), format=R.FMT_BGRA, w=w, h=h, as_color=1, field=par, frame=0)
            c.modulate()
            c.demodulate(noise)
            o = c.out.astype(np.int64)
            want[(img, par)] = (int(o.sum()), int((o * wn).sum()), c.get("hsync"), c.get("vsync"), c.get("rn"))
    for k in range(n):
        a, b, hs, vs, rn = want[(k % uniq, fields[k])]
        assert (int(sums[k, 0]), int(sums[k, 1])) == (a, b), "field %d: picture checksum differs from the oracle" % k
        assert (int(st[k, crtlib.ST_HSYNC]), int(st[k, crtlib.ST_VSYNC]), int(st[k, crtlib.ST_RN])) == (hs, vs, rn), "field %d state" % k


@pytest.mark.parametrize("name,shape", [("ntsc", 0), ("ntsc", 1), ("ntscbloom", 1), ("nes", 0), ("ntsc-1080p", 1), ("ntsc-many", 0)])
def test_fieldpass_is_graph_capturable(crtlib, name, shape):
    """crthip_fieldpass only enqueues kernels on the context's stream (no allocation, no synchronisation once the
    workspace is reserved -- the bloom build's sort scratch included), so a caller can capture the launch sequence into a
    HIP graph and replay it -- also as the very FIRST call of a context: the tables the context caches (skeleton fields,
    the NES sample table) are then built outside the capture.  The 1080p case runs the wide-run
    decoder (crt_decode4.hip); the bloom case is the one whose two memset nodes broke the second replay (round 4)."""
    import torch
    n, w, h = 6, 640, 480
    ow = oh = None
    if name.endswith("-1080p"):
        name, n, w, h = name[:-6], 3, 1920, 1080
    if name.endswith("-many"):
        # round 6: from 512 fields on the margin kernel runs on the context's internal stream beside the active-video kernel --
        # under capture a fork and a join inside the graph (crt_encode.hip, launch_encoder)
        name, n, w, h, ow, oh = name[:-5], 640, 64, 48, 96, 240
    nes = name == "nes"
    if nes:
        ppu = np.stack([R.synth_ppu(256, 240, 40 + k) for k in range(n)])
        full = torch.zeros((n, 241, 256), dtype=torch.int16, device="cuda:0")
        full[:, :240] = torch.from_numpy(ppu.astype(np.int16)).to("cuda:0")

        def settings():
            return crtlib.Settings(full[:, :240], hue=0, dot_crawl_offset=[k % 3 for k in range(n)])
    else:
        imgs = _padded(np.stack([R.synth_image(w, h, 4, 40 + k) for k in range(n)]))

        def settings():
            return crtlib.Settings(imgs, format=crtlib.FMT_BGRA, field=[k & 1 for k in range(n)], frame=0)

    def context():
        g = crtlib.CRT(n, ow or w, oh or h, crtlib.FMT_BGRA, name, device=0)
        g.scanlines = 1
        g.set_shape(shape)
        g.reserve(n)
        return g
    g = context()
    s = settings()
    p = g.params(s, 24)
    side = torch.cuda.Stream()
    g.use_stream(side)
    # eager: two consecutive field-passes (state carries over)
    g._load_field_state(s)
    torch.cuda.synchronize()
    state0 = g.state.clone()
    eager = []
    for _ in range(2):
        g.fieldpass(s, 24, params=p)
        g.synchronize()
        eager.append((g.out.clone(), g.state.clone()))
    # captured: the same launch sequence, replayed twice from the same initial state -- once on the context that has
    # run eagerly before, once on a FRESH context whose first call ever is the captured one
    for fresh in (False, True):
        c = context() if fresh else g
        c.use_stream(side)
        if fresh:
            c._load_field_state(s)
        c.state.copy_(state0)
        c.out.zero_()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            c.fieldpass(s, 24, params=p)
        for k in range(2):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(c.out, eager[k][0]), "fresh=%s replay %d: picture differs from the eager launch sequence" % (fresh, k)
            assert torch.equal(c.state, eager[k][1]), "fresh=%s replay %d: state differs" % (fresh, k)
        del graph
        c.use_stream(None)
        c.close()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
