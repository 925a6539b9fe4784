"""GPU tests of the drop-in layer: the reference's own API (crt_init / crt_modulate / crt_demodulate
on a caller-owned struct CRT) served by libntsccrt_hip_<sys>.so, compared bit-for-bit with the real
reference (oracle/_ref, prebuilt -- /root/reference is not needed at run time) or, where that is
missing, with the oracle.  Also runs the reference's UNCHANGED crt_main.c driver linked against the
HIP library next to the pure-reference binary and compares the written images byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import crtref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()


def _checker(name):
    return R.RefLib(name) if R.have_ref(name) else R.Oracle(name)


def test_crt_main_accumulate_loop_through_dropin():
    """crt_main.c:221-255 verbatim: blend=1, scanlines=1, 4 x (field, field^1), frame toggles;
    every struct member the caller can see is compared after every call."""
    hip = R.RefLib("ntsc", dropin=True)
    chk = _checker("ntsc")
    img = R.synth_image(640, 480, 4, 1, "bars")
    a, b = hip.new_crt(832, 624, R.FMT_BGRA), chk.new_crt(832, 624, R.FMT_BGRA)
    for c in (a, b):
        c.settings(img, format=R.FMT_BGRA, w=640, h=480, as_color=1, hue=0, raw=0)
        c.set("blend", 1)
        c.set("scanlines", 1)
    for err in range(4):
        for c in (a, b):
            c.modulate()
        np.testing.assert_array_equal(a.analog, b.analog)
        np.testing.assert_array_equal(a.ccf, b.ccf)
        for c in (a, b):
            c.demodulate(24)
        R.compare_state(a, b, "accumulate %d even" % err)
        for c in (a, b):
            c.sset("field", c.sget("field") ^ 1)
            c.modulate()
            c.demodulate(24)
            if err % 2 == 0:
                c.sset("frame", c.sget("frame") ^ 1)
        R.compare_state(a, b, "accumulate %d odd" % err)


def test_caller_edits_between_calls_are_honoured():
    """callers poke struct CRT directly (crt_main.c:317-391,:430): knobs, analog[], the out image."""
    hip = R.RefLib("ntsc", dropin=True)
    chk = _checker("ntsc")
    img = R.synth_image(320, 200, 3, 5)
    a, b = hip.new_crt(640, 480, R.FMT_RGB), chk.new_crt(640, 480, R.FMT_RGB)
    for c in (a, b):
        c.settings(img, format=R.FMT_RGB, w=320, h=200, as_color=1, raw=1)
        c.modulate()
        c.demodulate(10)
    R.compare_state(a, b, "first")
    for c in (a, b):
        c.analog[:] = 0                       # memset(crt.analog, 0, ...) of the live driver
        c.set("saturation", 17)
        c.set("brightness", -4)
        c.set("hue", 20)
        c.set("blend", 1)
        c.out[::7] = 99                       # scribble on the picture
        c.sset("raw", 0)
        c.modulate()
        c.demodulate(33)
    R.compare_state(a, b, "after edits")


def test_ntsc_pattern0_dropin():
    hip = R.RefLib("ntscp0", dropin=True)
    chk = _checker("ntscp0")
    img = R.synth_image(640, 480, 4, 11)
    a, b = hip.new_crt(640, 480, R.FMT_BGRA), chk.new_crt(640, 480, R.FMT_BGRA)
    for c in (a, b):
        c.settings(img, format=R.FMT_BGRA, w=640, h=480, as_color=1)
    for step in range(3):
        for c in (a, b):
            c.modulate()
            c.demodulate(24)
            c.sset("field", c.sget("field") ^ 1)
        R.compare_state(a, b, "ntscp0 step %d" % step)


@pytest.mark.parametrize("name,outsz", [("nes", (640, 480)), ("nesp0", (512, 480))])
def test_nes_dropin(name, outsz):
    hip = R.RefLib(name, dropin=True)
    chk = _checker(name)
    a, b = hip.new_crt(outsz[0], outsz[1], R.FMT_BGRA), chk.new_crt(outsz[0], outsz[1], R.FMT_BGRA)
    for step in range(4):
        ppu = R.synth_ppu(256, 240, 7 + step)
        pad = np.concatenate([ppu, ppu[-1:]], axis=0)
        for c in (a, b):
            c.settings(pad, w=256, h=240, dot_crawl_offset=step % 3, hue=(step * 40) % 360)
            c.modulate()
            c.demodulate([0, 12, 24, 5][step])
        R.compare_state(a, b, "%s step %d" % (name, step))


@pytest.mark.parametrize("aberration", [0, 1])
def test_vhs_dropin_shares_the_libc_rand_stream(aberration):
    """video_convert.c semantics: the program seeds rand(), crt_modulate draws the aberration height
    from it, crt_demodulate consumes ~500k values per field.  The HIP library borrows the generator
    from libc, advances it on the GPU and puts it back: after every call the NEXT rand() of the
    process must be what it would have been with the reference."""
    import ctypes as C
    libc = C.CDLL(None)
    hip = R.RefLib("vhs", dropin=True)
    chk = _checker("vhs")
    img = R.synth_image(832, 624, 4, 9)
    a, b = hip.new_crt(832, 624, R.FMT_BGRA), chk.new_crt(832, 624, R.FMT_BGRA)
    for c in (a, b):
        c.settings(img, format=R.FMT_BGRA, w=832, h=624, as_color=1, do_aberration=aberration)
    probes = []
    for c in (a, b):
        libc.srand(4242)
        seq = []
        for step in range(3):
            c.modulate()
            c.demodulate(12)
            seq.append(libc.rand())                       # the process's own next draw
            c.sset("field", c.sget("field") ^ 1)
        probes.append(seq)
    assert probes[0] == probes[1]
    np.testing.assert_array_equal(a.analog, b.analog)
    np.testing.assert_array_equal(a.inp, b.inp)
    for f in R.STATE_FIELDS:
        assert a.get(f) == b.get(f), f
    if not aberration:
        np.testing.assert_array_equal(a.out, b.out)


def _write_ppm(path, w, h, seed):
    img = R.synth_image(w, h, 3, seed, "bars")
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


@pytest.mark.parametrize("flags,outw,outh,noise,hue", [("-op", 640, 480, 0, 0), ("-o", 640, 480, 24, 0),
                                                       ("-om", 832, 624, 12, 90), ("-oa", 640, 480, 0, 0)])
def test_unchanged_crt_main_driver(tmp_path, flags, outw, outh, noise, hue):
    """BASELINE configs[0]: ./ntsc -op 640 480 0 0 in.ppm out.ppm -- same driver source, two libraries."""
    ref_cli = os.path.join(R.REF_DIR, "ntsc_cli")
    hip_cli = os.path.join(R.PKG_LIB, "ntsc_cli_hip")
    if not (os.path.exists(ref_cli) and os.path.exists(hip_cli)):
        pytest.skip("driver binaries not prebuilt (they are built where /root/reference exists)")
    src = str(tmp_path / "in.ppm")
    _write_ppm(src, 640, 480, 1)
    outs = []
    for exe, tag in ((ref_cli, "ref"), (hip_cli, "hip")):
        out = str(tmp_path / ("out_%s.ppm" % tag))
        r = subprocess.run([exe, flags, str(outw), str(outh), str(noise), str(hue), src, out],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1], "driver output differs between the reference and the HIP library"
