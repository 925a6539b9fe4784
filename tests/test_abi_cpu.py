"""CPU-side checks of the native library: it builds, loads, exports every symbol that
include/crt_hip.h declares, and its C89 host setup (crt_setup.c) derives the same constants
as the oracle.  No compute calls -- there is no GPU here."""
import ctypes as C
import numpy as np
import os
import re
import subprocess

import pytest

import crtref as R

ROOT = R.ROOT
PKG = os.path.join(ROOT, "ntsc-crt_amd")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import crtlib
    return crtlib


def _declared_functions(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crthip_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    L = lib.load_library()
    names = _declared_functions(os.path.join(ROOT, "include", "crt_hip.h"))
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert L.crthip_abi_version() == 6


def test_node_library_exports_its_header(lib):
    """include/crt_hip_node.h -> libcrthip_node.so (links RCCL; no compute calls here)"""
    path = os.path.join(ROOT, "ntsc-crt_amd", "lib", "libcrthip_node.so")
    names = [n for n in _declared_functions(os.path.join(ROOT, "include", "crt_hip_node.h")) if n.startswith("crthip_node_")]
    assert len(names) >= 10
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    for n in names:
        assert n in exported, n


def test_c_programs_over_the_node_library_are_built(lib):
    """tests/node_probe.c (parity of the multi-shard entries) and tools/node_bench.c (batches in flight through the C ABI
    alone) are plain C over include/crt_hip_node.h; the build links them against libcrthip_node.so and nothing else of ours"""
    for exe in ("node_probe", "node_bench"):
        path = os.path.join(ROOT, "ntsc-crt_amd", "lib", exe)
        assert os.path.exists(path), exe
        out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        wanted = set(line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip())
        ours = sorted(n for n in wanted if n.startswith("crthip_"))
        assert ours and all(n.startswith(("crthip_node_", "crthip_")) for n in ours)
        assert not any(n.startswith(("hip", "nccl", "orc_")) for n in wanted), "the C programs go through the C ABI only"


def test_struct_sizes_match_header(lib):
    hdr = '#include "crt_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu\\n", sizeof(crthip_params), sizeof(crthip_state), sizeof(crthip_line));return 0;}\n'
    exe = "/tmp/crthip_sizes"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                   input=hdr.encode(), check=True)
    a, b, c = map(int, subprocess.run([exe], capture_output=True, check=True).stdout.split())
    assert a == C.sizeof(lib.Params)
    assert b == 4 * lib.STATE_INTS
    assert c == 4 * lib.LINE_INTS


ALL_SYSTEMS = ["ntsc", "vhs", "nes", "nesp0", "ntscp0", "snes", "pv1k", "temp", "nesrgb"]


@pytest.mark.parametrize("hue", [0, 25, -42, -57, -79, -359, 700])
@pytest.mark.parametrize("name", ALL_SYSTEMS + ["ntscbloom", "pv1kbloom"])
def test_carrier_tables_and_geometry_match_the_oracle(lib, name, hue):
    """crthip_params_finalize on the host: the burst table (row = line class + dot_crawl_offset, or field == frame)
    must reproduce every burst sample the oracle's crt_modulate writes -- including negative hues, where C's
    truncating % and / matter -- and the active rectangle must be where the oracle puts the picture."""
    import numpy as np
    orc = R.Oracle(name)
    sysd = orc.sys
    w, h = 300, 200
    nes = orc.system == R.SYS_NES
    for dco, field, frame, raw in [(0, 0, 0, 0), (1, 1, 0, 1), (2, 1, 1, 0), (5, 0, 1, 1)]:
        c = orc.new_crt(640, 480, R.FMT_BGRA)
        if nes:
            img = np.full((h + 1, w), 0x30, dtype=np.uint16)
            c.settings(img, w=w, h=h, dot_crawl_offset=dco % 3, hue=hue)
        else:
            img = np.full((h + 1, w, 4), 255, dtype=np.uint8)
            kw = dict(format=R.FMT_BGRA, w=w, h=h, hue=hue)
            if orc.system != R.SYS_NESRGB:
                kw.update(as_color=1, field=field, frame=frame, raw=raw)
            if orc.system in R.DOT_CRAWL_SYSTEMS:
                kw.update(dot_crawl_offset=dco if orc.system in (R.SYS_PV1K, R.SYS_TEMP) else dco % 3)
            c.settings(img, **kw)
        c.modulate()
        p = lib.make_params(name, w=w, h=h, outw=640, outh=480, hue=hue, raw=0 if nes or orc.system == R.SYS_NESRGB else raw)
        an = c.analog.reshape(sysd.vres, sysd.hres)
        line_rows = orc.system in R.DOT_CRAWL_SYSTEMS
        eff = c.sget("dot_crawl_offset") if line_rows else 0
        for n in range(p.yo, p.yo + 6):
            row = (n % sysd.cc_vper) + eff if line_rows else int(field == frame)
            for t in range(sysd.cb_beg, sysd.cb_beg + sysd.cb_len):
                want = (sysd.blank_level + p.burst[row][t % sysd.cc_samples] * sysd.burst_level) >> 5
                assert an[n, t] == ((want + 128) % 256) - 128, (name, hue, dco, n, t)
        # geometry: the first active sample of the picture is white-ish, the sample before the rectangle is blank
        ys, xs = np.nonzero(an[:, sysd.av_beg - 8:] > 40)
        assert ys.min() == p.yo and ys.max() == p.yo + p.desth - 1, (name, raw)
        assert xs.min() + sysd.av_beg - 8 == p.xo, (name, raw, xs.min() + sysd.av_beg - 8, p.xo)


@pytest.mark.parametrize("name", ALL_SYSTEMS)
def test_host_setup_matches_oracle(lib, name):
    orc = R.Oracle(name)
    L = lib.load_library()
    sysid, pattern = lib.SYSTEMS[name]
    assert L.crthip_input_size(sysid, pattern) == orc.input_size
    assert L.crthip_hres(sysid, pattern) == orc.hres
    assert L.crthip_lines(sysid) == orc.bot - orc.top
    assert L.crthip_field_stride(sysid, pattern) >= orc.input_size + 16 + orc.av_len + 16
    p = lib.make_params(name, w=640, h=480, outw=832, outh=624, hue=25, mon_hue=340, brightness=4,
                        black_point=2, white_point=97)
    assert list(p.eq_lf) == list(orc.sys.eq_lf)
    assert list(p.eq_hf) == list(orc.sys.eq_hf)
    assert [list(r) for r in p.eq_g] == [list(r) for r in orc.sys.eq_g]
    assert list(p.iir_c) == list(orc.sys.iir_c)
    if name in ("ntsc", "vhs"):
        assert (p.destw, p.desth, p.xo, p.yo) == (753, 236, 156, 23)
    s, c = orc.sincos14(((340 % 360) + 33) * 8192 // 180)
    assert (p.huesn, p.huecs) == (s >> 11, c >> 11)
    assert p.bright == 4 - (orc.sys.black_level + 2)
    assert p.dx == ((orc.av_len - 1) << 12) // 832


@pytest.mark.parametrize("name", ["ntsc", "vhs", "snes", "pv1k", "temp", "nesrgb"])
def test_source_column_step_is_exact(lib, name):
    """crthip_params.col_step: the encoder's source column (x * w) / destw (crt_ntsc.c:272) as the high word of a running
    64-bit sum, one add per sample -- checked for every sample of odd, tiny and huge image widths, raw mode included"""
    for w in (1, 2, 3, 7, 16, 64, 255, 256, 320, 639, 640, 641, 753, 754, 832, 1283, 1920, 4096, 16383):
        for raw in (0, 1):
            p = lib.make_params(name, w=w, h=32, outw=640, outh=480, raw=raw)
            step = (p.col_step_hi << 32) | p.col_step_lo
            destw = p.destw
            assert destw > 0 and step == -((-(w << 32)) // destw)            # ceil(2^32 * w / destw)
            pos = 0
            for x in range(destw + 3):                                       # the tiled path runs up to 3 samples over
                assert pos >> 32 == (x * w) // destw, (w, raw, x)
                pos += step


def test_fir_flag_is_validated_by_finalize(lib):
    """CRTHIP_F_EQ_FIR(taps): 0 (stock equaliser) and the four kernels of crt_core.c:130-147 only"""
    for taps in (0, 4, 5, 6, 7):
        p = lib.make_params("ntsc", w=64, h=48, outw=64, outh=48, flags=taps << 8)
        assert p.eq_kernel == taps
    for taps in (1, 2, 3):
        with pytest.raises(ValueError):
            lib.make_params("ntsc", w=64, h=48, outw=64, outh=48, flags=taps << 8)


def test_sincos14_host_matches_oracle(lib):
    L = lib.load_library()
    orc = R.Oracle("ntsc")
    s, c = C.c_int(), C.c_int()
    for n in range(-17000, 34000, 11):
        L.crt_setup_sincos14(C.byref(s), C.byref(c), n)
        assert (s.value, c.value) == orc.sincos14(n)
    L.crt_setup_expx.restype = C.c_int
    orc.lib.orc_expx.restype = C.c_int
    for n in list(range(-30000, 30000, 37)) + [0, 1, -1, 2047, 2048, -2048]:
        assert L.crt_setup_expx(n) == orc.lib.orc_expx(n), n


def test_product_never_loads_the_oracle(lib):
    """The shipped package must not reference oracle/ (it is the checker, not a fallback)."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("libcrt_oracle", "crt_oracle", "oracle/_ref", "libref_", "orc_"):
                    assert needle not in txt, (f, needle)


def test_missing_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        lib.CRT(1, 640, 480)


SYSTEM_LINES = {"ntsc": (910, 262), "ntscp0": (912, 262), "snes": (909, 262), "temp": (910, 262), "pv1k": (1920, 262), "nesrgb": (909, 262)}


@pytest.mark.parametrize("name", sorted(SYSTEM_LINES))
def test_signal_layout_rules_and_margin_coverage(lib, name):
    """Round 6: the fused path keeps its signal in padded lines.  crthip_signal_layout_query is the host-side rule
    (no device): checked here against the rule restated, and -- for every geometry it calls padded -- the margin kernel's lane ->
    (line, column) mapping (k_margin_pad, crt_encode.hip; restated below) must cover every sample outside the active rectangle, none
    inside, copy the head of every line behind its predecessor up to `padv` exactly where the active-video kernel does not write,
    and never store past the pad."""
    L = lib.load_library()
    hres, vres = SYSTEM_LINES[name]
    pitch = (hres + 64 + 127) // 128 * 128
    padw = pitch - hres
    padc = min(padw, 96)
    seen_padded = seen_flat = 0
    for xoff in (0, 4, 8, 12, 16, 20, 24, 40, -40, -150):
        for raw, (w, h) in ((0, (640, 480)), (1, (200, 100)), (1, (753, 236))):
            for n, shape in ((4096, 0), (300, 0), (64, 0), (64, 1), (64, 2)):
                try:
                    p = lib.make_params(name, w=w, h=h, outw=640, outh=480, xoffset=xoff, raw=raw)
                except ValueError:
                    continue
                lay = (C.c_int * 4)()
                fs = C.c_size_t(0)
                rc = L.crthip_signal_layout_query(C.byref(p), n, shape, lay, C.byref(fs))
                assert rc in (0, 1)
                xo, yo, destw, desth = p.xo, p.yo, p.destw, p.desth
                wrap = max(0, xo + destw - hres)
                a = wrap + (padc - wrap) // 16 * 16
                padv = min(a, padc // 16 * 16)
                inside = xo >= 0 and yo >= 0 and destw > 0 and yo + desth + (1 if wrap else 0) <= vres
                want = (inside and wrap <= 16 and xo >= padw and destw >= 16 and padv >= 80 and not (shape == 0 and n <= 256))
                assert rc == int(want), (name, xoff, raw, n, shape, rc, xo, destw, wrap, padv)
                if not rc:
                    seen_flat += 1
                    assert lay[0] == hres and fs.value == L.crthip_field_stride(p.system, p.chroma_pattern)
                    continue
                seen_padded += 1
                assert (lay[0], lay[2], lay[3]) == (pitch, padv, wrap) and (lay[1] + xo) % 128 == 0 and 0 <= lay[1] < 128
                assert fs.value == (vres + 1 + 240) * pitch
                if n != 4096:
                    continue
                # --- k_margin_pad's mapping, restated ---
                cf = (hres + 15) // 16
                cl = (xo + 15) // 16
                right = hres - (xo + destw)
                cr = (right + 15) // 16 if right > 0 else 0
                per = cl + cr
                home = np.zeros((vres, hres), dtype=np.int32)
                copy = np.zeros((vres, padw), dtype=np.int32)
                for q0 in range(yo * cf + desth * per + (vres - yo - desth) * cf):
                    if q0 < yo * cf:
                        line, k, lo, hi = q0 // cf, q0 % cf, 0, hres
                    elif q0 < yo * cf + desth * per:
                        q = q0 - yo * cf
                        y, k = q // per, q % per
                        line = yo + y
                        if k < cl:
                            lo, hi = (wrap if y > 0 else 0), xo
                        else:
                            k, lo, hi = k - cl, min(xo + destw, hres), hres
                    else:
                        q = q0 - yo * cf - desth * per
                        r, k = q // cf, q % cf
                        line, lo, hi = yo + desth + r, (wrap if r == 0 else 0), hres
                    col, ln = lo + 16 * k, 16
                    if hi - lo >= 16:
                        col = min(col, hi - 16)
                    else:
                        if k > 0 or hi <= lo:
                            continue
                        ln = hi - lo
                    home[line, col:col + ln] += 1
                    if line >= 1 and col + ln <= padc:
                        copy[line - 1, col:col + ln] += 1
                active = np.zeros(vres * hres + 16, dtype=bool)
                for y in range(desth):
                    s0 = (y + yo) * hres + xo
                    active[s0:s0 + destw] = True
                active = active[:vres * hres].reshape(vres, hres)
                assert ((home > 0) == ~active).all(), (name, xoff, raw, "margin samples written / left out")
                # behind line l: the head of line l + 1, margin part by the margin kernel, overhang part by the active-video kernel
                want_copy = ~active[1:, :padv]
                assert ((copy[:-1, :padv] > 0) == want_copy).all(), (name, xoff, raw, "copies behind the lines")
                assert (active[1:, :padv].sum(axis=1) <= wrap).all() and not (copy[:, padc:] > 0).any()
    assert seen_padded > 10 and seen_flat > 10
