"""Drop-in boundary, CPU side: this repo's include/crt_core.h (+ per-system headers) must give
`struct CRT` / `struct NTSC_SETTINGS` the exact layout of the reference's headers, and the
reference's UNCHANGED drivers must compile and link against include/ + libntsccrt_hip_<sys>.so.
No GPU needed (nothing is executed that launches a kernel)."""
import os
import subprocess

import pytest

import crtref as R

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not R.have_ref("ntsc"), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


ALL_DROPINS = ["ntsc", "vhs", "nes", "nesp0", "ntscp0", "snes", "pv1k", "temp", "nesrgb",
               "ntscbloom", "vhsbloom", "snesbloom", "pv1kbloom",
               "vhslp", "vhsep", "vhslcg", "ntscnovsync", "ntscnohsync", "ntschipass", "nesborder"]


@needs_ref
@pytest.mark.parametrize("name", ALL_DROPINS)
def test_struct_layout_identical_to_reference(name):
    ref = R.RefLib(name)
    ours = R.RefLib(name, dropin=True)
    assert ours.sizeof_crt == ref.sizeof_crt
    assert ours.sizeof_settings == ref.sizeof_settings
    assert ours.off == ref.off
    assert ours.soff == ref.soff
    for f in ("hres", "vres", "input_size", "top", "bot", "vper", "av_beg", "av_len"):
        assert getattr(ours, f) == getattr(ref, f), f
    for fn in ("refp_sync_beg", "refp_bw_beg", "refp_cb_beg", "refp_system", "refp_cc_samples", "refp_do_bloom"):
        assert getattr(ours.lib, fn)() == getattr(ref.lib, fn)(), fn


@pytest.mark.parametrize("name", ALL_DROPINS)
def test_dropin_exports_the_reference_api(name):
    lib = os.path.join(R.PKG_LIB, R.DROPIN[name][0])
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for s in ("crt_init", "crt_resize", "crt_reset", "crt_modulate", "crt_demodulate", "crt_bpp4fmt", "crt_sincos14"):
        assert (" T " + s + "\n") in syms, (name, s)


@pytest.mark.skipif(not os.path.exists(REF + "/crt_main.c"), reason="no /root/reference")
def test_unchanged_reference_drivers_link_against_the_hip_library():
    """crt_main.c and extra/video_convert.c straight from /root/reference, our headers, our library."""
    for exe, log in R.build_driver_binaries():
        inc = os.path.join(R.ROOT, "include")
        assert os.path.join(inc, "crt_core.h") in log, "driver was not compiled against include/crt_core.h"
        assert REF + "/crt_core.h" not in log
        assert os.path.exists(exe)
