"""Build-system parity (reference CMakeLists.txt:62-95): `cmake -DCRT_SYSTEM=n [-DVIDEO=on]` must produce the same two
targets -- ntsc (crt_main.c) and ntsc_video (extra/video_convert.c) -- from the reference's UNCHANGED drivers, linked to
the HIP drop-in library of that system.  Configure + build only (no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(shutil.which("cmake") is None or not os.path.exists(REF + "/crt_main.c"),
                                reason="needs cmake and the reference checkout")


@pytest.mark.parametrize("defs,target,lib", [(["-DCRT_SYSTEM=0"], "ntsc", "libntsccrt_hip_ntsc.so"),
                                             (["-DCRT_SYSTEM=5", "-DVIDEO=on"], "ntsc_video", "libntsccrt_hip_vhs.so"),
                                             (["-DCRT_SYSTEM=2", "-DCRT_DO_BLOOM=1"], "ntsc", "libntsccrt_hip_pv1k_bloom.so"),
                                             (["-DCRT_SYSTEM=5", "-DVIDEO=on", "-DCRT_VARIANT=vhs_lp"], "ntsc_video", "libntsccrt_hip_vhs_lp.so"),
                                             (["-DCRT_SYSTEM=0", "-DCRT_VARIANT=ntsc_nohsync"], "ntsc", "libntsccrt_hip_ntsc_nohsync.so")])
def test_cmake_targets(tmp_path, defs, target, lib):
    import __graft_entry__ as g
    g.build()
    b = str(tmp_path / "build")
    r = subprocess.run(["cmake", "-S", ROOT, "-B", b, "-DNTSC_CRT_REFERENCE_DIR=" + REF] + defs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run(["cmake", "--build", b], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    exe = os.path.join(b, target)
    assert os.path.exists(exe)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert lib in ldd, ldd
