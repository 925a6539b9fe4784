"""Test-side ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``RefLib``  -- the real reference, compiled unmodified from /root/reference by
  ``oracle/Makefile`` into ``oracle/_ref/libref_<sys>.so`` (+ ``ref_probe.c``).
* ``Oracle``  -- our CPU restatement ``oracle/libcrt_oracle.so``.

Both expose the same tiny Python surface (``init / modulate / demodulate`` and the
observable state ``analog, inp, ccf, hsync, vsync, rn, out``) so parity tests read
the same against either.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

SYS_NTSC, SYS_NES, SYS_PV1K, SYS_SNES, SYS_TEMP, SYS_VHS, SYS_NESRGB = 0, 1, 2, 3, 4, 5, 6
FMT_RGB, FMT_BGR, FMT_ARGB, FMT_RGBA, FMT_ABGR, FMT_BGRA = range(6)
ORC_TAIL = 16

# name -> (system id, chroma pattern, _ref library)
SYSTEMS = {
    "ntsc": (SYS_NTSC, 1, "libref_ntsc.so"),
    "vhs": (SYS_VHS, 1, "libref_vhs.so"),
    "nes": (SYS_NES, 2, "libref_nes.so"),
    "nesp0": (SYS_NES, 0, "libref_nesp0.so"),
    "ntscp0": (SYS_NTSC, 0, "libref_ntscp0.so"),
    # USE_CONVOLUTION builds of the decoder (crt_core.c:85-147): FIR kernels instead of the 3-band equaliser
    "ntscfir7": (SYS_NTSC, 1, "libref_ntscfir7.so"),
    "ntscfir6": (SYS_NTSC, 1, "libref_ntscfir6.so"),
    "ntscfir5": (SYS_NTSC, 1, "libref_ntscfir5.so"),
    "ntscfir4": (SYS_NTSC, 1, "libref_ntscfir4.so"),
    # the remaining systems of crt_core.h:30-36
    "snes": (SYS_SNES, 1, "libref_snes.so"),
    "pv1k": (SYS_PV1K, 1, "libref_pv1k.so"),
    "temp": (SYS_TEMP, 1, "libref_temp.so"),
    "nesrgb": (SYS_NESRGB, 2, "libref_nesrgb.so"),
    # CRT_DO_BLOOM builds (crt_core.h:70 patched to 1)
    "ntscbloom": (SYS_NTSC, 1, "libref_ntscbloom.so"),
    "vhsbloom": (SYS_VHS, 1, "libref_vhsbloom.so"),
    "pv1kbloom": (SYS_PV1K, 1, "libref_pv1kbloom.so"),
    "snesbloom": (SYS_SNES, 1, "libref_snesbloom.so"),
    # VERDICT round 2, missing #3: the remaining build-time switches (oracle/Makefile: PATCHLIB)
    "vhslp": (SYS_VHS, 1, "libref_vhslp.so"),             # VHS_MODE VHS_LP, crt_ntscvhs.h:102-124
    "vhsep": (SYS_VHS, 1, "libref_vhsep.so"),             # VHS_MODE VHS_EP
    "vhslcg": (SYS_VHS, 1, "libref_vhslcg.so"),           # CRT_VHS_NOISE 0, crt_ntscvhs.h:29
    "ntscnovsync": (SYS_NTSC, 1, "libref_ntscnovsync.so"),   # CRT_DO_VSYNC 0, crt_core.h:71
    "ntscnohsync": (SYS_NTSC, 1, "libref_ntscnohsync.so"),   # CRT_DO_HSYNC 0, crt_core.h:72
    "ntschipass": (SYS_NTSC, 1, "libref_ntschipass.so"),     # HIPASS 1, crt_ntsc.c:115
    "nesborder": (SYS_NES, 2, "libref_nesborder.so"),        # NES_BORDER 1, crt_nes.c:69
}
# oracle switches of those builds (struct orc_sys members "set by the caller after orc_sys_init")
VARIANTS = {"vhslp": dict(vhs_mode=1), "vhsep": dict(vhs_mode=2), "vhslcg": dict(vhs_lcg_noise=1),
            "ntscnovsync": dict(no_vsync=1), "ntscnohsync": dict(no_hsync=1), "ntschipass": dict(hipass=1),
            "nesborder": dict(nes_border=1)}
EQ_KERNEL = {"ntscfir7": 7, "ntscfir6": 6, "ntscfir5": 5, "ntscfir4": 4}      # everything else: 0 (IIR)
DOT_CRAWL_SYSTEMS = (SYS_NES, SYS_NESRGB, SYS_SNES, SYS_PV1K, SYS_TEMP)       # NTSC_SETTINGS has dot_crawl_offset
PROGRESSIVE_SYSTEMS = (SYS_NES, SYS_NESRGB)                                   # no field / frame members


def is_bloom(name):
    return name.endswith("bloom")


def bpp4fmt(fmt):
    return 3 if fmt in (0, 1) else (4 if fmt in (2, 3, 4, 5) else 0)


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def build_ref():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True)


PKG_LIB = os.path.join(ROOT, "ntsc-crt_amd", "lib")
DROPIN = {"ntsc": ("libntsccrt_hip_ntsc.so", ["-DCRT_SYSTEM=0"]),
          "vhs": ("libntsccrt_hip_vhs.so", ["-DCRT_SYSTEM=5"]),
          "nes": ("libntsccrt_hip_nes.so", ["-DCRT_SYSTEM=1"]),
          "nesp0": ("libntsccrt_hip_nesp0.so", ["-DCRT_SYSTEM=1", "-DCRT_CHROMA_PATTERN=0"]),
          "ntscp0": ("libntsccrt_hip_ntscp0.so", ["-DCRT_SYSTEM=0", "-DCRT_CHROMA_PATTERN=0"]),
          "ntscfir7": ("libntsccrt_hip_ntsc_fir7.so", ["-DCRT_SYSTEM=0"]),
          "snes": ("libntsccrt_hip_snes.so", ["-DCRT_SYSTEM=3"]),
          "pv1k": ("libntsccrt_hip_pv1k.so", ["-DCRT_SYSTEM=2"]),
          "temp": ("libntsccrt_hip_temp.so", ["-DCRT_SYSTEM=4"]),
          "nesrgb": ("libntsccrt_hip_nesrgb.so", ["-DCRT_SYSTEM=6"]),
          "ntscbloom": ("libntsccrt_hip_ntsc_bloom.so", ["-DCRT_SYSTEM=0", "-DCRT_DO_BLOOM=1"]),
          "vhsbloom": ("libntsccrt_hip_vhs_bloom.so", ["-DCRT_SYSTEM=5", "-DCRT_DO_BLOOM=1"]),
          "snesbloom": ("libntsccrt_hip_snes_bloom.so", ["-DCRT_SYSTEM=3", "-DCRT_DO_BLOOM=1"]),
          "pv1kbloom": ("libntsccrt_hip_pv1k_bloom.so", ["-DCRT_SYSTEM=2", "-DCRT_DO_BLOOM=1"]),
          "vhslp": ("libntsccrt_hip_vhs_lp.so", ["-DCRT_SYSTEM=5", "-DVHS_MODE=1"]),
          "vhsep": ("libntsccrt_hip_vhs_ep.so", ["-DCRT_SYSTEM=5", "-DVHS_MODE=2"]),
          "vhslcg": ("libntsccrt_hip_vhs_lcg.so", ["-DCRT_SYSTEM=5", "-DCRT_VHS_NOISE=0"]),
          "ntscnovsync": ("libntsccrt_hip_ntsc_novsync.so", ["-DCRT_SYSTEM=0", "-DCRT_DO_VSYNC=0"]),
          "ntscnohsync": ("libntsccrt_hip_ntsc_nohsync.so", ["-DCRT_SYSTEM=0", "-DCRT_DO_HSYNC=0"]),
          "ntschipass": ("libntsccrt_hip_ntsc_hipass.so", ["-DCRT_SYSTEM=0", "-DCRT_HIPASS=1"]),
          "nesborder": ("libntsccrt_hip_nes_border.so", ["-DCRT_SYSTEM=1", "-DNES_BORDER=1"])}


def build_driver_binaries():
    """The reference's UNCHANGED drivers (crt_main.c, extra/video_convert.c, straight from /root/reference) compiled against
    include/ and linked with the HIP drop-in libraries -> ntsc-crt_amd/lib/ntsc_cli_hip & co. (tests/test_gpu_dropin.py,
    tools/time_cli.py run them on the GPU box, where the reference's sources do not exist).  Returns [(exe, gcc -H log)];
    raises if a link fails; [] where there is no /root/reference."""
    import subprocess
    import tempfile
    ref = "/root/reference"
    if not os.path.exists(ref + "/crt_main.c"):
        return []
    inc = os.path.join(ROOT, "include")
    done = []
    # `#include "crt_core.h"` looks next to the including file first; the drivers are therefore
    # reached through symlinks in a scratch directory, so the only crt_core.h found is include/'s.
    with tempfile.TemporaryDirectory(prefix="crtdrv") as tmp:
        for f in ("crt_main.c", "extra/video_convert.c"):
            os.symlink(os.path.join(ref, f), os.path.join(tmp, os.path.basename(f)))
        for out, defs, drv, lib in [
            ("ntsc_cli_hip", ["-DCRT_SYSTEM=0"], "crt_main.c", "ntsccrt_hip_ntsc"),
            ("ntscvhs_video_hip", ["-DCRT_SYSTEM=5"], "video_convert.c", "ntsccrt_hip_vhs"),
            ("ntsc_cli_snes_hip", ["-DCRT_SYSTEM=3"], "crt_main.c", "ntsccrt_hip_snes"),
            ("ntsc_cli_pv1k_hip", ["-DCRT_SYSTEM=2"], "crt_main.c", "ntsccrt_hip_pv1k"),
        ]:
            exe = os.path.join(PKG_LIB, out)
            cmd = ["gcc", "-O2", "-w", "-std=c89", "-H", "-I" + inc, "-I" + ref] + defs + ["-o", exe,
                   os.path.join(tmp, drv), ref + "/ppm_rw.c", ref + "/bmp_rw.c",
                   "-L" + PKG_LIB, "-l" + lib, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"]   # $ORIGIN like the Makefile: the snapshot may be unpacked anywhere
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("linking %s failed: %s" % (out, r.stderr[-2000:]))
            done.append((exe, r.stderr))
    return done


def build_dropin_probe(name):
    """tests/abi_probe.c compiled against THIS repo's include/crt_core.h and linked to the drop-in
    library: the same refp_* helper surface as oracle/_ref, but over the HIP implementation."""
    lib, defs = DROPIN[name]
    out = os.path.join(PKG_LIB, "libdropin_probe_%s.so" % name)
    src = os.path.join(ROOT, "tests", "abi_probe.c")
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(PKG_LIB, lib))):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-w", "-I" + os.path.join(ROOT, "include")] + defs +
                       ["-o", out, src, "-L" + PKG_LIB, "-l" + lib[3:-3], "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return out


def have_ref(name="ntsc"):
    return os.path.exists(os.path.join(REF_DIR, SYSTEMS[name][2]))


def fnv1a32(buf):
    """FNV-1a 32-bit over a bytes-like (the hash SURVEY.md section 8c quotes)."""
    h = 0x811C9DC5
    for b in bytes(buf):
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def lcg_bytes(n, seed):
    """Uniform bytes from the 32-bit LCG x <- x*1664525 + 1013904223 (x0 = seed); byte k is
    (x_{k+1} >> 8) & 0xff.  Vectorised: x_k = A_k*x0 + C_k with the affine
    powers composed per index bit."""
    mask = np.uint64(0xFFFFFFFF)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    A = np.ones(n, dtype=np.uint64)
    Cc = np.zeros(n, dtype=np.uint64)
    pa, pc = np.uint64(1664525), np.uint64(1013904223)
    bit = 0
    while (1 << bit) <= n:
        sel = ((idx >> np.uint64(bit)) & np.uint64(1)).astype(bool)
        A[sel] = (pa * A[sel]) & mask
        Cc[sel] = (pa * Cc[sel] + pc) & mask
        pc = (pa * pc + pc) & mask
        pa = (pa * pa) & mask
        bit += 1
    x = (A * np.uint64(seed & 0xFFFFFFFF) + Cc) & mask
    return ((x >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)


def synth_image(w, h, bpp, seed, kind="random"):
    """Synthetic input image (h, w, bpp) uint8."""
    if kind == "random":
        return lcg_bytes(w * h * bpp, seed).reshape(h, w, bpp).copy()
    if kind == "bars":
        img = np.zeros((h, w, bpp), dtype=np.uint8)
        cols = [(255, 255, 255), (255, 255, 0), (0, 255, 255), (0, 255, 0),
                (255, 0, 255), (255, 0, 0), (0, 0, 255), (0, 0, 0)]
        for k, c in enumerate(cols):
            x0, x1 = k * w // 8, (k + 1) * w // 8
            img[:, x0:x1, :3] = np.array(c, dtype=np.uint8)
        yy = np.arange(h)[:, None]
        xx = np.arange(w)[None, :]
        lo = h * 2 // 3
        grad = ((xx * 255) // max(w - 1, 1)).astype(np.uint8)
        xor = ((xx ^ yy) & 255).astype(np.uint8)
        for ch in range(min(3, bpp)):
            img[lo:, :, ch] = np.where(yy[lo:] < (lo + h) // 2, grad, xor[lo:])
        if bpp == 4:
            img[:, :, 3] = (seed * 37) & 255
        return img
    raise ValueError(kind)


def synth_ppu(w, h, seed):
    """NES PPU pixels: uniform in [0, 511]."""
    b = lcg_bytes(w * h * 2, seed).astype(np.uint16)
    return ((b[0::2] << 8 | b[1::2]) & 511).reshape(h, w).astype(np.uint16)


# ----------------------------------------------------------------------------
# the real reference
# ----------------------------------------------------------------------------
class RefLib:
    CRT_FIELDS = ["analog", "inp", "outw", "outh", "out_format", "out", "hue", "brightness",
                  "contrast", "saturation", "black_point", "white_point", "scanlines", "blend",
                  "v_fac", "ccf", "hsync", "vsync", "rn"]

    def __init__(self, name="ntsc", dropin=False):
        """dropin=False: the real reference (oracle/_ref).  dropin=True: this repo's drop-in library
        (ntsc-crt_amd/lib/libntsccrt_hip_<sys>.so) behind the same probe surface."""
        self.name = name
        self.system, self.pattern, libname = SYSTEMS[name]
        if dropin:
            self.lib = C.CDLL(build_dropin_probe(name), mode=os.RTLD_LOCAL)
        else:
            self.lib = C.CDLL(os.path.join(REF_DIR, libname), mode=os.RTLD_LOCAL)
        L = self.lib
        for f in ("sizeof_crt", "sizeof_settings"):
            getattr(L, "refp_" + f).restype = C.c_long
        self.off = {}
        for f in self.CRT_FIELDS:
            fn = getattr(L, "refp_off_" + f)
            fn.restype = C.c_long
            self.off[f] = fn()
        self.soff = {}
        names = ["data", "w", "h", "hue", "xoffset", "yoffset"]
        if self.system == SYS_NES:
            names += ["border_color", "dot_crawl_offset", "field_initialized"]
        elif self.system == SYS_NESRGB:
            names += ["format", "dot_crawl_offset", "field_initialized"]
        else:
            names += ["format", "raw", "as_color", "field", "frame", "iirs_initialized"]
        if self.system == SYS_VHS:
            names += ["do_aberration"]
        if self.system in (SYS_SNES, SYS_PV1K, SYS_TEMP):
            names += ["dot_crawl_offset"]
        for f in names:
            fn = getattr(L, "refp_soff_" + f)
            fn.restype = C.c_long
            self.soff[f] = fn()
        self.sizeof_crt = L.refp_sizeof_crt()
        self.sizeof_settings = L.refp_sizeof_settings()
        self.hres = L.refp_hres()
        self.vres = L.refp_vres()
        self.input_size = L.refp_input_size()
        self.top, self.bot = L.refp_top(), L.refp_bot()
        self.vper = L.refp_cc_vper()
        self.ccs = L.refp_cc_samples()
        self.av_beg, self.av_len = L.refp_av_beg(), L.refp_av_len()
        L.refp_time_fieldpasses.restype = C.c_double
        L.refp_time_fieldpasses.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.crt_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.crt_modulate.argtypes = [C.c_void_p, C.c_void_p]
        L.crt_demodulate.argtypes = [C.c_void_p, C.c_int]
        L.crt_sincos14.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]

    def srand(self, seed):
        self.lib.refp_srand(C.c_uint(seed))

    def new_crt(self, outw, outh, fmt, out=None):
        return RefCRT(self, outw, outh, fmt, out)


class RefCRT:
    """One ``struct CRT`` + one ``struct NTSC_SETTINGS`` of the reference."""

    def __init__(self, ref, outw, outh, fmt, out=None):
        self.ref = ref
        self.mem = np.zeros(ref.sizeof_crt, dtype=np.uint8)
        self.smem = np.zeros(ref.sizeof_settings, dtype=np.uint8)
        bpp = bpp4fmt(fmt) or 4
        self.out = out if out is not None else np.zeros(outw * outh * bpp, dtype=np.uint8)
        self._img = None
        ref.lib.crt_init(self.mem.ctypes.data, outw, outh, fmt, self.out.ctypes.data)

    # --- struct CRT field access -------------------------------------------------
    def _i32(self, name, count=1):
        o = self.ref.off[name]
        return self.mem[o:o + 4 * count].view(np.int32)

    def get(self, name):
        return int(self._i32(name)[0])

    def set(self, name, value):
        if name == "v_fac":
            o = self.ref.off[name]
            self.mem[o:o + 4].view(np.uint32)[0] = value
        else:
            self._i32(name)[0] = value

    @property
    def analog(self):
        o = self.ref.off["analog"]
        return self.mem[o:o + self.ref.input_size].view(np.int8)

    @property
    def inp(self):
        o = self.ref.off["inp"]
        return self.mem[o:o + self.ref.input_size].view(np.int8)

    @property
    def ccf(self):
        return self._i32("ccf", self.ref.vper * self.ref.ccs).reshape(self.ref.vper, self.ref.ccs)

    # --- settings ----------------------------------------------------------------
    def settings(self, img, **kw):
        """(Re)fill NTSC_SETTINGS from an image array and keyword fields."""
        self._img = np.ascontiguousarray(img)
        so = self.ref.soff
        self.smem[so["data"]:so["data"] + 8].view(np.uint64)[0] = self._img.ctypes.data
        for k, v in kw.items():
            o = so[k]
            if k == "border_color":
                self.smem[o:o + 4].view(np.uint32)[0] = v
            else:
                self.smem[o:o + 4].view(np.int32)[0] = v

    def sget(self, name):
        o = self.ref.soff[name]
        return int(self.smem[o:o + 4].view(np.int32)[0])

    def sset(self, name, value):
        o = self.ref.soff[name]
        self.smem[o:o + 4].view(np.int32)[0] = value

    # --- the hot path --------------------------------------------------------------
    def modulate(self):
        self.ref.lib.crt_modulate(self.mem.ctypes.data, self.smem.ctypes.data)

    def demodulate(self, noise):
        self.ref.lib.crt_demodulate(self.mem.ctypes.data, noise)

    def time_fieldpasses(self, noise, reps, interlaced):
        m, d = C.c_double(0), C.c_double(0)
        t = self.ref.lib.refp_time_fieldpasses(self.mem.ctypes.data, self.smem.ctypes.data,
                                               noise, reps, int(interlaced), C.byref(m), C.byref(d))
        return t, m.value, d.value


def reads_past_inp(orc, trace, vsync_found, hsync_before):
    """Does crt_demodulate, for the field whose per-line trace this is, read inp[] beyond the ORC_TAIL bytes behind it?  The
    reference then reads whatever follows `inp` in struct CRT / the heap (undefined behaviour: the oracle and the library see
    different bytes there, and so do two runs of the oracle on two machines).  Three reads can run off the end when the sync
    state is far from lock on the field's last analog line: the hsync search window (crt_core.c:437-445: from ln + hsync +
    SYNC_BEG - HWIN), the burst samples (:456-467: up to ln + (hsync aligned) + CB_BEG + CB_LEN) and the video window (:452-454,
    :534: pos + AV_LEN).  trace: Oracle demodulate(trace=True) rows (valid, pos, wave0, wave1, beg, nrows, hsync, dx, scanl)."""
    sd = orc.sys
    limit = sd.input_size + ORC_TAIL
    hs = int(hsync_before)
    for i in range(trace.shape[0]):
        if int(trace[i, 0]) != 1:
            continue                                       # skipped lines (beg >= outh) touch nothing
        ln = ((sd.top + i + int(vsync_found)) % sd.vres) * sd.hres
        if ln + hs + sd.sync_beg + sd.hsync_window > limit:
            return True
        hs = int(trace[i, 6])
        ha = hs & ~3 if sd.cc_samples == 4 else hs - hs % sd.cc_samples
        if ln + ha + sd.cb_beg + sd.cb_len > limit:
            return True
        if int(trace[i, 1]) + sd.av_len > limit:
            return True
    return False


# ----------------------------------------------------------------------------
# our CPU restatement
# ----------------------------------------------------------------------------
class OrcSys(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "system", "chroma_pattern", "hres", "vres", "input_size", "top", "bot", "lines",
        "cc_vper", "hsync_window", "vsync_window", "hsync_thresh", "vsync_thresh",
        "sync_beg", "bw_beg", "cb_beg", "av_beg", "av_len", "lav_beg", "vs_sep_end",
        "white_level", "burst_level", "black_level", "blank_level", "sync_level")] + [
        ("iir_c", C.c_int * 3), ("eq_lf", C.c_int * 3), ("eq_hf", C.c_int * 3),
        ("eq_g", (C.c_int * 3) * 3), ("eq_kernel", C.c_int), ("do_bloom", C.c_int),
        ("cc_samples", C.c_int), ("cb_len", C.c_int), ("enc_bandlimit", C.c_int), ("enc_field_rows", C.c_int),
        ("enc_line_rows", C.c_int), ("vert_step", C.c_int), ("burst_off", C.c_int), ("q_off", C.c_int),
        ("equ_a_lo", C.c_int), ("equ_a_hi", C.c_int), ("equ_b_lo", C.c_int), ("equ_b_hi", C.c_int),
        ("vs_lo", C.c_int), ("vs_hi", C.c_int), ("vs_by_field", C.c_int),
        ("vhs_lcg_noise", C.c_int), ("no_vsync", C.c_int), ("no_hsync", C.c_int), ("hipass", C.c_int), ("nes_border", C.c_int)]


class OrcCrt(C.Structure):
    _fields_ = [("analog", C.c_void_p), ("inp", C.c_void_p),
                ("outw", C.c_int), ("outh", C.c_int), ("out_format", C.c_int),
                ("out", C.c_void_p),
                ("hue", C.c_int), ("brightness", C.c_int), ("contrast", C.c_int),
                ("saturation", C.c_int), ("black_point", C.c_int), ("white_point", C.c_int),
                ("scanlines", C.c_int), ("blend", C.c_int), ("v_fac", C.c_uint),
                ("ccf", (C.c_int * 5) * 5), ("hsync", C.c_int), ("vsync", C.c_int),
                ("rn", C.c_int)]


class OrcSettings(C.Structure):
    _fields_ = [("data", C.c_void_p), ("format", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("raw", C.c_int), ("as_color", C.c_int), ("field", C.c_int), ("frame", C.c_int),
                ("hue", C.c_int), ("xoffset", C.c_int), ("yoffset", C.c_int),
                ("do_aberration", C.c_int), ("border_color", C.c_uint),
                ("dot_crawl_offset", C.c_int), ("initialized", C.c_int)]


class OrcLine(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("valid", "pos", "wave0", "wave1", "beg", "end", "hsync", "dx", "scanl")]


ORC_LINE_INTS = 9


_SET_ALIAS = {"iirs_initialized": "initialized", "field_initialized": "initialized"}


class Oracle:
    def __init__(self, name="ntsc"):
        self.name = name
        self.system, self.pattern, _ = SYSTEMS[name]
        path = os.path.join(ORACLE_DIR, "libcrt_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path, mode=os.RTLD_LOCAL)
        L = self.lib
        self.sys = OrcSys()
        L.orc_sys_init(C.byref(self.sys), self.system, self.pattern)
        self.sys.eq_kernel = EQ_KERNEL.get(name, 0)
        self.sys.do_bloom = int(is_bloom(name))
        for k, val in VARIANTS.get(name, {}).items():
            if k == "vhs_mode":
                L.orc_sys_set_vhs_mode(C.byref(self.sys), val)
            else:
                setattr(self.sys, k, val)
        for n in ("hres", "vres", "input_size", "top", "bot", "av_beg", "av_len"):
            setattr(self, n, getattr(self.sys, n))
        self.vper = self.sys.cc_vper
        self.ccs = self.sys.cc_samples
        L.orc_time_fieldpasses.restype = C.c_double
        L.orc_stage_noise.restype = C.c_int

    def srand(self, seed):
        C.CDLL(None).srand(C.c_uint(seed))

    def new_crt(self, outw, outh, fmt, out=None):
        return OracleCRT(self, outw, outh, fmt, out)

    def sincos14(self, n):
        s, c = C.c_int(), C.c_int()
        self.lib.orc_sincos14(C.byref(s), C.byref(c), C.c_int(n))
        return s.value, c.value

    def lcg_jump(self, k):
        m, a = C.c_uint(), C.c_uint()
        self.lib.orc_lcg_jump(C.c_uint(k), C.byref(m), C.byref(a))
        return m.value, a.value


class OracleCRT:
    def __init__(self, orc, outw, outh, fmt, out=None):
        self.ref = orc
        self.v = OrcCrt()
        self.s = OrcSettings()
        self._analog = np.zeros(orc.input_size, dtype=np.int8)
        self._inp = np.zeros(orc.input_size + ORC_TAIL, dtype=np.int8)
        bpp = bpp4fmt(fmt) or 4
        self.out = out if out is not None else np.zeros(outw * outh * bpp, dtype=np.uint8)
        self._img = None
        orc.lib.orc_crt_init(C.byref(orc.sys), C.byref(self.v), C.c_void_p(self._analog.ctypes.data),
                             C.c_void_p(self._inp.ctypes.data), outw, outh, fmt,
                             C.c_void_p(self.out.ctypes.data))
        self.trace = None

    def get(self, name):
        return int(getattr(self.v, name))

    def set(self, name, value):
        setattr(self.v, name, value)

    @property
    def analog(self):
        return self._analog

    @property
    def inp(self):
        return self._inp[:self.ref.input_size]

    @property
    def ccf(self):
        return np.array([[self.v.ccf[r][k] for k in range(self.ref.ccs)] for r in range(self.ref.vper)],
                        dtype=np.int32)

    def settings(self, img, **kw):
        self._img = np.ascontiguousarray(img)
        self.s.data = self._img.ctypes.data
        for k, val in kw.items():
            setattr(self.s, _SET_ALIAS.get(k, k), val)

    def sget(self, name):
        return int(getattr(self.s, _SET_ALIAS.get(name, name)))

    def sset(self, name, value):
        setattr(self.s, _SET_ALIAS.get(name, name), value)

    def modulate(self):
        self.ref.lib.orc_modulate(C.byref(self.ref.sys), C.byref(self.v), C.byref(self.s))

    def demodulate(self, noise, trace=False):
        if trace:
            n = self.ref.bot - self.ref.top
            arr = (OrcLine * n)()
            self.ref.lib.orc_demodulate_trace(C.byref(self.ref.sys), C.byref(self.v), noise, arr)
            self.trace = np.frombuffer(arr, dtype=np.int32).reshape(n, ORC_LINE_INTS).copy()
        else:
            self.ref.lib.orc_demodulate(C.byref(self.ref.sys), C.byref(self.v), noise)

    def time_fieldpasses(self, noise, reps, interlaced):
        t = self.ref.lib.orc_time_fieldpasses(C.byref(self.ref.sys), C.byref(self.v),
                                              C.byref(self.s), noise, reps, int(interlaced))
        return t, None, None


STATE_FIELDS = ("hsync", "vsync", "rn")


def compare_state(a, b, what=""):
    """Assert that two CRT wrappers (any mix of RefCRT / OracleCRT) agree bit-exactly."""
    np.testing.assert_array_equal(np.asarray(a.analog), np.asarray(b.analog), err_msg=what + " analog")
    np.testing.assert_array_equal(np.asarray(a.inp), np.asarray(b.inp), err_msg=what + " inp")
    np.testing.assert_array_equal(np.asarray(a.ccf), np.asarray(b.ccf), err_msg=what + " ccf")
    for f in STATE_FIELDS:
        assert a.get(f) == b.get(f), "%s %s: %d != %d" % (what, f, a.get(f), b.get(f))
    np.testing.assert_array_equal(a.out, b.out, err_msg=what + " out")
