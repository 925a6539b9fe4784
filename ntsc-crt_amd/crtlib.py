"""Host-side Python binding of the crthip_* C ABI (include/crt_hip.h).

PyTorch is used for plumbing only: device memory (torch tensors), streams and, in bench.py,
torch.distributed.  All computation happens in the hand-written HIP kernels of
lib/libcrthip.so; there is NO CPU or eager fallback here -- a missing library or a missing
GPU raises immediately.

The surface mirrors the reference's (crt_core.h:100-139): a ``CRT`` object is created with
the output geometry (crt_init), carries the monitor knobs as attributes (struct CRT members,
crt_core.h:77-86), and has ``modulate(settings)`` / ``demodulate(noise)``; the difference is
that it holds a *batch* of independent fields (one struct CRT each in the reference's terms).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(HERE, "lib")

SYSTEM_NTSC, SYSTEM_NES, SYSTEM_PV1K, SYSTEM_SNES, SYSTEM_TEMP, SYSTEM_VHS, SYSTEM_NESRGB = 0, 1, 2, 3, 4, 5, 6
FMT_RGB, FMT_BGR, FMT_ARGB, FMT_RGBA, FMT_ABGR, FMT_BGRA = range(6)
F_NES_SETUP, F_VHS_DRAW_ABERRATION, F_BLOOM, F_IMAGE_SPARE_ROW = 2, 4, 8, 16
F_NO_HSYNC, F_NO_VSYNC, F_VHS_LCG_NOISE, F_HIPASS, F_VHS_LP, F_VHS_EP, F_NES_BORDER = 0x20, 0x40, 0x80, 0x800, 0x2000, 0x4000, 0x1000
K_NAMES = ("template", "active", "noise", "sync", "decode")
MAX_VPER, MAX_CCS, CARRIER_ROWS = 5, 5, 10
STATE_INTS = 36      # sizeof(crthip_state) / 4
LINE_INTS = 8        # sizeof(crthip_line) / 4
# columns of the state tensor
ST_FIELD, ST_FRAME, ST_AUX, ST_HSYNC, ST_VSYNC, ST_RN, ST_CCF, ST_ODD = 0, 1, 2, 3, 4, 5, 6, 31
SHAPE_AUTO, SHAPE_LANE_PER_SCANLINE, SHAPE_SCANLINE_PARALLEL = 0, 1, 2

# name -> (CRT_SYSTEM, CRT_CHROMA_PATTERN); a "bloom" suffix selects the CRT_DO_BLOOM build (crt_core.h:70)
SYSTEMS = {"ntsc": (SYSTEM_NTSC, 1), "vhs": (SYSTEM_VHS, 1), "nes": (SYSTEM_NES, 2), "nesp0": (SYSTEM_NES, 0),
           "ntscp0": (SYSTEM_NTSC, 0), "snes": (SYSTEM_SNES, 1), "pv1k": (SYSTEM_PV1K, 1), "temp": (SYSTEM_TEMP, 1),
           "nesrgb": (SYSTEM_NESRGB, 2)}
DOT_CRAWL_SYSTEMS = (SYSTEM_NES, SYSTEM_NESRGB, SYSTEM_SNES, SYSTEM_PV1K, SYSTEM_TEMP)


# the reference's remaining build-time switches as names: base system + crthip_params.flags
VARIANTS = {"vhslp": ("vhs", F_VHS_LP), "vhsep": ("vhs", F_VHS_EP), "vhslcg": ("vhs", F_VHS_LCG_NOISE),
            "ntscnovsync": ("ntsc", F_NO_VSYNC), "ntscnohsync": ("ntsc", F_NO_HSYNC), "ntschipass": ("ntsc", F_HIPASS),
            "nesborder": ("nes", F_NES_BORDER)}


def split_system(name):
    """'ntscbloom' -> ('ntsc', True); a VARIANTS name -> its base system (variant_flags gives the flags)"""
    if name in VARIANTS:
        return VARIANTS[name][0], False
    if name.endswith("bloom"):
        return name[:-5], True
    return name, False


def variant_flags(name):
    return VARIANTS[name][1] if name in VARIANTS else 0


class Params(C.Structure):
    """crthip_params (include/crt_hip.h)."""
    _fields_ = [(n, C.c_int) for n in (
        "system", "chroma_pattern", "w", "h", "format", "raw", "as_color", "hue", "xoffset", "yoffset",
        "outw", "outh", "out_format", "mon_hue", "brightness", "contrast", "saturation",
        "black_point", "white_point", "scanlines", "blend")] + [
        ("v_fac", C.c_uint), ("noise", C.c_int), ("flags", C.c_int),
        ("finalized", C.c_int), ("in_bpp", C.c_int), ("out_bpp", C.c_int),
        ("destw", C.c_int), ("desth", C.c_int), ("xo", C.c_int), ("yo", C.c_int),
        ("burst", (C.c_int * MAX_CCS) * CARRIER_ROWS), ("modI", (C.c_int * MAX_CCS) * CARRIER_ROWS),
        ("modQ", (C.c_int * MAX_CCS) * CARRIER_ROWS),
        ("dem_cs", (C.c_int * MAX_CCS) * 2), ("dem_sn", (C.c_int * MAX_CCS) * 2),
        ("iir_c", C.c_int * 3), ("eq_lf", C.c_int * 3), ("eq_hf", C.c_int * 3),
        ("eq_g", (C.c_int * 3) * 3), ("huesn", C.c_int), ("huecs", C.c_int),
        ("bright", C.c_int), ("white", C.c_int), ("ire_base", C.c_int), ("dx", C.c_int),
        ("ratio", C.c_int), ("eq_kernel", C.c_int), ("bloom", C.c_int), ("bloom_max_e", C.c_int),
        ("col_step_lo", C.c_uint), ("col_step_hi", C.c_uint), ("loskip_wave_max", C.c_int), ("nes_border_color", C.c_int), ("reserved", C.c_int * 2)]


def bpp4fmt(fmt):
    return 3 if fmt in (0, 1) else (4 if fmt in (2, 3, 4, 5) else 0)


_LIB = None


def load_library():
    """dlopen lib/libcrthip.so (built by ``make -C ntsc-crt_amd`` / __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # CRTHIP_LIBDIR: another build of the same library (A/B measurements of a kernel change against the previous build)
    path = os.path.join(os.environ.get("CRTHIP_LIBDIR") or LIBDIR, "libcrthip.so")
    if not os.path.exists(path):
        raise RuntimeError("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    L = C.CDLL(path, mode=os.RTLD_GLOBAL)
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    PP = C.POINTER(Params)
    L.crthip_abi_version.restype = ci
    L.crthip_device_count.restype = ci
    L.crthip_params_default.argtypes = [PP, ci, ci]
    L.crthip_params_finalize.argtypes = [PP]
    L.crthip_input_size.argtypes = [ci, ci]
    L.crthip_hres.argtypes = [ci, ci]
    L.crthip_lines.argtypes = [ci]
    L.crthip_field_stride.argtypes = [ci, ci]
    L.crthip_field_stride.restype = sz
    L.crthip_create.argtypes = [C.POINTER(vp), ci, ci, ci]
    L.crthip_destroy.argtypes = [vp]
    L.crthip_destroy.restype = None
    L.crthip_set_stream.argtypes = [vp, vp]
    L.crthip_synchronize.argtypes = [vp]
    L.crthip_error_string.argtypes = [vp]
    L.crthip_error_string.restype = C.c_char_p
    L.crthip_reserve.argtypes = [vp, ci]
    L.crthip_fieldpass.argtypes = [vp, PP, ci, vp, sz, vp, sz, vp]
    L.crthip_modulate.argtypes = [vp, PP, ci, vp, sz, vp, vp]
    L.crthip_noise.argtypes = [vp, PP, ci, vp, vp, vp]
    L.crthip_sync.argtypes = [vp, PP, ci, vp, vp, vp]
    L.crthip_decode.argtypes = [vp, PP, ci, vp, vp, vp, sz]
    L.crthip_profile_enable.argtypes = [vp, ci]
    L.crthip_set_exact.argtypes = [vp, ci]
    L.crthip_vhs_history_from_seed.argtypes = [C.c_uint, C.POINTER(C.c_uint)]
    L.crthip_vhs_bind_history.argtypes = [vp, vp]
    L.crthip_set_overlap.argtypes = [vp, ci]
    L.crthip_set_shape.argtypes = [vp, ci]
    L.crthip_set_signal_tile.argtypes = [vp, ci]
    L.crthip_set_wide_lpw.argtypes = [vp, ci]
    L.crthip_set_signal_layout.argtypes = [vp, ci]
    L.crthip_signal_layout_query.argtypes = [PP, ci, ci, C.POINTER(ci), C.POINTER(sz)]
    L.crthip_fieldpass_signal.argtypes = [vp, ci, vp, C.POINTER(ci)]
    L.crthip_table_generation.argtypes = [vp]
    L.crthip_table_generation.restype = C.c_uint
    L.crthip_sequence.argtypes = [vp, PP, ci, vp, sz, vp, sz, vp, vp, C.POINTER(ci)]
    L.crthip_vhs_chain.argtypes = [vp, ci, vp, ci]
    L.crthip_seq_vhs_prechained.argtypes = [vp, ci]
    L.crthip_seq_encode.argtypes = [vp, PP, ci, ci, ci, vp, sz, vp]
    L.crthip_seq_sync.argtypes = [vp, PP, ci, vp, ci, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.crthip_seq_decode.argtypes = [vp, PP, ci, vp, sz, vp]
    L.crthip_seq_weave.argtypes = [vp, PP, ci, vp, sz, vp, ci]
    L.crthip_set_pixel_tile.argtypes = [vp, ci]
    L.crthip_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(ci)]
    _LIB = L
    return L


def make_params(system="ntsc", **kw):
    """crthip_params_default + user fields + crthip_params_finalize (host only, no GPU needed)."""
    L = load_library()
    name_in = system
    system, bloom = split_system(system)
    sysid, pattern = SYSTEMS[system]
    if bloom:
        kw["flags"] = kw.get("flags", 0) | F_BLOOM
    kw["flags"] = kw.get("flags", 0) | variant_flags(name_in)
    p = Params()
    rc = L.crthip_params_default(C.byref(p), sysid, pattern)
    if rc:
        raise ValueError("crthip_params_default failed (%d)" % rc)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    rc = L.crthip_params_finalize(C.byref(p))
    if rc:
        raise ValueError("crthip_params_finalize failed (%d)" % rc)
    return p


class Settings:
    """The reference's ``struct NTSC_SETTINGS`` for a batch: ``data`` is a device tensor holding n
    images ([n, h, w, bpp] uint8, or [n, h, w] int16/uint16 PPU pixels for NES); ``field`` /
    ``frame`` (or ``dot_crawl_offset`` for NES) may be ints or per-image sequences."""

    def __init__(self, data, format=FMT_BGRA, raw=0, as_color=1, field=0, frame=0, hue=0,
                 xoffset=0, yoffset=0, dot_crawl_offset=0, aberration=0, spare_row=None, border_color=0):
        self.data = data
        # CRTHIP_F_IMAGE_SPARE_ROW: every image is followed by one more readable row (the reference reads row h,
        # crt_ntsc.c:263).  None = detect from the tensor: stride(0) and the storage behind the last image cover it.
        self.spare_row = spare_row
        self.format, self.raw, self.as_color = format, raw, as_color
        self.field, self.frame, self.hue = field, frame, hue
        self.xoffset, self.yoffset = xoffset, yoffset
        self.dot_crawl_offset = dot_crawl_offset
        self.aberration = aberration
        self.border_color = border_color  # NES, NES_BORDER builds (crt_nes.h:136)
        self.initialized = 0          # iirs_initialized / field_initialized


class CRT:
    """A batch of n independent CRTs on one GPU (one ``struct CRT`` each in the reference)."""

    def __init__(self, n, outw, outh, out_format=FMT_BGRA, system="ntsc", device=0, out=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP path has no CPU fallback")
        self.torch = torch
        self.L = load_library()
        self.system = system
        self.base_system, self.bloom = split_system(system)
        self.sysid, self.pattern = SYSTEMS[self.base_system]
        self.dev = torch.device("cuda", device)
        self.n = n
        ctx = C.c_void_p()
        rc = self.L.crthip_create(C.byref(ctx), device, self.sysid, self.pattern)
        if rc:
            raise RuntimeError("crthip_create failed (%d): needs a gfx950 device" % rc)
        self.ctx = ctx
        self.input_size = self.L.crthip_input_size(self.sysid, self.pattern)
        self.hres = self.L.crthip_hres(self.sysid, self.pattern)
        self.lines = self.L.crthip_lines(self.sysid)
        self.fstride = self.L.crthip_field_stride(self.sysid, self.pattern)
        # crt_init (crt_core.c:263-289)
        self.outw, self.outh, self.out_format = outw, outh, out_format
        self.hue = self.brightness = self.black_point = 0
        self.saturation, self.contrast, self.white_point = 10, 180, 100
        self.scanlines = self.blend = 0
        self.v_fac = 0
        self.eq_fir = 0        # 0: the 3-band equaliser; 7/6/5/4: FIR kernel of a USE_CONVOLUTION build (crt_core.c:85-147)
        bpp = bpp4fmt(out_format) or 4
        self.out = out if out is not None else torch.zeros((n, outh, outw, bpp), dtype=torch.uint8, device=self.dev)
        self.state = torch.zeros((n, STATE_INTS), dtype=torch.int32, device=self.dev)
        self.state[:, ST_RN] = 194
        self._analog = None
        self._inp = None
        self._lines = None
        self._settings = None
        self.use_stream(None)
        self.vhs_hist = None
        if self.sysid == SYSTEM_VHS:
            # per-field rand() generator state (31-word history), see crthip_vhs_history_from_seed
            self.vhs_hist = torch.zeros((n, 32), dtype=torch.int32, device=self.dev)
            self._check(self.L.crthip_vhs_bind_history(self.ctx, C.c_void_p(self.vhs_hist.data_ptr())), "crthip_vhs_bind_history")
            self.srand([1] * n)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "ctx", None):
            self.L.crthip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.crthip_error_string(self.ctx).decode()))

    def use_stream(self, stream):
        """Launch on a torch stream (default: torch's current stream on this device)."""
        torch = self.torch
        s = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self._check(self.L.crthip_set_stream(self.ctx, C.c_void_p(s.cuda_stream)), "crthip_set_stream")

    def synchronize(self):
        self._check(self.L.crthip_synchronize(self.ctx), "crthip_synchronize")

    def reserve(self, n=None):
        self._check(self.L.crthip_reserve(self.ctx, n or self.n), "crthip_reserve")

    @property
    def analog(self):
        """Device analog[] of every field, [n, fstride] int8 (first input_size bytes are the field)."""
        if self._analog is None:
            self._analog = self.torch.zeros((self.n, self.fstride), dtype=self.torch.int8, device=self.dev)
        return self._analog

    @property
    def inp(self):
        if self._inp is None:
            self._inp = self.torch.zeros((self.n, self.fstride), dtype=self.torch.int8, device=self.dev)
        return self._inp

    @property
    def line_table(self):
        if self._lines is None:
            self._lines = self.torch.zeros((self.n, self.lines, LINE_INTS), dtype=self.torch.int32, device=self.dev)
        return self._lines

    def params(self, s, noise=0):
        """The batch-uniform parameter blob for settings ``s`` and this CRT's knobs."""
        d = s.data
        if self.sysid == SYSTEM_NES:
            h, w = int(d.shape[1]), int(d.shape[2])
        else:
            h, w = int(d.shape[1]), int(d.shape[2])
        flags = self.eq_fir << 8                                   # CRTHIP_F_EQ_FIR(taps)
        if getattr(s, "draw_aberration", 0):
            flags |= F_VHS_DRAW_ABERRATION                         # sequence mode
        if self._has_spare_row(s):
            flags |= F_IMAGE_SPARE_ROW
        if self.sysid in (SYSTEM_NES, SYSTEM_NESRGB) and not s.initialized:
            flags |= F_NES_SETUP
        return make_params(self.system, w=w, h=h, format=s.format, raw=s.raw, as_color=s.as_color, hue=s.hue,
                           xoffset=s.xoffset, yoffset=s.yoffset, outw=self.outw, outh=self.outh,
                           out_format=self.out_format, mon_hue=self.hue, brightness=self.brightness,
                           contrast=self.contrast, saturation=self.saturation, black_point=self.black_point,
                           white_point=self.white_point, scanlines=self.scanlines, blend=self.blend,
                           v_fac=self.v_fac, noise=noise, flags=flags, nes_border_color=getattr(s, "border_color", 0))

    def _has_spare_row(self, s):
        if s.spare_row is not None:
            return bool(s.spare_row)
        d = s.data
        row = d.stride(1) * d.element_size()
        need = (int(d.shape[1]) + 1) * row
        if d.shape[0] > 1 and d.stride(0) * d.element_size() < need:
            return False
        last = (d.storage_offset() + (int(d.shape[0]) - 1) * d.stride(0)) * d.element_size() + need
        return d.untyped_storage().nbytes() >= last

    def _load_field_state(self, s):
        torch = self.torch

        def col(v):
            if isinstance(v, int):
                return torch.full((self.n,), v, dtype=torch.int32, device=self.dev)
            return torch.as_tensor(list(v), dtype=torch.int32, device=self.dev)
        if self.sysid not in (SYSTEM_NES, SYSTEM_NESRGB):
            self.state[:, ST_FIELD] = col(s.field)
            self.state[:, ST_FRAME] = col(s.frame)
        self.state[:, ST_AUX] = col(s.dot_crawl_offset if self.sysid in DOT_CRAWL_SYSTEMS else s.aberration)

    def _image_stride(self, s):
        d = s.data
        assert d[0].is_contiguous() and d.device == self.dev   # images may be padded: only stride(0) is free
        return d.stride(0) * d.element_size()

    # ------------------------------------------------------------------ the hot path
    def modulate(self, s):
        """crt_modulate for every field of the batch (stage-level: analog[] is materialised)."""
        p = self.params(s)
        self._load_field_state(s)
        self._settings = s
        rc = self.L.crthip_modulate(self.ctx, C.byref(p), self.n, C.c_void_p(s.data.data_ptr()),
                                    self._image_stride(s), C.c_void_p(self.analog.data_ptr()),
                                    C.c_void_p(self.state.data_ptr()))
        self._check(rc, "crthip_modulate")
        s.initialized = 1

    def demodulate(self, noise):
        """crt_demodulate for every field (stage-level: noise -> sync -> decode on analog[])."""
        s = self._settings
        p = self.params(s, noise) if s is not None else make_params(
            self.system, w=1, h=1, outw=self.outw, outh=self.outh, out_format=self.out_format,
            mon_hue=self.hue, brightness=self.brightness, contrast=self.contrast, saturation=self.saturation,
            black_point=self.black_point, white_point=self.white_point, scanlines=self.scanlines,
            blend=self.blend, v_fac=self.v_fac, noise=noise, flags=self.eq_fir << 8)
        vp = C.c_void_p
        self._check(self.L.crthip_noise(self.ctx, C.byref(p), self.n, vp(self.analog.data_ptr()),
                                        vp(self.inp.data_ptr()), vp(self.state.data_ptr())), "crthip_noise")
        self._check(self.L.crthip_sync(self.ctx, C.byref(p), self.n, vp(self.inp.data_ptr()),
                                       vp(self.state.data_ptr()), vp(self.line_table.data_ptr())), "crthip_sync")
        self._check(self.L.crthip_decode(self.ctx, C.byref(p), self.n, vp(self.inp.data_ptr()),
                                         vp(self.line_table.data_ptr()), vp(self.out.data_ptr()),
                                         self.out.stride(0)), "crthip_decode")

    def fieldpass(self, s, noise, params=None):
        """modulate + demodulate fused for throughput: the encoder writes the noisy field directly,
        analog[] is never materialised, one launch sequence for the whole batch."""
        p = params if params is not None else self.params(s, noise)
        if params is None:
            self._load_field_state(s)
        rc = self.L.crthip_fieldpass(self.ctx, C.byref(p), self.n, C.c_void_p(s.data.data_ptr()),
                                     self._image_stride(s), C.c_void_p(self.out.data_ptr()),
                                     self.out.stride(0), C.c_void_p(self.state.data_ptr()))
        self._check(rc, "crthip_fieldpass")
        s.initialized = 1

    def sequence(self, s, noise, out_init=None):
        """The n images of ``s.data`` as n CONSECUTIVE fields of one television set (state and output
        buffer carried over, like extra/video_convert.c), processed in parallel.  ``self.state[0]`` holds
        the set's state before field 0; returns the number of sync fixed-point passes."""
        p = self.params(s, noise)
        self._load_field_state(s)
        passes = C.c_int(0)
        rc = self.L.crthip_sequence(self.ctx, C.byref(p), self.n, C.c_void_p(s.data.data_ptr()), self._image_stride(s),
                                    C.c_void_p(self.out.data_ptr()), self.out.stride(0),
                                    C.c_void_p(out_init.data_ptr()) if out_init is not None else None,
                                    C.c_void_p(self.state.data_ptr()), C.byref(passes))
        self._check(rc, "crthip_sequence")
        s.initialized = 1
        return passes.value

    # the phases of sequence(), for a video cut over several CRT objects / ranks (shard.sequence_sharded)
    def seq_encode(self, s, noise, first_index, rn0):
        """This object's n fields are fields [first_index, first_index + n) of the video; rn0 = the set's rn before field 0."""
        self._seq_p = self.params(s, noise)
        self._load_field_state(s)
        self._check(self.L.crthip_seq_encode(self.ctx, C.byref(self._seq_p), self.n, int(first_index), int(rn0) if rn0 < 2 ** 31 else int(rn0) - 2 ** 32,
                                             C.c_void_p(s.data.data_ptr()), self._image_stride(s), C.c_void_p(self.state.data_ptr())),
                    "crthip_seq_encode")
        s.initialized = 1

    def seq_sync(self, hsync_in, vsync_in):
        """The sync chain from the incoming pair; returns (hsync, vsync) after this object's last field."""
        ho, vo, ps = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.L.crthip_seq_sync(self.ctx, C.byref(self._seq_p), self.n, C.c_void_p(self.state.data_ptr()), int(hsync_in), int(vsync_in),
                                           C.byref(ho), C.byref(vo), C.byref(ps)), "crthip_seq_sync")
        return ho.value, vo.value

    def seq_decode(self):
        self._check(self.L.crthip_seq_decode(self.ctx, C.byref(self._seq_p), self.n, C.c_void_p(self.out.data_ptr()), self.out.stride(0),
                                             C.c_void_p(self.state.data_ptr())), "crthip_seq_decode")

    def seq_weave(self, out_init=None, patch_only=False):
        self._check(self.L.crthip_seq_weave(self.ctx, C.byref(self._seq_p), self.n, C.c_void_p(self.out.data_ptr()), self.out.stride(0),
                                            C.c_void_p(out_init.data_ptr()) if out_init is not None else None, int(bool(patch_only))),
                    "crthip_seq_weave")

    def last_picture(self):
        return self.out[self.n - 1]

    # ------------------------------------------------------------------ observation
    def srand(self, seeds):
        """VHS: put field k's rand() generator into the state right after srand(seeds[k])."""
        import numpy as np
        h = np.zeros((self.n, 32), dtype=np.uint32)
        buf = (C.c_uint * 31)()
        for k, sd in enumerate(seeds):
            self._check(self.L.crthip_vhs_history_from_seed(C.c_uint(sd & 0xffffffff), buf), "crthip_vhs_history_from_seed")
            h[k, :31] = np.frombuffer(buf, dtype=np.uint32)
        self.vhs_hist.copy_(self.torch.from_numpy(h.view(np.int32)).to(self.dev))

    def set_exact(self, on=True):
        """1/True: force the exact 32-bit-multiply kernels everywhere; 2: allow the 24-bit tier but not the
        64-bit-mad decoder tiers; 3: 64-bit-mad tier, but never drop the I/Q low cascades; 0: normal dispatch
        by proven operand range."""
        self._check(self.L.crthip_set_exact(self.ctx, int(on)), "crthip_set_exact")

    def set_shape(self, shape):
        """Kernel shape: SHAPE_AUTO (by batch size), SHAPE_LANE_PER_SCANLINE (throughput), SHAPE_SCANLINE_PARALLEL
        (latency; a DPP row of lanes per scanline)."""
        self._check(self.L.crthip_set_shape(self.ctx, int(shape)), "crthip_set_shape")

    def set_overlap(self, chunks):
        """fieldpass(): split the batch into `chunks` pieces alternating between two streams."""
        self._check(self.L.crthip_set_overlap(self.ctx, int(chunks)), "crthip_set_overlap")

    def table_generation(self):
        """rebuilds of the encoder's cached tables so far (a HIP graph captured at generation g replays with g's tables)"""
        return int(self.L.crthip_table_generation(self.ctx))

    def set_signal_tile(self, dwords):
        """fieldpass(): the encoder's signal tile -- 0 by batch size (default), 16 = 64-byte store pieces, 32 / 64 = the large ones"""
        self._check(self.L.crthip_set_signal_tile(self.ctx, int(dwords)), "crthip_set_signal_tile")

    def set_signal_layout(self, padded):
        """fieldpass(): 1 (default) = the signal between encoder and decoder in padded, aligned lines; 0 = the reference's flat layout"""
        self._check(self.L.crthip_set_signal_layout(self.ctx, int(bool(padded))), "crthip_set_signal_layout")

    def fieldpass_signal(self):
        """inp[] of the last fieldpass() in the reference's layout ([n, fstride] int8 like ``inp``) and whether it was kept padded"""
        dst = self.torch.zeros((self.n, self.fstride), dtype=self.torch.int8, device=self.dev)
        padded = C.c_int(0)
        self._check(self.L.crthip_fieldpass_signal(self.ctx, self.n, C.c_void_p(dst.data_ptr()), C.byref(padded)), "crthip_fieldpass_signal")
        return dst, bool(padded.value)

    def set_wide_lpw(self, lpw):
        """the wide-run decoder's scanlines per wavefront: 0 by batch size (default), 8 or 16 = always that instantiation"""
        self._check(self.L.crthip_set_wide_lpw(self.ctx, int(lpw)), "crthip_set_wide_lpw")

    def set_pixel_tile(self, px):
        self._check(self.L.crthip_set_pixel_tile(self.ctx, int(px)), "crthip_set_pixel_tile")

    def profile(self, on=True):
        self._check(self.L.crthip_profile_enable(self.ctx, int(on)), "crthip_profile_enable")

    def profile_read(self):
        ms = (C.c_double * 5)()
        cnt = (C.c_int * 5)()
        self._check(self.L.crthip_profile_read(self.ctx, ms, cnt), "crthip_profile_read")
        return {K_NAMES[k]: (ms[k], cnt[k]) for k in range(5)}

    def get(self, name):
        """hsync / vsync / rn of every field as a list (struct CRT members, crt_core.h:89-91)."""
        col = {"hsync": ST_HSYNC, "vsync": ST_VSYNC, "rn": ST_RN, "odd_field": ST_ODD}[name]
        return self.state[:, col].cpu().tolist()

    @property
    def ccf(self):
        return self.state[:, ST_CCF:ST_CCF + MAX_VPER * MAX_CCS].reshape(self.n, MAX_VPER, MAX_CCS).cpu().numpy()
