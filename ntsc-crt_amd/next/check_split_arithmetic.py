#!/usr/bin/env python3
"""CPU check of the ARITHMETIC of next/crt_encode_wide_draft.hip (not of the HIP file): the per-component work split -- each of
three lanes converts ONLY its own component, runs ONE one-pole low-pass, multiplies by its own carrier (luma: 2^16), the three
high words are summed -- restated in Python and compared with the oracle's crt_modulate output (analog[], no noise) sample for
sample over the active rectangle.  Uses the host setup of libcrthip.so (crthip_params_finalize: no GPU needed) for the
constants, exactly as the kernel would get them.   usage: python ntsc-crt_amd/next/check_split_arithmetic.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ntsc-crt_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import crtref as R
import crtlib


def s32(v):
    v &= 0xffffffff
    return v - (1 << 32) if v & 0x80000000 else v


def mul24(a, b):
    """v_mul_i32_i24: low 24 bits of each operand, sign-extended, 32-bit wrapped product"""
    def s24(x):
        x &= 0xffffff
        return x - (1 << 24) if x & 0x800000 else x
    return s32(s24(a) * s24(b))


def check(name, w, h, fmt, field, frame, rows_to_check):
    orc = R.Oracle(name)
    img = R.synth_image(w, h, 4, 4242, "random")
    c = orc.new_crt(w, h, R.FMT_BGRA)
    c.settings(np.concatenate([img, img[-1:]]), format=fmt, w=w, h=h, as_color=1, field=field, frame=frame)
    c.modulate()
    an = c.analog.reshape(orc.sys.vres, orc.sys.hres)
    p = crtlib.make_params(name, w=w, h=h, outw=w, outh=h, in_format=fmt) if "in_format" in crtlib.make_params.__code__.co_varnames \
        else crtlib.make_params(name, w=w, h=h, outw=w, outh=h)
    blue_low = fmt in (R.FMT_BGR, R.FMT_BGRA, R.FMT_ABGR)
    alpha_first = fmt in (R.FMT_ARGB, R.FMT_ABGR)
    K = [((7471, 38470, 19595) if blue_low else (19595, 38470, 7471)),
         ((-21103, -18022, 39059) if blue_low else (39059, -18022, -21103)),
         ((20382, -34275, 13894) if blue_low else (13894, -34275, 20382))]
    near_y = p.iir_c[0] >= 1024            # IIR_Y_NEAR of the system traits (checked at launch against the coefficients)
    M = [s32(((p.iir_c[0] - 2048) if near_y else p.iir_c[0]) << 21), s32(p.iir_c[1] << 21), s32(p.iir_c[2] << 21)]
    crow = int((field & 1) == (frame & 1))
    W = [[65536] * 4, [p.modI[crow][k] * 4096 for k in range(4)], [p.modQ[crow][k] * 4096 for k in range(4)]]
    step = (p.col_step_hi << 32) | p.col_step_lo
    bad = 0
    for y in rows_to_check:
        field_offset = (field * h + p.desth) // p.desth // 2
        sy = min((y * h) // p.desth + field_offset, h - 1)
        hstate = [0, 0, 0]
        cpos = 0
        for x in range(p.destw):
            col = cpos >> 32
            px = img[sy, col]
            b = [int(px[0]), int(px[1]), int(px[2])] if not alpha_first else [int(px[1]), int(px[2]), int(px[3])]
            t = 0
            for comp in range(3):
                v = (K[comp][0] * b[0] + K[comp][1] * b[1] + K[comp][2] * b[2]) >> 14
                d = v - hstate[comp]
                base = v if (comp == 0 and near_y) else hstate[comp]
                hstate[comp] = s32(((M[comp] * d) + (base << 32)) >> 32)          # hi32 of the 64-bit multiply-add
                t += mul24(hstate[comp], W[comp][x & 3]) >> 16
            ire = (mul24(t, p.white) + p.ire_base * 1024) >> 10
            ire = max(0, min(110, ire))
            want = int(an[y + p.yo, p.xo + x])
            if ire != want:
                bad += 1
                if bad < 5:
                    print("  MISMATCH", name, "row", y, "x", x, "draft", ire, "oracle", want)
            cpos += step
    print("%-6s %4dx%-4d fmt %d field %d: %d rows x %d samples, %d mismatches" % (name, w, h, fmt, field, len(rows_to_check), p.destw, bad))
    return bad


if __name__ == "__main__":
    total = 0
    total += check("ntsc", 1920, 1080, R.FMT_BGRA, 0, 0, [0, 1, 117, 235])
    total += check("ntsc", 1280, 720, R.FMT_RGBA, 1, 0, [0, 100, 235])
    total += check("ntsc", 2560, 1440, R.FMT_ARGB, 1, 1, [5, 200])
    total += check("vhs", 1920, 1080, R.FMT_ABGR, 0, 1, [3, 230])
    sys.exit(1 if total else 0)
