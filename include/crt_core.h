/*
 * crt_core.h -- drop-in replacement header for the public API of LMP88959/NTSC-CRT v2.3.2,
 * backed by the MI355X (gfx950) HIP implementation in libntsccrt_hip_<system>.so.
 *
 * This file is NOT the reference's header: it is written from scratch so that code written
 * against the reference (crt_main.c, extra/video_convert.c, emulator front ends) compiles and
 * links unchanged.  What has to be identical is therefore identical: the macro names the
 * callers use, the member names, order and types of `struct CRT` / `struct NTSC_SETTINGS`
 * (callers poke the members directly), and the seven entry points.
 * The layouts are pinned by _Static_assert-style checks in ntsc-crt_amd/csrc/crt_api.c and by
 * tests/test_dropin_layout.py against the reference compiled in oracle/_ref.
 *
 * As in the reference, the emulated system is a compile-time choice (-DCRT_SYSTEM=n, all seven systems of the
 * reference) and each system is a separate library: libntsccrt_hip_{ntsc,nes,pv1k,snes,temp,vhs,nesrgb}.so.
 * CRT_DO_BLOOM, an unguarded #define in the reference (crt_core.h:70), is a -D option here; the bloom builds are
 * libntsccrt_hip_<system>_bloom.so.
 */
#ifndef _CRT_CORE_H_
#define _CRT_CORE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* version of the reference API this header is compatible with */
#define CRT_MAJOR 2
#define CRT_MINOR 3
#define CRT_PATCH 2

/* selectable systems (values as in the reference) */
#define CRT_SYSTEM_NTSC     0
#define CRT_SYSTEM_NES      1
#define CRT_SYSTEM_PV1K     2
#define CRT_SYSTEM_SNES     3
#define CRT_SYSTEM_TEMP     4
#define CRT_SYSTEM_NTSCVHS  5
#define CRT_SYSTEM_NESRGB   6

#ifndef CRT_SYSTEM
#define CRT_SYSTEM CRT_SYSTEM_NTSC
#endif

#if (CRT_SYSTEM == CRT_SYSTEM_NTSC)
#include "crt_ntsc.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS)
#include "crt_ntscvhs.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_NES)
#include "crt_nes.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_SNES)
#include "crt_snes.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_PV1K)
#include "crt_pv1k.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_TEMP)
#include "crt_template.h"
#elif (CRT_SYSTEM == CRT_SYSTEM_NESRGB)
#include "crt_nesrgb.h"
#else
#error "No system defined: CRT_SYSTEM must be one of CRT_SYSTEM_NTSC .. CRT_SYSTEM_NESRGB (0-6)"
#endif

/* pixel byte orders; alpha is written as 0xff and never read */
#define CRT_PIX_FORMAT_RGB  0
#define CRT_PIX_FORMAT_BGR  1
#define CRT_PIX_FORMAT_ARGB 2
#define CRT_PIX_FORMAT_RGBA 3
#define CRT_PIX_FORMAT_ABGR 4
#define CRT_PIX_FORMAT_BGRA 5

/* decoder features.  CRT_DO_BLOOM (beam-energy dependent line width; black borders; not for the NES systems) is
 * 0 in the shipped reference; build with -DCRT_DO_BLOOM=1 and link libntsccrt_hip_<system>_bloom.so for the other one */
#ifndef CRT_DO_BLOOM
#define CRT_DO_BLOOM    0
#endif
/* look for VSYNC / HSYNC (crt_core.h:71-72): -DCRT_DO_VSYNC=0 / -DCRT_DO_HSYNC=0 give the reference's other builds */
#ifndef CRT_DO_VSYNC
#define CRT_DO_VSYNC    1
#endif
#ifndef CRT_DO_HSYNC
#define CRT_DO_HSYNC    1
#endif

/* One television set.  The caller owns this object and the `out` image; the library keeps
 * device-side mirrors and re-reads / writes back the host copy on every call, so direct
 * member access between calls behaves as with the CPU implementation. */
struct CRT {
    signed char analog[CRT_INPUT_SIZE];  /* composite signal of one field, IRE units     */
    signed char inp[CRT_INPUT_SIZE];     /* the same after the noisy channel             */

    int outw, outh;                      /* output image geometry                         */
    int out_format;                      /* CRT_PIX_FORMAT_*                              */
    unsigned char *out;                  /* output image (read back when blend != 0)      */

    int hue, brightness, contrast, saturation;
    int black_point, white_point;
    int scanlines;                       /* leave one dark row between scanlines          */
    int blend;                           /* average with the previous picture             */
    unsigned v_fac;                      /* extra vertical stretch                        */

    /* carried from field to field */
    int ccf[CRT_CC_VPER][CRT_CC_SAMPLES];
    int hsync, vsync;
    int rn;
};

extern void crt_init(struct CRT *v, int w, int h, int f, unsigned char *out);
extern void crt_resize(struct CRT *v, int w, int h, int f, unsigned char *out);
extern void crt_reset(struct CRT *v);
extern void crt_modulate(struct CRT *v, struct NTSC_SETTINGS *s);
extern void crt_demodulate(struct CRT *v, int noise);
extern int  crt_bpp4fmt(int format);

/* 14-bit angles: 16384 units per turn */
#define T14_2PI           16384
#define T14_MASK          (T14_2PI - 1)
#define T14_PI            (T14_2PI / 2)

extern void crt_sincos14(int *s, int *c, int n);

#ifdef __cplusplus
}
#endif

#endif
