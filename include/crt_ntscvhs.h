/*
 * crt_ntscvhs.h -- encoder settings for CRT_SYSTEM_NTSCVHS (drop-in for the reference's header
 * of the same name; written from scratch, see crt_core.h in this directory).
 */
#ifndef _CRT_NTSC_VHS_H_
#define _CRT_NTSC_VHS_H_

#ifdef __cplusplus
extern "C" {
#endif

#include "crt_rgb_timing.h"

/* 1 (shipped): the noise comes from the C library's rand() stream; -DCRT_VHS_NOISE=0: the LCG of the other systems
 * (crt_ntscvhs.h:29, crt_core.c:343-357) */
#ifndef CRT_VHS_NOISE
#define CRT_VHS_NOISE    1
#endif

/* tape speed (crt_ntscvhs.h:102-124): -DVHS_MODE=1 (LP) / 2 (EP); link the matching libntsccrt_hip_vhs_<lp|ep>.so */
#define VHS_SP 0
#define VHS_LP 1
#define VHS_EP 2
#ifndef VHS_MODE
#define VHS_MODE VHS_SP
#endif

/* encoder band limits in units of 10 Hz */
#if (VHS_MODE == VHS_SP)
#define Y_FREQ           300000
#define I_FREQ           62700
#define Q_FREQ           62700
#elif (VHS_MODE == VHS_LP)
#define Y_FREQ           240000
#define I_FREQ           40000
#define Q_FREQ           40000
#else
#define Y_FREQ           200000
#define I_FREQ           37000
#define Q_FREQ           37000
#endif

/* Zero the whole struct before first use (iirs_initialized is library state). */
struct NTSC_SETTINGS {
    const unsigned char *data;
    int format;
    int w, h;
    int raw;
    int as_color;
    int field;
    int frame;
    int hue;
    int xoffset;
    int yoffset;
    int do_aberration;          /* 1: tracking error band at the bottom of the picture */
    int iirs_initialized;
};

#ifdef __cplusplus
}
#endif
#endif
