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

#define CRT_VHS_NOISE    1

#define VHS_SP 0
#define VHS_LP 1
#define VHS_EP 2
#define VHS_MODE VHS_SP       /* the only tape speed this build provides */

/* encoder band limits in units of 10 Hz (SP) */
#define Y_FREQ           300000
#define I_FREQ           62700
#define Q_FREQ           62700

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
