/*
 * crt_ntsc.h -- encoder settings for CRT_SYSTEM_NTSC (drop-in for the reference's header of the
 * same name; written from scratch, see crt_core.h in this directory).
 */
#ifndef _CRT_NTSC_H_
#define _CRT_NTSC_H_

#ifdef __cplusplus
extern "C" {
#endif

#include "crt_rgb_timing.h"

/* encoder band limits in units of 10 Hz */
#define Y_FREQ           420000
#define I_FREQ           150000
#define Q_FREQ           55000

/* Zero the whole struct before first use (iirs_initialized is library state). */
struct NTSC_SETTINGS {
    const unsigned char *data;  /* source image                                  */
    int format;                 /* its CRT_PIX_FORMAT_*                          */
    int w, h;                   /* its size                                      */
    int raw;                    /* 1: copy 1:1 instead of fitting to the raster   */
    int as_color;               /* 0: monochrome                                 */
    int field;                  /* 0 even, 1 odd                                 */
    int frame;                  /* 0 even, 1 odd                                 */
    int hue;                    /* degrees                                       */
    int xoffset;                /* samples                                       */
    int yoffset;                /* lines                                         */
    int iirs_initialized;
};

#ifdef __cplusplus
}
#endif
#endif
