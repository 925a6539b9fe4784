/*
 * crt_nesrgb.h -- timing and encoder settings for CRT_SYSTEM_NESRGB (drop-in for the reference's header of the same
 * name; written from scratch, see crt_core.h in this directory): an RGB image encoded with the NES's line timing and
 * chroma artifacts (progressive, no band limit).  Same raster as crt_nes.h, WHITE_LEVEL 100.
 */
#ifndef _CRT_NESRGB_H_
#define _CRT_NESRGB_H_

#ifdef __cplusplus
extern "C" {
#endif

/* 0 vertical, 1 checkerboard, 2 sawtooth chroma (the NES' own 227.3 cycles per line) */
#ifndef CRT_CHROMA_PATTERN
#define CRT_CHROMA_PATTERN 2
#endif
#if (CRT_CHROMA_PATTERN == 1)
#define CRT_CC_LINE 2275
#elif (CRT_CHROMA_PATTERN == 2)
#define CRT_CC_LINE 2273
#else
#define CRT_CC_LINE 2280
#endif

#define CRT_CB_FREQ      4
#define CRT_HRES         (CRT_CC_LINE * CRT_CB_FREQ / 10)
#define CRT_VRES         262
#define CRT_INPUT_SIZE   (CRT_HRES * CRT_VRES)

#define CRT_TOP          15
#define CRT_BOT          255
#define CRT_LINES        (CRT_BOT - CRT_TOP)

#define CRT_CC_SAMPLES   4
#define CRT_CC_VPER      3

#define CRT_HSYNC_WINDOW 6
#define CRT_VSYNC_WINDOW 6
#define CRT_HSYNC_THRESH 4
#define CRT_VSYNC_THRESH 94

/* horizontal line budget in PPU pixels (341 per line) */
#define LINE_BEG         0
#define FP_PPUpx         9
#define SYNC_PPUpx       25
#define BW_PPUpx         4
#define CB_PPUpx         15
#define BP_PPUpx         5
#define PS_PPUpx         1
#define LB_PPUpx         15
#define AV_PPUpx         256
#define RB_PPUpx         11
#define HB_PPUpx         (FP_PPUpx + SYNC_PPUpx + BW_PPUpx + CB_PPUpx + BP_PPUpx)
#define LINE_PPUpx       (HB_PPUpx + PS_PPUpx + LB_PPUpx + AV_PPUpx + RB_PPUpx)
#define PPUpx2pos(PPUpx) ((PPUpx) * CRT_HRES / LINE_PPUpx)
#define FP_BEG           PPUpx2pos(0)
#define SYNC_BEG         PPUpx2pos(FP_PPUpx)
#define BW_BEG           PPUpx2pos(FP_PPUpx + SYNC_PPUpx)
#define CB_BEG           PPUpx2pos(FP_PPUpx + SYNC_PPUpx + BW_PPUpx)
#define BP_BEG           PPUpx2pos(FP_PPUpx + SYNC_PPUpx + BW_PPUpx + CB_PPUpx)
#define LAV_BEG          PPUpx2pos(HB_PPUpx)
#define AV_BEG           PPUpx2pos(HB_PPUpx + PS_PPUpx + LB_PPUpx)
#define AV_LEN           PPUpx2pos(AV_PPUpx)
#define CB_CYCLES        10

#define L_FREQ           1431818

/* signal levels, IRE */
#define WHITE_LEVEL      100
#define BURST_LEVEL      30
#define BLACK_LEVEL      0
#define BLANK_LEVEL      0
#define SYNC_LEVEL       -37

/* Zero the whole struct before first use (field_initialized is library state). */
struct NTSC_SETTINGS {
    const unsigned char *data;
    int format;
    int w, h;
    int dot_crawl_offset;       /* 0..2 */
    int hue;
    int xoffset;
    int yoffset;
    int field_initialized;
};

#ifdef __cplusplus
}
#endif
#endif
