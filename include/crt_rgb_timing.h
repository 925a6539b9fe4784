/*
 * crt_rgb_timing.h -- line timing shared by the two RGB-input systems of this build
 * (standard NTSC and NTSC-VHS).  Macro names follow the reference headers (crt_ntsc.h,
 * crt_ntscvhs.h) because callers and the analog-dump path use them (crt_main.c:261-267).
 */
#ifndef CRT_RGB_TIMING_H
#define CRT_RGB_TIMING_H

/* colour-subcarrier cycles per line x10: 2275 -> checkerboard chroma, 2280 -> vertical */
#ifndef CRT_CHROMA_PATTERN
#define CRT_CHROMA_PATTERN 1
#endif
#if (CRT_CHROMA_PATTERN == 1)
#define CRT_CC_LINE 2275
#else
#define CRT_CC_LINE 2280
#endif

#define CRT_CB_FREQ      4                                  /* samples per subcarrier cycle  */
#define CRT_HRES         (CRT_CC_LINE * CRT_CB_FREQ / 10)   /* samples per line              */
#define CRT_VRES         262                                /* lines per field               */
#define CRT_INPUT_SIZE   (CRT_HRES * CRT_VRES)

#define CRT_TOP          21                                 /* first / one-past-last picture line */
#define CRT_BOT          261
#define CRT_LINES        (CRT_BOT - CRT_TOP)

#define CRT_CC_SAMPLES   4
#define CRT_CC_VPER      1

#define CRT_HSYNC_WINDOW 8
#define CRT_VSYNC_WINDOW 8
#define CRT_HSYNC_THRESH 4
#define CRT_VSYNC_THRESH 94

/* horizontal line budget in nanoseconds */
#define LINE_BEG         0
#define FP_ns            1500
#define SYNC_ns          4700
#define BW_ns            600
#define CB_ns            2500
#define BP_ns            1600
#define AV_ns            52600
#define HB_ns            (FP_ns + SYNC_ns + BW_ns + CB_ns + BP_ns)
#define LINE_ns          (HB_ns + AV_ns)
#define ns2pos(ns)       ((ns) * CRT_HRES / LINE_ns)
#define FP_BEG           ns2pos(0)
#define SYNC_BEG         ns2pos(FP_ns)
#define BW_BEG           ns2pos(FP_ns + SYNC_ns)
#define CB_BEG           ns2pos(FP_ns + SYNC_ns + BW_ns)
#define BP_BEG           ns2pos(FP_ns + SYNC_ns + BW_ns + CB_ns)
#define AV_BEG           ns2pos(HB_ns)
#define AV_LEN           ns2pos(AV_ns)
#define CB_CYCLES        10

#define L_FREQ           1431818

/* signal levels, IRE */
#define WHITE_LEVEL      100
#define BURST_LEVEL      20
#define BLACK_LEVEL      7
#define BLANK_LEVEL      0
#define SYNC_LEVEL       -40

#endif
