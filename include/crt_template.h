/*
 * crt_template.h -- timing and encoder settings for CRT_SYSTEM_TEMP, the reference's "template" system (drop-in
 * for the reference's header of the same name; written from scratch, see crt_core.h in this directory): standard
 * NTSC line timing, 227.5 colour cycles per line handled as a 2-line chroma period with per-line carrier tables,
 * band limit on.
 */
#ifndef _CRT_TEMP_H_
#define _CRT_TEMP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_CC_LINE      2275

#define CRT_CB_FREQ      4
#define CRT_HRES         (CRT_CC_LINE * CRT_CB_FREQ / 10)
#define CRT_VRES         262
#define CRT_INPUT_SIZE   (CRT_HRES * CRT_VRES)

#define CRT_TOP          21
#define CRT_BOT          261
#define CRT_LINES        (CRT_BOT - CRT_TOP)

#define CRT_CC_SAMPLES   4
#define CRT_CC_VPER      2

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

#define CRT_DO_BANDLIMITING 1
#define L_FREQ           1431818
#define Y_FREQ           420000
#define I_FREQ           150000
#define Q_FREQ           55000

/* signal levels, IRE */
#define WHITE_LEVEL      100
#define BURST_LEVEL      20
#define BLACK_LEVEL      7
#define BLANK_LEVEL      0
#define SYNC_LEVEL       -40
#define IRE_MAX          110
#define IRE_MIN          0

#define Q_OFFSET         (-90)
#define HUE_OFFSET       (-60)

#define SYNC_REGION_LO   3
#define SYNC_REGION_HI   6
#define EQU_REGION_A_LO  0
#define EQU_REGION_A_HI  2
#define EQU_REGION_B_LO  7
#define EQU_REGION_B_HI  9

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
    int dot_crawl_offset;       /* 0..5 */
    int iirs_initialized;
};

#ifdef __cplusplus
}
#endif
#endif
