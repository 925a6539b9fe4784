/*
 * crt_snes.h -- timing and encoder settings for CRT_SYSTEM_SNES (drop-in for the reference's header of the same
 * name; written from scratch, see crt_core.h in this directory).  An RGB image encoded on the NES/SNES line raster
 * (341 PPU pixels per line, 227.3 colour cycles per line -> a 3-line chroma period), NTSC signal levels, no band
 * limit.
 */
#ifndef _CRT_SNES_H_
#define _CRT_SNES_H_

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_CC_LINE      2273

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

#define CRT_DO_BANDLIMITING 0
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

#define Q_OFFSET         (-90)   /* Q carrier relative to I, degrees */
#define HUE_OFFSET       (210)   /* burst relative to I, degrees     */

/* lines (inclusive) carrying vertical sync / equalising pulses */
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
    int field;                  /* unused by this system */
    int frame;                  /* unused by this system */
    int hue;
    int xoffset;
    int yoffset;
    int dot_crawl_offset;       /* 0..3 */
    int iirs_initialized;
};

#ifdef __cplusplus
}
#endif
#endif
