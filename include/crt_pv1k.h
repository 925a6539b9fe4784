/*
 * crt_pv1k.h -- timing and encoder settings for CRT_SYSTEM_PV1K, the Casio PV-1000 (drop-in for the reference's
 * header of the same name; written from scratch, see crt_core.h in this directory): FIVE samples per colour cycle
 * (1920 samples per line), a 5-line chroma period, line budget counted in groups of four dots.
 */
#ifndef _CRT_PV1K_H_
#define _CRT_PV1K_H_

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_CC_LINE      2304

#define CRT_CB_FREQ      5
#define CRT_HRES         (CRT_CC_LINE * CRT_CB_FREQ / 6)
#define CRT_VRES         262
#define CRT_INPUT_SIZE   (CRT_HRES * CRT_VRES)

#define CRT_TOP          21
#define CRT_BOT          261
#define CRT_LINES        (CRT_BOT - CRT_TOP)

#define CRT_CC_SAMPLES   5
#define CRT_CC_VPER      5

#define CRT_HSYNC_WINDOW 8
#define CRT_VSYNC_WINDOW 8
#define CRT_HSYNC_THRESH 4
#define CRT_VSYNC_THRESH 94

/* horizontal line budget: one dot is 223 ns, everything is a multiple of 4 dots */
#define DOT_ns           223
#define DOTx4_ns         892
#define LINE_BEG         0
#define FP_ns            (3 * DOTx4_ns)
#define SYNC_ns          (3 * DOTx4_ns)
#define BW_ns            (2 * DOTx4_ns)
#define CB_ns            (4 * DOTx4_ns)
#define BP_ns            (4 * DOTx4_ns)
#define AV_ns            (55 * DOTx4_ns)
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
#define Y_FREQ           420000
#define I_FREQ           150000
#define Q_FREQ           55000

/* signal levels, IRE */
#define WHITE_LEVEL      100
#define BURST_LEVEL      20
#define BLACK_LEVEL      7
#define BLANK_LEVEL      0
#define SYNC_LEVEL       -40

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
