/*
 * crt_setup.h -- host-side (C89) setup arithmetic shared by the drop-in layer and the
 * batch C ABI: system timing tables, the fixed-point sin/cos and exp of the reference,
 * and crthip_params_finalize().  No device code, no allocation.
 */
#ifndef CRT_SETUP_H
#define CRT_SETUP_H

#include "crt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Line timing / level constants that the reference bakes in per CRT_SYSTEM through
 * crt_ntsc.h:23-109, crt_ntscvhs.h:23-130 and crt_nes.h:30-130. */
struct crt_sysdef {
    int system, chroma_pattern;
    int hres, vres, input_size;
    int top, bot, lines;
    int vper;                        /* CRT_CC_VPER */
    int hsync_window, vsync_window;
    int hsync_thresh, vsync_thresh;  /* already multiplied by sync_level */
    int sync_beg, bw_beg, cb_beg, av_beg, av_len;
    int vs_sep_end;                  /* NES: PPUpx2pos(327), crt_nes.c:95 */
    int white_level, burst_level, black_level, blank_level, sync_level;
    int y_freq, i_freq, q_freq;      /* encoder band limits (0 for NES) */
};

/* returns 0, or CRTHIP_E_ARG for a system outside SURVEY.md section 8 */
int crt_sysdef_get(struct crt_sysdef *d, int system, int chroma_pattern);

void crt_setup_sincos14(int *s, int *c, int n);   /* crt_core.c:42-61 */
int  crt_setup_expx(int n);                       /* crt_ntsc.c:41-83 */
int  crt_setup_bpp4fmt(int format);               /* crt_core.c:63-78 */

/* VHS rand() model (see crt_setup.c) */
void crt_setup_vhs_power(unsigned long k, unsigned c[31]);
void crt_setup_vhs_power_table(unsigned long first, unsigned long step, int count, unsigned *rows);

#define CRTHIP_PARAMS_MAGIC 0x43525431            /* "CRT1" */

#ifdef __cplusplus
}
#endif
#endif
