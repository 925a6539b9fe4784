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
    int y_freq, i_freq, q_freq;      /* encoder band limits (0: no band limit) */
    int cc_samples, cb_len;          /* CRT_CC_SAMPLES; CB_CYCLES * CRT_CB_FREQ */
    /* what differs between the RGB encoders crt_ntsc.c / crt_ntscvhs.c / crt_snes.c / crt_template.c / crt_pv1k.c /
     * crt_nesrgb.c */
    int ppu_input;                   /* NES: the image is 9-bit PPU pixels (crt_nes.c) */
    int nes_timing;                  /* NES, NES-RGB: setup_field skeleton, full-height progressive geometry */
    int field_rows;                  /* source row offset by field parity (crt_ntsc.c:258) */
    int line_rows;                   /* carrier tables per line class + dot_crawl_offset (crt_snes.c:171-183) */
    int vert_step;                   /* degrees per line class */
    int burst_off, q_off;            /* burst / Q carrier angle relative to the I carrier, degrees */
    int hue_in_mod;                  /* the encoder hue rotates the modulation carriers too (not NES-RGB, crt_nesrgb.c:72-77) */
    int equ_a_lo, equ_a_hi, equ_b_lo, equ_b_hi;   /* equalising-pulse lines, inclusive */
    int vs_lo, vs_hi, vs_by_field;   /* vertical sync lines, inclusive; odd-field pattern (crt_ntsc.c:219-223) */
    int ccf_row_shift;               /* ccf preset row = (line + shift) % vper: 3 in crt_snes.c:240, 0 in crt_nes.c:177 */
};

/* returns 0, or CRTHIP_E_ARG for a system outside SURVEY.md section 8 */
int crt_sysdef_get(struct crt_sysdef *d, int system, int chroma_pattern);

void crt_setup_sincos14(int *s, int *c, int n);   /* crt_core.c:42-61 */
int  crt_setup_expx(int n);                       /* crt_ntsc.c:41-83 */
int  crt_setup_bpp4fmt(int format);               /* crt_core.c:63-78 */

/* decoder envelope helpers (see crt_setup.c) */
void crt_setup_signal_range(const crthip_params *p, int *lo, int *hi);
int  crt_setup_loskip_bound(int lo, int hi);

/* VHS rand() model (see crt_setup.c) */
void crt_setup_vhs_power(unsigned long k, unsigned c[31]);
void crt_setup_vhs_power_table(unsigned long first, unsigned long step, int count, unsigned *rows);

#define CRTHIP_PARAMS_MAGIC 0x43525431            /* "CRT1" */

#ifdef __cplusplus
}
#endif
#endif
