/*
 * crt_setup.c -- C89 host code: everything the reference computes once per call or
 * once per crt_init on the CPU stays on the CPU here too (last
 * bullet): timing tables, carrier phases, filter coefficients, geometry.  The result
 * is the `crthip_params` blob handed to the HIP kernels (and broadcast over RCCL in
 * the multi-GPU driver).
 */
#include "crt_setup.h"

#include <string.h>

/* ------------------------------------------------------------------------- */
/* fixed-point trig: 14-bit angle (16384 = one turn) -> 15-bit sine / cosine   */
/* reference: crt_core.c:19-61                                                */
/* ------------------------------------------------------------------------- */

/* first quadrant in 16 steps, one guard entry for the interpolation */
static const int sine_knots[18] = {
    0, 3208, 6392, 9512, 12536, 15440, 18200, 20784, 23168,
    25328, 27240, 28896, 30272, 31352, 32136, 32608, 32768, 32608
};

static int
sine_q1(int a)
{
    int k = (a >> 8) & 255;
    int t = a & 255;
    return sine_knots[k] + (((sine_knots[k + 1] - sine_knots[k]) * t) >> 8);
}

void
crt_setup_sincos14(int *s, int *c, int n)
{
    int a, sn, cs;

    n &= 16383;
    a = n & 8191;                 /* position inside the half turn */
    if (a < 4096) {
        sn = sine_q1(a);
        cs = sine_q1(4096 - a);
    } else {
        sn = sine_q1(8192 - a);
        cs = -sine_q1(a - 4096);
    }
    if (n & 8192) {               /* second half turn: both change sign */
        sn = -sn;
        cs = -cs;
    }
    *s = sn;
    *c = cs;
}

int
crt_setup_bpp4fmt(int format)
{
    switch (format) {
    case CRTHIP_FMT_RGB:
    case CRTHIP_FMT_BGR:
        return 3;
    case CRTHIP_FMT_ARGB:
    case CRTHIP_FMT_RGBA:
    case CRTHIP_FMT_ABGR:
    case CRTHIP_FMT_BGRA:
        return 4;
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------------- */
/* Q11 exponential, reference: crt_ntsc.c:25-83                                */
/* ------------------------------------------------------------------------- */
#define Q11 2048

int
crt_setup_expx(int n)
{
    /* e^0 .. e^4 in Q11 */
    static const int e_int[5] = { Q11, 5567, 15133, 41135, 111817 };
    int inverse, ip, r, series, term, fact, k;

    if (n == 0) {
        return Q11;
    }
    inverse = (n < 0);
    if (inverse) {
        n = -n;
    }
    ip = n >> 11;
    r = Q11;
    for (k = ip / 4; k > 0; k--) {
        r = (r * e_int[4]) >> 11;
    }
    if (ip & 3) {
        r = (r * e_int[ip & 3]) >> 11;
    }
    /* Taylor series of the fractional part; same truncation rule as the reference */
    n &= Q11 - 1;
    series = 0;
    term = Q11;
    fact = 1;
    for (k = 1; k < 17; k++) {
        series += term / fact;
        term = (term * n) >> 11;
        fact *= k;
        if (fact > term || term <= 0 || fact <= 0) {
            break;
        }
    }
    r = (r * series) >> 11;
    if (inverse) {
        r = (Q11 << 11) / r;
    }
    return r;
}

/* ------------------------------------------------------------------------- */
/* system tables                                                              */
/* ------------------------------------------------------------------------- */
#define L_FREQ 1431818

int
crt_sysdef_get(struct crt_sysdef *d, int system, int chroma_pattern)
{
    int cc_line;

    memset(d, 0, sizeof(*d));
    d->system = system;
    d->chroma_pattern = chroma_pattern;
    d->vres = 262;
    d->cc_samples = 4;
    d->cb_len = 40;
    d->burst_off = 33;
    d->q_off = -90;
    d->hue_in_mod = 1;
    d->equ_a_lo = 0; d->equ_a_hi = 3; d->equ_b_lo = 7; d->equ_b_hi = 9;     /* crt_ntsc.c:211 */
    d->vs_lo = 4; d->vs_hi = 6; d->vs_by_field = 1;                            /* crt_ntsc.c:217-223 */
    if (system == CRTHIP_SYSTEM_NTSC || system == CRTHIP_SYSTEM_NTSCVHS || system == CRTHIP_SYSTEM_TEMP) {
        /* crt_ntsc.h:25-109 (= crt_ntscvhs.h, crt_template.h): times in ns on a 63500 ns line */
        const int line_ns = 63500;
        if (system == CRTHIP_SYSTEM_TEMP) {
            if (chroma_pattern != 1) {
                return CRTHIP_E_ARG;
            }
        } else if (chroma_pattern != 0 && chroma_pattern != 1) {
            return CRTHIP_E_ARG;
        }
        cc_line = (chroma_pattern == 1) ? 2275 : 2280;
        d->hres = cc_line * 4 / 10;
        d->top = 21;
        d->bot = 261;
        d->vper = (system == CRTHIP_SYSTEM_TEMP) ? 2 : 1;
        d->hsync_window = 8;
        d->vsync_window = 8;
        d->white_level = 100;
        d->burst_level = 20;
        d->black_level = 7;
        d->blank_level = 0;
        d->sync_level = -40;
        d->sync_beg = 1500 * d->hres / line_ns;
        d->bw_beg = 6200 * d->hres / line_ns;
        d->cb_beg = 6800 * d->hres / line_ns;
        d->av_beg = 10900 * d->hres / line_ns;
        d->av_len = 52600 * d->hres / line_ns;
        d->field_rows = 1;
        if (system == CRTHIP_SYSTEM_NTSCVHS) {   /* VHS_SP, crt_ntscvhs.h:109-113 */
            d->y_freq = 300000;
            d->i_freq = 62700;
            d->q_freq = 62700;
        } else {                                 /* crt_ntsc.h:99-102, crt_template.h:107-110 */
            d->y_freq = 420000;
            d->i_freq = 150000;
            d->q_freq = 55000;
        }
        if (system == CRTHIP_SYSTEM_TEMP) {      /* crt_template.h:133-146, crt_template.c:171-183,240 */
            d->line_rows = 1;
            d->burst_off = -60 - 90;
            d->equ_a_hi = 2;
            d->vs_lo = 3;
            d->ccf_row_shift = 3;
        }
    } else if (system == CRTHIP_SYSTEM_PV1K) {
        /* crt_pv1k.h:24-98: 5 samples per chroma cycle; times in units of 4 dots (892 ns), 71 units per line */
        const int unit = 892, line_ns = 71 * 892;
        if (chroma_pattern != 1) {
            return CRTHIP_E_ARG;
        }
        d->hres = 2304 * 5 / 6;
        d->top = 21;
        d->bot = 261;
        d->vper = 5;
        d->cc_samples = 5;
        d->cb_len = 50;
        d->hsync_window = 8;
        d->vsync_window = 8;
        d->white_level = 100;
        d->burst_level = 20;
        d->black_level = 7;
        d->blank_level = 0;
        d->sync_level = -40;
        d->sync_beg = 3 * unit * d->hres / line_ns;
        d->bw_beg = 6 * unit * d->hres / line_ns;
        d->cb_beg = 8 * unit * d->hres / line_ns;
        d->av_beg = 16 * unit * d->hres / line_ns;
        d->av_len = 55 * unit * d->hres / line_ns;
        d->y_freq = 420000;
        d->i_freq = 150000;
        d->q_freq = 55000;
        d->field_rows = 1;
        d->line_rows = 1;
        d->burst_off = -72;                      /* crt_pv1k.c:172: n - step */
        d->q_off = 90;
        d->equ_a_hi = -1;                        /* crt_pv1k.c:197: only lines 7..9 */
        d->vs_lo = 258;                          /* crt_pv1k.c:204 */
        d->vs_hi = 260;
        d->ccf_row_shift = 3;
    } else if (system == CRTHIP_SYSTEM_NES || system == CRTHIP_SYSTEM_NESRGB || system == CRTHIP_SYSTEM_SNES) {
        /* crt_nes.h:30-126, crt_nesrgb.h, crt_snes.h:24-116: positions in PPU pixels on a 341 px line */
        const int line_px = 341;
        if (system == CRTHIP_SYSTEM_SNES) {
            if (chroma_pattern != 1) {
                return CRTHIP_E_ARG;
            }
            cc_line = 2273;
        } else {
            if (chroma_pattern < 0 || chroma_pattern > 2) {
                return CRTHIP_E_ARG;
            }
            cc_line = (chroma_pattern == 1) ? 2275 : ((chroma_pattern == 2) ? 2273 : 2280);
        }
        d->hres = cc_line * 4 / 10;
        d->top = 15;
        d->bot = 255;
        d->vper = 3;
        d->hsync_window = 6;
        d->vsync_window = 6;
        d->blank_level = 0;
        if (system == CRTHIP_SYSTEM_SNES) {
            d->white_level = 100;
            d->burst_level = 20;
            d->black_level = 7;
            d->sync_level = -40;
            d->burst_off = 210 - 90;             /* crt_snes.c:176: n - step + HUE_OFFSET */
            d->equ_a_hi = 2;                     /* crt_snes.h:137-146 */
            d->vs_lo = 3;
            d->vs_by_field = 0;                  /* crt_snes.c:216-218: even pattern only */
            d->ccf_row_shift = 3;
        } else {
            d->white_level = (system == CRTHIP_SYSTEM_NES) ? 110 : 100;
            d->burst_level = 30;
            d->black_level = 0;
            d->sync_level = -37;
            d->nes_timing = 1;
            d->ppu_input = (system == CRTHIP_SYSTEM_NES);
            if (system == CRTHIP_SYSTEM_NESRGB) {
                d->burst_off = 90 + 33;          /* crt_nesrgb.c:72 */
                d->hue_in_mod = 0;
            }
        }
        d->line_rows = 1;
        d->sync_beg = 9 * d->hres / line_px;
        d->bw_beg = 34 * d->hres / line_px;
        d->cb_beg = 38 * d->hres / line_px;
        d->av_beg = 74 * d->hres / line_px;
        d->av_len = 256 * d->hres / line_px;
        d->vs_sep_end = 327 * d->hres / line_px;
    } else {
        return CRTHIP_E_ARG;
    }
    d->vert_step = d->vper > 1 ? 360 / d->vper : 0;
    if (system == CRTHIP_SYSTEM_PV1K) {
        d->vert_step = 360 * 2 / 5;              /* crt_pv1k.c:168 */
    }
    d->input_size = d->hres * d->vres;
    d->lines = d->bot - d->top;
    d->hsync_thresh = 4 * d->sync_level;     /* CRT_HSYNC_THRESH */
    d->vsync_thresh = 94 * d->sync_level;    /* CRT_VSYNC_THRESH */
    return CRTHIP_OK;
}

int
crthip_input_size(int system, int chroma_pattern)
{
    struct crt_sysdef d;
    if (crt_sysdef_get(&d, system, chroma_pattern) != CRTHIP_OK) {
        return 0;
    }
    return d.input_size;
}

int
crthip_hres(int system, int chroma_pattern)
{
    struct crt_sysdef d;
    if (crt_sysdef_get(&d, system, chroma_pattern) != CRTHIP_OK) {
        return 0;
    }
    return d.hres;
}

int
crthip_lines(int system)
{
    struct crt_sysdef d;
    if (crt_sysdef_get(&d, system, (system == CRTHIP_SYSTEM_NES || system == CRTHIP_SYSTEM_NESRGB) ? 2 : 1) != CRTHIP_OK) {
        return 0;
    }
    return d.lines;
}

size_t
crthip_field_stride(int system, int chroma_pattern)
{
    /* INPUT_SIZE + room for the mirrored struct tail and for the widest out-of-contract
     * window (pos <= INPUT_SIZE-1, window AV_LEN, 16-byte vector reads), rounded to 256 */
    struct crt_sysdef d;
    size_t n;
    if (crt_sysdef_get(&d, system, chroma_pattern) != CRTHIP_OK) {
        return 0;
    }
    n = (size_t) d.input_size;
    n += (size_t) (d.av_len > 960 ? 2048 : 1024);
    return (n + 255) & ~(size_t) 255;
}

/* ------------------------------------------------------------------------- */
/* parameter blob                                                             */
/* ------------------------------------------------------------------------- */

int
crthip_params_default(crthip_params *p, int system, int chroma_pattern)
{
    struct crt_sysdef d;

    if (p == 0 || crt_sysdef_get(&d, system, chroma_pattern) != CRTHIP_OK) {
        return CRTHIP_E_ARG;
    }
    memset(p, 0, sizeof(*p));
    p->system = system;
    p->chroma_pattern = chroma_pattern;
    p->format = CRTHIP_FMT_BGRA;
    p->out_format = CRTHIP_FMT_BGRA;
    p->as_color = 1;
    /* crt_reset, crt_core.c:250-261 */
    p->saturation = 10;
    p->contrast = 180;
    p->white_point = 100;
    return CRTHIP_OK;
}

/* one-pole low-pass coefficient, crt_ntsc.c:98-106 (init_iir) */
static int
lowpass_coef(int limit)
{
    int rate = (L_FREQ << 9) / limit;
    return Q11 - crt_setup_expx(-((6434 << 9) / rate));
}

int
crthip_params_finalize(crthip_params *p)
{
    /* crt_core.c:272-286: band edges in kHz and band gains (Q16) for Y, I, Q */
    static const int edge_khz[3][2] = { { 1500, 3000 }, { 80, 1150 }, { 80, 1000 } };
    static const int band_gain4[3][3] = {
        { 65536, 8192, 9175 }, { 65536, 65536, 1311 }, { 65536, 65536, 0 }
    };
    static const int band_gain5[3][3] = {           /* CRT_CC_SAMPLES == 5, crt_core.c:281-283 */
        { 65536, 12192, 7775 }, { 65536, 65536, 1311 }, { 65536, 65536, 0 }
    };
    struct crt_sysdef d;
    int k, r, sn, cs, rows;

    if (p == 0 || crt_sysdef_get(&d, p->system, p->chroma_pattern) != CRTHIP_OK) {
        return CRTHIP_E_ARG;
    }
    p->finalized = 0;
    p->out_bpp = crt_setup_bpp4fmt(p->out_format);
    if (p->outw <= 0 || p->outh <= 0 || p->w <= 0 || p->h <= 0) {
        return CRTHIP_E_ARG;
    }
    p->eq_kernel = (p->flags & CRTHIP_F_EQ_FIR_MASK) >> 8;      /* crt_core.c:85-88 */
    if (p->eq_kernel != 0 && (p->eq_kernel < 4 || p->eq_kernel > 7)) {
        return CRTHIP_E_ARG;
    }
    p->bloom = (p->flags & CRTHIP_F_BLOOM) != 0;                /* crt_core.h:70 */
    if (p->bloom && (d.nes_timing || p->eq_kernel != 0)) {
        return CRTHIP_E_ARG;                                    /* "does not work for NES" (crt_core.h:70) */
    }
    if ((p->flags & (CRTHIP_F_VHS_LP | CRTHIP_F_VHS_EP | CRTHIP_F_VHS_LCG_NOISE)) && p->system != CRTHIP_SYSTEM_NTSCVHS) {
        return CRTHIP_E_ARG;
    }
    if ((p->flags & CRTHIP_F_VHS_LP) && (p->flags & CRTHIP_F_VHS_EP)) {
        return CRTHIP_E_ARG;
    }
    if ((p->flags & CRTHIP_F_NES_BORDER) && p->system != CRTHIP_SYSTEM_NES) {
        return CRTHIP_E_ARG;
    }

    memset(p->burst, 0, sizeof(p->burst));
    memset(p->modI, 0, sizeof(p->modI));
    memset(p->modQ, 0, sizeof(p->modQ));
    memset(p->dem_cs, 0, sizeof(p->dem_cs));
    memset(p->dem_sn, 0, sizeof(p->dem_sn));
    p->iir_c[0] = p->iir_c[1] = p->iir_c[2] = 0;
    if (d.ppu_input) {
        /* crt_nes.c:110-136.  Row r = (line % 3) + dot_crawl_offset: the reference reduces the ANGLE mod 360 with
         * C's truncating %, not the row, so the unreduced row keeps negative hues exact */
        p->in_bpp = 2;
        p->destw = d.av_len;
        p->desth = d.lines;
        p->xo = (d.av_beg + p->xoffset) & ~3;
        p->yo = d.top + p->yoffset;
        for (r = 0; r < CRTHIP_CARRIER_ROWS; r++) {
            for (k = 0; k < 4; k++) {
                int ang = (p->hue + k * 90 + r * 120 + 33) % 360;
                crt_setup_sincos14(&sn, &cs, ang * 8192 / 180);
                p->burst[r][k] = sn >> 10;
            }
        }
    } else {
        const int step = 360 / d.cc_samples;
        p->in_bpp = crt_setup_bpp4fmt(p->format);
        /* geometry: crt_ntsc.c:132-133, 148-173, 194-203; crt_nesrgb.c:51-52, 85-89 */
        p->destw = d.av_len;
        if (d.nes_timing) {
            p->desth = d.lines;
        } else if (p->bloom) {
            p->destw = (d.av_len * 55500) >> 16;
            p->desth = (d.lines * 63500) >> 16;
        } else {
            p->desth = (d.lines * 64500) >> 16;
        }
        if (p->raw && !d.nes_timing) {
            if (p->w < p->destw) {
                p->destw = p->w;
            }
            if (p->h < p->desth) {
                p->desth = p->h;
            }
        }
        p->xo = d.av_beg + p->xoffset + (d.av_len - p->destw) / 2;
        p->yo = d.top + p->yoffset + (d.lines - p->desth) / 2;
        if (d.line_rows && !d.nes_timing) {
            p->xo = p->xo - (p->xo % d.cc_samples);             /* crt_snes.c:201 */
        } else {
            p->xo &= ~3;
        }
        if (p->as_color || d.nes_timing) {                      /* NES-RGB has no as_color member: always colour */
            rows = d.line_rows ? CRTHIP_CARRIER_ROWS : 1;
            for (r = 0; r < rows; r++) {
                for (k = 0; k < d.cc_samples; k++) {
                    /* crt_ntsc.c:175-182; crt_snes.c:171-182; crt_pv1k.c:167-178; crt_nesrgb.c:67-78 */
                    int ang = r * d.vert_step + k * step;
                    int hue_mod = d.hue_in_mod ? p->hue : 0;
                    crt_setup_sincos14(&sn, &cs, (ang + p->hue + d.burst_off) * 8192 / 180);
                    p->burst[r][k] = sn >> 10;
                    crt_setup_sincos14(&sn, &cs, (ang + hue_mod) * 8192 / 180);
                    p->modI[r][k] = sn >> 10;
                    crt_setup_sincos14(&sn, &cs, (ang + hue_mod + d.q_off) * 8192 / 180);
                    p->modQ[r][k] = sn >> 10;
                }
            }
            if (!d.line_rows) {
                /* row 1 = the lines of a field whose parity equals the frame's (inv_phase, crt_ntsc.c:199-200,
                 * 243-247): burst advanced by half a cycle, modulation carriers negated -- CRT_CHROMA_PATTERN 1 only */
                for (k = 0; k < 4; k++) {
                    if (p->chroma_pattern == 1) {
                        p->burst[1][k] = p->burst[0][(k + 2) & 3];
                        p->modI[1][k] = -p->modI[0][k];
                        p->modQ[1][k] = -p->modQ[0][k];
                    } else {
                        p->burst[1][k] = p->burst[0][k];
                        p->modI[1][k] = p->modI[0][k];
                        p->modQ[1][k] = p->modQ[0][k];
                    }
                }
            }
        }
        if (d.y_freq != 0) {
            int yf = d.y_freq, jf = d.i_freq, qf = d.q_freq;
            if (p->system == CRTHIP_SYSTEM_NTSCVHS && (p->flags & CRTHIP_F_VHS_LP)) {        /* crt_ntscvhs.h:114-118 */
                yf = 240000; jf = 40000; qf = 40000;
            } else if (p->system == CRTHIP_SYSTEM_NTSCVHS && (p->flags & CRTHIP_F_VHS_EP)) { /* crt_ntscvhs.h:119-123 */
                yf = 200000; jf = 37000; qf = 37000;
            }
            p->iir_c[0] = lowpass_coef(yf);
            p->iir_c[1] = lowpass_coef(jf);
            p->iir_c[2] = lowpass_coef(qf);
        }
    }

    /* crt_core.c:171-196 with EQ_P = 16: cut-off as 2*sin(pi*f/rate), Q16 */
    for (k = 0; k < 3; k++) {
        int f_lo = d.hres * (edge_khz[k][0] * 100) / L_FREQ;
        int f_hi = d.hres * (edge_khz[k][1] * 100) / L_FREQ;
        crt_setup_sincos14(&sn, &cs, 8192 * f_lo / d.hres);
        p->eq_lf[k] = 2 * (sn << 1);
        crt_setup_sincos14(&sn, &cs, 8192 * f_hi / d.hres);
        p->eq_hf[k] = 2 * (sn << 1);
        for (r = 0; r < 3; r++) {
            p->eq_g[k][r] = d.cc_samples == 5 ? band_gain5[k][r] : band_gain4[k][r];
        }
    }

    /* crt_core.c:305, 318-320, 404-405, 528 */
    crt_setup_sincos14(&sn, &cs, ((p->mon_hue % 360) + 33) * 8192 / 180);
    p->huesn = sn >> 11;
    p->huecs = cs >> 11;
    if (d.cc_samples == 5) {
        /* crt_core.c:484,497-505: demodulation carriers rotated by the monitor hue, one entry per sample phase */
        int ang = p->mon_hue % 360;
        for (k = 0; k < 5; k++) {
            crt_setup_sincos14(&sn, &cs, ang * 8192 / 180);
            p->dem_cs[0][k] = cs;
            p->dem_sn[0][k] = sn;
            crt_setup_sincos14(&sn, &cs, (ang + 90) * 8192 / 180);
            p->dem_cs[1][k] = cs;
            p->dem_sn[1][k] = sn;
            ang += 360 / 5;
        }
    }
    p->bright = p->brightness - (d.black_level + p->black_point);
    p->white = d.white_level * p->white_point / 100;
    p->ire_base = d.black_level + p->black_point;
    p->dx = ((d.av_len - 1) << 12) / p->outw;
    p->ratio = (((p->outh << 16) / d.lines) + 32768) >> 16;
    {   /* ceil(2^32 * w / destw) by long division (C89: no 64-bit type) */
        unsigned dw = (unsigned) (p->destw > 0 ? p->destw : 1), r = (unsigned) p->w % dw, lo = 0;
        int i;
        p->col_step_hi = (unsigned) p->w / dw;
        for (i = 0; i < 32; i++) {
            r <<= 1;
            lo <<= 1;
            if (r >= dw) {
                r -= dw;
                lo |= 1u;
            }
        }
        if (r != 0 && ++lo == 0) {
            p->col_step_hi++;
        }
        p->col_step_lo = lo;
    }
    p->bloom_max_e = (128 + (p->noise / 2)) * d.av_len;         /* crt_core.c:400 */
    if (p->bloom && (p->bloom_max_e <= 0 || p->noise < 0)) {
        /* max_e <= 0: the reference divides by zero (:522).  noise < 0 shrinks max_e, the beam-energy term then drives
         * line_w past AV_LEN + 16 and scanL (:521) below zero: the reference reads in front of its line buffer
         * (undefined there), the row decoder would leave its LDS window -- refused */
        return CRTHIP_E_ARG;
    }
    p->loskip_wave_max = 65532;                                 /* |s| <= 127, see crt_hip.h */
    p->finalized = CRTHIP_PARAMS_MAGIC;
    return CRTHIP_OK;
}

/* NES composite level of PPU pixel p at phase `phase`, crt_nes.c:21-61 (host copy: only used to bound the signal) */
static int
nes_level(int p, int phase)
{
    static const int active[6] = { 0300, 0100, 0500, 0400, 0600, 0200 };
    static const int ire[16] = {
        -12042, 0, 34406, 81427,        /* low, no emphasis  */
        -17203, -8028, 19497, 57342,    /* low, emphasis     */
        43581, 75693, 112965, 112965,   /* high, no emphasis */
        26951, 52181, 83721, 83721      /* high, emphasis    */
    };
    int hue = p & 15, high, emph;

    if (hue >= 14) {
        return 0;
    }
    high = ((hue + phase) % 12) < 6;
    if (hue == 0) {
        high = 1;
    }
    if (hue == 13) {
        high = 0;
    }
    emph = (p & 0700 & active[(phase >> 1) % 6]) != 0;
    return ire[(high << 3) + (emph << 2) + ((p >> 4) & 3)];
}

/*
 * Range [*lo, *hi] of every sample of a field as crt_modulate leaves it in a crt_init-clean analog[] and the noise stage
 * of crt_demodulate hands it on (crt_core.c:359-366).  The struct bytes mirrored behind inp[] (CRTHIP_TAIL) are NOT
 * covered: a line whose window reaches them keeps the any-signal bound (line_tier_flags, crt_dev.h).  Used by
 * the fused entry points to widen the decoder's no-low-cascade envelope (crthip_params.loskip_wave_max); VHS, whose
 * noise gain is data dependent (crt_core.c:349-357), reports the full range.
 */
void
crt_setup_signal_range(const crthip_params *p, int *lo, int *hi)
{
    struct crt_sysdef d;
    int a_lo = 0, a_hi = 0, r, k, v, t0, t1;

    *lo = -128;
    *hi = 127;
    if (crt_sysdef_get(&d, p->system, p->chroma_pattern) != CRTHIP_OK ||
        (p->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE))) {
        return;
    }
    /* skeleton: blank, sync tip, burst (crt_ntsc.c:205-252, crt_nes.c:81-104) */
    if (d.sync_level < a_lo) a_lo = d.sync_level;
    if (d.blank_level > a_hi) a_hi = d.blank_level;
    for (r = 0; r < CRTHIP_CARRIER_ROWS; r++) {
        for (k = 0; k < CRTHIP_MAX_CCS; k++) {
            v = (int) (signed char) ((d.blank_level + p->burst[r][k] * d.burst_level) >> 5);
            if (v < a_lo) a_lo = v;
            if (v > a_hi) a_hi = v;
        }
    }
    /* active video */
    if (d.ppu_input) {
        int px, ph;
        for (px = 0; px < 512; px++) {                      /* crt_nes.c:162-193, every pixel value at every phase */
            for (ph = 0; ph < 12; ph++) {
                int ire = d.black_level + p->black_point + nes_level(px, ph) + nes_level(px, ph + 1) +
                          nes_level(px, ph + 2) + nes_level(px, ph + 3);
                v = (int) (signed char) ((ire * p->white_point / 100) >> 12);
                if (v < a_lo) a_lo = v;
                if (v > a_hi) a_hi = v;
            }
        }
    } else {
        if (110 > a_hi) a_hi = 110;                         /* clamped to 0..110, crt_ntsc.c:319-320 */
    }
    /* noise term ((byte - 0x7f) * noise) >> 8, byte = 0..255 (range check first: the products below must not overflow) */
    if (p->noise > (1 << 20) || p->noise < -(1 << 20)) {
        return;
    }
    t0 = (-127 * p->noise) >> 8;
    t1 = (128 * p->noise) >> 8;
    a_lo += t0 < t1 ? t0 : t1;
    a_hi += t0 < t1 ? t1 : t0;
    if (a_lo < -127) a_lo = -127;                           /* crt_core.c:363-364 */
    if (a_hi > 127) a_hi = 127;
    *lo = a_lo;
    *hi = a_hi;
}

/* largest |wave| for which the chroma equaliser input u = (s * wave) >> 9, s in [lo, hi], keeps every state of the
 * cascades inside the hull H of the inputs and 0 with max|H| <= 32767 and width(H) <= 32767 (see eq_step64 in
 * crt_decode.hip): |u| <= |s| * W / 512 + 1 */
int
crt_setup_loskip_bound(int lo, int hi)
{
    int m, wd;
    long a, b;

    if (lo > 0) lo = 0;
    if (hi < 0) hi = 0;
    m = -lo > hi ? -lo : hi;
    wd = hi - lo;
    if (m <= 0 || wd <= 0) {
        return 65532;
    }
    a = (32766L * 512L) / m;
    b = (32765L * 512L) / wd;
    if (b < a) a = b;
    if (a < 65532L) a = 65532L;
    if (a > 120000L) a = 120000L;                           /* T0_WAVE_MAX: the 64-bit-mad tiers end there anyway */
    return (int) a;
}

/* ------------------------------------------------------------------------- */
/* VHS: the C library's rand() (crt_core.c:344,349-351; crt_ntscvhs.c:206)    */
/* ------------------------------------------------------------------------- */
/*
 * The reference's VHS noise is whatever the process's rand() returns; on the platform the reference
 * is built for (glibc) that is random()'s TYPE_3 additive feedback generator:
 *     y[n] = y[n-31] + y[n-3]  (mod 2^32),   rand() = y[n] >> 1
 * seeded by srand(seed): y[0] = seed, y[i] = 16807*y[i-1] mod (2^31-1) for i < 31, y[31..33] =
 * y[0..2], and the first 310 outputs discarded.  A generator state is handed to the kernels as its
 * "history": the 31 values y[n-31 .. n-1] that precede the next output.
 */
int
crthip_vhs_history_from_seed(unsigned seed, unsigned hist[31])
{
    unsigned y[34 + 310 + 31];
    int i;
    long word;

    if (hist == 0) {
        return CRTHIP_E_ARG;
    }
    if (seed == 0) {
        seed = 1;
    }
    y[0] = seed;
    word = (long) (int) seed;
    for (i = 1; i < 31; i++) {
        /* Schrage: 16807 * word mod 2147483647 without overflow (glibc srandom_r) */
        long hi = word / 127773;
        long lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) {
            word += 2147483647;
        }
        y[i] = (unsigned) word;
    }
    for (i = 31; i < 34; i++) {
        y[i] = y[i - 31];
    }
    for (i = 34; i < 34 + 310; i++) {
        y[i] = y[i - 31] + y[i - 3];
    }
    /* the next output is y[344]; its history is y[313 .. 343] */
    for (i = 0; i < 31; i++) {
        hist[i] = y[313 + i];
    }
    return CRTHIP_OK;
}

/* coefficients of x^k modulo x^31 - x^28 - 1 over Z/2^32: y[a+k] = sum_m c[m] * y[a+m] */
static void
poly_mul_mod(const unsigned *a, const unsigned *b, unsigned *out)
{
    unsigned t[61];
    int i, j;

    for (i = 0; i < 61; i++) {
        t[i] = 0;
    }
    for (i = 0; i < 31; i++) {
        if (a[i] == 0) {
            continue;
        }
        for (j = 0; j < 31; j++) {
            t[i + j] += a[i] * b[j];
        }
    }
    for (i = 60; i >= 31; i--) {       /* x^i = x^(i-3) + x^(i-31) */
        t[i - 3] += t[i];
        t[i - 31] += t[i];
    }
    for (i = 0; i < 31; i++) {
        out[i] = t[i];
    }
}

void
crt_setup_vhs_power(unsigned long k, unsigned c[31])
{
    unsigned base[31], acc[31], tmp[31];
    int i;

    for (i = 0; i < 31; i++) {
        base[i] = 0;
        acc[i] = 0;
    }
    base[1] = 1;    /* x   */
    acc[0] = 1;     /* 1   */
    while (k) {
        if (k & 1ul) {
            poly_mul_mod(acc, base, tmp);
            memcpy(acc, tmp, sizeof(acc));
        }
        poly_mul_mod(base, base, tmp);
        memcpy(base, tmp, sizeof(base));
        k >>= 1;
    }
    memcpy(c, acc, sizeof(acc));
}

/* rows[q] = coefficients of x^(first + q*step), q = 0 .. count-1 */
void
crt_setup_vhs_power_table(unsigned long first, unsigned long step, int count, unsigned *rows)
{
    unsigned stepc[31];
    int q;

    crt_setup_vhs_power(step, stepc);
    crt_setup_vhs_power(first, rows);
    for (q = 1; q < count; q++) {
        poly_mul_mod(rows + 31 * (q - 1), stepc, rows + 31 * q);
    }
}
