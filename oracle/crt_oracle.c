/*
 * crt_oracle.c -- TEST INFRASTRUCTURE ONLY (see crt_oracle.h).
 *
 * Stage-structured CPU restatement of the reference's per-field composite
 * encode -> noisy channel -> decode path.  Every function cites the reference
 * file:line (under /root/reference) whose arithmetic it follows.  All integer
 * arithmetic is 32-bit two's complement with wrap (-fwrapv), `>>` of negatives is
 * arithmetic, `/` and `%` truncate toward zero -- exactly what the reference gets
 * from gcc/clang on x86-64.
 *
 * Stages (names used throughout the repo):
 *   M2 carriers  M4 blanking/sync/burst skeleton  M5 active video  M6 ccf preset
 *   D1 noise  D2 vsync  D3/D4 row map  D5 hsync  D6 burst lock  D7 carrier table
 *   D8 equalisers  D9 resample+YIQ->RGB  D10 row duplication
 */
#include "crt_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define POSMOD(x, n) ((((x) % (n)) + (n)) % (n))   /* crt_core.c:17 */

/* ------------------------------------------------------------------------- */
/* L0: fixed-point trig / exp                                                 */
/* ------------------------------------------------------------------------- */

/* crt_core.c:19-24: quarter sine sampled every 256/16384 of a turn, 15-bit */
static const int quarter_sine15[18] = {
    0x0000, 0x0c88, 0x18f8, 0x2528, 0x30f8, 0x3c50, 0x4718, 0x5130, 0x5a80,
    0x62f0, 0x6a68, 0x70e0, 0x7640, 0x7a78, 0x7d88, 0x7f60, 0x8000, 0x7f60
};

/* crt_core.c:26-39: linear interpolation inside one of 16 segments */
static int
sine_seg(int n)
{
    int frac = n & 0xff;
    int seg = (n >> 8) & 0xff;
    int lo = quarter_sine15[seg];
    int hi = quarter_sine15[seg + 1];
    return lo + (((hi - lo) * frac) >> 8);
}

/* crt_core.c:42-61 */
void
orc_sincos14(int *s, int *c, int n)
{
    int half, sn, cs;

    n &= 16383;
    half = n & 8191;
    if (half >= 4096) {
        cs = -sine_seg(half - 4096);
        sn = sine_seg(8192 - half);
    } else {
        cs = sine_seg(4096 - half);
        sn = sine_seg(half);
    }
    if (n >= 8192) {
        cs = -cs;
        sn = -sn;
    }
    *s = sn;
    *c = cs;
}

/* crt_core.c:63-78 */
int
orc_bpp4fmt(int format)
{
    if (format == 0 || format == 1) return 3;
    if (format >= 2 && format <= 5) return 4;
    return 0;
}

/* crt_ntsc.c:25-83: Q11 e^x: integer part from a table of e^1..e^4, fraction
 * from the Taylor series with early termination */
int
orc_expx(int n)
{
    static const int epow[5] = { 2048, 5567, 15133, 41135, 111817 };
    int negative, whole, res, term, sum, fact, k;

    if (n == 0) return 2048;
    negative = n < 0;
    if (negative) n = -n;
    whole = n >> 11;
    res = 2048;
    for (k = 0; k < whole / 4; k++) res = (res * epow[4]) >> 11;
    whole &= 3;
    if (whole > 0) res = (res * epow[whole]) >> 11;

    n &= 2047;
    term = 2048;
    sum = 0;
    fact = 1;
    for (k = 1; k < 17; k++) {
        sum += term / fact;
        term = (term * n) >> 11;
        fact *= k;
        if (fact > term || term <= 0 || fact <= 0) break;
    }
    res = (res * sum) >> 11;
    if (negative) res = (2048 << 11) / res;
    return res;
}

/* crt_ntsc.c:98-106 init_iir */
static int
iir_coef(int freq, int limit)
{
    int rate = (freq << 9) / limit;
    return 2048 - orc_expx(-((6434 << 9) / rate));
}

/* ------------------------------------------------------------------------- */
/* system tables                                                              */
/* ------------------------------------------------------------------------- */

void
orc_sys_init(struct orc_sys *sys, int system, int chroma_pattern)
{
    static const int khz[3][2] = { {1500, 3000}, {80, 1150}, {80, 1000} }; /* crt_core.c:272-286 */
    static const int gains4[3][3] = { {65536, 8192, 9175}, {65536, 65536, 1311}, {65536, 65536, 0} };
    static const int gains5[3][3] = { {65536, 12192, 7775}, {65536, 65536, 1311}, {65536, 65536, 0} };
    const int l_freq = 1431818;
    int cc_line, k, b, sn, cs;
    int yf = 420000, ifr = 150000, qf = 55000;       /* crt_ntsc.h:99-102 and every other RGB header */

    memset(sys, 0, sizeof(*sys));
    sys->system = system;
    sys->chroma_pattern = chroma_pattern;
    sys->vres = 262;
    sys->cc_samples = 4;
    sys->cb_len = 40;                                /* CB_CYCLES 10 * CRT_CB_FREQ 4 */
    if (system == ORC_SYS_NES || system == ORC_SYS_NESRGB || system == ORC_SYS_SNES) {
        /* crt_nes.h:30-126, crt_nesrgb.h (same timing, WHITE_LEVEL 100), crt_snes.h:24-116 (same PPU-pixel timing,
         * NTSC levels): positions in PPU pixels on a 341 px line */
        const int line_px = 9 + 25 + 4 + 15 + 5 + 1 + 15 + 256 + 11; /* 341 */
        cc_line = system == ORC_SYS_SNES ? 2273 : (chroma_pattern == 1 ? 2275 : (chroma_pattern == 2 ? 2273 : 2280));
        sys->hres = cc_line * 4 / 10;
        sys->top = 15;
        sys->bot = 255;
        sys->cc_vper = 3;
        sys->hsync_window = 6;
        sys->vsync_window = 6;
        if (system == ORC_SYS_SNES) {
            sys->white_level = 100; sys->burst_level = 20; sys->black_level = 7; sys->sync_level = -40;
        } else {
            sys->white_level = system == ORC_SYS_NES ? 110 : 100;
            sys->burst_level = 30; sys->black_level = 0; sys->sync_level = -37;
        }
        sys->blank_level = 0;
#define PPU2POS(p) ((p) * sys->hres / line_px)
        sys->sync_beg = PPU2POS(9);
        sys->bw_beg = PPU2POS(9 + 25);
        sys->cb_beg = PPU2POS(9 + 25 + 4);
        sys->lav_beg = PPU2POS(58);
        sys->av_beg = PPU2POS(58 + 1 + 15);
        sys->av_len = PPU2POS(256);
        sys->vs_sep_end = PPU2POS(327);  /* crt_nes.c:95 */
#undef PPU2POS
    } else if (system == ORC_SYS_PV1K) {
        /* crt_pv1k.h:24-98: 5 samples per chroma cycle, times in units of 4 dots (892 ns) */
        const int d4 = 892;
        const int line_ns = (3 + 3 + 2 + 4 + 4 + 55) * d4;
        sys->hres = 2304 * 5 / 6;
        sys->top = 21;
        sys->bot = 261;
        sys->cc_vper = 5;
        sys->cc_samples = 5;
        sys->cb_len = 50;
        sys->hsync_window = 8;
        sys->vsync_window = 8;
        sys->white_level = 100; sys->burst_level = 20; sys->black_level = 7; sys->blank_level = 0; sys->sync_level = -40;
#define NS2POS(ns) ((ns) * sys->hres / line_ns)
        sys->sync_beg = NS2POS(3 * d4);
        sys->bw_beg = NS2POS(6 * d4);
        sys->cb_beg = NS2POS(8 * d4);
        sys->av_beg = NS2POS(16 * d4);
        sys->av_len = NS2POS(55 * d4);
#undef NS2POS
    } else {
        /* crt_ntsc.h:25-109 (crt_ntscvhs.h identical except Y/I/Q_FREQ; crt_template.h: CC_LINE 2275, VPER 2) */
        const int line_ns = 1500 + 4700 + 600 + 2500 + 1600 + 52600; /* 63500 */
        cc_line = (system == ORC_SYS_TEMP || chroma_pattern == 1) ? 2275 : 2280;
        sys->hres = cc_line * 4 / 10;
        sys->top = 21;
        sys->bot = 261;
        sys->cc_vper = system == ORC_SYS_TEMP ? 2 : 1;
        sys->hsync_window = 8;
        sys->vsync_window = 8;
        sys->white_level = 100;
        sys->burst_level = 20;
        sys->black_level = 7;
        sys->blank_level = 0;
        sys->sync_level = -40;
#define NS2POS(ns) ((ns) * sys->hres / line_ns)
        sys->sync_beg = NS2POS(1500);
        sys->bw_beg = NS2POS(1500 + 4700);
        sys->cb_beg = NS2POS(1500 + 4700 + 600);
        sys->av_beg = NS2POS(1500 + 4700 + 600 + 2500 + 1600);
        sys->av_len = NS2POS(52600);
#undef NS2POS
        if (system == ORC_SYS_VHS) { yf = 300000; ifr = 62700; qf = 62700; } /* VHS_SP, crt_ntscvhs.h:109-113 */
    }
    /* what differs between the RGB encoders (crt_ntsc.c / crt_ntscvhs.c / crt_snes.c / crt_template.c / crt_pv1k.c /
     * crt_nesrgb.c) */
    sys->enc_bandlimit = system == ORC_SYS_NTSC || system == ORC_SYS_VHS || system == ORC_SYS_TEMP || system == ORC_SYS_PV1K;
    sys->enc_field_rows = sys->enc_bandlimit;        /* the same four systems: crt_ntsc.c:258, crt_template.c:255, crt_pv1k.c:243 */
    sys->enc_line_rows = system == ORC_SYS_SNES || system == ORC_SYS_TEMP || system == ORC_SYS_PV1K || system == ORC_SYS_NESRGB;
    sys->vert_step = sys->cc_vper > 1 ? 360 / sys->cc_vper : 0;
    if (system == ORC_SYS_PV1K) sys->vert_step = 360 * 2 / 5;                 /* crt_pv1k.c:168 */
    sys->burst_off = 33; sys->q_off = -90;                                     /* crt_ntsc.c:177-182 */
    if (system == ORC_SYS_SNES) { sys->burst_off = 210 - 90; sys->q_off = -90; }   /* crt_snes.c:176-181: n - step + HUE_OFFSET */
    if (system == ORC_SYS_TEMP) { sys->burst_off = -60 - 90; sys->q_off = -90; }   /* crt_template.c:176-181 */
    if (system == ORC_SYS_PV1K) { sys->burst_off = -72; sys->q_off = 90; }         /* crt_pv1k.c:172-177 */
    if (system == ORC_SYS_NESRGB) { sys->burst_off = 90 + 33; sys->q_off = -90; }  /* crt_nesrgb.c:72-77 */
    sys->equ_a_lo = 0; sys->equ_a_hi = 3; sys->equ_b_lo = 7; sys->equ_b_hi = 9;   /* crt_ntsc.c:211 */
    sys->vs_lo = 4; sys->vs_hi = 6; sys->vs_by_field = 1;                          /* crt_ntsc.c:217-223 */
    if (system == ORC_SYS_SNES || system == ORC_SYS_TEMP) {                        /* crt_snes.h:137-146 */
        sys->equ_a_hi = 2; sys->vs_lo = 3; sys->vs_by_field = system == ORC_SYS_TEMP;
    }
    if (system == ORC_SYS_PV1K) {                                                  /* crt_pv1k.c:197,204 */
        sys->equ_a_lo = 0; sys->equ_a_hi = -1; sys->vs_lo = 258; sys->vs_hi = 260;
    }
    if (sys->enc_bandlimit) {
        sys->iir_c[0] = iir_coef(l_freq, yf);
        sys->iir_c[1] = iir_coef(l_freq, ifr);
        sys->iir_c[2] = iir_coef(l_freq, qf);
    }
    sys->input_size = sys->hres * sys->vres;
    sys->lines = sys->bot - sys->top;
    sys->hsync_thresh = 4 * sys->sync_level;
    sys->vsync_thresh = 94 * sys->sync_level;

    /* crt_core.c:171-196 init_eq with EQ_P 16, called from crt_init :272-286 */
    for (k = 0; k < 3; k++) {
        int f_lo = sys->hres * (khz[k][0] * 100) / l_freq;
        int f_hi = sys->hres * (khz[k][1] * 100) / l_freq;
        orc_sincos14(&sn, &cs, 8192 * f_lo / sys->hres);
        sys->eq_lf[k] = 2 * (sn << 1);
        orc_sincos14(&sn, &cs, 8192 * f_hi / sys->hres);
        sys->eq_hf[k] = 2 * (sn << 1);
        for (b = 0; b < 3; b++) sys->eq_g[k][b] = sys->cc_samples == 5 ? gains5[k][b] : gains4[k][b];
    }
    sys->eq_kernel = 0;
    sys->do_bloom = 0;
}

/* crt_ntscvhs.h:102-124: Y / I / Q band limits of the three tape speeds */
void
orc_sys_set_vhs_mode(struct orc_sys *sys, int mode)
{
    static const int freq[3][3] = { { 300000, 62700, 62700 }, { 240000, 40000, 40000 }, { 200000, 37000, 37000 } };
    int k;
    if (sys->system != ORC_SYS_VHS || mode < 0 || mode > 2) return;
    for (k = 0; k < 3; k++) sys->iir_c[k] = iir_coef(1431818, freq[mode][k]);
}

/* crt_core.c:241-289 crt_init = memset + crt_resize + crt_reset + rn seed */
void
orc_crt_init(const struct orc_sys *sys, struct orc_crt *v,
             int8_t *analog, int8_t *inp, int w, int h, int f, uint8_t *out)
{
    memset(v, 0, sizeof(*v));
    v->analog = analog;
    v->inp = inp;
    memset(analog, 0, (size_t) sys->input_size);
    memset(inp, 0, (size_t) sys->input_size + ORC_TAIL);
    v->outw = w;
    v->outh = h;
    v->out_format = f;
    v->out = out;
    v->saturation = 10;
    v->contrast = 180;
    v->white_point = 100;
    v->rn = 194;
}

/* ------------------------------------------------------------------------- */
/* encoder                                                                    */
/* ------------------------------------------------------------------------- */

static void
fill_run(int8_t *line, int from, int to, int level)
{
    int t;
    for (t = from; t < to; t++) line[t] = (int8_t) level;
}

/* read one pixel as r,g,b for the 6 byte orders, crt_ntsc.c:278-305 */
static void
fetch_rgb(const uint8_t *p, int format, int *r, int *g, int *b)
{
    switch (format) {
    case 0: case 3: *r = p[0]; *g = p[1]; *b = p[2]; break; /* RGB, RGBA */
    case 1: case 5: *r = p[2]; *g = p[1]; *b = p[0]; break; /* BGR, BGRA */
    case 2:         *r = p[1]; *g = p[2]; *b = p[3]; break; /* ARGB */
    case 4:         *r = p[3]; *g = p[2]; *b = p[1]; break; /* ABGR */
    default:        *r = *g = *b = 0; break;
    }
}

/* RGB -> YIQ, Q14 (crt_ntsc.c:306-310, identical in every RGB encoder) */
static void
rgb_to_yiq(int r, int g, int b, int *fy, int *fi, int *fq)
{
    *fy = (19595 * r + 38470 * g + 7471 * b) >> 14;
    *fi = (39059 * r - 18022 * g - 21103 * b) >> 14;
    *fq = (13894 * r - 34275 * g + 20382 * b) >> 14;
}

/* The RGB encoders: crt_ntsc.c:128-330, crt_ntscvhs.c:129-338, crt_snes.c:125-327, crt_template.c:125-337,
 * crt_pv1k.c:121-321.  They are one routine with the per-system switches of struct orc_sys. */
static void
modulate_rgb(const struct orc_sys *sys, struct orc_crt *v, struct orc_settings *s)
{
    const int hres = sys->hres, ccs = sys->cc_samples, vper = sys->cc_vper;
    int destw = sys->av_len;
    int desth = (sys->lines * 64500) >> 16;
    int burst[ORC_MAX_VPER][ORC_MAX_CCS], modI[ORC_MAX_VPER][ORC_MAX_CCS], modQ[ORC_MAX_VPER][ORC_MAX_CCS];
    int preset[ORC_MAX_VPER][ORC_MAX_CCS];
    int x, y, n, k, xo, yo, bpp, inv_phase = 0, ph = 1, sn, cs, aberration = 0;
    int white;
    const uint8_t *data = (const uint8_t *) s->data;

    memset(preset, 0, sizeof(preset));
    s->initialized = 1;                              /* M0, crt_ntsc.c:142-147 (coefs live in sys) */
    if (sys->do_bloom) {                             /* M1, crt_ntsc.c:148-161 */
        destw = (sys->av_len * 55500) >> 16;
        desth = (sys->lines * 63500) >> 16;
        if (s->raw) {
            destw = s->w < destw ? s->w : destw;
            desth = s->h < desth ? s->h : desth;
        }
    } else if (s->raw) {                             /* crt_ntsc.c:163-172 */
        destw = s->w < sys->av_len ? s->w : sys->av_len;
        desth = s->h < desth ? s->h : desth;
    }
    memset(burst, 0, sizeof(burst));
    memset(modI, 0, sizeof(modI));
    memset(modQ, 0, sizeof(modQ));
    if (s->as_color) {                               /* M2 */
        const int step = 360 / ccs;
        for (y = 0; y < vper; y++) {
            /* crt_ntsc.c:174-188 (one row);  crt_snes.c:171-183, crt_pv1k.c:167-179 (one row per line class) */
            int vert = sys->enc_line_rows ? (y + s->dot_crawl_offset) * sys->vert_step : 0;
            for (k = 0; k < ccs; k++) {
                int ang = vert + s->hue + k * step;
                orc_sincos14(&sn, &cs, (ang + sys->burst_off) * 8192 / 180);
                burst[y][k] = sn >> 10;
                orc_sincos14(&sn, &cs, ang * 8192 / 180);
                modI[y][k] = sn >> 10;
                orc_sincos14(&sn, &cs, (ang + sys->q_off) * 8192 / 180);
                modQ[y][k] = sn >> 10;
            }
        }
    }
    bpp = orc_bpp4fmt(s->format);
    if (bpp == 0) return;                            /* crt_ntsc.c:190-193 */

    xo = sys->av_beg + s->xoffset + (sys->av_len - destw) / 2;   /* M3, crt_ntsc.c:194-203 */
    yo = sys->top + s->yoffset + (sys->lines - desth) / 2;
    s->field &= 1;
    s->frame &= 1;
    if (!sys->enc_line_rows) {
        inv_phase = (s->field == s->frame);
        if (sys->chroma_pattern == 1) ph = (inv_phase & 1) ? -1 : 1;
        xo &= ~3;
    } else {
        xo = xo - (xo % ccs);                        /* crt_snes.c:201 */
    }

    if (sys->system == ORC_SYS_VHS && s->do_aberration) {         /* crt_ntscvhs.c:205-207 */
        aberration = ((rand() % 12) - 8) + 14;
    }

    for (n = 0; n < sys->vres; n++) {                /* M4, crt_ntsc.c:205-252 */
        int8_t *line = v->analog + n * hres;
        if ((n >= sys->equ_a_lo && n <= sys->equ_a_hi) || (n >= sys->equ_b_lo && n <= sys->equ_b_hi)) {   /* equalising pulses */
            fill_run(line, 0, 4 * hres / 100, sys->sync_level);
            fill_run(line, 4 * hres / 100, 50 * hres / 100, sys->blank_level);
            fill_run(line, 50 * hres / 100, 54 * hres / 100, sys->sync_level);
            fill_run(line, 54 * hres / 100, hres, sys->blank_level);
        } else if (n >= sys->vs_lo && n <= sys->vs_hi) {   /* vertical sync, field-dependent where the system says so */
            int a = ((sys->vs_by_field && s->field == 1) ? 4 : 46) * hres / 100;
            fill_run(line, 0, a, sys->sync_level);
            fill_run(line, a, 50 * hres / 100, sys->blank_level);
            fill_run(line, 50 * hres / 100, 96 * hres / 100, sys->sync_level);
            fill_run(line, 96 * hres / 100, hres, sys->blank_level);
        } else {                                     /* ordinary line */
            int t0 = 0;
            if (n < sys->vres - aberration) {        /* always true unless VHS aberration */
                fill_run(line, 0, sys->sync_beg, sys->blank_level);
                fill_run(line, sys->sync_beg, sys->bw_beg, sys->sync_level);
                t0 = sys->bw_beg;
            }
            fill_run(line, t0, sys->av_beg, sys->blank_level);
            if (n < sys->top) fill_run(line, sys->av_beg, hres, sys->blank_level);
            for (k = sys->cb_beg; k < sys->cb_beg + sys->cb_len; k++) {
                int cb;
                if (sys->enc_line_rows) cb = burst[n % vper][k % ccs];                     /* crt_snes.c:238 */
                else cb = sys->chroma_pattern == 1 ? burst[0][(k + inv_phase * 2) % 4] : burst[0][k % 4];
                line[k] = (int8_t) ((sys->blank_level + cb * sys->burst_level) >> 5);
                if (sys->enc_line_rows) preset[(n + 3) % vper][k % ccs] = line[k];        /* crt_snes.c:240 */
                else preset[0][k % 4] = line[k];
            }
        }
    }
    if (sys->system == ORC_SYS_VHS) v->hsync = 0;   /* crt_ntscvhs.c:258-259 */

    white = sys->white_level * v->white_point / 100;
    for (y = 0; y < desth; y++) {                    /* M5, crt_ntsc.c:254-324 */
        int hy = 0, hi = 0, hq = 0;                  /* the three 1-pole states, reset per line */
        int field_offset = sys->enc_field_rows ? (s->field * s->h + desth) / desth / 2 : 0;
        int sy = (y * s->h) / desth + field_offset;
        int row = sys->enc_line_rows ? (y + yo) % vper : 0;       /* crt_snes.c:258 */
        if (sy >= s->h) sy = s->h;                   /* (sic) crt_ntsc.c:263 */
        sy *= s->w;
        for (x = 0; x < destw; x++) {
            int r, g, b, fy, fi, fq, ire, xoff;
            fetch_rgb(data + (((x * s->w) / destw) + sy) * bpp, s->format, &r, &g, &b);
            rgb_to_yiq(r, g, b, &fy, &fi, &fq);
            ire = sys->black_level + v->black_point;
            xoff = (x + xo) % ccs;
            if (sys->enc_bandlimit) {                /* iirf, crt_ntsc.c:117-126 */
                hy += ((fy - hy) * sys->iir_c[0]) >> 11;
                hi += ((fi - hi) * sys->iir_c[1]) >> 11;
                hq += ((fq - hq) * sys->iir_c[2]) >> 11;
            } else {                                 /* crt_snes.c:113-122 with CRT_DO_BANDLIMITING 0 */
                hy = fy; hi = fi; hq = fq;
            }
            {
                /* iirf's return value: the state, or with HIPASS 1 (crt_ntsc.c:121-122) input minus state */
                const int hp = sys->hipass && sys->enc_bandlimit;
                const int oy = hp ? fy - hy : hy, oi = hp ? fi - hi : hi, oq = hp ? fq - hq : hq;
                fy = oy;
                if (sys->enc_line_rows) {
                    fi = oi * modI[row][xoff] >> 4;
                    fq = oq * modQ[row][xoff] >> 4;
                } else {
                    fi = oi * ph * modI[0][xoff] >> 4;
                    fq = oq * ph * modQ[0][xoff] >> 4;
                }
            }
            ire += (fy + fi + fq) * white >> 10;
            if (ire < 0) ire = 0;
            if (ire > 110) ire = 110;
            v->analog[(x + xo) + (y + yo) * hres] = (int8_t) ire;
        }
    }
    for (n = 0; n < vper; n++) {                     /* M6, crt_ntsc.c:325-329 / vhs :332-336 / crt_snes.c:321-325 */
        for (k = 0; k < ccs; k++) v->ccf[n][k] = sys->system == ORC_SYS_VHS ? 0 : preset[n][k] << 7;
    }
}

/* sync skeleton of the NES-timed systems, crt_nes.c:81-104 / crt_nesrgb.c:24-46 (setup_field) */
static void
nes_setup_field(const struct orc_sys *sys, struct orc_crt *v)
{
    int n;
    for (n = 0; n < sys->vres; n++) {
        int8_t *line = v->analog + n * sys->hres;
        fill_run(line, 0, sys->sync_beg, sys->blank_level);
        fill_run(line, sys->sync_beg, n >= 259 ? sys->vs_sep_end : sys->bw_beg, sys->sync_level);
        fill_run(line, n >= 259 ? sys->vs_sep_end : sys->bw_beg, sys->hres, sys->blank_level);
    }
}

/* crt_nesrgb.c:48-170: an RGB image encoded with the NES's line timing; no band limit, progressive */
static void
modulate_nesrgb(const struct orc_sys *sys, struct orc_crt *v, struct orc_settings *s)
{
    const int hres = sys->hres;
    const uint8_t *data = (const uint8_t *) s->data;
    int burst[3][4], modI[3][4], modQ[3][4], preset[3][4];
    int x, y, n, k, xo, yo, sn, cs, bpp, white;

    memset(preset, 0, sizeof(preset));
    if (!s->initialized) {
        nes_setup_field(sys, v);
        s->initialized = 1;
    }
    for (y = 0; y < 3; y++) {                        /* :67-78: the hue only rotates the burst */
        int rot = (y + s->dot_crawl_offset) * 120;
        for (k = 0; k < 4; k++) {
            n = rot + k * 90;
            orc_sincos14(&sn, &cs, (s->hue + 90 + n + 33) * 8192 / 180);
            burst[y][k] = sn >> 10;
            orc_sincos14(&sn, &cs, n * 8192 / 180);
            modI[y][k] = sn >> 10;
            orc_sincos14(&sn, &cs, (n - 90) * 8192 / 180);
            modQ[y][k] = sn >> 10;
        }
    }
    bpp = orc_bpp4fmt(s->format);
    if (bpp == 0) return;                            /* :80-83 */
    xo = (sys->av_beg + s->xoffset) & ~3;            /* :85-89 */
    yo = sys->top + s->yoffset;
    white = sys->white_level * v->white_point / 100;

    for (y = 0; y < sys->lines; y++) {               /* :91-163 */
        int sy = (y * s->h) / sys->lines;
        int8_t *line;
        if (sy >= s->h) sy = s->h;
        if (sy < 0) sy = 0;
        n = y + yo;
        line = v->analog + n * hres;
        n %= 3;
        for (k = sys->cb_beg; k < sys->cb_beg + sys->cb_len; k++) {
            int cb = burst[n][k % 4];
            line[k] = (int8_t) ((sys->blank_level + cb * sys->burst_level) >> 5);
            preset[n][k % 4] = line[k];
        }
        sy *= s->w;
        for (x = 0; x < sys->av_len; x++) {
            int r, g, b, fy, fi, fq, ire, xoff;
            fetch_rgb(data + (((x * s->w) / sys->av_len) + sy) * bpp, s->format, &r, &g, &b);
            rgb_to_yiq(r, g, b, &fy, &fi, &fq);
            ire = sys->black_level + v->black_point;
            xoff = (x + xo) % 4;
            fi = fi * modI[n][xoff] >> 4;
            fq = fq * modQ[n][xoff] >> 4;
            ire += (fy + fi + fq) * white >> 10;
            if (ire < 0) ire = 0;
            if (ire > 110) ire = 110;
            v->analog[(x + xo) + (y + yo) * hres] = (int8_t) ire;
        }
    }
    for (n = 0; n < 3; n++) {                        /* :165-169 */
        for (k = 0; k < 4; k++) v->ccf[n][k] = preset[n][k] << 7;
    }
}

/* crt_nes.c:21-61: one quarter-sample of the PPU's square-wave output */
static int
ppu_level(int p, int phase)
{
    static const int level[16] = {
        -12042, 0, 34406, 81427,        /* low,  colours x0..x3 */
        -17203, -8028, 19497, 57342,    /* low,  emphasised */
        43581, 75693, 112965, 112965,   /* high */
        26951, 52181, 83721, 83721      /* high, emphasised */
    };
    static const int emph_mask[6] = { 0300, 0100, 0500, 0400, 0600, 0200 };
    int hue = p & 15;
    int high, emph;

    if (hue >= 14) return 0;
    if (hue == 0) high = 1;
    else if (hue == 13) high = 0;
    else high = ((hue + phase) % 12) < 6;
    emph = ((p & 0700) & emph_mask[(phase >> 1) % 6]) > 0;
    return level[high * 8 + emph * 4 + ((p >> 4) & 3)];
}

/* crt_nes.c:81-201 (NES_OPTIMIZED 1, NES_BORDER 0) */
static void
modulate_nes(const struct orc_sys *sys, struct orc_crt *v, struct orc_settings *s)
{
    static const int phasetab[3] = { 0, 4, 8 };
    const int hres = sys->hres;
    const uint16_t *data = (const uint16_t *) s->data;
    int burst[3][4], preset[3][4];
    int x, y, n, k, xo, yo, sn, cs;

    memset(preset, 0, sizeof(preset));
    if (!s->initialized) {                           /* setup_field, :81-104 */
        nes_setup_field(sys, v);
        s->initialized = 1;
    }
    for (y = 0; y < 3; y++) {                        /* :123-130 */
        int rot = (y + s->dot_crawl_offset) * 120;
        for (k = 0; k < 4; k++) {
            n = (s->hue + k * 90 + rot + 33) % 360;
            orc_sincos14(&sn, &cs, n * 8192 / 180);
            burst[y][k] = sn >> 10;
        }
    }
    xo = (sys->av_beg + s->xoffset) & ~3;            /* :132-136 */
    yo = sys->top + s->yoffset;

    if (sys->nes_border) {                           /* NES_BORDER 1, :138-160: written on every call, before the picture */
        for (n = sys->top; n <= sys->bot + 2; n++) {
            int8_t *line = v->analog + n * hres;
            int phase = phasetab[(n + s->dot_crawl_offset) % 3] + 6;
            int t;
            for (t = sys->lav_beg; t < hres; t++) {
                int p = t == sys->lav_beg ? 0xf0 : (int) s->border_color;
                int ire = sys->black_level + v->black_point;
                ire += ppu_level(p, phase + 0);
                ire += ppu_level(p, phase + 1);
                ire += ppu_level(p, phase + 2);
                ire += ppu_level(p, phase + 3);
                ire = (ire * v->white_point / 100) >> 12;
                line[t] = (int8_t) ire;
                phase += 3;
            }
        }
    }
    for (y = 0; y < sys->lines; y++) {               /* :162-194 */
        int sy = (y * s->h) / sys->lines;
        int phase;
        int8_t *line;
        if (sy >= s->h) sy = s->h;
        if (sy < 0) sy = 0;
        n = y + yo;
        line = v->analog + n * hres;
        for (k = sys->cb_beg; k < sys->cb_beg + 40; k++) {
            int cb = burst[n % 3][k % 4];
            line[k] = (int8_t) ((sys->blank_level + cb * sys->burst_level) >> 5);
            preset[n % 3][k % 4] = line[k];
        }
        sy *= s->w;
        phase = phasetab[(y + yo + s->dot_crawl_offset) % 3];
        for (x = 0; x < sys->av_len; x++) {
            int p = data[((x * s->w) / sys->av_len) + sy];
            int ire = sys->black_level + v->black_point;
            ire += ppu_level(p, phase + 0);
            ire += ppu_level(p, phase + 1);
            ire += ppu_level(p, phase + 2);
            ire += ppu_level(p, phase + 3);
            ire = (ire * v->white_point / 100) >> 12;
            v->analog[(x + xo) + (y + yo) * hres] = (int8_t) ire;
            phase += 3;
        }
    }
    for (n = 0; n < 3; n++) {                        /* :196-200 */
        for (k = 0; k < 4; k++) v->ccf[n][k] = preset[n][k] << 7;
    }
}

void
orc_modulate(const struct orc_sys *sys, struct orc_crt *v, struct orc_settings *s)
{
    if (sys->system == ORC_SYS_NES) modulate_nes(sys, v, s);
    else if (sys->system == ORC_SYS_NESRGB) modulate_nesrgb(sys, v, s);
    else modulate_rgb(sys, v, s);
}

/* ------------------------------------------------------------------------- */
/* decoder                                                                    */
/* ------------------------------------------------------------------------- */

/* affine jump of the noise LCG x -> 214019 x + 140327895 (crt_core.c:359) by k
 * steps: x_k = mul * x_0 + add (mod 2^32).  Not in the reference (which iterates);
 * the GPU path relies on it, so the oracle pins it against plain iteration. */
void
orc_lcg_jump(unsigned k, unsigned *mul, unsigned *add)
{
    unsigned am = 214019u, ac = 140327895u;  /* current power: x -> am x + ac */
    unsigned rm = 1u, rc = 0u;               /* accumulated map */
    while (k) {
        if (k & 1u) { rm = am * rm; rc = am * rc + ac; }
        ac = am * ac + ac;
        am = am * am;
        k >>= 1;
    }
    *mul = rm;
    *add = rc;
}

/* D1, crt_core.c:343-367.  Returns the new rn. */
int
orc_stage_noise(const struct orc_sys *sys, const int8_t *analog, int8_t *inp, int rn, int noise)
{
    int i, vhs_line = 0;
    const int vhs = sys->system == ORC_SYS_VHS && !sys->vhs_lcg_noise;      /* :343 (CRT_SYSTEM == NTSCVHS) && CRT_VHS_NOISE */

    if (vhs) vhs_line = ((rand() % 8) - 4) + 14;                            /* :344 */
    for (i = 0; i < sys->input_size; i++) {
        int nn = noise, s;
        if (vhs) {                                                          /* :349-357 */
            rn = rand();
            if (i > (sys->input_size - sys->hres * (16 + ((rand() % 20) - 10))) &&
                i < (sys->input_size - sys->hres * (5 + ((rand() % 8) - 4)))) {
                int sn, cs;
                orc_sincos14(&sn, &cs, ((i * vhs_line) / sys->hres) * 8192 / 180);
                nn = cs >> 8;
            }
        } else {
            rn = (int) (214019u * (unsigned) rn + 140327895u);              /* :359 */
        }
        s = analog[i] + (((((rn >> 16) & 0xff) - 0x7f) * nn) >> 8);         /* :362 */
        if (s > 127) s = 127;
        if (s < -127) s = -127;
        inp[i] = (int8_t) s;
    }
    return rn;
}

/* D2, crt_core.c:379-396 */
static void
stage_vsync(const struct orc_sys *sys, const int8_t *inp, int *vsync, int *odd_field)
{
    int i, j = 0, line = 0;

    for (i = -sys->vsync_window; i < sys->vsync_window; i++) {
        const int8_t *sig;
        int acc = 0;
        line = POSMOD(*vsync + i, sys->vres);
        sig = inp + line * sys->hres;
        for (j = 0; j < sys->hres; j++) {
            acc += sig[j];
            if (acc <= sys->vsync_thresh) goto found;
        }
    }
found:
    *vsync = line;
    *odd_field = j > sys->hres / 2;
}

struct eq_state { int lo[4], hi[4], hist[7]; };

/* eqf of a USE_CONVOLUTION build, crt_core.c:119-147: 7-deep input history, one of four symmetric kernels */
static int
eq_fir(int taps, struct eq_state *f, int s)
{
    int *h = f->hist;
    int i;

    for (i = 6; i > 0; i--) h[i] = h[i - 1];
    h[0] = s;
    switch (taps) {
    case 7:  return (s + h[6] + ((h[1] + h[5]) * 4) + ((h[2] + h[4]) * 7) + (h[3] * 8)) >> 5;   /* 1 4 7 8 7 4 1 */
    case 6:  return (s + h[5] + 3 * (h[1] + h[4]) + 4 * (h[2] + h[3])) >> 4;                    /* 1 3 4 4 3 1   */
    case 5:  return (s + h[4] + ((h[1] + h[2] + h[3]) << 1)) >> 3;                              /* 1 2 2 2 1     */
    default: return (s + h[3] + h[1] + h[2]) >> 2;                                              /* 1 1 1 1       */
    }
}

/* eqf, crt_core.c:206-233 */
static int
eq_step(const struct orc_sys *sys, int which, struct eq_state *f, int s)
{
    const int lf = sys->eq_lf[which], hf = sys->eq_hf[which];
    const int *g = sys->eq_g[which];
    int k, band0, band1, band2;

    if (sys->eq_kernel) return eq_fir(sys->eq_kernel, f, s);

    f->lo[0] += (lf * (s - f->lo[0]) + 32768) >> 16;
    f->hi[0] += (hf * (s - f->hi[0]) + 32768) >> 16;
    for (k = 1; k < 4; k++) {
        f->lo[k] += (lf * (f->lo[k - 1] - f->lo[k]) + 32768) >> 16;
        f->hi[k] += (hf * (f->hi[k - 1] - f->hi[k]) + 32768) >> 16;
    }
    band0 = (f->lo[3] * g[0]) >> 16;
    band1 = ((f->hi[3] - f->lo[3]) * g[1]) >> 16;
    band2 = ((f->hist[2] - f->hi[3]) * g[2]) >> 16;
    f->hist[2] = f->hist[1];
    f->hist[1] = f->hist[0];
    f->hist[0] = s;
    return band0 + band1 + band2;
}

/* write / read one output pixel, crt_core.c:584-656 */
static int
load_px(const uint8_t *p, int format)
{
    switch (format) {
    case 0: case 3: return p[0] << 16 | p[1] << 8 | p[2];
    case 1: case 5: return p[2] << 16 | p[1] << 8 | p[0];
    case 2:         return p[1] << 16 | p[2] << 8 | p[3];
    case 4:         return p[3] << 16 | p[2] << 8 | p[1];
    default:        return 0;
    }
}

static void
store_px(uint8_t *p, int format, int rgb)
{
    uint8_t r = (uint8_t) (rgb >> 16), g = (uint8_t) (rgb >> 8), b = (uint8_t) rgb;
    switch (format) {
    case 0: p[0] = r; p[1] = g; p[2] = b; break;
    case 3: p[0] = r; p[1] = g; p[2] = b; p[3] = 0xff; break;
    case 1: p[0] = b; p[1] = g; p[2] = r; break;
    case 5: p[0] = b; p[1] = g; p[2] = r; p[3] = 0xff; break;
    case 2: p[0] = 0xff; p[1] = r; p[2] = g; p[3] = b; break;
    case 4: p[0] = 0xff; p[1] = b; p[2] = g; p[3] = r; break;
    default: break;
    }
}

void
orc_demodulate_trace(const struct orc_sys *sys, struct orc_crt *v, int noise, struct orc_line *trace)
{
    /* crt_core.c:295-297: function-static, AV_LEN + 1 entries, zero until written.  With CRT_DO_BLOOM the filter
     * loop stops one sample early (:518, R = scanR >> 12 = AV_LEN - 1), so entry AV_LEN - 1 is never written in a
     * bloom build and stays 0 -- the array is cleared per call here to model exactly that (entries below the
     * line's first sample are stale in the reference but never read: the resampler starts at scanL) */
    static int yq[3][2048];
    const int hres = sys->hres, av_len = sys->av_len, ccs = sys->cc_samples;
    int bpp, pitch, huesn, huecs, bright, odd_field = 0, ratio, field_rows, line;
    int max_e = 0, prev_e = 0;

    bpp = orc_bpp4fmt(v->out_format);
    if (bpp == 0) return;                                                    /* :312-315 */
    pitch = v->outw * bpp;
    bright = v->brightness - (sys->black_level + v->black_point);            /* :305 */
    orc_sincos14(&huesn, &huecs, ((v->hue % 360) + 33) * 8192 / 180);        /* D0, :318-320 */
    huesn >>= 11;
    huecs >>= 11;

    /* the bytes the reference finds behind inp[] (struct members outw..pad) */
    memcpy(v->inp + sys->input_size + 0, &v->outw, 4);
    memcpy(v->inp + sys->input_size + 4, &v->outh, 4);
    memcpy(v->inp + sys->input_size + 8, &v->out_format, 4);
    memset(v->inp + sys->input_size + 12, 0, ORC_TAIL - 12);

    if (sys->no_vsync) {                                                     /* :323-341: the field parity from the clean signal */
        stage_vsync(sys, v->analog, &v->vsync, &odd_field);
        v->vsync = -3;
    }
    v->rn = orc_stage_noise(sys, v->analog, v->inp, v->rn, noise);           /* D1 */
    if (!sys->no_vsync) stage_vsync(sys, v->inp, &v->vsync, &odd_field);     /* D2 */

    if (sys->do_bloom) {                                                     /* :399-402 */
        max_e = (128 + (noise / 2)) * av_len;
        prev_e = 16384 / 8;
        memset(yq, 0, sizeof(yq));
    }
    ratio = (v->outh << 16) / sys->lines;                                    /* D3, :403-407 */
    ratio = (ratio + 32768) >> 16;
    field_rows = odd_field * (ratio / 2);

    for (line = sys->top; line < sys->bot; line++) {
        struct orc_line *tr = trace ? &trace[line - sys->top] : 0;
        struct eq_state ey, ei, eq;
        const int8_t *sig;
        int beg, end, ln, acc, i, xpos, ypos, pos, *ccr, align, dci, dcq;
        int wave_i[ORC_MAX_CCS], wave_q[ORC_MAX_CCS];
        int dx, scanl, first, last, row;
        unsigned upos, scanr;
        uint8_t *dst, *dst_end;

        if (tr) memset(tr, 0, sizeof(*tr));
        /* D4, :428-432.  v_fac is `unsigned`, so the reference evaluates these in
         * unsigned arithmetic and converts back on assignment */
        beg = (int) ((unsigned) (line - sys->top + 0) * ((unsigned) v->outh + v->v_fac)
                     / (unsigned) sys->lines + (unsigned) field_rows);
        end = (int) ((unsigned) (line - sys->top + 1) * ((unsigned) v->outh + v->v_fac)
                     / (unsigned) sys->lines + (unsigned) field_rows);
        if (beg >= v->outh) continue;
        if (end > v->outh) end = v->outh;

        ln = POSMOD(line + v->vsync, sys->vres) * hres;                      /* D5, :437-450 */
        sig = v->inp + ln + v->hsync;
        acc = 0;
        for (i = -sys->hsync_window; i < sys->hsync_window; i++) {
            acc += sig[sys->sync_beg + i];
            if (acc <= sys->hsync_thresh) break;
        }
        v->hsync = sys->no_hsync ? 0 : POSMOD(i + v->hsync, hres);           /* :446-450 */

        xpos = POSMOD(sys->av_beg + v->hsync - 3, hres);                     /* D6, :452-467 */
        ypos = POSMOD(line + v->vsync + 3, sys->vres);
        pos = xpos + ypos * hres;
        ccr = v->ccf[ypos % sys->cc_vper];
        sig = v->inp + ln + (ccs == 4 ? (v->hsync & ~3) : v->hsync - (v->hsync % ccs));
        for (i = sys->cb_beg; i < sys->cb_beg + sys->cb_len; i++) {
            ccr[i % ccs] = ccr[i % ccs] * 127 / 128 + sig[i];
        }

        align = POSMOD(v->hsync, ccs);                                       /* D7 */
        if (ccs == 4) {                                                      /* :471-479 */
            dci = ccr[(align + 1) & 3] - ccr[(align + 3) & 3];
            dcq = ccr[(align + 2) & 3] - ccr[(align + 0) & 3];
            wave_i[0] = ((dci * huecs - dcq * huesn) >> 4) * v->saturation;
            wave_i[1] = ((dcq * huecs + dci * huesn) >> 4) * v->saturation;
            wave_i[2] = -wave_i[0];
            wave_i[3] = -wave_i[1];
            for (i = 0; i < 4; i++) wave_q[i] = wave_i[(i + 3) & 3];         /* :541-542: Q reads wave[(i + 3) & 3] */
            if (tr) { tr->wave0 = wave_i[0]; tr->wave1 = wave_i[1]; }
        } else {                                                             /* :480-510, 5 samples per cycle */
            int ang = v->hue % 360;
            int peak_a = align + ccs / 4, peak_b = align, sn, cs;
            int dci_a = ccr[peak_a % ccs];
            int dci_b = (ccr[(peak_a + ccs / 2) % ccs] + ccr[(peak_a + ccs / 2 + 1) % ccs]) / 2;
            int dcq_a = ccr[(peak_b + ccs / 2) % ccs];
            int dcq_b = ccr[peak_b % ccs];
            dci = dci_a - dci_b;
            dcq = dcq_a - dcq_b;
            for (i = 0; i < ccs; i++) {
                orc_sincos14(&sn, &cs, ang * 8192 / 180);
                wave_i[i] = ((dci * cs + dcq * sn) >> 15) * v->saturation;
                orc_sincos14(&sn, &cs, (ang + 90) * 8192 / 180);
                wave_q[i] = ((dci * cs + dcq * sn) >> 15) * v->saturation;
                ang += 360 / ccs;
            }
            if (tr) { tr->wave0 = dci; tr->wave1 = dcq; }
        }

        sig = v->inp + pos;
        if (sys->do_bloom) {                                                 /* :512-526 */
            int sum = 0, line_w;
            for (i = 0; i < av_len; i++) sum += sig[i];
            prev_e = (prev_e * 123 / 128) + ((((max_e >> 1) - sum) << 10) / max_e);
            line_w = (av_len * 112 / 128) + (prev_e >> 9);
            dx = (line_w << 12) / v->outw;
            scanl = ((av_len / 2) - (line_w >> 1) + 8) << 12;
            scanr = (unsigned) (av_len - 1) << 12;
            first = scanl >> 12;
            last = (int) (scanr >> 12);
        } else {                                                             /* :528-532 */
            dx = ((av_len - 1) << 12) / v->outw;
            scanl = 0;
            scanr = (unsigned) (av_len - 1) << 12;
            first = 0;
            last = av_len;
        }
        if (tr) {
            tr->valid = 1; tr->pos = pos; tr->beg = beg; tr->end = end; tr->hsync = v->hsync;
            tr->dx = dx; tr->scanl = scanl;
        }

        memset(&ey, 0, sizeof(ey));                                          /* D8, :534-549 */
        memset(&ei, 0, sizeof(ei));
        memset(&eq, 0, sizeof(eq));
        for (i = first; i < last; i++) {
            yq[0][i] = eq_step(sys, 0, &ey, sig[i] + bright) << 4;
            yq[1][i] = eq_step(sys, 1, &ei, sig[i] * wave_i[i % ccs] >> 9) >> 3;
            yq[2][i] = eq_step(sys, 2, &eq, sig[i] * wave_q[i % ccs] >> 9) >> 3;
        }

        dst = v->out + beg * pitch;                                          /* D9, :552-659 */
        dst_end = dst + pitch;
        for (upos = (unsigned) scanl; upos < scanr && dst < dst_end; upos += (unsigned) dx, dst += bpp) {
            int R, L, sidx, y, ci, cq, r, g, b, rgb;
            R = (int) (upos & 0xfff);
            L = 0xfff - R;
            sidx = (int) (upos >> 12);
            y = ((yq[0][sidx] * L) >> 2) + ((yq[0][sidx + 1] * R) >> 2);
            ci = ((yq[1][sidx] * L) >> 14) + ((yq[1][sidx + 1] * R) >> 14);
            cq = ((yq[2][sidx] * L) >> 14) + ((yq[2][sidx + 1] * R) >> 14);
            r = (((y + 3879 * ci + 2556 * cq) >> 12) * v->contrast) >> 8;
            g = (((y - 1126 * ci - 2605 * cq) >> 12) * v->contrast) >> 8;
            b = (((y - 4530 * ci + 7021 * cq) >> 12) * v->contrast) >> 8;
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            g = g < 0 ? 0 : (g > 255 ? 255 : g);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            rgb = r << 16 | g << 8 | b;
            if (v->blend) {                                                  /* :584-609 */
                int old = load_px(dst, v->out_format);
                rgb = ((rgb & 0xfefeff) >> 1) + ((old & 0xfefeff) >> 1);
            }
            store_px(dst, v->out_format, rgb);
        }
        for (row = beg + 1; row < end - v->scanlines; row++) {               /* D10, :661-664 */
            memcpy(v->out + row * pitch, v->out + (row - 1) * pitch, (size_t) pitch);
        }
    }
}

void
orc_demodulate(const struct orc_sys *sys, struct orc_crt *v, int noise)
{
    orc_demodulate_trace(sys, v, noise, 0);
}

double
orc_time_fieldpasses(const struct orc_sys *sys, struct orc_crt *v, struct orc_settings *s,
                     int noise, int reps, int interlaced)
{
    struct timespec a, b;
    int k;

    clock_gettime(CLOCK_MONOTONIC, &a);
    for (k = 0; k < reps; k++) {
        orc_modulate(sys, v, s);
        orc_demodulate(sys, v, noise);
        if (interlaced && sys->system != ORC_SYS_NES && sys->system != ORC_SYS_NESRGB) {
            s->field ^= 1;
            if ((k & 1) == 0) s->frame ^= 1;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
}
