/*
 * abi_probe.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny shim over "crt_core.h" -- WHICHEVER crt_core.h the include path provides:
 *   (a) compiled together with the untouched reference sources (found through -I$(REF), never
 *       copied) into oracle/_ref/libref_<sys>.so, and
 *   (b) compiled against this repo's own include/crt_core.h and linked to the drop-in
 *       libntsccrt_hip_<sys>.so (tests/crtref.py: build_dropin_probe), so both can be driven
 *       and compared through the same Python wrapper.
 * It adds a few helper entry points next to the crt_* symbols so the tests / bench can
 *   - learn sizeof/offsetof of the reference's struct CRT / NTSC_SETTINGS
 *     (they depend on CRT_SYSTEM, crt_core.h:43-59),
 *   - time the reference's crt_modulate + crt_demodulate on the host cores
 *     (bench.py "cpu_baseline", kind = "reference").
 * Nothing here is part of the shipped product.
 */
#include "crt_core.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int refp_system(void)            { return CRT_SYSTEM; }
int refp_hres(void)              { return CRT_HRES; }
int refp_vres(void)              { return CRT_VRES; }
int refp_input_size(void)        { return CRT_INPUT_SIZE; }
int refp_top(void)               { return CRT_TOP; }
int refp_bot(void)               { return CRT_BOT; }
int refp_cc_vper(void)           { return CRT_CC_VPER; }
int refp_av_beg(void)            { return AV_BEG; }
int refp_av_len(void)            { return AV_LEN; }
int refp_sync_beg(void)          { return SYNC_BEG; }
int refp_bw_beg(void)            { return BW_BEG; }
int refp_cb_beg(void)            { return CB_BEG; }

long refp_sizeof_crt(void)       { return (long) sizeof(struct CRT); }
long refp_sizeof_settings(void)  { return (long) sizeof(struct NTSC_SETTINGS); }

#define OFF(name, member) \
    long refp_off_##name(void) { return (long) offsetof(struct CRT, member); }
OFF(analog, analog)
OFF(inp, inp)
OFF(outw, outw)
OFF(outh, outh)
OFF(out_format, out_format)
OFF(out, out)
OFF(hue, hue)
OFF(brightness, brightness)
OFF(contrast, contrast)
OFF(saturation, saturation)
OFF(black_point, black_point)
OFF(white_point, white_point)
OFF(scanlines, scanlines)
OFF(blend, blend)
OFF(v_fac, v_fac)
OFF(ccf, ccf)
OFF(hsync, hsync)
OFF(vsync, vsync)
OFF(rn, rn)

#define SOFF(name, member) \
    long refp_soff_##name(void) { return (long) offsetof(struct NTSC_SETTINGS, member); }
SOFF(data, data)
SOFF(w, w)
SOFF(h, h)
SOFF(hue, hue)
SOFF(xoffset, xoffset)
SOFF(yoffset, yoffset)
#if (CRT_SYSTEM == CRT_SYSTEM_NES)
SOFF(border_color, border_color)
SOFF(dot_crawl_offset, dot_crawl_offset)
SOFF(field_initialized, field_initialized)
#elif (CRT_SYSTEM == CRT_SYSTEM_NESRGB)
SOFF(format, format)
SOFF(dot_crawl_offset, dot_crawl_offset)
SOFF(field_initialized, field_initialized)
#else
SOFF(format, format)
SOFF(raw, raw)
SOFF(as_color, as_color)
SOFF(field, field)
SOFF(frame, frame)
SOFF(iirs_initialized, iirs_initialized)
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS)
SOFF(do_aberration, do_aberration)
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_SNES) || (CRT_SYSTEM == CRT_SYSTEM_PV1K) || (CRT_SYSTEM == CRT_SYSTEM_TEMP)
SOFF(dot_crawl_offset, dot_crawl_offset)
#endif
int refp_cc_samples(void)        { return CRT_CC_SAMPLES; }
int refp_do_bloom(void)          { return CRT_DO_BLOOM; }

void refp_srand(unsigned seed) { srand(seed); }

/* Time `reps` field-passes (modulate + demodulate) of the reference on the
 * calling thread.  Field parity alternates per pass like video_convert.c
 * (:261-267) when `interlaced` is set.  Returns seconds (CLOCK_MONOTONIC);
 * *mod_s / *dem_s get the split. */
double
refp_time_fieldpasses(struct CRT *v, struct NTSC_SETTINGS *s, int noise,
                      int reps, int interlaced, double *mod_s, double *dem_s)
{
    struct timespec a, b, c;
    double tm = 0.0, td = 0.0;
    int k;

    for (k = 0; k < reps; k++) {
        clock_gettime(CLOCK_MONOTONIC, &a);
        crt_modulate(v, s);
        clock_gettime(CLOCK_MONOTONIC, &b);
        crt_demodulate(v, noise);
        clock_gettime(CLOCK_MONOTONIC, &c);
        tm += (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
        td += (double) (c.tv_sec - b.tv_sec) + 1e-9 * (double) (c.tv_nsec - b.tv_nsec);
#if (CRT_SYSTEM != CRT_SYSTEM_NES) && (CRT_SYSTEM != CRT_SYSTEM_NESRGB)
        if (interlaced) {
            s->field ^= 1;
            if ((k & 1) == 0) {
                s->frame ^= 1;
            }
        }
#else
        (void) interlaced;
#endif
    }
    if (mod_s) { *mod_s = tm; }
    if (dem_s) { *dem_s = td; }
    return tm + td;
}
