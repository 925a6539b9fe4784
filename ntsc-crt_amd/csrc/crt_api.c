/*
 * crt_api.c -- the reference's public API (crt_core.h:100-139 of LMP88959/NTSC-CRT) as a thin
 * C89 client of the crthip_* C ABI.  One library per CRT_SYSTEM (-DCRT_SYSTEM=0|1|5), exactly
 * like the reference is one build per CRT_SYSTEM.
 *
 * The caller owns `struct CRT`, the input image and the output image and may read or write any
 * member between calls (crt_main.c:235-236, :262-264, :317-391, :430).  This layer therefore
 * keeps NO authoritative state on the device: every call uploads what the kernels read
 * (analog[], the output image, hsync/vsync/rn/ccf, the knobs) and downloads what the reference
 * would have written (analog[] and ccf after crt_modulate; inp[], the output image, ccf, hsync,
 * vsync, rn after crt_demodulate).  Device buffers are cached per `struct CRT *`.
 *
 * CRTHIP_LAZY_MIRROR=1|2 (environment, opt-in -- SURVEY.md 8b "lazy mirror"): the device copies of analog[] and of the
 * output image become authoritative between calls.  What the caller does NOT get any more: crt.analog after
 * crt_modulate and crt.inp after crt_demodulate are not refreshed on the host (crt_main.c's -a dump needs the strict
 * mode).  What stays safe: analog[] and the output image are re-uploaded whenever the host copy differs from what this
 * layer last saw there.  analog[] is compared byte for byte with a shadow copy of what this layer last saw (all of it,
 * in both modes); because a lazy crt_modulate leaves the host copy stale, it also stamps a 16-byte marker into
 * crt.analog[0..15], so that a caller's memset / repaint is seen even when it restores the very bytes the stale copy
 * held (crt_main.c:430 after crt_init).  Marker gone = the caller replaced the field: upload it all; marker intact but
 * other bytes changed = the caller edited samples of a field whose current content lives on the device: exactly the
 * changed bytes are patched into the device copy.  The output image and the input image are judged by a hash, sampled
 * (1: every 16th 64-byte block + both ends) or complete (2); the input image is re-uploaded unless pointer, geometry
 * and hash are unchanged (crt_main.c modulates the same picture 8 times).  The caller's image and output buffers are
 * page-locked on first sight (hipHostRegister) so that the remaining copies run as direct DMA; the registrations are
 * dropped again on crt_init / crt_resize of that set and when its slot is recycled.
 *
 * Errors: the reference API is all-void and has no error channel.  A HIP failure or a missing
 * gfx950 device is reported on stderr and the process aborts -- there is no CPU fallback.
 */
#include "crt_core.h"
#include "crt_hip.h"
#include "crt_setup.h"

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- layout pins (C89 static assertions) ----------------------------------------------- */
#define PIN(name, cond) typedef char crt_pin_##name[(cond) ? 1 : -1]
PIN(analog_first, offsetof(struct CRT, analog) == 0);
PIN(inp_follows_analog, offsetof(struct CRT, inp) == CRT_INPUT_SIZE);
PIN(outw_follows_inp, offsetof(struct CRT, outw) == ((2 * CRT_INPUT_SIZE + 3) & ~3));
PIN(out_is_pointer_aligned, offsetof(struct CRT, out) % sizeof(void *) == 0);
PIN(ccf_shape, sizeof(((struct CRT *) 0)->ccf) == CRT_CC_VPER * CRT_CC_SAMPLES * sizeof(int));
PIN(rn_last, offsetof(struct CRT, rn) + sizeof(int) <= sizeof(struct CRT));
#if (CRT_SYSTEM == CRT_SYSTEM_NTSC) && (CRT_CHROMA_PATTERN == 1)
PIN(ntsc_sizeof_crt, sizeof(void *) != 8 || sizeof(struct CRT) == 476928);    /* SURVEY.md section 8b */
PIN(ntsc_sizeof_settings, sizeof(void *) != 8 || sizeof(struct NTSC_SETTINGS) == 56);
PIN(ntsc_hres, CRT_HRES == 910 && AV_LEN == 753 && AV_BEG == 156);
#endif

/* CRT_SYSTEM_* and CRTHIP_SYSTEM_* share their values (crt_core.h:30-36) */
#define SYS_ID CRT_SYSTEM
PIN(system_ids, CRT_SYSTEM_NTSC == CRTHIP_SYSTEM_NTSC && CRT_SYSTEM_NES == CRTHIP_SYSTEM_NES &&
                CRT_SYSTEM_PV1K == CRTHIP_SYSTEM_PV1K && CRT_SYSTEM_SNES == CRTHIP_SYSTEM_SNES &&
                CRT_SYSTEM_TEMP == CRTHIP_SYSTEM_TEMP && CRT_SYSTEM_NTSCVHS == CRTHIP_SYSTEM_NTSCVHS &&
                CRT_SYSTEM_NESRGB == CRTHIP_SYSTEM_NESRGB);
/* the headers of SNES / PV-1000 / template have no CRT_CHROMA_PATTERN (crt_snes.h:24): one raster each */
#ifndef CRT_CHROMA_PATTERN
#define CRT_CHROMA_PATTERN 1
#endif
#define HAS_DOT_CRAWL ((CRT_SYSTEM == CRT_SYSTEM_NES) || (CRT_SYSTEM == CRT_SYSTEM_NESRGB) || (CRT_SYSTEM == CRT_SYSTEM_SNES) || \
                       (CRT_SYSTEM == CRT_SYSTEM_PV1K) || (CRT_SYSTEM == CRT_SYSTEM_TEMP))

/* ---- device-side cache, one slot per struct CRT ------------------------------------------- */
#define MAX_SLOTS 32

struct slot {
    struct CRT *host;
    signed char *d_analog, *d_inp;
    crthip_state *d_state;
    crthip_line *d_lines;
    unsigned char *d_img, *d_out;
    size_t img_cap, out_cap;
    /* lazy mirror: what the host copies looked like when this layer last read or wrote them */
    int analog_valid, out_valid, img_valid;
    unsigned long last_use;               /* g_tick of the slot's last call: the least recently used slot is the one recycled */
    int analog_dev_newer;                 /* lazy: the device copy holds a crt_modulate result the host copy does not */
    signed char *analog_shadow;           /* lazy: crt.analog[] as this layer last saw / left it on the host */
    unsigned long out_hash, img_hash;
    const void *img_ptr, *out_ptr, *pinned_img, *pinned_out;
    size_t img_bytes, out_bytes, pinned_img_bytes, pinned_out_bytes;
};

static crthip_ctx *g_ctx;
static unsigned long g_tick;
static int g_lazy = -1;             /* CRTHIP_LAZY_MIRROR: 0 strict (default), 1 sampled hashes, 2 full hashes */

static int
lazy_mode(void)
{
    if (g_lazy < 0) {
        const char *e = getenv("CRTHIP_LAZY_MIRROR");
        g_lazy = e ? atoi(e) : 0;
        if (g_lazy < 0 || g_lazy > 2) {
            g_lazy = 0;
        }
    }
    return g_lazy;
}

/* 64-bit multiplicative hash of a buffer: all of it (mode 2) or every 16th 64-byte block plus both ends (mode 1) */
static unsigned long
buf_hash(const void *buf, size_t bytes, int mode)
{
    const unsigned long *w = (const unsigned long *) buf;
    const unsigned char *b = (const unsigned char *) buf;
    size_t nw = bytes / sizeof(unsigned long), i, k;
    unsigned long h = 0x9e3779b97f4a7c15ul ^ (unsigned long) bytes;

    if (((size_t) buf & (sizeof(unsigned long) - 1)) != 0) {
        for (i = 0; i < bytes; i += (mode == 2 ? 1 : 61)) {       /* unaligned buffer: bytewise */
            h = (h ^ b[i]) * 0x100000001b3ul;
        }
        return h;
    }
    if (mode == 2 || nw < 1024) {
        for (i = 0; i < nw; i++) {
            h = (h ^ w[i]) * 0x100000001b3ul;
        }
    } else {
        for (i = 0; i + 8 <= nw; i += 128) {                     /* 8 words out of every 128 */
            for (k = 0; k < 8; k++) {
                h = (h ^ w[i + k]) * 0x100000001b3ul;
            }
        }
        for (i = nw - 32; i < nw; i++) {
            h = (h ^ w[i]) * 0x100000001b3ul;
        }
    }
    for (i = nw * sizeof(unsigned long); i < bytes; i++) {
        h = (h ^ b[i]) * 0x100000001b3ul;
    }
    return h;
}

/* page-lock a caller buffer once (best effort: a failure only means staged copies) */
static void
pin_buffer(const void **pinned, size_t *pinned_bytes, const void *ptr, size_t bytes)
{
    if (*pinned == ptr && *pinned_bytes == bytes) {
        return;
    }
    if (*pinned) {
        crthip_host_unregister(g_ctx, (void *) *pinned);
    }
    *pinned = 0;
    *pinned_bytes = 0;
    if (crthip_host_register(g_ctx, (void *) ptr, bytes) == CRTHIP_OK) {
        *pinned = ptr;
        *pinned_bytes = bytes;
    }
}
static struct slot g_slots[MAX_SLOTS];
static size_t g_fstride;

/* drop the page-lock registrations of a slot (the caller may free or reallocate those buffers: crt_init, crt_resize, a
 * recycled slot); failures are ignored -- an address that is no longer mapped cannot be unregistered */
static void
unpin_slot(struct slot *sl, int img_too)
{
    if (g_ctx == 0) {
        return;
    }
    if (sl->pinned_out) {
        (void) crthip_host_unregister(g_ctx, (void *) sl->pinned_out);
        sl->pinned_out = 0;
        sl->pinned_out_bytes = 0;
    }
    if (img_too && sl->pinned_img) {
        (void) crthip_host_unregister(g_ctx, (void *) sl->pinned_img);
        sl->pinned_img = 0;
        sl->pinned_img_bytes = 0;
    }
}

static void
fatal(const char *what, int rc)
{
    fprintf(stderr, "ntsc-crt (HIP/gfx950): %s failed (%d): %s\n", what, rc,
            g_ctx ? crthip_error_string(g_ctx) : "no context");
    fprintf(stderr, "ntsc-crt (HIP/gfx950): this library has no CPU fallback -- aborting\n");
    abort();
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != CRTHIP_OK) fatal(#call, rc_); } while (0)

static void *
dev_alloc(size_t bytes)
{
    void *p = crthip_malloc(g_ctx, bytes);
    if (p == 0) {
        fatal("crthip_malloc", CRTHIP_E_NOMEM);
    }
    CHECK(crthip_memset(g_ctx, p, 0, bytes));
    return p;
}

static struct slot *
get_slot(struct CRT *v)
{
    int i;
    struct slot *free_slot = 0;

    if (g_ctx == 0) {
        const char *dev = getenv("CRTHIP_DEVICE");
        int rc = crthip_create(&g_ctx, dev ? atoi(dev) : 0, SYS_ID, CRT_CHROMA_PATTERN);
        if (rc != CRTHIP_OK) {
            g_ctx = 0;
            fatal("crthip_create (is an MI355X / gfx950 visible?)", rc);
        }
        g_fstride = crthip_field_stride(SYS_ID, CRT_CHROMA_PATTERN);
    }
    for (i = 0; i < MAX_SLOTS; i++) {
        if (g_slots[i].host == v) {
            g_slots[i].last_use = ++g_tick;
            return &g_slots[i];
        }
        if (g_slots[i].host == 0 && free_slot == 0) {
            free_slot = &g_slots[i];
        }
    }
    if (free_slot == 0) {
        /* recycle slot 0: its buffers stay allocated and are simply re-filled on the next call */
        /* recycle the least recently used slot.  Lazy mirror: if the evicted set's last crt_modulate result exists on the
         * device only, it is lost here -- the set's struct may be long gone (a stack object), so nothing is written back
         * to it; should the set call again, the intact marker in its analog[] without a slot is caught in
         * sync_analog_to_device */
        free_slot = &g_slots[0];
        for (i = 1; i < MAX_SLOTS; i++) {
            if (g_slots[i].last_use < free_slot->last_use) {
                free_slot = &g_slots[i];
            }
        }
        unpin_slot(free_slot, 1);
    }
    free_slot->host = v;
    free_slot->last_use = ++g_tick;
    free_slot->analog_valid = free_slot->out_valid = free_slot->img_valid = 0;
    free_slot->analog_dev_newer = 0;
    if (free_slot->d_analog == 0) {
        free_slot->d_analog = (signed char *) dev_alloc(g_fstride + 4096);
        free_slot->d_inp = (signed char *) dev_alloc(g_fstride + 4096);
        free_slot->d_state = (crthip_state *) dev_alloc(sizeof(crthip_state));
        free_slot->d_lines = (crthip_line *) dev_alloc(sizeof(crthip_line) * CRT_LINES);
    }
    return free_slot;
}

static void
ensure(unsigned char **buf, size_t *cap, size_t need)
{
    if (*cap < need) {
        if (*buf) {
            crthip_free(g_ctx, *buf);
        }
        *buf = (unsigned char *) dev_alloc(need);
        *cap = need;
    }
}

static void
state_to_device(struct slot *sl, const struct CRT *v, int field, int frame, int aux)
{
    crthip_state st;
    int r, k;

    memset(&st, 0, sizeof(st));
    st.field = field;
    st.frame = frame;
    st.aux = aux;
    st.hsync = v->hsync;
    st.vsync = v->vsync;
    st.rn = v->rn;
    for (r = 0; r < CRT_CC_VPER; r++) {
        for (k = 0; k < CRT_CC_SAMPLES; k++) {
            st.ccf[r][k] = v->ccf[r][k];
        }
    }
    CHECK(crthip_upload(g_ctx, sl->d_state, &st, sizeof(st)));
}

static void
state_from_device(struct slot *sl, struct CRT *v, int sync_too)
{
    crthip_state st;
    int r, k;

    CHECK(crthip_download(g_ctx, &st, sl->d_state, sizeof(st)));
    for (r = 0; r < CRT_CC_VPER; r++) {
        for (k = 0; k < CRT_CC_SAMPLES; k++) {
            v->ccf[r][k] = st.ccf[r][k];
        }
    }
    v->hsync = st.hsync;
    if (sync_too) {
        v->vsync = st.vsync;
        v->rn = st.rn;
    }
}

/* the knobs the kernels need, from the caller's struct */
static void
monitor_params(crthip_params *p, const struct CRT *v)
{
    CHECK(crthip_params_default(p, SYS_ID, CRT_CHROMA_PATTERN));
    p->outw = v->outw > 0 ? v->outw : 1;
    p->outh = v->outh > 0 ? v->outh : 1;
    p->out_format = v->out_format;
    p->mon_hue = v->hue;
    p->brightness = v->brightness;
    p->contrast = v->contrast;
    p->saturation = v->saturation;
    p->black_point = v->black_point;
    p->white_point = v->white_point;
    p->scanlines = v->scanlines;
    p->blend = v->blend;
    p->v_fac = v->v_fac;
    p->w = 1;
    p->h = 1;
#ifdef CRT_EQ_FIR_TAPS
    /* stand-in for a USE_CONVOLUTION build of the reference (crt_core.c:85-88): 7, 6, 5 or 4 */
    p->flags |= CRTHIP_F_EQ_FIR(CRT_EQ_FIR_TAPS);
#endif
#if CRT_DO_BLOOM
    p->flags |= CRTHIP_F_BLOOM;                     /* crt_core.h:70 */
#endif
    /* the reference's other build-time switches, selected like there: by the macros this file is compiled with */
#if !CRT_DO_VSYNC
    p->flags |= CRTHIP_F_NO_VSYNC;                  /* crt_core.h:71 */
#endif
#if !CRT_DO_HSYNC
    p->flags |= CRTHIP_F_NO_HSYNC;                  /* crt_core.h:72 */
#endif
#ifdef CRT_HIPASS
    p->flags |= CRTHIP_F_HIPASS;                    /* HIPASS 1, crt_ntsc.c:115 */
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS)
#if !CRT_VHS_NOISE
    p->flags |= CRTHIP_F_VHS_LCG_NOISE;             /* crt_ntscvhs.h:29 */
#endif
#if (VHS_MODE == VHS_LP)
    p->flags |= CRTHIP_F_VHS_LP;
#elif (VHS_MODE == VHS_EP)
    p->flags |= CRTHIP_F_VHS_EP;
#endif
#endif
}

/*
 * The C library's rand() stream belongs to the host program (video_convert.c seeds it; the VHS build of the
 * reference draws from it in crt_modulate and crt_demodulate, crt_core.c:344-351).  The ROCm runtime also draws
 * from it while it initialises (seen on the first HIP call of a process: the stream came back shifted), so the
 * program's generator is PARKED for the duration of every call into HIP:
 *   setstate(scratch) makes libc use a scratch state and returns the program's own state array; word 0 of that
 *   array is the info word (5 * rear_index + type, refreshed by that very setstate call), words 1..31 the ring.
 * The VHS build reads the 31-value history out of the parked array, advances it on the GPU and writes it back
 * before the array is re-installed with setstate(), which re-reads the info word.  glibc's default TYPE_3
 * generator only (the reference's platform); anything else aborts.
 */
extern char *setstate(char *state);        /* POSIX (stdlib.h hides it under -std=c89) */

static int *
park_libc_rand(void)
{
    static int scratch[34];
    int *lib;

    scratch[0] = 3;                                 /* TYPE_3, rear index 0 */
    lib = (int *) setstate((char *) scratch);
    if (lib == 0) {
        fatal("setstate", CRTHIP_E_ARG);
    }
    return lib;
}

static void
unpark_libc_rand(int *lib)
{
    setstate((char *) lib);
}

#if (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS) && CRT_VHS_NOISE
static unsigned *d_hist_buf;

/* the generator's history y[n-31 .. n-1] in logical order, out of the parked state array */
static void
read_libc_rand(int *lib, unsigned hist[31])
{
    int info, rear, j;

    info = lib[0];
    if (info % 5 != 3) {
        unpark_libc_rand(lib);
        fatal("rand() is not glibc's TYPE_3 generator", CRTHIP_E_ARG);
    }
    rear = info / 5;
    for (j = 0; j < 31; j++) {
        hist[j] = (unsigned) lib[1 + (rear + 3 + j) % 31];
    }
}

static void
write_libc_rand(int *lib, const unsigned hist[31])
{
    int j;

    for (j = 0; j < 31; j++) {
        lib[1 + (3 + j) % 31] = (int) hist[j];
    }
    lib[0] = 3;                                     /* rear index 0, TYPE_3 */
}
#endif

/* analog[] before a kernel reads it: strict mode uploads the caller's copy every time; lazy mode only what the caller
 * changed since this layer last looked (see the header comment) */
static const unsigned char LAZY_MARK[16] = { 'c', 'r', 't', 'h', 'i', 'p', 0x7f, 0x80, 'l', 'a', 'z', 'y', 0x81, 0x7e, 0x5a, 0xa5 };

static void
sync_analog_to_device(struct slot *sl, const struct CRT *v)
{
    if (!lazy_mode()) {
        CHECK(crthip_upload(g_ctx, sl->d_analog, v->analog, CRT_INPUT_SIZE));
        return;
    }
    if (sl->analog_shadow == 0) {
        sl->analog_shadow = (signed char *) malloc(CRT_INPUT_SIZE);
        if (sl->analog_shadow == 0) {
            fatal("malloc (analog shadow)", CRTHIP_E_NOMEM);
        }
        sl->analog_valid = 0;
    }
    if (sl->analog_valid && memcmp(v->analog, sl->analog_shadow, CRT_INPUT_SIZE) == 0) {
        return;                                     /* untouched since this layer last saw it */
    }
    if (!sl->analog_valid && memcmp(v->analog, LAZY_MARK, sizeof(LAZY_MARK)) == 0) {
        /* the marker of a lazy crt_modulate, but no device copy behind it: this set's slot was recycled (more than
         * MAX_SLOTS sets alive in lazy mode) and its last field went with it.  Uploading the stale host copy, marker
         * bytes included, would silently decode the wrong signal: refuse loudly. */
        fatal("lazy mirror: this struct CRT lost its device copy (more than 32 sets in use); run with CRTHIP_LAZY_MIRROR=0", CRTHIP_E_ARG);
    }
    if (sl->analog_valid && sl->analog_dev_newer && memcmp(v->analog, LAZY_MARK, sizeof(LAZY_MARK)) == 0) {
        /* marker intact: sparse edits on top of a stale host copy -> patch exactly the changed bytes into the device copy */
        static signed char *tmp;
        long i;
        if (tmp == 0) {
            tmp = (signed char *) malloc(CRT_INPUT_SIZE);
            if (tmp == 0) {
                fatal("malloc (analog patch buffer)", CRTHIP_E_NOMEM);
            }
        }
        CHECK(crthip_download(g_ctx, tmp, sl->d_analog, CRT_INPUT_SIZE));
        for (i = 0; i < CRT_INPUT_SIZE; i++) {
            if (v->analog[i] != sl->analog_shadow[i]) {
                tmp[i] = v->analog[i];
            }
        }
        CHECK(crthip_upload(g_ctx, sl->d_analog, tmp, CRT_INPUT_SIZE));
        memcpy(sl->analog_shadow, v->analog, CRT_INPUT_SIZE);
        return;                                     /* the device copy stays newer than the host's */
    }
    CHECK(crthip_upload(g_ctx, sl->d_analog, v->analog, CRT_INPUT_SIZE));
    memcpy(sl->analog_shadow, v->analog, CRT_INPUT_SIZE);
    sl->analog_valid = 1;
    sl->analog_dev_newer = 0;
}

/* after a lazy crt_modulate: crt.analog on the host is stale from here on.  The marker makes that visible to this layer
 * (and to anyone who dumps the array); the reference would have written the new field there */
static void
mark_analog_stale(struct slot *sl, struct CRT *v)
{
    memcpy(v->analog, LAZY_MARK, sizeof(LAZY_MARK));
    memcpy(sl->analog_shadow, LAZY_MARK, sizeof(LAZY_MARK));
    sl->analog_dev_newer = 1;
}

/* ---- public API ---------------------------------------------------------------------------- */

extern int
crt_bpp4fmt(int format)
{
    return crt_setup_bpp4fmt(format);
}

extern void
crt_sincos14(int *s, int *c, int n)
{
    crt_setup_sincos14(s, c, n);
}

extern void
crt_resize(struct CRT *v, int w, int h, int f, unsigned char *out)
{
    int i;

    for (i = 0; i < MAX_SLOTS; i++) {           /* a new output buffer: the old one may be freed by the caller */
        if (g_slots[i].host == v && (const void *) out != g_slots[i].pinned_out) {
            unpin_slot(&g_slots[i], 0);
        }
    }
    v->outw = w;
    v->outh = h;
    v->out_format = f;
    v->out = out;
}

extern void
crt_reset(struct CRT *v)
{
    v->hue = 0;
    v->saturation = 10;
    v->brightness = 0;
    v->contrast = 180;
    v->black_point = 0;
    v->white_point = 100;
    v->hsync = 0;
    v->vsync = 0;
}

extern void
crt_init(struct CRT *v, int w, int h, int f, unsigned char *out)
{
    int i;

    memset(v, 0, sizeof(struct CRT));
    crt_resize(v, w, h, f, out);
    crt_reset(v);
    v->rn = 194;
    for (i = 0; i < MAX_SLOTS; i++) {           /* lazy mirror: a re-initialised set starts from its host copies again */
        if (g_slots[i].host == v) {
            g_slots[i].analog_valid = g_slots[i].out_valid = g_slots[i].img_valid = 0;
            g_slots[i].analog_dev_newer = 0;
            unpin_slot(&g_slots[i], 1);
        }
    }
    /* the equaliser coefficients the reference sets up here are derived per call on the host
     * (crthip_params_finalize); nothing else to do until the first modulate / demodulate */
}

extern void
crt_modulate(struct CRT *v, struct NTSC_SETTINGS *s)
{
    struct slot *sl;
    crthip_params p;
    struct crt_sysdef sd;
    size_t img_bytes;
    long end;
    int field = 0, frame = 0, aux = 0;
    int *lib;

    monitor_params(&p, v);
    p.w = s->w;
    p.h = s->h;
    p.hue = s->hue;
    p.xoffset = s->xoffset;
    p.yoffset = s->yoffset;
#if HAS_DOT_CRAWL
    aux = s->dot_crawl_offset;
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NES) && defined(NES_BORDER) && NES_BORDER
    p.flags |= CRTHIP_F_NES_BORDER;                 /* crt_nes.c:69 */
    p.nes_border_color = (int) s->border_color;
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NES) || (CRT_SYSTEM == CRT_SYSTEM_NESRGB)
    /* crt_nes.c:118-121, crt_nesrgb.c:63-66: the sync skeleton is only written on the first call */
    if (!s->field_initialized) {
        p.flags |= CRTHIP_F_NES_SETUP;
    }
    /* crt_nes.c:118-121 / crt_nesrgb.c:63-66 write the skeleton before any of their early returns; here it is written
     * by the launch below, so field_initialized is only set once that launch is certain: a call that returns early
     * (unknown pixel format, empty image) leaves it 0 and the next valid call writes the skeleton */
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NES)
    img_bytes = (size_t) s->w * (size_t) s->h * 2;
#else
    p.format = s->format;
#if (CRT_SYSTEM != CRT_SYSTEM_NESRGB)
    p.raw = s->raw;
    p.as_color = s->as_color;
    s->iirs_initialized = 1;
#endif
    if (crt_setup_bpp4fmt(s->format) == 0) {
        return;     /* like the reference: nothing else happens for an unknown pixel format (NES-RGB: the skeleton is
                       then written by the first call with a valid one, field_initialized stays 0) */
    }

#if (CRT_SYSTEM != CRT_SYSTEM_NESRGB)
    s->field &= 1;
    s->frame &= 1;
    field = s->field;
    frame = s->frame;
#endif
#if (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS)
    if (s->do_aberration) {
        aux = ((rand() % 12) - 8) + 14;     /* same libc stream as the reference consumes */
    }
#endif
    img_bytes = (size_t) s->w * (size_t) s->h * (size_t) crt_setup_bpp4fmt(s->format);
#endif
    if (s->w <= 0 || s->h <= 0 || s->data == 0) {
        return;
    }
    CHECK(crthip_params_finalize(&p));
    /* The reference writes analog[(x + xo) + (y + yo) * CRT_HRES] whatever the offsets are.  Rectangles that merely
     * run over the end of a line are reproduced (flat index); what would leave analog[] -- the reference then
     * scribbles over inp[] and the rest of the struct, e.g. with video_convert.c's uninitialised xoffset/yoffset
     * (extra/video_convert.c:153) -- is refused: warn once, leave analog[] alone. */
    CHECK(crt_sysdef_get(&sd, SYS_ID, CRT_CHROMA_PATTERN));
    end = (long) p.yo * sd.hres + p.xo + (long) (p.desth - 1) * sd.hres + p.destw;
    if (p.xo < 0 || p.yo < 0 || p.destw <= 0 || p.desth <= 0 || end > (long) sd.input_size) {
        static int warned;
        if (!warned) {
            fprintf(stderr, "ntsc-crt (HIP/gfx950): crt_modulate: xoffset %d / yoffset %d put the picture outside "
                            "analog[] -- field left unchanged (the reference would write out of bounds)\n",
                    s->xoffset, s->yoffset);
            warned = 1;
        }
        return;
    }

#if (CRT_SYSTEM == CRT_SYSTEM_NES) || (CRT_SYSTEM == CRT_SYSTEM_NESRGB)
    s->field_initialized = 1;
#endif
    lib = park_libc_rand();
    sl = get_slot(v);
    /* The reference clamps the source row with `if (sy >= h) sy = h` (crt_ntsc.c:263) and so reads the image row
     * BEHIND the caller's buffer on odd fields of small raw images.  That memory is not ours to read; the device copy
     * gets one spare row of zeros instead -- what a freshly mapped heap block behind a calloc'ed image holds, i.e. what
     * the reference's drivers see in practice (crt_main.c -r; tests/test_gpu_dropin.py) -- re-zeroed on every call. */
    {
        const size_t row_bytes = img_bytes / (size_t) s->h;
        const int lazy = lazy_mode();
        unsigned long h = 0;
        int same = 0;
        if (lazy) {
            h = buf_hash(s->data, img_bytes, lazy);
            same = sl->img_valid && sl->img_ptr == (const void *) s->data && sl->img_bytes == img_bytes && sl->img_hash == h &&
                   sl->img_cap >= img_bytes + row_bytes + 256;
            if (!same) {
                pin_buffer(&sl->pinned_img, &sl->pinned_img_bytes, s->data, img_bytes);
            }
        }
        if (!same) {
            ensure(&sl->d_img, &sl->img_cap, img_bytes + row_bytes + 256);
            CHECK(crthip_upload(g_ctx, sl->d_img, s->data, img_bytes));
            CHECK(crthip_memset(g_ctx, sl->d_img + img_bytes, 0, row_bytes));
            sl->img_valid = lazy != 0;
            sl->img_ptr = s->data;
            sl->img_bytes = img_bytes;
            sl->img_hash = h;
        }
        p.flags |= CRTHIP_F_IMAGE_SPARE_ROW;
    }
    sync_analog_to_device(sl, v);
    state_to_device(sl, v, field, frame, aux);
    CHECK(crthip_modulate(g_ctx, &p, 1, sl->d_img, 0, sl->d_analog, sl->d_state));
    if (!lazy_mode()) {
        CHECK(crthip_download(g_ctx, v->analog, sl->d_analog, CRT_INPUT_SIZE));
    } else {
        mark_analog_stale(sl, v);
    }
    state_from_device(sl, v, 0);
    unpark_libc_rand(lib);
}

extern void
crt_demodulate(struct CRT *v, int noise)
{
    struct slot *sl;
    crthip_params p;
    size_t out_bytes;
    int bpp;
    int *lib;

    bpp = crt_setup_bpp4fmt(v->out_format);
    if (bpp == 0) {
        return;
    }
    if (v->outw <= 0 || v->outh <= 0 || v->out == 0) {
        return;
    }
    monitor_params(&p, v);
    p.noise = noise;
    CHECK(crthip_params_finalize(&p));

    lib = park_libc_rand();
    sl = get_slot(v);
    out_bytes = (size_t) v->outw * (size_t) v->outh * (size_t) bpp;
    {
        const int lazy = lazy_mode();
        int same = 0;
        if (lazy) {
            same = sl->out_valid && sl->out_ptr == (const void *) v->out && sl->out_bytes == out_bytes &&
                   sl->out_cap >= out_bytes + 256 && sl->out_hash == buf_hash(v->out, out_bytes, lazy);
            if (!same) {
                pin_buffer(&sl->pinned_out, &sl->pinned_out_bytes, v->out, out_bytes);
            }
        }
        ensure(&sl->d_out, &sl->out_cap, out_bytes + 256);
        sync_analog_to_device(sl, v);
        if (!same) {
            CHECK(crthip_upload(g_ctx, sl->d_out, v->out, out_bytes));
        }
    }
    state_to_device(sl, v, 0, 0, 0);
#if (CRT_SYSTEM == CRT_SYSTEM_NTSCVHS) && CRT_VHS_NOISE
    {
        unsigned hist[32];
        read_libc_rand(lib, hist);
        if (d_hist_buf == 0) {
            d_hist_buf = (unsigned *) dev_alloc(32 * sizeof(unsigned));
        }
        CHECK(crthip_upload(g_ctx, d_hist_buf, hist, 31 * sizeof(unsigned)));
        CHECK(crthip_vhs_bind_history(g_ctx, d_hist_buf));
        CHECK(crthip_noise(g_ctx, &p, 1, sl->d_analog, sl->d_inp, sl->d_state));
        CHECK(crthip_download(g_ctx, hist, d_hist_buf, 31 * sizeof(unsigned)));
        write_libc_rand(lib, hist);
    }
#else
    CHECK(crthip_noise(g_ctx, &p, 1, sl->d_analog, sl->d_inp, sl->d_state));
#endif
    CHECK(crthip_sync(g_ctx, &p, 1, sl->d_inp, sl->d_state, sl->d_lines));
    CHECK(crthip_decode(g_ctx, &p, 1, sl->d_inp, sl->d_lines, sl->d_out, out_bytes));
    if (!lazy_mode()) {
        CHECK(crthip_download(g_ctx, v->inp, sl->d_inp, CRT_INPUT_SIZE));
    }
    CHECK(crthip_download(g_ctx, v->out, sl->d_out, out_bytes));
    if (lazy_mode()) {
        sl->out_valid = 1;
        sl->out_ptr = v->out;
        sl->out_bytes = out_bytes;
        sl->out_hash = buf_hash(v->out, out_bytes, lazy_mode());
    }
    state_from_device(sl, v, 1);
    unpark_libc_rand(lib);
}
