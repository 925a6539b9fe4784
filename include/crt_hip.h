/*
 * crt_hip.h -- C ABI of the MI355X (gfx950) implementation of NTSC-CRT's per-field
 * composite encode -> noisy channel -> decode hot path.
 *
 * Two layers live behind this ABI:
 *
 *  (1) The reference's own entry points (crt_core.h:100-139 of LMP88959/NTSC-CRT):
 *      crt_init / crt_resize / crt_reset / crt_modulate / crt_demodulate /
 *      crt_bpp4fmt / crt_sincos14, with layout-identical `struct CRT` and
 *      `struct NTSC_SETTINGS` (include/crt_core.h and friends).  They are exported
 *      by libntsccrt_hip_<system>.so, one library per CRT_SYSTEM exactly like the
 *      reference is one build per CRT_SYSTEM (crt_core.h:39-59), so the unchanged
 *      crt_main.c / extra/video_convert.c drivers link against it.
 *
 *  (2) The device-resident batch ABI below (crthip_*), exported by libcrthip.so.
 *      It has no counterpart in the reference (which has no batching, SURVEY.md
 *      section 2); layer (1) is a thin C89 client of it.  Plain pointers and sizes
 *      only; device pointers are raw `void *` (e.g. torch.Tensor.data_ptr()).
 *
 * Unit of work: a FIELD-PASS = one crt_modulate (crt_ntsc.c:128 / crt_ntscvhs.c:129 /
 * crt_nes.c:106) followed by one crt_demodulate (crt_core.c:291) on one image.
 * All arithmetic is the reference's 32-bit integer fixed point; results are
 * bit-exact with the CPU path (tests/).
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRTHIP_ABI_VERSION 6

/* CRT_SYSTEM_* of crt_core.h:30-36 */
#define CRTHIP_SYSTEM_NTSC    0
#define CRTHIP_SYSTEM_NES     1
#define CRTHIP_SYSTEM_PV1K    2   /* Casio PV-1000: 5 samples per chroma cycle (crt_pv1k.h, crt_core.c:480-510) */
#define CRTHIP_SYSTEM_SNES    3
#define CRTHIP_SYSTEM_TEMP    4   /* crt_template.c */
#define CRTHIP_SYSTEM_NTSCVHS 5
#define CRTHIP_SYSTEM_NESRGB  6

#define CRTHIP_MAX_VPER  5    /* largest CRT_CC_VPER (PV-1000) */
#define CRTHIP_MAX_CCS   5    /* largest CRT_CC_SAMPLES (PV-1000) */
#define CRTHIP_DCO_MAX   5    /* dot_crawl_offset values 0..5 are tabulated exactly (the headers document 0-5) */
#define CRTHIP_CARRIER_ROWS (CRTHIP_MAX_VPER + CRTHIP_DCO_MAX)   /* rows of the per-line-class carrier tables */

/* CRT_PIX_FORMAT_* of crt_core.h:62-67 */
#define CRTHIP_FMT_RGB  0
#define CRTHIP_FMT_BGR  1
#define CRTHIP_FMT_ARGB 2
#define CRTHIP_FMT_RGBA 3
#define CRTHIP_FMT_ABGR 4
#define CRTHIP_FMT_BGRA 5

/* Bytes of `struct CRT` behind inp[] (outw, outh, out_format, zeroed padding) that the
 * reference's filter window can over-read deterministically when ypos == VRES-1 and
 * hsync >= 5 (crt_core.c:511,539-543).  Mirrored behind every
 * device inp[] so those pixels stay bit-exact.  Deeper over-reads are undefined
 * behaviour in the reference and outside the parity contract. */
#define CRTHIP_TAIL 16

/* error codes (0 = ok).  The reference API is all-void (no error channel,
 * crt_core.h:100-139); layer (1) therefore reports on stderr and aborts. */
#define CRTHIP_OK            0
#define CRTHIP_E_ARG        -1   /* bad argument / unsupported configuration */
#define CRTHIP_E_NODEVICE   -2   /* no HIP device, or device is not gfx950 */
#define CRTHIP_E_HIP        -3   /* HIP runtime error, see crthip_error_string */
#define CRTHIP_E_NOMEM      -4

/* crthip_params.flags */
#define CRTHIP_F_NES_SETUP    2  /* crthip_modulate, NES: NTSC_SETTINGS.field_initialized == 0, i.e. also
                                    write the whole-field sync skeleton (setup_field, crt_nes.c:81-104) */
/* Decoder filter of a USE_CONVOLUTION build of the reference (crt_core.c:85-147): a symmetric FIR kernel of
 * 7 (USE_7_SAMPLE_KERNEL, weights 1 4 7 8 7 4 1), 6, 5 or 4 taps instead of the 3-band IIR equaliser.
 * flags |= CRTHIP_F_EQ_FIR(7); 0 taps (the default) = the equaliser of the reference's stock build. */
#define CRTHIP_F_VHS_DRAW_ABERRATION 4  /* crthip_sequence, VHS: draw every field's aberration height (state.aux)
                                           from the rand() stream like crt_modulate does with do_aberration
                                           (crt_ntscvhs.c:205-207) instead of taking state.aux from the caller */
/* CRT_DO_BLOOM build of the reference (crt_core.h:70; crt_core.c:399-402,512-526; encoder geometry crt_ntsc.c:148-161) */
#define CRTHIP_F_BLOOM        8
/* Every image is followed by one more readable row (image_stride covers h + 1 rows, also behind the last image).
 * The reference clamps the source row with `if (sy >= h) sy = h` (crt_ntsc.c:263, sic) and so reads row h -- one
 * past the image -- for odd fields of raw images with h <= desth.  With this flag the kernels read that row like
 * the reference does; without it they read row h - 1 instead and never touch memory behind an image. */
#define CRTHIP_F_IMAGE_SPARE_ROW 16
/* Further build-time switches of the reference as run-time flags (each is a #define in its sources / headers; 0 = the
 * shipped build).  All of them reproduce the corresponding rebuilt reference bit for bit (tests/, oracle/_ref): */
#define CRTHIP_F_NO_HSYNC     0x20    /* CRT_DO_HSYNC 0 (crt_core.h:72; crt_core.c:446-450): hsync = 0 after every line      */
#define CRTHIP_F_NO_VSYNC     0x40    /* CRT_DO_VSYNC 0 (crt_core.h:71; crt_core.c:323-341): field parity found in the CLEAN
                                         signal, vsync = -3 from then on                                                    */
#define CRTHIP_F_VHS_LCG_NOISE 0x80   /* CRT_VHS_NOISE 0 (crt_ntscvhs.h:29; crt_core.c:343-357): the VHS build with the LCG
                                         noise of every other system, no rand() stream                                      */
#define CRTHIP_F_NES_BORDER   0x1000  /* NES_BORDER 1 (crt_nes.c:69,138-160): crthip_params.nes_border_color right of the picture,
                                         lines CRT_TOP .. CRT_BOT + 2, rewritten by every crt_modulate                      */
#define CRTHIP_F_HIPASS       0x800   /* HIPASS 1 (crt_ntsc.c:115-126 and its siblings): iirf returns s - h ("for debugging") */
#define CRTHIP_F_VHS_LP       0x2000  /* VHS_MODE VHS_LP / VHS_EP (crt_ntscvhs.h:102-124): the encoder's band limits of the  */
#define CRTHIP_F_VHS_EP       0x4000  /*   other two tape speeds                                                            */
#define CRTHIP_F_EQ_FIR(taps)  ((taps) << 8)
#define CRTHIP_F_EQ_FIR_MASK   (7 << 8)

/*
 * Everything that is uniform over a batch of field-passes.  Plain old data, no
 * pointers: this is also the blob rank 0 broadcasts over RCCL in the multi-GPU
 * driver.  Fill the USER part, then call crthip_params_finalize(), which derives
 * the rest on the host exactly as the reference does per call (crt_ntsc.c:142-203,
 * crt_core.c:272-280, :305-320, :403-407, :528).
 */
typedef struct crthip_params {
    /* ---- user part ------------------------------------------------------- */
    int system;           /* CRTHIP_SYSTEM_*                                   */
    int chroma_pattern;   /* CRT_CHROMA_PATTERN (crt_ntsc.h:25, crt_nes.h:30)  */
    /* encoder input: struct NTSC_SETTINGS minus data/field/frame
     * (crt_ntsc.h:111-124, crt_ntscvhs.h:133-147, crt_nes.h:132-143) */
    int w, h;             /* image size                                        */
    int format;           /* CRTHIP_FMT_* of the input image (ignored for NES) */
    int raw, as_color;
    int hue;              /* encoder hue                                       */
    int xoffset, yoffset;
    /* decoder: the caller-visible part of struct CRT (crt_core.h:77-86) */
    int outw, outh, out_format;
    int mon_hue, brightness, contrast, saturation;
    int black_point, white_point;
    int scanlines, blend;
    unsigned v_fac;
    int noise;            /* crt_demodulate's `noise` argument                 */
    int flags;            /* CRTHIP_F_*                                        */
    /* ---- derived part (crthip_params_finalize) --------------------------- */
    int finalized;        /* magic, set by crthip_params_finalize              */
    int in_bpp, out_bpp;
    int destw, desth, xo, yo;      /* crt_ntsc.c:132-133,163-173,194-203       */
    /* carrier tables [row][t % CC_SAMPLES] (crt_ntsc.c:174-188, crt_snes.c:171-183, crt_nes.c:123-130).
     * row = (line % CC_VPER) + dot_crawl_offset for the systems whose carriers depend on the line class (NES,
     * NES-RGB, SNES, template, PV-1000; the unreduced sum, so that negative angles truncate exactly like the
     * reference's); NTSC / VHS: row = (field == frame), row 1 carrying the inverted line phase of crt_ntsc.c:199-200 */
    int burst[CRTHIP_CARRIER_ROWS][CRTHIP_MAX_CCS];
    int modI[CRTHIP_CARRIER_ROWS][CRTHIP_MAX_CCS];
    int modQ[CRTHIP_CARRIER_ROWS][CRTHIP_MAX_CCS];
    int dem_cs[2][CRTHIP_MAX_CCS]; /* 5-sample demodulator: cos / sin of hue + i*72 (I) and + 90 (Q), crt_core.c:497-505 */
    int dem_sn[2][CRTHIP_MAX_CCS];
    int iir_c[3];                  /* crt_ntsc.c:98-106, Q11                   */
    int eq_lf[3], eq_hf[3];        /* crt_core.c:171-196, Q16                  */
    int eq_g[3][3];                /* crt_core.c:278-280                       */
    int huesn, huecs;              /* crt_core.c:318-320, 4-bit                */
    int bright;                    /* crt_core.c:305                           */
    int white;                     /* WHITE_LEVEL*white_point/100, crt_ntsc.c:318 */
    int ire_base;                  /* BLACK_LEVEL + black_point, crt_ntsc.c:311 */
    int dx;                        /* crt_core.c:528                           */
    int ratio;                     /* crt_core.c:404-405                       */
    int eq_kernel;                 /* 0 or the FIR taps from flags (validated)   */
    int bloom;                     /* flags & CRTHIP_F_BLOOM                     */
    int bloom_max_e;               /* crt_core.c:400                             */
    /* source column of destination sample x, crt_ntsc.c:272: x * w / destw == (x * col_step) >> 32 with
     * col_step = ceil(2^32 * w / destw), exact while x * destw < 2^32 */
    unsigned col_step_lo, col_step_hi;
    /* Decoder envelope: the largest carrier amplitude |wave| for which the I / Q equalisers' low cascades
     * can be dropped.  crthip_params_finalize sets the bound that holds for ANY inp[] (|s| <= 127: 65 532); the fused entry
     * points (crthip_fieldpass, crthip_sequence), which produce inp[] themselves, raise it from the signal range they know
     * (sync level ... white level + noise term).  A performance hint only: every setting gives identical pictures. */
    int loskip_wave_max;
    int nes_border_color; /* user: NTSC_SETTINGS.border_color (crt_nes.h:136), drawn with CRTHIP_F_NES_BORDER */
    int reserved[2];
} crthip_params;

/*
 * Per-field state, device resident, one entry per field-pass of a batch.  These are
 * the members of struct CRT / NTSC_SETTINGS that vary from field to field and carry
 * over between calls.
 */
typedef struct crthip_state {
    int field, frame;     /* in : NTSC_SETTINGS.field / .frame (NTSC, VHS)     */
    int aux;              /* in : dot_crawl_offset (NES, NES-RGB, SNES, template, PV-1000); VHS aberration lines */
    int hsync, vsync;     /* i/o: crt_core.h:89                                */
    int rn;               /* i/o: crt_core.h:91                                */
    int ccf[CRTHIP_MAX_VPER][CRTHIP_MAX_CCS];   /* i/o: crt_core.h:88 (rows >= CRT_CC_VPER, columns >= CRT_CC_SAMPLES unused) */
    int odd_field;        /* out: field parity found by the vsync search        */
    int reserved[4];
} crthip_state;           /* 144 bytes */

/* What the serial sync chain (crt_core.c:428-479) hands to the filter stage, per
 * decoded line (CRT_TOP..CRT_BOT-1), device resident. */
typedef struct crthip_line {
    int pos;              /* first sample of the active window in inp[] (:454) */
    int wave0, wave1;     /* wave[0], wave[1] (:476-477); [2],[3] = negations.  5-sample systems: dci, dcq (:493-494) */
    int beg;              /* first output row (:428)                            */
    int nrows;            /* rows written: 1 + duplicates (:661-664); 0 = line skipped (:431) */
    int hsync;            /* hsync after this line (diagnostic)                 */
    int dx, scanl;        /* resampler step / start, 12-bit fraction (:528-529; per line with bloom, :519-521) */
} crthip_line;            /* 32 bytes */

typedef struct crthip_ctx crthip_ctx;

/* kernels, for crthip_profile_read() */
#define CRTHIP_K_TEMPLATE 0   /* M4: blanking / sync / burst skeleton           */
#define CRTHIP_K_ACTIVE   1   /* M5: RGB->YIQ, band-limit, quadrature modulate  */
#define CRTHIP_K_NOISE    2   /* D1: channel noise                              */
#define CRTHIP_K_SYNC     3   /* D2-D7: vsync, hsync, burst lock (serial chain) */
#define CRTHIP_K_DECODE   4   /* D8-D10: equalisers, resample, YIQ->RGB, rows   */
#define CRTHIP_K_COUNT    5

int  crthip_abi_version(void);
int  crthip_device_count(void);

/* Host-only helpers (no device needed). */
int  crthip_params_default(crthip_params *p, int system, int chroma_pattern);
int  crthip_params_finalize(crthip_params *p);
int  crthip_input_size(int system, int chroma_pattern);      /* CRT_INPUT_SIZE */
int  crthip_hres(int system, int chroma_pattern);            /* CRT_HRES       */
int  crthip_lines(int system);                               /* CRT_LINES      */
size_t crthip_field_stride(int system, int chroma_pattern);  /* bytes between consecutive
                                   fields in analog[] / inp[] device buffers (>= INPUT_SIZE+CRTHIP_TAIL) */

/* VHS only.  The reference's VHS noise is the C library's rand() stream (crt_core.c:344-351); on
 * glibc that is y[n] = y[n-31] + y[n-3], rand() = y[n] >> 1.  The kernels take the generator state
 * of every field as its 31-word history (the values preceding the next output), in/out, device
 * resident, 32 words apart.  crthip_vhs_history_from_seed gives the history right after srand(seed). */
int  crthip_vhs_history_from_seed(unsigned seed, unsigned hist[31]);
int  crthip_vhs_bind_history(crthip_ctx *ctx, unsigned *d_hist);     /* n x 32 words, device */

/* Context = one device + one stream + the jump tables of the noise LCG. */
int  crthip_create(crthip_ctx **out, int device, int system, int chroma_pattern);
void crthip_destroy(crthip_ctx *ctx);
int  crthip_set_stream(crthip_ctx *ctx, void *hip_stream);   /* hipStream_t; NULL = the default stream,
                                                                  which is also the initial setting */
int  crthip_synchronize(crthip_ctx *ctx);
const char *crthip_error_string(const crthip_ctx *ctx);

/* Pre-allocate the per-field workspace (inp[], line table) for up to n fields so that
 * crthip_fieldpass() performs no allocation inside a timed region. */
int  crthip_reserve(crthip_ctx *ctx, int n_fields);

/*
 * One batch of n independent field-passes, everything device resident:
 *   d_images : n images, image k at d_images + k*image_stride (w*h*bpp bytes, or
 *              w*h u16 PPU pixels for NES)
 *   d_out    : n output images, k at d_out + k*out_stride (outw*outh*bpp bytes);
 *              read as well as written when blend != 0, and rows this field does not
 *              own keep their contents (crt_core.c:431,:608,:662)
 *   d_state  : n crthip_state, updated in place
 * Each field starts from a crt_init-clean analog[] (all zero where crt_modulate does
 * not write, crt_ntsc.c:236-238).  Asynchronous on the context's stream: once crthip_reserve
 * has sized the workspace a call allocates nothing and does not synchronise, so the launch
 * sequence can be captured into a HIP graph -- the first call of a context included: the
 * tables the context caches (skeleton fields, the NES sample table) are then built at once
 * on a stream of the library's own, outside the capture, and the graph holds only the
 * per-call kernels.  (A graph reads the tables of the settings it was captured with: do not
 * replay it after calls with other settings went through the same context.)
 */
int  crthip_fieldpass(crthip_ctx *ctx, const crthip_params *p, int n,
                      const void *d_images, size_t image_stride,
                      void *d_out, size_t out_stride,
                      crthip_state *d_state);

/*
 * SEQUENCE mode: n consecutive fields of ONE television set -- what the reference's
 * batch driver does (extra/video_convert.c:246-277:  for every frame: crt_modulate; crt_demodulate;
 * write the output image), with the sync state and the output buffer carried from field to field,
 * yet processed by parallel kernels (DESIGN.md "Sequence mode").
 *   d_state[k].field / .frame / .aux : encoder inputs of field k;  d_state[0].hsync/.vsync/.rn : the
 *   set's state before field 0; on return d_state[k] holds the state after field k.
 *   d_out image k = the (single) output buffer as it stands after field k; d_out_init = its content
 *   before field 0 (NULL = zeros, i.e. calloc as in the drivers).
 * blend != 0 (crt_main.c:235) makes the picture a recurrence over the fields: the fields are then decoded in
 * parallel and folded into each other by one small pass per field (needs outh + v_fac >= CRT_LINES).
 * *passes (optional) receives the number of sync fixed-point passes that were needed.
 * VHS build: the fields also share ONE rand() stream.  Entry 0 of the bound history array
 * (crthip_vhs_bind_history) is the generator before field 0; a serial pre-pass walks the stream's
 * data-dependent part for all fields (about 0.15 ms per field) and on return entry k is the generator after
 * field k.  With CRTHIP_F_VHS_DRAW_ABERRATION the aberration heights are drawn from the stream as well.
 */
int  crthip_sequence(crthip_ctx *ctx, const crthip_params *p, int n,
                     const void *d_images, size_t image_stride,
                     void *d_out, size_t out_stride, const void *d_out_init,
                     crthip_state *d_state, int *passes);

/*
 * The phases of crthip_sequence, for hosts that cut ONE video over several contexts / devices / processes
 * (include/crt_hip_node.h; ntsc-crt_amd/shard.py over torch.distributed).  A shard holds the n consecutive fields
 * [first_index, first_index + n) of the video; crthip_sequence == encode(0, rn0) + sync + decode + weave on one context.
 *   encode : state[k].rn = the set's generator before field first_index + k (closed form from rn0, the generator before
 *            field 0 of the VIDEO); all fields encoded, noise fused.  VHS: first_index must be 0 (one rand() stream).
 *   sync   : the sync chain from the incoming (hsync_in, vsync_in) = the set's state before the shard's first field;
 *            returns the pair after its last field.  May be called again with another incoming pair (the predecessor
 *            shard's final state became known): it then restarts from its previous finals.
 *   decode : rn after each field, every field decoded (without blend).
 *   weave  : image k = the output buffer after field k, given d_out_init = the buffer before the shard's first field
 *            (NULL = zeros).  patch_only != 0 (blend == 0 only): the images were woven before with a placeholder init;
 *            only the rows no field of the shard wrote are taken from d_out_init now.
 *
 * VHS build: a video's fields share ONE rand() stream, so a shard cannot start in the middle of it -- unless the stream was
 * walked ahead for the whole video first: crthip_vhs_chain does that for n consecutive fields on ONE context (entry 0 of its
 * bound history array = the generator before field 0; on return entry k = the generator at the start of field k, and with
 * draw_aberration state[k].aux = the aberration height crt_modulate draws, crt_ntscvhs.c:205-207).  A context whose bound
 * histories (and aux) were filled from that array is told so with crthip_seq_vhs_prechained(ctx, 1); its crthip_seq_encode
 * then accepts any first_index and does not walk the stream again.  (include/crt_hip_node.h does all of this.)
 */
int  crthip_vhs_chain(crthip_ctx *ctx, int n, crthip_state *d_state, int draw_aberration);
int  crthip_seq_vhs_prechained(crthip_ctx *ctx, int on);
int  crthip_seq_encode(crthip_ctx *ctx, const crthip_params *p, int n, int first_index, int rn0,
                       const void *d_images, size_t image_stride, crthip_state *d_state);
int  crthip_seq_sync(crthip_ctx *ctx, const crthip_params *p, int n, crthip_state *d_state, int hsync_in, int vsync_in,
                     int *hsync_out, int *vsync_out, int *passes);
int  crthip_seq_decode(crthip_ctx *ctx, const crthip_params *p, int n, void *d_out, size_t out_stride, crthip_state *d_state);
int  crthip_seq_weave(crthip_ctx *ctx, const crthip_params *p, int n, void *d_out, size_t out_stride,
                      const void *d_out_init, int patch_only);

/*
 * Stage-level entry points (used by the drop-in layer, which must keep the host's
 * struct CRT coherent between crt_modulate and crt_demodulate, and by the stage
 * parity tests).  d_analog / d_inp hold n fields at crthip_field_stride() spacing.
 */
/* crt_modulate: writes exactly the samples the reference writes (crt_ntsc.c:205-324),
 * leaving all other samples of d_analog untouched; updates state.ccf (and VHS hsync). */
int  crthip_modulate(crthip_ctx *ctx, const crthip_params *p, int n,
                     const void *d_images, size_t image_stride,
                     signed char *d_analog, crthip_state *d_state);
/* crt_demodulate D1: d_inp = clamp(d_analog + noise), state.rn advanced (crt_core.c:346-367) */
int  crthip_noise(crthip_ctx *ctx, const crthip_params *p, int n,
                  const signed char *d_analog, signed char *d_inp, crthip_state *d_state);
/* crt_demodulate D2-D7: vsync, per-line hsync / burst lock -> line table (crt_core.c:379-479) */
int  crthip_sync(crthip_ctx *ctx, const crthip_params *p, int n,
                 const signed char *d_inp, crthip_state *d_state, crthip_line *d_lines);
/* crt_demodulate D8-D10 (crt_core.c:534-664) */
int  crthip_decode(crthip_ctx *ctx, const crthip_params *p, int n,
                   const signed char *d_inp, const crthip_line *d_lines,
                   void *d_out, size_t out_stride);

/* crthip_fieldpass may cut a batch into `chunks` pieces that alternate between the caller's
 * stream and an internal stream (fenced by events on both sides), so that the latency-bound
 * kernels of one piece overlap the ALU-bound kernels of the next.  1 = off, 0 = chosen per launch (default). */
int  crthip_set_overlap(crthip_ctx *ctx, int chunks);

/* Kernel shape.  0 (default): chosen per launch -- lane-per-scanline kernels (64 scanlines per wavefront, the
 * throughput shape) when the batch fills the chip, scanline-parallel kernels (a DPP row of 16 or 32 lanes per
 * scanline: one filter stage per lane, samples handed on with row_shr, pixels emitted by the whole wavefront from
 * LDS -- the latency shape) for small batches.  1 / 2 force the throughput / latency shape (tests, tuning).
 * Bloom builds (a resampler geometry per scanline) take the throughput shape after a counting sort of the batch's scanlines
 * by beam width, so that the 64 scanlines of a wavefront share one geometry (crt_decode3.hip).
 * The automatic choice minimises the time of ONE batch (encoder: latency shape up to 256 fields, decoder up to 128).  A
 * caller that keeps several batches in flight hides the latency anyway and is better off forcing the
 * throughput shape from 128 fields up (256 fields, three in flight: 1.13 M against 1.01 M fields/s). */
int  crthip_set_shape(crthip_ctx *ctx, int shape);


/* HIP graphs: the encoder's cached tables (blanking / sync / burst skeleton, NES sample table) are rebuilt when the settings they
 * derive from change; this counts the rebuilds.  A graph captured at generation g keeps replaying with generation g's tables --
 * they stay allocated until crthip_destroy and are never written again -- i.e. with the settings it was captured with; compare
 * the value at capture with the current one to know whether a kept graph still matches the context's latest settings. */
unsigned crthip_table_generation(const crthip_ctx *ctx);

/* Decoder output tile: 16 or 32 pixels per row and flush (0 = choose by output width, default). */
int  crthip_set_pixel_tile(crthip_ctx *ctx, int pixels);
/* Encoder signal tile of the fused path: how many bytes of a scanline leave the encoder per store piece.  0 (default) = by
 * batch size (64-byte pieces; 128 / 256-byte pieces for batches that keep the chip full at the occupancy their LDS tile
 * leaves), 16 = always the 64-byte pieces, 32 or 64 = always the large ones.  Same bytes either way (tests, A/B measurements).
 * Environment: CRTHIP_SIG_TILE sets the default of new contexts. */
int  crthip_set_signal_tile(crthip_ctx *ctx, int dwords);
/* Signal layout of the fused path (round 6).  The reference addresses analog[] / inp[] flat: line n of a field starts at sample
 * n * CRT_HRES (crt_ntsc.c:322, crt_core.c:438-461), and so do the stage-level entry points above and the drop-in libraries.  The
 * signal crthip_fieldpass keeps in its own workspace between its encoder and its decoder is not visible to anybody; by default
 * (1) it lives in PADDED lines -- 1024 bytes apart (2048 for the PV-1000's 1920-sample lines), active rows on 128-byte boundaries,
 * the head of every line repeated behind its predecessor so that windows over a line end stay contiguous -- which is what lets the
 * encoder store whole aligned cache lines.  0 = the flat layout there too (A/B measurements; also what the library takes by itself
 * where the padded layout does not reach or does not pay: row overhangs of more than 16 samples, x offsets that put the row before
 * column 114, the rand()-noise VHS build, CRT_DO_VSYNC 0, the NES's PPU-pixel encoder, and the batches of up to 256 fields that go to
 * the scanline-parallel encoder).  Same pictures and states either way.  Environment: CRTHIP_SIG_PAD sets the default
 * of new contexts.
 * crthip_fieldpass_signal: the noisy signal of the first n fields of the context's LAST crthip_fieldpass, repacked into the
 * reference's layout (n fields at crthip_field_stride() spacing: CRT_INPUT_SIZE samples + the CRTHIP_TAIL mirror), i.e. what
 * crt_demodulate leaves in CRT.inp -- for parity tests of the fused path; *padded (optional) tells which layout it came from. */
int  crthip_set_signal_layout(crthip_ctx *ctx, int padded);
/* Host only (no device needed): which layout a crthip_fieldpass of n_fields with these (finalized) parameters takes under the default
 * switches and kernel shape `shape` (crthip_set_shape) -- returns 1 = padded, 0 = flat, < 0 = error; layout[] = { bytes between line
 * starts, bytes in front of line 0, valid copy bytes behind every line, samples of an active row beyond its line's end }, *field_stride
 * = bytes between the fields of the workspace.  (What crthip_reserve sizes the workspace for, and what the CPU tests check the
 * geometry rules with.) */
int  crthip_signal_layout_query(const crthip_params *p, int n_fields, int shape, int layout[4], size_t *field_stride);
int  crthip_fieldpass_signal(crthip_ctx *ctx, int n, signed char *d_inp_flat, int *padded);
/* Wide-run decoder (wide pictures, crt_decode4.hip): scanlines per wavefront.  0 (default) = by batch size (8 below 96 fields of
 * 1920x1080, 16 from there on), 8 / 16 = always that instantiation.  Same pictures either way (tests pin each instantiation to the
 * oracle, A/B measurements).  Environment: CRTHIP_WIDE_LPW sets the default of new contexts. */
int  crthip_set_wide_lpw(crthip_ctx *ctx, int scanlines_per_wave);

/* The decoder and encoder normally run kernels whose multiplies are the full-rate 24-bit
 * instructions; they are dispatched only where every operand is proven to fit (DESIGN.md,
 * "24-bit multiply envelope"), everything else goes to the exact 32-bit instantiation.  Both
 * give identical results; this switch forces the 32-bit kernels everywhere (tests, debugging).
 * on: 0 automatic, 1 exact kernels only, 2 no 64-bit-mad tiers, 3 never drop the I/Q low cascades (= 2 since round 3). */
int  crthip_set_exact(crthip_ctx *ctx, int on);

/* Per-kernel timing with HIP events on the context's stream (bench.py roofline leg).
 * While enabled every launch is bracketed by an event pair. */
int  crthip_profile_enable(crthip_ctx *ctx, int on);
int  crthip_profile_read(crthip_ctx *ctx, double total_ms[CRTHIP_K_COUNT],
                         int launches[CRTHIP_K_COUNT]);   /* synchronises; resets */

/* Raw device memory helpers for hosts without a tensor library (the C89 drop-in layer). */
void *crthip_malloc(crthip_ctx *ctx, size_t bytes);
void  crthip_free(crthip_ctx *ctx, void *d_ptr);
int   crthip_upload(crthip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int   crthip_download(crthip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int   crthip_memset(crthip_ctx *ctx, void *d_dst, int value, size_t bytes);
/* page-lock / release a host buffer so that crthip_upload / crthip_download to it run as direct DMA */
int   crthip_host_register(crthip_ctx *ctx, void *h_ptr, size_t bytes);
int   crthip_host_unregister(crthip_ctx *ctx, void *h_ptr);

#ifdef __cplusplus
}
#endif
#endif
