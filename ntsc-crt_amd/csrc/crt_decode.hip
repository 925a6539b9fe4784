/* crt_decode.hip -- host side of the lane-per-scanline decoder (kernel: crt_decode_lane.h).  See crt_dev.h. */
#include "crt_decode_lane.h"

/* host half of the decoder envelopes: the batch-wide floor of the decoder tier.
 * tier 0 additionally needs the luma coefficients in [2^15, 1.5*2^16) and the chroma ones below 2^15 */
static int decoder_min_tier(const crthip_ctx *c, const crthip_params *p)
{
    const int b = p->bright < 0 ? -p->bright : p->bright;
    const long ct = p->contrast < 0 ? -(long) p->contrast : (long) p->contrast;
    if (c->force_exact || b > FAST_BRIGHT_MAX || ct >= (1 << 23)) return 3;
    /* the 64-bit-mad tiers: luma coefficients near 2^16 (the form x' = hi32(((c - 2^16) << 16) * d + {2^31, u})), chroma ones
     * below 2^15 */
    bool coef_ok = p->eq_lf[0] >= 32768 && p->eq_lf[0] < 98304 && p->eq_hf[0] >= 32768 && p->eq_hf[0] < 98304 &&
                   p->eq_lf[1] > 0 && p->eq_lf[1] < 32768 && p->eq_hf[1] > 0 && p->eq_hf[1] < 32768 &&
                   p->eq_lf[2] > 0 && p->eq_lf[2] < 32768 && p->eq_hf[2] > 0 && p->eq_hf[2] < 32768;
    /* ... and no luma stage may wrap in the reference: stage k of a cascade with alpha = c / 2^16 obeys
     * |x_k| <= (alpha |x_(k-1)| + 1) / (1 - |1 - alpha|), its product is c * (x_(k-1) - x_k) */
    for (int cas = 0; cas < 2 && coef_ok; cas++) {
        const double cf = cas ? p->eq_hf[0] : p->eq_lf[0], al = cf / 65536.0, den = 1.0 - (al > 1.0 ? al - 1.0 : 1.0 - al);
        double prev = 127.0 + b, worst = 0.0;
        for (int k = 0; k < 4; k++) {
            const double cur = (al * prev + 1.0) / den + 1.0;
            if (prev + cur > worst) worst = prev + cur;
            prev = cur;
        }
        if (cf * worst + 32768.0 >= 2147483647.0) coef_ok = false;
    }
    if (c->no_tier0 || !coef_ok || b > T0_BRIGHT_MAX) return 2;
    /* ... and ((v >> 12) * contrast) >> 8 must not wrap in the reference, because tiers 0 / 1 take it from an exact 64-bit
     * product (D9): |luma| <= 10 U + 64 with U = 127 + |bright| (both luma gain sets), |chroma| <= 3870,
     * |v| <= 4 * 4095 * |luma| + (4530 + 7021) * |chroma| */
    const long vmax = 16380L * (10L * (127 + b) + 64) + 11551L * 3870L;
    if (((vmax >> 12) + 1) * ct >= (1L << 31)) return 2;
    return c->no_loskip ? 2 : 0;                /* crthip_set_exact(3): keep the I/Q low cascades = the 24-bit tier */
}

int crt_run_decode(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                   const crthip_line *d_lines, void *d_out, size_t ostride, size_t fstride)
{
    /* fstride: bytes between the fields of d_inp -- the flat layout's (0) or the padded one's of the fused path (crt_dev.h, sig_layout);
     * crthip_line.pos is an offset into the field either way, the kernels do not know the difference */
    if (!fstride) fstride = c->fstride;
    if (p->dx <= 0)     /* more than 4096 output pixels per sample: the resampler's step (crt_core.c:528) rounds to 0 */
        return set_err(c, CRTHIP_E_ARG, "outw too large for the 12-bit resampler (dx == 0)", hipSuccess);
    /* kernel shape (crthip_set_shape): the FIR build only exists in the lane-per-scanline shape; a bloom build has a
     * per-scanline resampler geometry, which the scanline-parallel shape takes as it comes and the lane-per-scanline shape
     * after sorting the lines by their width (crt_decode3.hip) */
    if (c->sd.cc_samples != 4 && p->eq_kernel)
        return set_err(c, CRTHIP_E_ARG, "the FIR decoder (USE_CONVOLUTION build) is not available for the 5-sample system", hipSuccess);
    {
        /* k_decode has the band gains of crt_core.c:272-286 compiled in */
        const int five = c->sd.cc_samples == 5;
        const int want[3][3] = { { 65536, five ? 12192 : 8192, five ? 7775 : 9175 }, { 65536, 65536, 1311 }, { 65536, 65536, 0 } };
        for (int k = 0; k < 3; k++)
            for (int b = 0; b < 3; b++)
                if (p->eq_g[k][b] != want[k][b])
                    return set_err(c, CRTHIP_E_ARG, "equaliser gains differ from crt_core.c:272-286", hipSuccess);
    }
    /* ... and a WIDE picture leaves the scanline-parallel shape early: from WIDE_SHAPE_MIN_FIELDS fields on the wide-run decoder
     * (crt_decode4.hip: 16 scanlines per wave, so 32 fields are already two waves per CU) is the faster one -- 1080p x 64: 0.178 ->
     * 0.135 ms, x 128: 0.359 -> 0.191, x 32: 0.126 / 0.130, x 16: 0.114 / 0.126 (profiles/r04_experiments.txt, section 15) */
    const bool wide_px = c->px_tile ? c->px_tile >= 32 : p->outw >= 1280;
    const bool rows_shape = !p->eq_kernel && (c->shape == 2 || (c->shape == 0 && n <= ROWS_SHAPE_MAX_FIELDS &&
                            !(n >= WIDE_SHAPE_MIN_FIELDS && !p->bloom && crt_decode_wide_ok(c, p, decoder_min_tier(c, p), wide_px))));
    if (rows_shape) return crt_run_decode_rows(c, p, n, d_inp, d_lines, d_out, ostride, fstride);
    if (p->bloom) return crt_run_decode_bloom_lanes(c, p, n, d_inp, d_lines, d_out, ostride, decoder_min_tier(c, p), fstride);
    /* FIR build: the filters only add, their outputs stay inside the hull of the inputs, so the 24-bit envelope
     * of tier 2 carries over (tier 4); beyond it the exact instantiation (tier 5) */
    const int min_tier = p->eq_kernel ? (decoder_min_tier(c, p) == 3 ? 5 : 4) : decoder_min_tier(c, p);
    const bool wide = wide_px;
    const bool use_wide = crt_decode_wide_ok(c, p, min_tier, wide);
    /* lines per output row when the picture is shorter than the raster: one pass per rank */
    const unsigned span = (unsigned) p->outh + p->v_fac;
    const int passes = span >= (unsigned) c->sd.lines ? 1 : (int) (((unsigned) c->sd.lines + span - 1) / (span ? span : 1));
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        {
        const int total = n * S::LINES;
        const block_order bo = make_block_order((total + 63) / 64, c->dec_order_env == -1 ? n : c->dec_order_env);
        const dim3 grid(bo.grid), block(64);
        unsigned char *o = (unsigned char *) d_out;
        ProfScope ps(c, CRTHIP_K_DECODE);
        for (int rank = 0; rank < passes; rank++) {
#define CRTHIP_LAUNCH_DECODE(TG, B3) \
    do { if constexpr (S::CCS != 4 && TG == 2) break; /* no FIR build of the 5-sample system */ \
         else if (wide) hipLaunchKernelGGL((k_decode<S, TG, B3, 32>), grid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, min_tier, rank, (const int *) nullptr, bo.K, bo.per); \
         else hipLaunchKernelGGL((k_decode<S, TG, B3, 16>), grid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, min_tier, rank, (const int *) nullptr, bo.K, bo.per); } while (0)
            /* wide pictures in tiers 0 / 1: the 16-scanlines-per-wave kernel with 1 KB row runs (crt_decode4.hip); the groups
             * of the higher tiers stay with k_decode below */
            if (use_wide) {
                const int rc = crt_run_decode_wide(c, p, n, d_inp, d_lines, d_out, ostride, min_tier, rank, fstride);
                if (rc) return rc;
                CRTHIP_LAUNCH_DECODE(1, false);
                continue;
            }
            /* one launch per tier GROUP that can be populated (0: tiers 0 / 1, 1: tiers 2 / 3, 2: the FIR tiers); waves of the
             * other group leave at once */
            if (p->out_bpp == 3) {
                if (min_tier <= 1) CRTHIP_LAUNCH_DECODE(0, true);
                if (min_tier <= 3) CRTHIP_LAUNCH_DECODE(1, true);
                if (min_tier >= 4) CRTHIP_LAUNCH_DECODE(2, true);
            } else {
                if (min_tier <= 1) CRTHIP_LAUNCH_DECODE(0, false);
                if (min_tier <= 3) CRTHIP_LAUNCH_DECODE(1, false);
                if (min_tier >= 4) CRTHIP_LAUNCH_DECODE(2, false);
            }
#undef CRTHIP_LAUNCH_DECODE
        }
        return CRTHIP_OK;
        }
    });
}

