/* crt_decode2.hip -- D8-D10 in the SCANLINE-PARALLEL shape: a DPP row of lanes per scanline, one filter stage per
 * lane.  See crt_dev.h for the lane-per-scanline shape (crt_decode.hip), which is the throughput path.
 *
 * Why a second shape.  The equaliser (crt_core.c:206-233) is 6 cascades (Y/I/Q x low/high) of 4 one-pole stages; stage
 * k of sample x needs stage k-1 of sample x and its own state of sample x-1 -- a systolic array.  With one lane per
 * STAGE, bank b (lanes 4b..4b+3 of a 16-lane DPP row) works on sample t-b at step t and takes its input from the lane
 * four to its left (`row_shr:4`, one DPP move, no LDS), so a scanline advances one sample per ~8 instructions instead
 * of one per ~60, and a single field occupies 60-120 wavefronts instead of 4: this is the shape for small batches
 * (drop-in calls, configs[2]'s 64 frames per GPU) where lane-per-scanline leaves the chip empty.  The price is ~2-3x
 * more issued instructions per scanline, so large batches stay on the lane-per-scanline kernels.
 *
 * Because the pixels are no longer produced inside a lane's serial loop but by the whole wavefront from an LDS ring
 * of y/i/q samples, the resampler geometry may differ per scanline: this kernel is also THE decoder of the
 * CRT_DO_BLOOM build (per-line dx / scanL, crt_core.c:512-526) at every batch size, and it knows the
 * 5-samples-per-cycle system (PV-1000, crt_core.c:480-510, 544-549), whose large batches go to k_decode's exact tier.  All arithmetic is the reference's wrapping 32-bit arithmetic (v_mul_lo_u32), no
 * operand envelope is needed; the NARROW variant only drops the I/Q low cascades where DESIGN.md proves them dead.
 *
 * Work of a wavefront per tile of TS samples (everything wave-synchronous, workgroup = one wave):
 *   prep     u[ch][x]   = s + bright | s * waveI[i % CCS] >> 9 | s * waveQ[i % CCS] >> 9       -> LDS ring
 *   filter   TS steps of { read u (bank 0), DPP hand-over, stage update, bank 3 writes its cascade output }
 *   combine  y/i/q[x]   = band gains over (lo3, hi3, u[x-3])  (crt_core.c:218-232)               -> LDS ring
 *   pixels   every output pixel whose two taps are now available, 64 pixels per pass, rows duplicated (D9, D10)
 */
#include "crt_dev.h"

#define DPP_ROW_SHR4 0x114

template <class S, bool NARROW, bool BPP3, int TS_>
__global__ void __launch_bounds__(64)
k_decode_row(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
             const crthip_line *__restrict__ lines, unsigned char *__restrict__ outp, size_t ostride,
             int want_rank, int force_wide)
{
    constexpr int LPL = NARROW ? 16 : 32;               /* lanes per scanline */
    constexpr int LPW = 64 / LPL;                       /* scanlines per wavefront */
    constexpr int CCS = S::CCS;
    constexpr int TS = TS_, RING = 2 * TS_;             /* samples per tile; LDS ring length (two tiles) */
    constexpr int WINB = ((S::AV_LEN + 31) / 16) * 16;  /* bytes of a line's sample window in LDS */
    __shared__ __attribute__((aligned(16))) signed char s_inp[LPW][WINB];
    __shared__ int s_u[LPW][3][RING + 1];                 /* filter inputs per channel */
    constexpr int NC = NARROW ? 4 : 6;                  /* cascades per scanline: [Ylo, Yhi, Ihi, Qhi] or [Ylo, Yhi, Ilo, Ihi, Qlo, Qhi] */
    constexpr int C_IHI = NARROW ? 2 : 3, C_QHI = NARROW ? 3 : 5;
    __shared__ int s_c[LPW][NC][RING + 1];                /* cascade outputs, slot (x + 3) % RING */
    __shared__ int s_yiq[LPW][3][RING + 1];    /* + 1: (scanline, channel) rows start on different LDS banks */
    __shared__ int s_wave[LPW][2][8];                   /* demodulation carriers per sample phase */
    __shared__ int s_sink[64 + TS];                     /* where the lanes that are not the last stage of a cascade "store" */

    const int lane = threadIdx.x;
    const int lrow = lane >> 4, jj = lane & 15, bank = jj >> 2, cidx = jj & 3;
    const int ls = NARROW ? lrow : lrow >> 1;           /* my scanline slot in the wave */
    const int total = n_fields * S::LINES;
    const int gl = blockIdx.x * LPW + ls;               /* global scanline */
    const bool live = gl < total;

    /* which of the two instantiations owns this group of 4 scanlines: NARROW iff none of them is flagged */
    {
        const int g4 = (blockIdx.x * LPW / 4) * 4 + lrow;
        int fl = 0;
        if (g4 < total) fl = lines[g4].nrows & ((int) CRTHIP_LINE_KEEPLO | CRTHIP_LINE_NOT64 | CRTHIP_LINE_EXACT);
        const bool grp_wide = force_wide || __ballot(fl != 0) != 0ull;
        if (grp_wide == NARROW) return;
    }
    crthip_line lp;
    lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0; lp.hsync = 0; lp.dx = 0; lp.scanl = 0;
    if (live) lp = lines[gl];
    int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    if (((lp.nrows >> CRTHIP_LINE_RANK_SHIFT) & CRTHIP_LINE_RANK_MASK) != want_rank) nrows = 0;
    if (__ballot(nrows > 0) == 0ull) return;
    const int f = live ? gl / S::LINES : 0;

    /* the line's sample range [first, last) (crt_core.c:518-519 with bloom, :530-531 without) */
    const bool bloom = P.bloom != 0;
    const int first = bloom ? (lp.scanl >> 12) : 0;
    const int last = bloom ? S::AV_LEN - 1 : S::AV_LEN;
    const int W = nrows > 0 ? last - first : 0;
    int wmax = W;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(wmax, o); wmax = v > wmax ? v : wmax; }

    /* per-lane role: cascade (channel, low/high) and stage */
    int ch, is_hi;
    bool active;
    if (NARROW) { ch = cidx == 0 ? 0 : cidx == 1 ? 0 : cidx == 2 ? 1 : 2; is_hi = cidx != 0; active = true; }
    else if ((lrow & 1) == 0) { ch = cidx >> 1; is_hi = cidx & 1; active = true; }
    else { ch = 2; is_hi = cidx & 1; active = cidx < 2; }
    const int coef = is_hi ? (ch == 0 ? P.eq_hf[0] : ch == 1 ? P.eq_hf[1] : P.eq_hf[2])
                           : (ch == 0 ? P.eq_lf[0] : ch == 1 ? P.eq_lf[1] : P.eq_lf[2]);
    const int *my_u = &s_u[ls][ch][0];
    const bool writes = active && bank == 3;
    /* every lane stores after every step (no divergent control flow inside the systolic loop); only the last stage of a
     * live cascade stores into its ring, everybody else into the sink */
    int *my_c = writes ? &s_c[ls][NARROW ? cidx : ch * 2 + is_hi][0] : &s_sink[lane];
    const int c_half = writes ? TS : 0;                 /* the sink has no halves */

    /* sample window -> LDS (16-byte pieces, LPL lanes per scanline) */
    {
        const signed char *src = inp + (size_t) f * fstride + lp.pos + first;
        for (int off = (lane & (LPL - 1)) * 16; off < W; off += LPL * 16)
            *(v4i *) &s_inp[ls][off] = load16u(src + off);
    }
    /* demodulation carriers of the line: crt_core.c:476-479 (wave[], Q reads wave[(i + 3) & 3]) / :497-505 */
    if ((lane & (LPL - 1)) == 0) {
        if constexpr (CCS == 4) {
            s_wave[ls][0][0] = lp.wave0; s_wave[ls][0][1] = lp.wave1; s_wave[ls][0][2] = -lp.wave0; s_wave[ls][0][3] = -lp.wave1;
            s_wave[ls][1][0] = -lp.wave1; s_wave[ls][1][1] = lp.wave0; s_wave[ls][1][2] = lp.wave1; s_wave[ls][1][3] = -lp.wave0;
        } else {
#pragma unroll
            for (int i = 0; i < CCS; i++) {
                s_wave[ls][0][i] = ((lp.wave0 * P.dem_cs[0][i] + lp.wave1 * P.dem_sn[0][i]) >> 15) * P.saturation;
                s_wave[ls][1][i] = ((lp.wave0 * P.dem_cs[1][i] + lp.wave1 * P.dem_sn[1][i]) >> 15) * P.saturation;
            }
        }
    }
    wave_lds_fence();

    /* In the wave-wide phases (prep, combine, pixels) a scanline keeps its LPL lanes: lane jl of the scanline handles
     * samples / pixels jl, jl + LPL, ... of it, so every per-scanline quantity is simply the lane's own copy. */
    const int jl = lane & (LPL - 1);
    constexpr int SPL = TS >= LPL ? TS / LPL : 1;        /* samples per lane and tile in prep / combine */
    const bool jl_on = jl * SPL < TS;                    /* (a 32-lane scanline has idle lanes when the tile is 16 samples) */
    constexpr int bpp = BPP3 ? 3 : 4;
    const size_t pitch = (size_t) P.outw * bpp;
    const unsigned long long dst_row = (unsigned long long) (outp + (size_t) f * ostride + (size_t) lp.beg * pitch);
    const unsigned slot_shift = (unsigned) (ls * LPL);
    const unsigned long long slot_mask = LPL == 32 ? 0xffffffffull : 0xffffull;
    int pxn = 0;                                         /* next pixel of my scanline (the same in all its lanes) */

    const int bright = P.bright, contrast = P.contrast;
    const unsigned psel = P.out_format == CRTHIP_FMT_BGRA ? 0x03020100u : P.out_format == CRTHIP_FMT_RGBA ? 0x03000102u
                        : P.out_format == CRTHIP_FMT_ARGB ? 0x00010203u : 0x02010003u /* ABGR */;
    const unsigned usel = P.out_format == CRTHIP_FMT_BGRA ? 0x03020100u : P.out_format == CRTHIP_FMT_RGBA ? 0x03000102u
                        : P.out_format == CRTHIP_FMT_ARGB ? 0x00010203u : 0x00030201u /* ABGR */;
    const bool rgb_order = P.out_format == CRTHIP_FMT_RGB;
    const bool blend = P.blend != 0;
    const unsigned scanr = (unsigned) (S::AV_LEN - 1) << 12;
    const int outw = P.outw;
    const unsigned dx = (unsigned) lp.dx;
    /* band gains, crt_core.c:272-286: compile-time per CC_SAMPLES (the host refuses a blob that says otherwise), so
     * the gains 65536 and 8192 turn into bit-field extracts */
    constexpr int GY1 = CCS == 5 ? 12192 : 8192, GY2 = CCS == 5 ? 7775 : 9175, GI2 = 1311;

    /* the demodulation carrier of my samples: with 4 samples per cycle the phase of sample t0 + jl * SPL + k does not
     * depend on the tile (TS is a multiple of 4), so it is a per-lane constant */
    int wIk[SPL], wQk[SPL];
#pragma unroll
    for (int k = 0; k < SPL; k++) {
        const int ph = (first + jl * SPL + k) & 3;
        wIk[k] = CCS == 4 ? s_wave[ls][0][ph] : 0;
        wQk[k] = CCS == 4 ? s_wave[ls][1][ph] : 0;
    }

    int xs = 0;                                          /* my stage's state */
    int cdone = 0;                                       /* samples combined so far (local index) */
    for (int t0 = 0; t0 < wmax + 3; t0 += TS) {
        /* ---- prep: filter inputs of samples [t0, t0 + TS) of my scanline, all three channels ---- */
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            if (!jl_on) break;
            const int xl = t0 + jl * SPL + k;
            const int sm = xl < WINB ? s_inp[ls][xl] : 0;
            int wi = wIk[k], wq = wQk[k];
            if constexpr (CCS != 4) {
                const int ph = (first + xl) % CCS;
                wi = s_wave[ls][0][ph];
                wq = s_wave[ls][1][ph];
            }
            s_u[ls][0][xl & (RING - 1)] = sm + bright;                     /* crt_core.c:540 */
            s_u[ls][1][xl & (RING - 1)] = (sm * wi) >> 9;                  /* :541 */
            s_u[ls][2][xl & (RING - 1)] = (sm * wq) >> 9;                  /* :542 */
        }
        wave_lds_fence();
        /* ---- filter: TS systolic steps ---- */
        {
            const int hb = t0 & TS;                      /* which half of the ring this tile uses */
            int ub[TS];                                  /* the tile's inputs up front: no LDS latency inside the chain */
#pragma unroll
            for (int k = 0; k < TS; k++) ub[k] = my_u[hb + k];
            int *cdst = my_c + ((t0 & TS) ? c_half : 0);
#pragma unroll
            for (int k = 0; k < TS; k++) {
                const int tin = __builtin_amdgcn_update_dpp(ub[k], xs, DPP_ROW_SHR4, 0xf, 0xf, false);
                xs += (int) ((unsigned) coef * (unsigned) (tin - xs) + 32768u) >> 16;     /* crt_core.c:211-217 */
                cdst[k] = xs;                            /* bank 3 at step t holds the cascade output of sample t - 3 */
            }
        }
        wave_lds_fence();
        /* ---- combine: band gains, crt_core.c:218-232 ---- */
        const int xready = t0 + TS - 3;                  /* cascade outputs exist for samples < xready */
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            const int x = cdone + jl * SPL + k;
            if (x < xready && jl_on) {
                const int sl = (x + 3) & (RING - 1), sp = (x - 3) & (RING - 1);
                const int ylo = s_c[ls][0][sl], yhi = s_c[ls][1][sl], ihi = s_c[ls][C_IHI][sl], qhi = s_c[ls][C_QHI][sl];
                const int uy = x >= 3 ? s_u[ls][0][sp] : 0;
                const int ui = x >= 3 ? s_u[ls][1][sp] : 0;
                int yv = ((ylo * 65536) >> 16) + (((yhi - ylo) * GY1) >> 16) + (((uy - yhi) * GY2) >> 16);
                int iv, qv;
                if (NARROW) {
                    /* gains (65536, 65536, g2) and |lo3|, |hi3 - lo3| < 2^15: low + mid band == hi3 */
                    iv = ihi + (((ui - ihi) * GI2) >> 16);
                    qv = qhi;                                              /* Q: high-band gain 0 */
                } else {
                    const int ilo = s_c[ls][NARROW ? 0 : 2][sl], qlo = s_c[ls][NARROW ? 0 : 4][sl];
                    iv = ((ilo * 65536) >> 16) + (((ihi - ilo) * 65536) >> 16) + (((ui - ihi) * GI2) >> 16);
                    qv = ((qlo * 65536) >> 16) + (((qhi - qlo) * 65536) >> 16);
                }
                s_yiq[ls][0][x & (RING - 1)] = yv << 4;   /* crt_core.c:540-542 */
                s_yiq[ls][1][x & (RING - 1)] = iv >> 3;
                s_yiq[ls][2][x & (RING - 1)] = qv >> 3;
            }
        }
        cdone = xready > 0 ? xready : 0;
        wave_lds_fence();
        /* ---- pixels: D9 / D10, crt_core.c:552-664, every pixel whose right tap exists now; LPL pixels of every
         *      scanline of the wave per pass ---- */
        {
            const int xlim = cdone < W ? cdone : W;                      /* local samples [0, xlim) are in the ring */
            const bool complete = cdone >= W;
            for (;;) {
                const int px = pxn + jl;
                const unsigned upos = (unsigned) lp.scanl + (unsigned) px * dx;
                const int sa = (int) (upos >> 12) - first;                 /* local index of the left tap */
                /* right tap: inside the ring, or -- bloom only -- the never-written entry AV_LEN - 1 (zero) */
                const bool tail0 = bloom && complete && sa + 1 == W;
                const bool ok = nrows > 0 && px < outw && upos < scanr && (int) dx > 0 && sa >= 0 && (sa + 1 < xlim || tail0);
                const unsigned long long m = __ballot(ok);
                if (m == 0ull) break;
                if (ok) {
                    const int R = (int) (upos & 0xfffu), L = 0xfff - R;
                    const int ia = sa & (RING - 1), ib = (sa + 1) & (RING - 1);
                    const int ay = s_yiq[ls][0][ia], ai = s_yiq[ls][1][ia], aq = s_yiq[ls][2][ia];
                    const int by = tail0 ? 0 : s_yiq[ls][0][ib], bi = tail0 ? 0 : s_yiq[ls][1][ib], bq = tail0 ? 0 : s_yiq[ls][2][ib];
                    const int yy = ((ay * L) >> 2) + ((by * R) >> 2);
                    const int ii = ((ai * L) >> 14) + ((bi * R) >> 14);
                    const int qq = ((aq * L) >> 14) + ((bq * R) >> 14);
                    int r = (((yy + 3879 * ii + 2556 * qq) >> 12) * contrast) >> 8;
                    int g = (((yy - 1126 * ii - 2605 * qq) >> 12) * contrast) >> 8;
                    int b = (((yy - 4530 * ii + 7021 * qq) >> 12) * contrast) >> 8;
                    r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                    unsigned rgb = (unsigned) (((r << 8 | g) << 8) | b);
                    const unsigned long long d = dst_row + (size_t) px * bpp;
                    if (!BPP3) {
                        if (blend) {
                            const unsigned oldw = gload32(d);
                            const unsigned old = __builtin_amdgcn_perm(oldw, oldw, usel);
                            rgb = ((rgb & 0xfefeffu) >> 1) + ((old & 0xfefeffu) >> 1);
                        }
                        const unsigned full = 0xff000000u | rgb;
                        const unsigned o = __builtin_amdgcn_perm(full, full, psel);
                        for (int dup = 0; dup < nrows; dup++) gstore32(d + (size_t) dup * pitch, o);
                    } else {
                        if (blend) {
                            const unsigned o0 = gload8(d), o1 = gload8(d + 1), o2 = gload8(d + 2);
                            const unsigned old = rgb_order ? (o0 << 16 | o1 << 8 | o2) : (o2 << 16 | o1 << 8 | o0);
                            rgb = ((rgb & 0xfefeffu) >> 1) + ((old & 0xfefeffu) >> 1);
                        }
                        const unsigned c0 = rgb_order ? rgb >> 16 : rgb, c2 = rgb_order ? rgb : rgb >> 16;
                        for (int dup = 0; dup < nrows; dup++) {
                            const unsigned long long dd = d + (size_t) dup * pitch;
                            gstore8(dd, c0); gstore8(dd + 1, rgb >> 8); gstore8(dd + 2, c2);
                        }
                    }
                }
                /* my scanline's share of the pass (its pixels are a prefix of its lanes: upos grows with px) */
                pxn += __popcll((m >> slot_shift) & slot_mask);
            }
        }
        wave_lds_fence();
    }
}

/* Launch the scanline-parallel decoder: both instantiations (groups of 4 scanlines pick theirs), one pass per rank. */
int crt_run_decode_rows(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                        const crthip_line *d_lines, void *d_out, size_t ostride, size_t fstride)
{
    if (p->eq_kernel)
        return set_err(c, CRTHIP_E_ARG, "the FIR decoder (USE_CONVOLUTION build) has no scanline-parallel kernel", hipSuccess);
    /* the NARROW variant relies on chroma gains (65536, 65536, x) and coefficients below 2^15 (tier 0) */
    const bool narrow_ok = !c->force_exact && !c->no_loskip && p->eq_g[1][0] == 65536 && p->eq_g[1][1] == 65536 &&
                           p->eq_g[2][0] == 65536 && p->eq_g[2][1] == 65536 && p->eq_lf[1] > 0 && p->eq_lf[1] < 32768 &&
                           p->eq_hf[1] > 0 && p->eq_hf[1] < 32768 && p->eq_lf[2] > 0 && p->eq_lf[2] < 32768 &&
                           p->eq_hf[2] > 0 && p->eq_hf[2] < 32768;
    {
        /* the kernel has the band gains of crt_core.c:272-286 compiled in */
        const int five = c->sd.cc_samples == 5;
        const int want[3][3] = { { 65536, five ? 12192 : 8192, five ? 7775 : 9175 }, { 65536, 65536, 1311 }, { 65536, 65536, 0 } };
        for (int k = 0; k < 3; k++)
            for (int b = 0; b < 3; b++)
                if (p->eq_g[k][b] != want[k][b])
                    return set_err(c, CRTHIP_E_ARG, "equaliser gains differ from crt_core.c:272-286", hipSuccess);
    }
    const unsigned span = (unsigned) p->outh + p->v_fac;
    const int passes = span >= (unsigned) c->sd.lines ? 1 : (int) (((unsigned) c->sd.lines + span - 1) / (span ? span : 1));
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        const int total = n * S::LINES;
        unsigned char *o = (unsigned char *) d_out;
        ProfScope ps(c, CRTHIP_K_DECODE);
        for (int rank = 0; rank < passes; rank++) {
#define CRTHIP_LAUNCH_ROWS_T(B3, TSV) \
    do { if (narrow_ok) hipLaunchKernelGGL((k_decode_row<S, true, B3, TSV>), dim3((total + 3) / 4), dim3(64), 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, rank, 0); \
         hipLaunchKernelGGL((k_decode_row<S, false, B3, TSV>), dim3((total + 1) / 2), dim3(64), 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, rank, narrow_ok ? 0 : 1); } while (0)
            /* tile of 16 samples: 9 KB of LDS per wave instead of 14 (4 waves per SIMD instead of 2.75) but twice the
             * per-tile overheads: better once the launch has a few waves per SIMD (measured: batch 64
             * -22 %, batch 16 and batch 1 +10 %, profiles/r02_shape_sweep.txt); CRTHIP_ROW_TILE=16|32 overrides */
            if (c->row_tile == 16 || (c->row_tile == 0 && n >= 48)) { if (p->out_bpp == 3) CRTHIP_LAUNCH_ROWS_T(true, 16); else CRTHIP_LAUNCH_ROWS_T(false, 16); }
            else { if (p->out_bpp == 3) CRTHIP_LAUNCH_ROWS_T(true, 32); else CRTHIP_LAUNCH_ROWS_T(false, 32); }
#undef CRTHIP_LAUNCH_ROWS_T
        }
        return CRTHIP_OK;
    });
}
