/* crt_decode4.hip -- D8-D10 for WIDE pictures: 16 scanlines per wavefront, the four filter cascades of a scanline on four
 * lanes, pixels emitted lane-per-pixel in 1 KB runs of one picture row.  See crt_dev.h / crt_decode_lane.h for the
 * lane-per-scanline decoder (the throughput shape at 640x480) and crt_decode2.hip for the scanline-parallel one (small batches).
 *
 * Why a third shape (round 4; profiles/r04_store_patterns.txt, r04_experiments.txt section 9).  At 1920x1080 the decoder is bound
 * by its picture stores -- 6.5 MB per field, rows duplicated 3-4 times (crt_core.c:661-664).  A lane-per-scanline wave owns 64
 * scanlines = 288 picture rows and can only buffer 32 pixels of each before it has to store: 128-byte runs, for which the memory
 * system gives 5.2 TB/s; for 1 KB runs it gives 5.85.  Emitting a 256-pixel run needs ~105 samples of y/i/q of that scanline at
 * once, i.e. 600-800 bytes of LDS per scanline -- 38-51 KB for 64 scanlines.  So this kernel takes 16 scanlines per wave and gets its
 * 64 lanes busy in the filter stage by giving every scanline FOUR lanes, one per cascade that tiers 0 / 1 run (luma low, luma
 * high, I high, Q high: crt_decode_lane.h, eq_step64_yiq -- four independent chains of four one-pole stages): a lane runs 12
 * stage instructions per sample instead of 48, the band sums (crt_core.c:218-232) are formed with one quad DPP exchange, and
 * every lane files its result into the scanline's rings in LDS with one 16-bit store at its own offset.  The pixel stage then
 * walks the 16 scanlines one after the other, lane = four consecutive pixels: the two taps from the rings (offsets and weights
 * computed once per run: they do not depend on the scanline), the same packed-chroma / v_dot2 / 64-bit-mad arithmetic as the
 * lane-per-scanline decoder (crt_core.c:555-562), one 16-byte store per lane = 1 KB of ONE row per instruction, repeated for the
 * duplicated rows.
 *
 * Measured (profiles/r04_experiments.txt, section 10; 1920x1080 x 2048): 2.6-2.85 ms against 2.95-3.1 for the lane-per-scanline
 * kernel on the same box.  The kernel costs its stores -- 2.46 ms with everything else cut out, 5.4 TB/s -- plus the part of its
 * arithmetic (1.6 ms alone) that does not hide under them; neither fewer instructions nor more waves moved it further.
 *
 * Scope: the 4-samples-per-cycle systems, tiers 0 / 1 (what a wide picture at ordinary knobs runs in), 4-byte pixel formats, no
 * blend, no bloom, a resampler step small enough for the ring (outw >= ~1650).  Everything else stays with k_decode; the tier of a
 * 64-scanline group is decided exactly as there, so the two kernels partition the batch between them.  Same arithmetic, bit for
 * bit (tests/test_gpu_parity.py: every 1080p case runs through here).
 */
#include "crt_decode_lane.h"
#include <type_traits>

#define WIDE_LPW   16                 /* scanlines per wave (throughput); small batches: 8, see k_decode_wide */
#define WIDE_RING  128                /* samples per scanline in the y/i/q ring */
#define WIDE_PXT   256                /* pixels per row run: 64 lanes x 4 */

/* a 16-bit LDS store by the lanes of `lanes` only (the others keep their registers and the LDS pipe free); the wave's stores and
 * loads reach the LDS in program order, so the compiler not counting this one only makes its waits conservative */
__device__ __forceinline__ void lds_store16_lanes(unsigned addr, int v, unsigned long long lanes)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tds_write_b16 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(addr), "v"(v), "s"(lanes) : "memory", "scc");      /* s_and_b64 writes SCC */
}

/* v_mad_i64_i32 with a per-lane multiplier */
__device__ __forceinline__ long mad64_vv(int d, int m, long acc)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(d), "v"(m), "v"(acc));
    return r;
}

/* LPW = scanlines per wave.  16 fills the filter stage's 64 lanes (four cascades per scanline).  8 leaves half of them idle there
 * but makes twice as many waves, each half as long: a batch of 64 fields of 1920x1080 is 960 waves of 16 scanlines on 1024 SIMDs --
 * less than one wave per SIMD, each walking its 16 x 1920 pixels alone -- and 1920 waves of 8 (VERDICT round 4, item 3b;
 * profiles/r05_experiments.txt section 6). */
template <class S, int LPW>
__global__ void __launch_bounds__(64)
k_decode_wide(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
              const crthip_line *__restrict__ lines, unsigned char *__restrict__ outp, size_t ostride, int min_tier,
              int want_rank, int n_px, unsigned ppos_end, int order_k, int order_per)
{
    static_assert(S::CCS == 4, "tiers 0 / 1 of the 4-samples-per-cycle systems");
    static_assert(LPW == 16 || LPW == 8, "scanlines per wave");
    constexpr int RING = WIDE_RING;
    constexpr int IN_DW = 16, IN_STRIDE = IN_DW + 1;      /* (IN_STRIDE: the allocation; rows are addressed by TileRows, crt_dev.h) */
    using TIN = TileRows<IN_DW>;
    /* LDS, 13.5 KB (12 waves per CU): the { q, i } ring -- one dword per sample --, the luma ring -- 16-bit values, scanlines l and
     * l + 8 sharing a dword so that both rings are addressed by sample * 4 -- and the input tile (64 samples per scanline).
     * A scanline's ring is 129 dwords apart from the next and the luma ring starts 16 banks after the chroma ring: the filter stage
     * stores sample x of all 16 scanlines with ONE instruction, and at a stride of 128 dwords its 32-lane halves met on a single bank
     * (SQ_LDS_BANK_CONFLICT: 80 % of all LDS cycles, the filter stage alone 1.35 ms at 1080p x 2048 -- profiles/r04_experiments.txt 10).
     * What the counter still shows (121 M of 308 M LDS cycles per launch, profiles/r06_1080p_sq_counters.json) is by design: the I and
     * the Q lane of a scanline file the two halves of ONE dword (two 16-bit stores to one bank; a single store would cost two more
     * vector instructions per sample and lane), and the pixel stage's taps are a gather -- lane l reads sample 1.57 l of the ring, so
     * lanes 20 apart meet on a bank (2-way; 16 reads beside 110 vector instructions per scanline and run).  Neither is on the
     * kernel's critical resource (its picture stores); the input tile's rows are conflict-free since round 6 (TileRows). */
    constexpr int RSTRIDE = RING + 1;
    constexpr int OFF_IQ = 0, OFF_Y = LPW * RSTRIDE * 4 + (LPW == 8 ? 32 : 0), OFF_IN = OFF_Y + 8 * RSTRIDE * 4 + (LPW == 8 ? 32 : 0);
    static_assert((OFF_Y / 4) % 32 == 16 && OFF_IN % 16 == 0, "bank offset of the luma ring, alignment of the input tile");
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[OFF_IN + LPW * IN_STRIDE * 4];
    unsigned *const s_in = (unsigned *) (s_mem + OFF_IN);

    const int lane = threadIdx.x, l = lane >> 2, c = lane & 3; /* my scanline of the wave, my cascade */
    /* which LPW scanlines: crt_dev.h, block_item (field-interleaved by default: the waves resident together then write rows of
     * different pictures -- 1080p x 2048: 2.70 -> 2.35 ms, profiles/r06_1080p_placement.txt) */
    static_assert(S::LINES % LPW == 0, "scanline groups do not straddle fields");
    const int bid = block_item(blockIdx.x, order_k, order_per);
    if (bid * LPW >= n_fields * S::LINES) return;
    const bool has_line = l < LPW;                             /* (LPW 8: the upper half-wave has no scanline of its own; it emits pixels) */
    const int total = n_fields * S::LINES;
    int tier;
    {
        /* the tier of the 64-scanline group my scanlines belong to: the decision of k_decode, flag for flag; tiers 0 and 1 are
         * both decoded here (they differ in ONE instruction of the filter stage, chosen per wave), the groups of tiers 2 / 3 are
         * k_decode's */
        const int g = (bid * LPW) / 64 * 64 + lane;
        int fl = 0;
        if (g < total) fl = lines[g].nrows;
        tier = __ballot(fl & CRTHIP_LINE_EXACT) ? 3 : __ballot(fl & CRTHIP_LINE_NOT64) ? 2 : __ballot(fl & CRTHIP_LINE_WIDE) ? 1 : 0;
        if (tier < 2 && __ballot(fl & (int) CRTHIP_LINE_KEEPLO) != 0ull) tier = 2;
        if (tier < min_tier) tier = min_tier;
        if (tier > 1) return;
    }
    const int gl = bid * LPW + l;
    const bool live = has_line && gl < total;
    crthip_line lp;
    lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0; lp.hsync = 0; lp.dx = 0; lp.scanl = 0;
    if (live) lp = lines[gl];
    int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    if (!live || ((lp.nrows >> CRTHIP_LINE_RANK_SHIFT) & CRTHIP_LINE_RANK_MASK) != want_rank) nrows = 0;
    if (__ballot(nrows > 0) == 0ull) return;
    const int f = live ? gl / S::LINES : 0;
    const size_t pitch = (size_t) P.outw * 4;
    const unsigned long long src = (unsigned long long) (inp + (size_t) f * fstride + (nrows > 0 ? lp.pos : 0));
    const unsigned long long dst = (unsigned long long) (outp + (size_t) f * ostride + (size_t) (nrows > 0 ? lp.beg : 0) * pitch);

    /* my cascade: input multiplier by sample phase (luma: 2^16, i.e. the sample itself; chroma: the demodulation carriers << 7,
     * crt_core.c:476-479, 541-542), input offset, stage multiplier and form (eq_step64_yiq: coefficients >= 2^15 take x' = u + ...),
     * top-band gain, output shift */
    const int w0 = lp.wave0 * 128, w1 = lp.wave1 * 128;
    int wk[4];
    wk[0] = c < 2 ? 65536 : c == 2 ? w0 : -w1;
    wk[1] = c < 2 ? 65536 : c == 2 ? w1 : w0;
    wk[2] = c < 2 ? 65536 : c == 2 ? -w0 : w1;
    wk[3] = c < 2 ? 65536 : c == 2 ? -w1 : -w0;
    const int bl = c < 2 ? P.bright : 0;
    const bool near1 = c < 2;
    const int M = c == 0 ? (P.eq_lf[0] - 65536) * 65536 : c == 1 ? (P.eq_hf[0] - 65536) * 65536 : c == 2 ? P.eq_hf[1] * 65536 : P.eq_hf[2] * 65536;
    const int g2 = c == 1 ? 9175 : c == 2 ? 1311 : 0;          /* crt_core.c:272-286 (the host refuses other gains) */
    const int sh = c == 1 ? 0 : 3;                             /* luma stays unshifted (see k_decode, D9), chroma >> 3 */
    /* where my result goes: the luma-high lane files y, the chroma lanes their half of the { q, i } dword; the luma-low lane's
     * output only feeds its neighbour (it sits out the store) */
    const unsigned wbase = c == 1 ? OFF_Y + (l & 7) * (RSTRIDE * 4) + (l >> 3) * 2 : OFF_IQ + l * (RSTRIDE * 4) + (c == 2 ? 2 : 0);

    int x0 = 0, x1 = 0, x2 = 0, x3 = 0;                        /* my four stages */
    int h0 = 0, h1 = 0, h2 = 0;                                /* my input history (top band, crt_core.c:229-231) */

    const unsigned dx = (unsigned) P.dx;
    const int contrast12 = P.contrast * 4096;
    long alpha_pair = (long) 0xff00ul << 32;
    asm volatile("" : "+v"(alpha_pair));
    const unsigned psel = pack_selector(P.out_format);

    constexpr int NQ = (S::AV_LEN + 3) / 4, NT = (NQ + IN_DW - 1) / IN_DW;
    v4i nxt = gload16u(src + c * 16);                          /* input tile 0: lane (l, c) moves piece c of scanline l */
    int have_tile = -1;
    int xq = 0;                                                /* next dword of samples to filter (wave-uniform) */

    for (int px0 = 0; px0 < n_px; px0 += WIDE_PXT) {
        /* ---- filter: every sample the run's pixels need (taps idx and idx + 1, crt_core.c:555-558) ---- */
        const int pxl = px0 + WIDE_PXT - 1 < n_px - 1 ? px0 + WIDE_PXT - 1 : n_px - 1;
        const int x_need = (int) (((unsigned) pxl * dx) >> 12) + 2;
        auto filter_run = [&](auto tier_tag) {
        constexpr int TIER = decltype(tier_tag)::value;
        while (xq * 4 < x_need && xq < NQ) {
            const int t = xq >> 4;
            if (t != have_tile) {
                wave_lds_fence();
                if (has_line) {
                    unsigned *d = s_in + TIN::row(l) + c * 4;
                    d[0] = (unsigned) nxt.x; d[1] = (unsigned) nxt.y; d[2] = (unsigned) nxt.z; d[3] = (unsigned) nxt.w;
                }
                wave_lds_fence();
                have_tile = t;
                if (t + 1 < NT) nxt = gload16u(src + (t + 1) * (IN_DW * 4) + c * 16);
            }
            const int word = (int) s_in[TIN::row(has_line ? l : 0) + (xq & (IN_DW - 1))];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = xq * 4 + k;
                const int s = (word << (24 - 8 * k)) >> 24;
                /* my cascade's input: s + bright | (s * wave) >> 9, as the high word of the product with the pre-scaled multiplier */
                const int ut = TIER == 0 ? __mul24(s, wk[k]) : mul_lo_mad64(s, wk[k]);      /* tier 1: carriers << 7 beyond 24 bits */
                const int u = add_hiword(bl, ut);
                /* four stages, eq_step64_yiq: x' = hi32(M * (in - x) + {2^31, near1 ? in : x}) */
#define WIDE_STAGE(X, IN) X = hi32(mad64_vv((IN) - X, M, pair_of(near1 ? (IN) : X)))
                WIDE_STAGE(x0, u);
                WIDE_STAGE(x1, x0);
                WIDE_STAGE(x2, x1);
                WIDE_STAGE(x3, x2);
#undef WIDE_STAGE
                /* band sums (crt_core.c:218-232; eq_step64_yiq): top band on my own history; the luma-high lane
                 * takes the luma-low lane's output from its left neighbour in the quad, the others take themselves (difference 0) */
                const int r = x3 + (__mul24(h2 - x3, g2) >> 16);
                h2 = h1; h1 = h0; h0 = u;
                const int nb = __builtin_amdgcn_mov_dpp(x3, 0xe0 /* quad_perm:[0,0,2,3] */, 0xf, 0xf, true);
                const int tt = x3 - nb;
                const int out = (r - tt + (tt >> 3)) >> sh;
                lds_store16_lanes(wbase + (unsigned) (x & (RING - 1)) * 4u, out, LPW == 16 ? 0xeeeeeeeeeeeeeeeeull : 0x00000000eeeeeeeeull);
            }
            xq++;
        }
        };
        if (tier == 0) filter_run(std::integral_constant<int, 0>{}); else filter_run(std::integral_constant<int, 1>{});
        wave_lds_fence();
        /* ---- pixels: scanline after scanline, lane = four consecutive pixels of the run ---- */
        const int px = px0 + 4 * lane;
        /* what does not depend on the scanline: the taps' ring offsets and weights (scaled by 4, as in k_decode) and how many of
         * my pixels exist (px * dx < ppos_end  <=>  px < n_px) */
        unsigned o0[4], o1[4];
        int R4[4], L4[4];
        {
            unsigned ppos = __umul24((unsigned) px, dx);       /* px < 2^24, dx < 2^24 (host-checked) */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                o0[j] = (ppos >> 10) & ((RING - 1) << 2);
                o1[j] = ((ppos + 4096u) >> 10) & ((RING - 1) << 2);
                R4[j] = (int) ((ppos & 0xfffu) << 2);
                L4[j] = 0x3ffc - R4[j];
                ppos += dx;
            }
        }
        const int have = n_px - px;
        if (have > 0) {
#pragma unroll
            for (int ll = 0; ll < LPW; ll++) {
                const int nr = __builtin_amdgcn_readlane(nrows, 4 * ll);
                if (nr == 0) continue;
                const unsigned dlo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) dst, 4 * ll);
                const unsigned dhi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (dst >> 32), 4 * ll);
                const unsigned long long drow = ((unsigned long long) dhi << 32 | dlo) + (unsigned long long) px * 4;
                const unsigned char *const ry = s_mem + (OFF_Y + (ll & 7) * (RSTRIDE * 4) + (ll >> 3) * 2);
                const unsigned char *const riq = s_mem + (OFF_IQ + ll * (RSTRIDE * 4));
                unsigned v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int py = *(const short *) (ry + o0[j]), cy = *(const short *) (ry + o1[j]);
                    const unsigned e0 = *(const unsigned *) (riq + o0[j]), e1 = *(const unsigned *) (riq + o1[j]);
                    const int pq = (int) (short) e0, pi = (int) e0 >> 16, cq = (int) (short) e1, ci = (int) e1 >> 16;
                    /* crt_core.c:556-562 exactly as in k_decode (tiers 0 / 1): weights scaled by 4, chroma packed, one v_dot2 per
                     * colour row, contrast as a pre-shifted 64-bit multiply-add with the alpha riding on the red row */
                    const int yy = __mul24(cy, R4[j]) + __mul24(py, L4[j]);
                    int iq = add_hiwords(__mul24(pq, L4[j]), __mul24(cq, R4[j]));
                    iq = add_hiwords_to_hi(iq, __mul24(pi, L4[j]), __mul24(ci, R4[j]));
                    const int vr = dot2_vs(iq, (3879 << 16) | 2556, yy);
                    const int vg = dot2_vs(iq, (int) (((unsigned) -1126 << 16) | ((unsigned) -2605 & 0xffffu)), yy);
                    const int vb = dot2_vs(iq, (int) (((unsigned) -4530 << 16) | 7021u), yy);
                    int r8 = pair_hi(mad64_vs(vr & ~0xfff, contrast12, alpha_pair));
                    int g8 = pair_hi(mad64_vs0(vg & ~0xfff, contrast12));
                    int b8 = pair_hi(mad64_vs0(vb & ~0xfff, contrast12));
                    r8 = clampi(r8, 0xff00, 0xffff); g8 = clampi(g8, 0, 255); b8 = clampi(b8, 0, 255);
                    const unsigned rgb = lshl_or(lshl_or((unsigned) r8, 8, (unsigned) g8), 8, (unsigned) b8);   /* 0xffRRGGBB */
                    v[j] = __builtin_amdgcn_perm(rgb, rgb, psel);                  /* (the identity for the native order) */
                }
                for (int dup = 0; dup < nr; dup++) {
                    const unsigned long long dd = drow + (size_t) dup * pitch;
                    if (have >= 4) {
                        v4i o; o.x = (int) v[0]; o.y = (int) v[1]; o.z = (int) v[2]; o.w = (int) v[3];
                        gstore16u_nt(dd, o);
                    } else {
                        gstore32(dd, v[0]);
                        if (have > 1) gstore32(dd + 4, v[1]);
                        if (have > 2) gstore32(dd + 8, v[2]);
                    }
                }
            }
        }
        wave_lds_fence();
    }
}

/* the configurations this kernel takes (everything else: k_decode) */
bool crt_decode_wide_ok(const crthip_ctx *c, const crthip_params *p, int min_tier, bool wide)
{
    if (!c->wide_decode || !wide || c->sd.cc_samples != 4 || p->out_bpp != 4 || p->blend || p->bloom || p->eq_kernel || min_tier > 1) return false;
    if (p->dx <= 0 || p->dx >= (1 << 24) || p->outw >= (1 << 22)) return false;
    /* a run of 256 pixels must fit the ring with the filter's look-ahead: taps up to ((255 dx) >> 12) + 1 samples apart + 3 */
    return ((255ll * p->dx) >> 12) + 8 <= WIDE_RING;
}

int crt_run_decode_wide(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, const crthip_line *d_lines,
                        void *d_out, size_t ostride, int min_tier, int rank, size_t fstride)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        if constexpr (S::CCS == 4) {
            const int total = n * S::LINES;
            /* 8 scanlines per wave while 16 would leave SIMDs without a wave of their own twice over (see the kernel);
             * CRTHIP_WIDE_LPW=8|16 pins it (A/B) */
            const int lpw = c->wide_lpw_env ? c->wide_lpw_env : (total / WIDE_LPW < WIDE_LPW8_MAX_WAVES ? 8 : 16);
            const dim3 block(64);
            /* pixels the reference emits per scanline: px * dx < min(dx * outw, (AV_LEN - 1) << 12)  (crt_core.c:528-531, 555) */
            const unsigned long long all = (unsigned long long) (unsigned) p->dx * (unsigned) p->outw;
            const unsigned scan_r = (unsigned) (S::AV_LEN - 1) << 12;
            const unsigned ppos_end = all < scan_r ? (unsigned) all : scan_r;
            const int n_px = (int) ((ppos_end + (unsigned) p->dx - 1) / (unsigned) p->dx);
            /* workgroup order: one stride per field unless CRTHIP_WIDE_ORDER says otherwise (1 = in order, K = that many strides) */
            const int nblk = (total + lpw - 1) / lpw;
            const block_order bo = make_block_order(nblk, c->wide_order_env == 0 || c->wide_order_env == -1 ? n : c->wide_order_env);
            const dim3 ogrid(bo.grid);
            if (lpw == 8)
                hipLaunchKernelGGL((k_decode_wide<S, 8>), ogrid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, (unsigned char *) d_out, ostride,
                                   min_tier, rank, n_px, ppos_end, bo.K, bo.per);
            else
                hipLaunchKernelGGL((k_decode_wide<S, 16>), ogrid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, (unsigned char *) d_out, ostride,
                                   min_tier, rank, n_px, ppos_end, bo.K, bo.per);
        }
        return CRTHIP_OK;
    });
}
