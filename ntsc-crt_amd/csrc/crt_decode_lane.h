/* crt_decode_lane.h -- D8-D10: equalisers, resampling, YIQ->RGB, row duplication: the lane-per-scanline decoder kernel.
 * Instantiated by crt_decode.hip (fixed resampler geometry) and crt_decode3.hip (CRT_DO_BLOOM: per-scanline geometry).
 * See crt_dev.h. */
#ifndef CRT_DECODE_LANE_H
#define CRT_DECODE_LANE_H
#include "crt_dev.h"
#include <type_traits>

/* ------------------------------------------------------------------------- */
/* D8-D10: equalisers + resample + YIQ->RGB, one lane per CRT line              */
/* ------------------------------------------------------------------------- */
/*
 * Multiplies.  The reference multiplies 32x32->32 with wrap-around.  gfx950's
 * v_mul_lo_u32 does exactly that but runs at quarter rate; v_mul_i32_i24 /
 * v_mad_i32_i24 run at full rate and return the low 32 bits of the 48-bit product
 * of the operands' low 24 bits (sign-extended) -- identical to the wrapped 32-bit
 * product WHENEVER both operands are within [-2^23, 2^23).  FAST=true uses them and
 * is only dispatched when that range is proven (see fast_path_ok() below and
 * DESIGN.md "24-bit multiply envelope"); lines outside the envelope are flagged by
 * k_sync (CRTHIP_LINE_EXACT) and re-run by the FAST=false instantiation.
 */

struct Eq3 { int lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3, h0, h1, h2; };

/* eqf, crt_core.c:206-233.  Band gains are the compile-time constants of crt_core.c:278-280
 * (G0 is always 65536: (x * 65536) >> 16 wraps to the sign-extended low half of x). */
template <bool FAST, int G1, int G2>
__device__ __forceinline__ int eq_step(Eq3 &f, const int lf, const int hf, const int s)
{
    f.lo0 += (mulq<FAST>(lf, s - f.lo0) + 32768) >> 16;
    f.hi0 += (mulq<FAST>(hf, s - f.hi0) + 32768) >> 16;
    f.lo1 += (mulq<FAST>(lf, f.lo0 - f.lo1) + 32768) >> 16;
    f.hi1 += (mulq<FAST>(hf, f.hi0 - f.hi1) + 32768) >> 16;
    f.lo2 += (mulq<FAST>(lf, f.lo1 - f.lo2) + 32768) >> 16;
    f.hi2 += (mulq<FAST>(hf, f.hi1 - f.hi2) + 32768) >> 16;
    f.lo3 += (mulq<FAST>(lf, f.lo2 - f.lo3) + 32768) >> 16;
    f.hi3 += (mulq<FAST>(hf, f.hi2 - f.hi3) + 32768) >> 16;
    int r = (f.lo3 * 65536) >> 16;
    if (G1 == 65536 || G1 == 8192) r += ((f.hi3 - f.lo3) * G1) >> 16;      /* shifts / bit-field extract */
    else r += mulq<FAST>(f.hi3 - f.lo3, G1) >> 16;
    if (G2 != 0) {
        r += mulq<FAST>(f.h2 - f.hi3, G2) >> 16;
        f.h2 = f.h1; f.h1 = f.h0; f.h0 = s;
    }
    return r;
}

/*
 * Tier 0 of the decoder: one v_mad_i64_i32 per filter stage.
 *   x' = x + ((c*(u-x) + 2^15) >> 16)  ==  hi32( (c<<16)*(u-x) + {lo: 2^31, hi: x} )          c < 2^15
 *   and, because x + (u-x) = u,        ==  hi32( ((c-2^16)<<16)*(u-x) + {lo: 2^31, hi: u} )   2^15 <= c < 1.5*2^16
 * -- multiply, rounding, shift and accumulate in ONE 4-cycle instruction (measured: v_mad_i64_i32 issues
 * like v_mad_i32_i24, profiles/r01_valu_issue_rates.txt), i.e. v_sub + v_mov(lo = 2^31) + v_mad_i64_i32
 * = 8 cycles per stage instead of 10.  Every state lives in the HIGH half of a register pair whose low
 * half is re-armed with 2^31 after each update, so a pair can serve as addend of its own stage (small c)
 * or of the next stage (c near 2^16).  The 64-bit product is exact, whereas the reference's 32-bit one
 * wraps: equal only while |c*(u-x)| + 2^15 < 2^31, which is what the tier-0 envelope guarantees
 * (DESIGN.md): luma |s+bright| <= 2727, chroma |wave| <= 120000.  Luma coefficients are near 2^16,
 * chroma ones below 2^15 for every system of this build (checked on the host).
 */
#define KROUND64 0x80000000ul
__device__ __forceinline__ int hi32(long v) { return (int) (v >> 32); }
__device__ __forceinline__ long pair_of(int v) { return (long) (((unsigned long) (unsigned) v << 32) | KROUND64); }
__device__ __forceinline__ long rearm(long v) { return (long) (((unsigned long) v & 0xffffffff00000000ul) | KROUND64); }
__device__ __forceinline__ long mad64(int d, int cc, long acc)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(d), "s"(cc), "v"(acc));
    return r;
}
struct Eq64 { long lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3; int h0, h1, h2; };
__device__ __forceinline__ void eq64_reset(Eq64 &f)
{
    f.lo0 = f.lo1 = f.lo2 = f.lo3 = f.hi0 = f.hi1 = f.hi2 = f.hi3 = (long) KROUND64;
    f.h0 = f.h1 = f.h2 = 0;
}
/* Tiers 0 and 1: the three equalisers of one sample.  ylfm ... qhfm: pre-shifted multipliers (see above); the luma coefficients are
 * >= 2^15 (x' = u + ...), the chroma ones below (x' = x + ...).
 * The band gains (crt_core.c:203-206) are applied as (r * g) >> 16 per band IN 32-BIT WRAPPING ARITHMETIC, i.e.
 * a gain of 65536 is "sign-extend the low 16 bits".  Inside the envelopes that is the identity:
 *   luma   |lo3|, |hi3| <= 2727                     -> low band = lo3, mid band (gain 8192) = (hi3 - lo3) >> 3
 *   chroma gains (65536, 65536, g2): low + mid = lo3 + (hi3 - lo3) = hi3 whenever |lo3|, |hi3 - lo3| < 2^15.
 *          Every stage output stays inside the hull of its inputs (0 < c < 2^16, round-to-nearest never overshoots), the input
 *          is |s * wave >> 9| <= 16383 (|wave| <= LOSKIP_WAVE_MAX), hence |lo3| <= 16383 and |hi3 - lo3| <= 32766: the four low
 *          stages of I and Q feed nothing and are not computed (lines that need them are flagged CRTHIP_LINE_KEEPLO -> tier 2).
 * Chroma input: u = (s * wave) >> 9 is handed over as the product with the carrier pre-scaled by 2^7, ut = s * (wave << 7) =
 * u * 2^16 + fraction, and every consumer takes its high word inside the subtraction it feeds (SDWA): the shift costs no
 * instruction.  |ut| <= 127 * 120 000 * 128 < 2^31 in these tiers; in tier 0 (|wave| <= 65 532) wave << 7 is still a 24-bit
 * multiplier.  The 3-deep input history (crt_core.c:229-231) holds the products likewise.
 * (r6) The four cascades that remain -- luma low, luma high, I high, Q high -- are independent of each other and are advanced SIDE
 * BY SIDE, stage by stage (four differences, four multiply-adds, the re-arming moves), pinned in that order: written cascade after
 * cascade every instruction waited on the one issued just before it (profiles/r02_valu_mixed_sequences.txt: 3.54 against 3.37
 * cycles per instruction at 4 waves per SIMD).  k_decode 1.594 -> 1.552 ms at 640x480 x 4096, -1.7 ... -3.1 % at every batch size
 * measured (profiles/r06_ab_cascades_side_by_side.txt). */
template <int G1, int G2>
__device__ __forceinline__ void eq_step64_yiq(Eq64 &y, Eq64 &ci_, Eq64 &cq_, const int ylfm, const int yhfm, const int ihfm, const int qhfm,
                                              const long sp, const int uti, const int utq, int &cy, int &ci, int &cq)
{
#define EQ64_PIN() __builtin_amdgcn_sched_barrier(0)
    int d0 = hi32(sp) - hi32(y.lo0), d1 = hi32(sp) - hi32(y.hi0), d2 = sub_hiword(uti, hi32(ci_.hi0)), d3 = sub_hiword(utq, hi32(cq_.hi0));
    EQ64_PIN();
    long a0 = mad64(d0, ylfm, sp), a1 = mad64(d1, yhfm, sp), a2 = mad64(d2, ihfm, ci_.hi0), a3 = mad64(d3, qhfm, cq_.hi0);
    EQ64_PIN();
    y.lo0 = rearm(a0); y.hi0 = rearm(a1); ci_.hi0 = rearm(a2); cq_.hi0 = rearm(a3);
    EQ64_PIN();
#define EQ64_STAGES(P, N)                                                                                                        \
    d0 = hi32(y.lo##P) - hi32(y.lo##N); d1 = hi32(y.hi##P) - hi32(y.hi##N);                                                      \
    d2 = hi32(ci_.hi##P) - hi32(ci_.hi##N); d3 = hi32(cq_.hi##P) - hi32(cq_.hi##N);                                              \
    EQ64_PIN();                                                                                                                  \
    a0 = mad64(d0, ylfm, y.lo##P); a1 = mad64(d1, yhfm, y.hi##P); a2 = mad64(d2, ihfm, ci_.hi##N); a3 = mad64(d3, qhfm, cq_.hi##N); \
    EQ64_PIN();                                                                                                                  \
    y.lo##N = rearm(a0); y.hi##N = rearm(a1); ci_.hi##N = rearm(a2); cq_.hi##N = rearm(a3);                                       \
    EQ64_PIN();
    EQ64_STAGES(0, 1)
    EQ64_STAGES(1, 2)
    EQ64_STAGES(2, 3)
#undef EQ64_STAGES
#undef EQ64_PIN
    const int lo3 = hi32(y.lo3), hi3 = hi32(y.hi3);
    int r;
    if (G1 == 8192) r = lo3 + ((hi3 - lo3) >> 3);                      /* luma envelope, see above */
    else r = ((lo3 * 65536) >> 16) + (__mul24(hi3 - lo3, G1) >> 16);
    r += __mul24(y.h2 - hi3, G2) >> 16;
    y.h2 = y.h1; y.h1 = y.h0; y.h0 = hi32(sp);
    cy = r;
    const int i3 = hi32(ci_.hi3);
    ci = i3 + (__mul24(sub_hiword(ci_.h2, i3), 1311) >> 16);            /* I: top band gain 1311, Q: none (crt_core.c:272-286) */
    ci_.h2 = ci_.h1; ci_.h1 = ci_.h0; ci_.h0 = uti;
    cq = hi32(cq_.hi3);
}

/* eqf of a USE_CONVOLUTION build of the reference (crt_core.c:119-147): a symmetric FIR kernel over a 7-deep
 * input history.  The four kernels factor into running sums,
 *     4 taps  1 1 1 1        = box4
 *     5 taps  1 2 2 2 1      = box2 * box4
 *     6 taps  1 3 4 4 3 1    = box2 * box2 * box4
 *     7 taps  1 4 7 8 7 4 1  = box2 * box2 * box2 * box4          (then >> 2 + number of box2 stages)
 * which is the same integer sum (adds only, no rounding before the final shift; histories start at 0 like the
 * reference's): M box2 stages, each  t = x + prev, prev = x,  then the box4 as  acc += t - ring[n & 3]  with a
 * ring of 4 that the 4-samples-per-dword unrolling indexes statically -- no history shifting at all. */
struct Fir { int p1, p2, p3, acc, r0, r1, r2, r3; };
template <int M>
__device__ __forceinline__ int fir_step(Fir &f, int x, const int K /* sample index & 3: a constant after unrolling */)
{
    if (M >= 1) { const int t = x + f.p1; f.p1 = x; x = t; }
    if (M >= 2) { const int t = x + f.p2; f.p2 = x; x = t; }
    if (M >= 3) { const int t = x + f.p3; f.p3 = x; x = t; }
    int &slot = K == 0 ? f.r0 : K == 1 ? f.r1 : K == 2 ? f.r2 : f.r3;
    f.acc += x - slot;
    slot = x;
    return f.acc >> (2 + M);
}

/* byte selectors for v_perm_b32: 0xffRRGGBB (bytes B,G,R,ff) <-> the four 4-byte output formats,
 * crt_core.c:587-656 */
__device__ __forceinline__ unsigned pack_selector(int format)
{
    return format == CRTHIP_FMT_BGRA ? 0x03020100u : format == CRTHIP_FMT_RGBA ? 0x03000102u
         : format == CRTHIP_FMT_ARGB ? 0x00010203u : 0x02010003u /* ABGR */;
}
__device__ __forceinline__ unsigned unpack_selector(int format)
{
    return format == CRTHIP_FMT_BGRA ? 0x03020100u : format == CRTHIP_FMT_RGBA ? 0x03000102u
         : format == CRTHIP_FMT_ARGB ? 0x00010203u : 0x00030201u /* ABGR */;
}

/*
 * Global memory traffic of the lane-per-line kernels.  A lane walks along its own scanline, so a
 * wave's 64 lanes touch 64 different rows: per-lane loads/stores would move 16 bytes out of every
 * 128-byte line at a time (measured: 3-8x the algorithmic HBM traffic, profiles/r01_v1_*).  Instead
 * all global I/O goes through LDS tiles [64 rows][TILE] that the wave fills / drains COOPERATIVELY
 * with row-contiguous 16-byte pieces (8 lanes x 16 B = one 128-byte line of one row per 8 lanes),
 * while each lane reads / writes only its own row of the tile.  Row addressing: TileRows (crt_dev.h) -- (r6) one pad dword per
 * row PAIR for the 16-dword tiles, which is conflict-free for both access directions; the odd row stride TILE + 1 of rounds 1-5
 * was that only for the lane-per-row direction (the counters said so: profiles/r05_headline_sq_counters.json).
 * Workgroup = one wave, so __syncthreads() is only an ordering fence between the two phases.
 */
/* decoder input tile: IN_TILE_DW dwords (64 samples) per row.  8 dwords would buy a fifth wave per SIMD (fewer
 * VGPRs, less LDS) but measured slower for narrow and wide pictures alike (A/B on one box) */
/* decoder output tile: PXT pixels per row (16: 64-byte store pieces, less LDS -> more waves, best for
 * narrow pictures that are ALU bound; 32: full 128-byte lines per store piece group, best for wide
 * pictures that lean on HBM write bandwidth) */
/* TIER: 0 = 64-bit-mad stages without the I/Q low cascades, carrier products by v_mul_i32_i24 (|wave| <= 65532);
 * 1 = the same with the carrier products by v_mad_i64_i32 (|wave| <= crthip_params.loskip_wave_max <= 120000: a carrier
 * << 7 is no 24-bit multiplier any more, the cascades can still be dropped -- the NES at its default saturation);
 * 2 = 24-bit mads, all cascades (also every line flagged CRTHIP_LINE_KEEPLO),
 * 3 = exact 32-bit multiplies; 4 / 5 = the FIR kernels of a USE_CONVOLUTION build (P.eq_kernel taps; the whole
 * batch) with 24-bit / exact 32-bit multiplies around them; a wave of 64 lines is decoded by the kernel
 * of its tier = max(tier flagged by k_hsync_wave from its carrier amplitude, min_tier of the batch);
 * want_rank: only lines of this collision rank (always 0 unless outh + v_fac < LINES).
 * BLOOM (CRT_DO_BLOOM build, crt_core.c:512-526): every scanline has its own resampler step and start, a function of its
 * width line_w alone.  The 64 lines of a wave are then not neighbours but lines of EQUAL line_w, gathered by `perm` (the
 * counting sort of crt_decode3.hip; -1 = padding), so the pixel schedule stays wave-uniform; the filters run over
 * [scanL >> 12, AV_LEN - 1) only and the entry AV_LEN - 1 of the reference's yiq buffer is never written (zero). */
/* TG = tier GROUP of this launch: 0 -> tiers 0 / 1, 1 -> tiers 2 / 3, 2 -> tiers 4 / 5 (FIR).  A wave takes the loop compiled for
 * its own tier -- two copies of the line loop in one kernel, chosen once per wave -- so that a field-pass launches two decoder
 * kernels where it launched four, three of which usually had nothing to do (5 us each in a 230 us step). */
#ifndef DEC_IN_TILE_DW
#define DEC_IN_TILE_DW 16                  /* (A/B builds: -DDEC_IN_TILE_DW=8 -DDEC_WAVES_PER_EU=5, profiles/r06_decode_occupancy.txt) */
#endif
#ifdef DEC_WAVES_PER_EU
#define DEC_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(DEC_WAVES_PER_EU, DEC_WAVES_PER_EU)))
#else
#define DEC_OCCUPANCY_ATTR
#endif
template <class S, int TG, bool BPP3, int PXT, bool BLOOM = false>
__global__ void __launch_bounds__(64) DEC_OCCUPANCY_ATTR
k_decode(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
         const crthip_line *__restrict__ lines, unsigned char *__restrict__ outp, size_t ostride, int min_tier,
         int want_rank, const int *__restrict__ perm, int order_k, int order_per)
{
    constexpr int TLO = 2 * TG;
    constexpr int IN_TILE_DW = DEC_IN_TILE_DW, IN_PIECES = IN_TILE_DW / 4;
    using TIN = TileRows<IN_TILE_DW>;                 /* row addressing of both tiles: conflict-free lane-per-row AND cooperatively (crt_dev.h) */
    __shared__ unsigned s_in[TIN::DWORDS];
    constexpr int PX_TILE = PXT, PX_PIECES = PXT / 4;
    using TPX = TileRows<PXT>;
    __shared__ unsigned s_px[TPX::DWORDS];
    __shared__ unsigned long long s_src[64], s_dst[64];
    __shared__ int s_nrows[64];

    const int lane = threadIdx.x;
    int gid = block_item(blockIdx.x, order_k, order_per) * 64 + lane;     /* workgroup order: crt_dev.h */
    if (BLOOM) gid = perm[gid];
    const bool live = BLOOM ? gid >= 0 : gid < n_fields * S::LINES;
    crthip_line lp;
    lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0; lp.hsync = 0;
    const int f = live ? gid / S::LINES : 0;
    if (live) lp = lines[gid];
    /* the WAVE's tier = the highest one any of its lines needs (a higher tier decodes lower-tier lines just
     * as exactly), at least the batch-wide floor from the host (brightness, contrast): a wave with mixed lines
     * runs once, not once per tier */
    int tier = __ballot(lp.nrows & CRTHIP_LINE_EXACT) ? 3 : __ballot(lp.nrows & CRTHIP_LINE_NOT64) ? 2
             : __ballot(lp.nrows & CRTHIP_LINE_WIDE) ? (BLOOM && S::CCS != 5 ? 2 : 1) : 0;    /* bloom build: tier 1 only for the 5-sample system */
    /* a line that needs its I/Q low cascades takes the 24-bit tier (tiers 0 and 1 have none) */
    if (tier < 2 && __ballot(lp.nrows & (int) CRTHIP_LINE_KEEPLO) != 0ull) tier = 2;
    if (min_tier >= 4) tier = (min_tier == 5 || tier == 3) ? 5 : 4;     /* FIR build: 24-bit envelope as for tier 2 */
    else if (tier < min_tier) tier = min_tier;
    if (tier != TLO && tier != TLO + 1) return;
    int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    const int rank = (lp.nrows >> CRTHIP_LINE_RANK_SHIFT) & CRTHIP_LINE_RANK_MASK;
    if (!live || rank != want_rank) nrows = 0;
    if (__ballot(nrows > 0) == 0ull) return;          /* whole wave has nothing to do */
    /* Bloom: the sort hands every wave lines of ONE geometry (see crt_decode3.hip for why their width is bounded).  Should
     * a wave ever hold more than one -- a line table that did not come from k_bloom, through crthip_decode -- it is decoded
     * in rounds, one geometry at a time, the other lanes idling: correct for any table, free for the tables that occur. */
    /* 5 samples per chroma cycle: the line's carrier pairs by sample phase, [lane][phase]{I, Q} at an odd row stride (below) */
    constexpr int W5_STRIDE = 11;
    __shared__ int s_w5[S::CCS == 5 ? 64 * W5_STRIDE : 1];
  auto decode_lines = [&](auto tier_tag) {
    constexpr int TIER = decltype(tier_tag)::value;
    constexpr bool FAST = TIER <= 2 || TIER == 4;   /* 24-bit multiplies outside the filter stages */
    constexpr bool FIR = TIER >= 4;
    int nrows_todo = nrows;
  do {
    int scanl_u = 0, dx_u = P.dx;
    if (BLOOM) {
        const int src = __ffsll((unsigned long long) __ballot(nrows_todo > 0)) - 1;
        scanl_u = __builtin_amdgcn_readlane(lp.scanl, src);         /* crt_core.c:519-526 */
        dx_u = __builtin_amdgcn_readlane(lp.dx, src);
        nrows = (lp.scanl == scanl_u && lp.dx == dx_u) ? nrows_todo : 0;
        if (nrows > 0) nrows_todo = 0;
    }
    const bool act = nrows > 0;
    constexpr int bpp = BPP3 ? 3 : 4;
    const size_t pitch = (size_t) P.outw * bpp;
    s_src[lane] = (unsigned long long) (inp + (size_t) f * fstride + (act ? lp.pos : 0));
    s_dst[lane] = (unsigned long long) (outp + (size_t) f * ostride + (size_t) (act ? lp.beg : 0) * pitch);
    s_nrows[lane] = nrows;
    wave_lds_fence();

    /* tiers 0 and 1 multiply by the carriers scaled by 2^7 (eq_step64_yiq) */
    constexpr int WSCALE = TIER <= 1 ? 128 : 1;
    const int w0 = lp.wave0 * WSCALE, w1 = lp.wave1 * WSCALE, nw0 = -lp.wave0 * WSCALE, nw1 = -lp.wave1 * WSCALE;
    /* 5 samples per chroma cycle (PV-1000, crt_core.c:497-505, 544-549): the line table carries dci / dcq, the carriers
     * of the five sample phases follow from them with the blob's cos / sin tables; the phase of a sample is wave-uniform */
    /* ... so the carrier pair of a sample is ONE LDS read at a scalar offset ([lane][phase]{I, Q}, odd row stride), where
     * selecting among five per-lane registers costs four v_cndmask per carrier (round 3: 3.35 -> 3.08 ms per 4096 fields) */
    if constexpr (S::CCS == 5) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            s_w5[lane * W5_STRIDE + 2 * i] = ((lp.wave0 * P.dem_cs[0][i] + lp.wave1 * P.dem_sn[0][i]) >> 15) * P.saturation * WSCALE;
            s_w5[lane * W5_STRIDE + 2 * i + 1] = ((lp.wave0 * P.dem_cs[1][i] + lp.wave1 * P.dem_sn[1][i]) >> 15) * P.saturation * WSCALE;
        }
    }
    const int first = scanl_u >> 12;               /* first filtered sample, :525, :539 */
    const int xq0 = first >> 2;                    /* ... and its dword */
    int ph5 = (xq0 * 4) % 5;                       /* sample index % 5 */
    /* luma band gains of the build, crt_core.c:272-286 */
    constexpr int GY1 = S::CCS == 5 ? 12192 : 8192, GY2 = S::CCS == 5 ? 7775 : 9175;
    const int bright = P.bright, contrast = P.contrast;
    const int ylf = P.eq_lf[0], yhf = P.eq_hf[0], ilf = P.eq_lf[1], ihf = P.eq_hf[1], qlf = P.eq_lf[2], qhf = P.eq_hf[2];
    const unsigned psel = pack_selector(P.out_format), usel = unpack_selector(P.out_format);
    const bool rgb_order = P.out_format == CRTHIP_FMT_RGB;
    const bool blend = P.blend != 0;
    Eq3 ey = {}, ei = {}, eq = {};
    Fir fy = {}, fi = {}, fq = {};
    const int fir_m = P.eq_kernel - 4;             /* box2 stages of the FIR kernel (wave-uniform) */
    Eq64 wy, wi_, wq_;                             /* tier 0 state (register pairs) */
    eq64_reset(wy); eq64_reset(wi_); eq64_reset(wq_);
    /* tier 0 multipliers: luma coefficients are 2^16 + c', chroma ones < 2^15 (host-checked) */
    const int ylfm = (ylf - 65536) << 16, yhfm = (yhf - 65536) << 16;
    const int ihfm = ihf << 16, qhfm = qhf << 16;              /* (the chroma low cascades are not computed in these tiers) */
    int py = 0, pi = 0, pq = 0;                    /* yiq of the previous sample */

    /* wave-uniform output pixel schedule, crt_core.c:528-531,555-562 */
    /* pixel px sits at ppos = px * dx and is emitted after sample (ppos >> 12) + 1; both loop conditions of
     * crt_core.c:555 (px < outw, pos < scan_r) fold into one bound on ppos (64-bit product: dx * outw < 2^32 only
     * just) */
    const unsigned long long ppos_all = (unsigned long long) (unsigned) dx_u * (unsigned) P.outw + (unsigned) scanl_u;
    const unsigned scan_r = (unsigned) (S::AV_LEN - 1) << 12;
    const unsigned ppos_end = ppos_all < scan_r ? (unsigned) ppos_all : scan_r;
    const unsigned dx = (unsigned) dx_u;
    unsigned ppos = (unsigned) scanl_u;
    const int outw = P.outw;
    int px = 0, px0 = 0;                           /* next pixel; first pixel of the tile being filled */
    int tile_end = PX_TILE < outw ? PX_TILE : outw;
    /* tiers 0 / 1: contrast as a pre-shifted 64-bit-mad multiplier, the opaque alpha riding on the red row */
    const int contrast12 = contrast * 4096;
    long alpha_pair = (long) 0xff00ul << 32;
    asm volatile("" : "+v"(alpha_pair));

    /* cooperative input tile: piece = 16 bytes, IN_PIECES pieces per row, 64 / IN_PIECES rows per load instruction */
    const int in_row = lane / IN_PIECES, in_piece = lane % IN_PIECES;
    constexpr int IN_ROWS = 64 / IN_PIECES;            /* rows covered by one load instruction */
    const int t0 = xq0 / IN_TILE_DW;               /* bloom: the tiles before the first filtered sample are never read */
    v4i stage[IN_PIECES];
#pragma unroll
    for (int i = 0; i < IN_PIECES; i++) {
        stage[i] = gload16u(s_src[i * IN_ROWS + in_row] + t0 * (IN_TILE_DW * 4) + in_piece * 16);
    }
    constexpr int NQ = (S::AV_LEN + 3) / 4;        /* dwords per line window; the last one may run past
                                                      AV_LEN: the filters are causal, the extra samples feed nothing */
    constexpr int NT = (NQ + IN_TILE_DW - 1) / IN_TILE_DW;
    for (int t = t0; t < NT; t++) {
        /* stash tile t (already in registers), then start fetching tile t+1 */
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < IN_PIECES; i++) {
            unsigned *d = s_in + TIN::row(i * IN_ROWS + in_row) + in_piece * 4;
            d[0] = (unsigned) stage[i].x; d[1] = (unsigned) stage[i].y; d[2] = (unsigned) stage[i].z; d[3] = (unsigned) stage[i].w;
        }
        wave_lds_fence();
        if (t + 1 < NT) {
#pragma unroll
            for (int i = 0; i < IN_PIECES; i++) {
                stage[i] = gload16u(s_src[i * IN_ROWS + in_row] + (t + 1) * (IN_TILE_DW * 4) + in_piece * 16);
            }
        }
        const int xq_end = (t + 1) * IN_TILE_DW < NQ ? (t + 1) * IN_TILE_DW : NQ;
        for (int xq = t == t0 ? xq0 : t * IN_TILE_DW; xq < xq_end; xq++) {
            const int word = (int) s_in[TIN::row(lane) + (xq - t * IN_TILE_DW)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = xq * 4 + k;
                const int s = (word << (24 - 8 * k)) >> 24;
                /* D8, crt_core.c:539-543; wave[] = {w0, w1, -w0, -w1}: I uses wave[x&3], Q wave[(x+3)&3] */
                int wi = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? nw0 : nw1;
                int wq = k == 0 ? nw1 : k == 1 ? w0 : k == 2 ? w1 : nw0;
                if constexpr (S::CCS == 5) {
                    wi = s_w5[lane * W5_STRIDE + 2 * ph5];    /* ph5 is wave-uniform */
                    wq = s_w5[lane * W5_STRIDE + 2 * ph5 + 1];
                    ph5 = ph5 == 4 ? 0 : ph5 + 1;
                }
                if (BLOOM && x < first) continue;  /* wave-uniform: the filters start at scanL >> 12, crt_core.c:539 */
                int cy, ci, cq;
                if (TIER == 0) {
                    /* luma stays unshifted here: (y << 4) * w >> 2 == (y * w) << 2 while nothing wraps, see D9 */
                    eq_step64_yiq<GY1, GY2>(wy, wi_, wq_, ylfm, yhfm, ihfm, qhfm, pair_of(s + bright), __mul24(s, wi), __mul24(s, wq), cy, ci, cq);
                    ci >>= 3; cq >>= 3;                        /* wi, wq: carriers << 7 here */
                } else if (TIER == 1) {
                    /* carriers << 7 beyond 24 bits: the products come from the 64-bit multiply-add (exact low dword,
                     * |s * (wave << 7)| < 2^31 up to |wave| = 120000) */
                    eq_step64_yiq<GY1, GY2>(wy, wi_, wq_, ylfm, yhfm, ihfm, qhfm, pair_of(s + bright), mul_lo_mad64(s, wi), mul_lo_mad64(s, wq), cy, ci, cq);
                    ci >>= 3; cq >>= 3;
                } else if (FIR) {
                    const int uy = s + bright, ui = mulq<FAST>(s, wi) >> 9, uq = mulq<FAST>(s, wq) >> 9;
#define CRT_FIR3(M) do { cy = fir_step<M>(fy, uy, k) << 4; ci = fir_step<M>(fi, ui, k) >> 3; cq = fir_step<M>(fq, uq, k) >> 3; } while (0)
                    if (fir_m == 3) CRT_FIR3(3); else if (fir_m == 2) CRT_FIR3(2); else if (fir_m == 1) CRT_FIR3(1); else CRT_FIR3(0);
#undef CRT_FIR3
                } else {
                    cy = eq_step<FAST, GY1, GY2>(ey, ylf, yhf, s + bright) << 4;
                    ci = eq_step<FAST, 65536, 1311>(ei, ilf, ihf, mulq<FAST>(s, wi) >> 9) >> 3;
                    cq = eq_step<FAST, 65536, 0>(eq, qlf, qhf, mulq<FAST>(s, wq) >> 9) >> 3;
                }
                if (BLOOM && x >= S::AV_LEN - 1) { cy = 0; ci = 0; cq = 0; }     /* never written: crt_core.c:526, 539 */
                /* D9: every output pixel whose left tap is sample x-1 is now computable: pixel px sits at ppos = px * dx and
                 * needs samples ppos >> 12 and (ppos >> 12) + 1, i.e. it is emitted at the first x with ppos < x << 12
                 * (pixels are emitted in order, so everything below (x - 1) << 12 is already out) */
                const unsigned lim_x = (unsigned) x << 12;
                const unsigned lim = lim_x < ppos_end ? lim_x : ppos_end;
                while (ppos < lim) {
                    const int R = (int) (ppos & 0xfffu), L = 0xfff - R;
                    int yy;
                    if (TIER <= 1) {
                        /* crt_core.c:556: (py * L >> 2) + (cy * R >> 2) with py, cy = luma << 4.  |luma| <= 4173
                         * inside the tier's envelope, so no product wraps and both shifts are exact */
                        yy = mad24_vs(cy, R << 2, mulq_vs<true>(py, L << 2));
                    } else {
                        yy = (mulq_vs<FAST>(py, L) >> 2) + (mulq_vs<FAST>(cy, R) >> 2);
                    }
                    unsigned rgb;
                    if (TIER <= 1) {
                        /* crt_core.c:557-558: (pi * L >> 14) + (ci * R >> 14).  With the weights scaled by 4 each shift
                         * is "take the high word", and both ride on the add.  |chroma| <= 2^13 inside the tier's
                         * envelope (|wave| <= 120000: inputs |s * wave >> 9| < 2^15, outputs >> 3), weights < 2^14.
                         * Both results fit 16 bits, so they are formed PACKED -- q in the low half, i in the high half
                         * (the second add writes only WORD_1) -- and each colour row of crt_core.c:560-562 is one
                         * v_dot2_i32_i16 by its packed coefficient pair on top of the luma. */
                        int iq = add_hiwords(mulq_vs<true>(pq, L << 2), mulq_vs<true>(cq, R << 2));
                        iq = add_hiwords_to_hi(iq, mulq_vs<true>(pi, L << 2), mulq_vs<true>(ci, R << 2));
                        const int vr = dot2_vs(iq, (3879 << 16) | 2556, yy);
                        const int vg = dot2_vs(iq, (int) (((unsigned) -1126 << 16) | ((unsigned) -2605 & 0xffffu)), yy);
                        const int vb = dot2_vs(iq, (int) (((unsigned) -4530 << 16) | 7021u), yy);
                        /* ((v >> 12) * contrast) >> 8 == hi32((v & ~0xfff) * (contrast << 12)): one v_mad_i64_i32 instead of
                         * shift, multiply, shift.  The 64-bit product is exact where the reference's 32-bit one wraps: equal
                         * inside the envelope (|v| < 2^27.3, |contrast| <= 32768, host-checked).  The red row carries the
                         * opaque alpha along: + 0xff00 through the addend, clamped to [0xff00, 0xffff]. */
                        int r = pair_hi(mad64_vs(vr & ~0xfff, contrast12, alpha_pair));
                        int g = pair_hi(mad64_vs0(vg & ~0xfff, contrast12));
                        int b = pair_hi(mad64_vs0(vb & ~0xfff, contrast12));
                        r = clampi(r, 0xff00, 0xffff); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                        rgb = lshl_or(lshl_or((unsigned) r, 8, (unsigned) g), 8, (unsigned) b);       /* 0xffRRGGBB */
                    } else {
                        const int ii = (mulq_vs<FAST>(pi, L) >> 14) + (mulq_vs<FAST>(ci, R) >> 14);
                        const int qq = (mulq_vs<FAST>(pq, L) >> 14) + (mulq_vs<FAST>(cq, R) >> 14);
                        int r, g, b;
                        if (FAST) {
                            r = mulq<true>(mad24_vs(qq, 2556, mad24_vs(ii, 3879, yy)) >> 12, contrast) >> 8;
                            g = mulq<true>(mad24_vs(qq, -2605, mad24_vs(ii, -1126, yy)) >> 12, contrast) >> 8;
                            b = mulq<true>(mad24_vs(qq, 7021, mad24_vs(ii, -4530, yy)) >> 12, contrast) >> 8;
                        } else {
                            r = ((yy + 3879 * ii + 2556 * qq) >> 12) * contrast >> 8;
                            g = ((yy - 1126 * ii - 2605 * qq) >> 12) * contrast >> 8;
                            b = ((yy - 4530 * ii + 7021 * qq) >> 12) * contrast >> 8;
                        }
                        r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                        rgb = lshl_or(lshl_or((unsigned) r, 8, (unsigned) g), 8, (unsigned) b);
                    }
                    s_px[TPX::row(lane) + (px - px0)] = rgb;
                    ppos += dx;
                    px++;
                    if (px == tile_end) {
                        /* drain the pixel tile: pixels [px0, px0+cnt) of every row, + D10 duplicates (:661-664) */
                        const int cnt = px - px0;
                        wave_lds_fence();
                        if (!BPP3) {
                            const int orow_ = lane / PX_PIECES, piece = lane % PX_PIECES;  /* 4 pixels = 16 bytes per piece */
                            const int have = cnt - piece * 4;                      /* pixels of this piece that exist */
#pragma unroll 2
                            for (int i = 0; i < PX_PIECES; i++) {
                                const int rr_ = i * (64 / PX_PIECES) + orow_;
                                const int nr = s_nrows[rr_];
                                if (nr > 0 && have > 0) {
                                    const unsigned long long d = s_dst[rr_] + (size_t) (px0 + piece * 4) * 4;
                                    const unsigned *sp = s_px + TPX::row(rr_) + piece * 4;
                                    unsigned v[4] = { sp[0], sp[1], sp[2], sp[3] };
                                    if (blend) {
#pragma unroll
                                        for (int c = 0; c < 4; c++) {
                                            if (c < have) {
                                                const unsigned oldw = gload32(d + 4 * c);
                                                const unsigned old = __builtin_amdgcn_perm(oldw, oldw, usel);
                                                v[c] = ((v[c] & 0xfefeffu) >> 1) + ((old & 0xfefeffu) >> 1);
                                            }
                                        }
                                    }
                                    if (TIER > 1 || blend) {                       /* tiers 0 / 1 carry the alpha byte already */
#pragma unroll
                                        for (int c = 0; c < 4; c++) v[c] |= 0xff000000u;
                                    }
                                    if (psel != 0x03020100u) {                     /* BGRA is the in-register order */
#pragma unroll
                                        for (int c = 0; c < 4; c++) v[c] = __builtin_amdgcn_perm(v[c], v[c], psel);
                                    }
                                    for (int dup = 0; dup < nr; dup++) {
                                        const unsigned long long dd = d + (size_t) dup * pitch;
                                        if (have >= 4) {
                                            v4i o; o.x = (int) v[0]; o.y = (int) v[1]; o.z = (int) v[2]; o.w = (int) v[3];
#if defined(DEC_DBG) && DEC_DBG == 1          /* measurement build: no picture stores (the alpha byte is never 0) */
                                            if (o.x != 0) continue;
#endif
                                            gstore16u_nt(dd, o);   /* nontemporal: plain stores measured 5 % slower here and slow the encoder down too (profiles/r03_1080p_experiments.txt) */
                                        } else {
                                            gstore32(dd, v[0]);
                                            if (have > 1) gstore32(dd + 4, v[1]);
                                            if (have > 2) gstore32(dd + 8, v[2]);
                                        }
                                    }
                                }
                            }
                        } else {
                            /* 3-byte formats: one pixel per lane, PX_TILE pixels of 64/PX_TILE rows per pass */
                            const int half = lane / PX_TILE, c = lane % PX_TILE;
                            for (int i = 0; i < PX_TILE; i++) {
                                const int rr_ = i * (64 / PX_TILE) + half;
                                const int nr = s_nrows[rr_];
                                if (nr > 0 && c < cnt) {
                                    const unsigned long long d = s_dst[rr_] + (size_t) (px0 + c) * 3;
                                    int rgb3 = (int) s_px[TPX::row(rr_) + c];
                                    if (blend) {
                                        const int o0 = (int) gload8(d), o1 = (int) gload8(d + 1), o2 = (int) gload8(d + 2);
                                        const int old = rgb_order ? (o0 << 16 | o1 << 8 | o2) : (o2 << 16 | o1 << 8 | o0);
                                        rgb3 = ((rgb3 & 0xfefeff) >> 1) + ((old & 0xfefeff) >> 1);
                                    }
                                    const unsigned char c0 = (unsigned char) (rgb_order ? rgb3 >> 16 : rgb3);
                                    const unsigned char c2 = (unsigned char) (rgb_order ? rgb3 : rgb3 >> 16);
                                    for (int dup = 0; dup < nr; dup++) {
                                        const unsigned long long dd = d + (size_t) dup * pitch;
                                        gstore8(dd, c0); gstore8(dd + 1, (unsigned) (rgb3 >> 8)); gstore8(dd + 2, c2);
                                    }
                                }
                            }
                        }
                        wave_lds_fence();
                        px0 = px;
                        tile_end = px0 + PX_TILE < outw ? px0 + PX_TILE : outw;
                    }
                }
                py = cy; pi = ci; pq = cq;
            }
        }
    }
    wave_lds_fence();
  } while (BLOOM && __ballot(nrows_todo > 0) != 0ull);
  };
    if (tier == TLO) decode_lines(std::integral_constant<int, TLO>{});
    else decode_lines(std::integral_constant<int, TLO + 1>{});
}


#endif
