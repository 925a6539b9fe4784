/*
 * crt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) + the crthip_* C ABI.
 *
 * The reference (LMP88959/NTSC-CRT, C89, single threaded) processes one field with
 * strictly serial per-scanline recurrences that floor after every multiply
 * (crt_ntsc.c:117-126 iirf, crt_core.c:206-233 eqf), so no parallel scan can be
 * bit-exact.  What IS independent: scanlines (filter state is reset per line,
 * crt_ntsc.c:267-269, crt_core.c:534-536) and fields.  The design is therefore
 *
 *     one LANE per scanline, 64 scanlines per wavefront, many fields per launch
 *
 * with wave-uniform control flow: all 64 lanes are at the same sample x at the same
 * time, so everything that depends only on x (source column, carrier phase, which
 * output pixels become ready and their interpolation weights) lives in SGPRs / the
 * scalar unit and costs no vector issue slots.
 *
 * Kernels (stage names M0-M6 / D0-D10 as in DESIGN.md):
 *   k_template  M4      blanking / sync / burst skeleton        (elementwise)
 *   k_active    M5      RGB->YIQ, 3x 1-pole IIR, quadrature mod  (lane per image row)
 *   k_noise     D1      LCG noise via affine jump-ahead tables   (elementwise)
 *   k_sync      D2-D7   vsync, hsync, burst lock                 (wave per field, serial chain)
 *   k_decode    D8-D10  3x 3-band IIR equaliser, resample, YIQ->RGB, row duplication
 *                                                                (lane per CRT line)
 * Integer-only; signed overflow wraps, >> of negatives is arithmetic, / truncates --
 * identical to the reference on x86-64.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "crt_hip.h"
#include "crt_setup.h"

/* ------------------------------------------------------------------------- */
/* compile-time system tables (device side); cross-checked against the C89    */
/* host table crt_sysdef_get() when a context is created                      */
/* ------------------------------------------------------------------------- */
template <int CC_LINE>
struct RgbTiming {   /* crt_ntsc.h:25-109 / crt_ntscvhs.h */
    static constexpr int HRES = CC_LINE * 4 / 10;
    static constexpr int VRES = 262;
    static constexpr int INPUT_SIZE = HRES * VRES;
    static constexpr int TOP = 21, BOT = 261, LINES = BOT - TOP;
    static constexpr int VPER = 1;
    static constexpr int HWIN = 8, VWIN = 8;
    static constexpr int WHITE = 100, BURST = 20, BLACK = 7, BLANK = 0, SYNC = -40;
    static constexpr int HTHR = 4 * SYNC, VTHR = 94 * SYNC;
    static constexpr int SYNC_BEG = 1500 * HRES / 63500;
    static constexpr int BW_BEG = 6200 * HRES / 63500;
    static constexpr int CB_BEG = 6800 * HRES / 63500;
    static constexpr int AV_BEG = 10900 * HRES / 63500;
    static constexpr int AV_LEN = 52600 * HRES / 63500;
    static constexpr int VS_SEP_END = 0;
    static constexpr bool IS_NES = false;
};
template <int CC_LINE>
struct NesTiming {   /* crt_nes.h:30-126 */
    static constexpr int HRES = CC_LINE * 4 / 10;
    static constexpr int VRES = 262;
    static constexpr int INPUT_SIZE = HRES * VRES;
    static constexpr int TOP = 15, BOT = 255, LINES = BOT - TOP;
    static constexpr int VPER = 3;
    static constexpr int HWIN = 6, VWIN = 6;
    static constexpr int WHITE = 110, BURST = 30, BLACK = 0, BLANK = 0, SYNC = -37;
    static constexpr int HTHR = 4 * SYNC, VTHR = 94 * SYNC;
    static constexpr int SYNC_BEG = 9 * HRES / 341;
    static constexpr int BW_BEG = 34 * HRES / 341;
    static constexpr int CB_BEG = 38 * HRES / 341;
    static constexpr int AV_BEG = 74 * HRES / 341;
    static constexpr int AV_LEN = 256 * HRES / 341;
    static constexpr int VS_SEP_END = 327 * HRES / 341;
    static constexpr bool IS_NES = true;
};
struct SysNTSC : RgbTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSC, PATTERN = 1; static constexpr bool IS_VHS = false; };
struct SysNTSC0 : RgbTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSC, PATTERN = 0; static constexpr bool IS_VHS = false; };
struct SysVHS : RgbTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSCVHS, PATTERN = 1; static constexpr bool IS_VHS = true; };
struct SysVHS0 : RgbTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSCVHS, PATTERN = 0; static constexpr bool IS_VHS = true; };
struct SysNES2 : NesTiming<2273> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 2; static constexpr bool IS_VHS = false; };
struct SysNES1 : NesTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 1; static constexpr bool IS_VHS = false; };
struct SysNES0 : NesTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 0; static constexpr bool IS_VHS = false; };

static_assert(SysNTSC::HRES == 910 && SysNTSC::AV_BEG == 156 && SysNTSC::AV_LEN == 753 &&
              SysNTSC::SYNC_BEG == 21 && SysNTSC::CB_BEG == 97, "NTSC timing (SURVEY.md section 8)");
static_assert(SysNES2::HRES == 909 && SysNES2::AV_LEN == 682 && SysNES0::HRES == 912 &&
              SysNES0::AV_LEN == 684, "NES timing (SURVEY.md section 8)");

#define CRTHIP_LINE_EXACT 0x40000000      /* bit in crthip_line.nrows: outside the 24-bit envelope */
#define CRTHIP_LINE_NROWS_MASK 0xffff     /* crthip_line.nrows bits 0-15: rows written              */
#define CRTHIP_LINE_RANK_SHIFT 16         /* bits 16-27: rank among lines starting on the same row   */
#define CRTHIP_LINE_RANK_MASK  0xfff
#define CRTHIP_LINE_WIDE  0x10000000      /* bit 28: chroma too strong to drop the I/Q low cascades (decoder tier 0) */
#define CRTHIP_LINE_NOT64 0x20000000      /* bit 29: outside the no-wrap envelope of the 64-bit-mad decoder */
#define LOSKIP_WAVE_MAX   65532           /* |wave[k]| bound of decoder tier 0: |s*wave >> 9| <= 16383 */
#define T0_WAVE_MAX       120000          /* |wave[k]| bound of decoder tiers 0 and 1 */
#define T0_BRIGHT_MAX     2600            /* |bright| bound of decoder tiers 0 and 1  */
#define FAST_WAVE_MAX     524288          /* |wave[k]| bound of the fast decoder, 2^19 */
#define FAST_BRIGHT_MAX   130000          /* |bright| bound of the fast decoder        */
#define CB_SAMPLES 40            /* CB_CYCLES * CRT_CB_FREQ, crt_ntsc.h:89 */
#define LCG_MUL 214019u          /* crt_core.c:359 */
#define LCG_ADD 140327895u

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) unaligned16 { v4i v; };
struct __attribute__((packed)) unaligned4 { int v; };

/* The same through an explicit global (address space 1) pointer: addresses that went through LDS
 * as integers would otherwise be treated as generic ("flat") and unaligned 16-byte accesses to them
 * get split into dwords, because flat could mean LDS, where misaligned wide accesses are illegal. */
typedef __attribute__((address_space(1))) unaligned16 g_unaligned16;
typedef __attribute__((address_space(1))) unsigned g_u32;
typedef __attribute__((address_space(1))) unsigned char g_u8;
__device__ __forceinline__ v4i gload16u(unsigned long long a) { return ((const g_unaligned16 *) a)->v; }
__device__ __forceinline__ void gstore16u(unsigned long long a, v4i v) { ((g_unaligned16 *) a)->v = v; }
__device__ __forceinline__ unsigned gload32(unsigned long long a) { return *(const g_u32 *) a; }
__device__ __forceinline__ void gstore32(unsigned long long a, unsigned v) { *(g_u32 *) a = v; }
__device__ __forceinline__ unsigned gload8(unsigned long long a) { return *(const g_u8 *) a; }
__device__ __forceinline__ void gstore8(unsigned long long a, unsigned v) { *(g_u8 *) a = (unsigned char) v; }

__device__ __forceinline__ v4i load16u(const void *p) { return ((const unaligned16 *) p)->v; }
__device__ __forceinline__ void store16u(void *p, v4i v) { ((unaligned16 *) p)->v = v; }
__device__ __forceinline__ int load4u(const void *p) { return ((const unaligned4 *) p)->v; }
__device__ __forceinline__ void store4u(void *p, int v) { ((unaligned4 *) p)->v = v; }

__device__ __forceinline__ int posmod(int x, int n) { return ((x % n) + n) % n; }  /* crt_core.c:17 */
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* full-rate 24-bit multiply when FAST (operands proven inside [-2^23, 2^23)), else the
 * quarter-rate exact 32-bit one -- see the comment above k_decode */
template <bool FAST> __device__ __forceinline__ int mulq(int a, int b)
{
    if (FAST) return __mul24(a, b);
    return a * b;
}

/* vgpr * sgpr.  __mul24's sign-extension of a LOOP-CARRIED operand gets hoisted to its definition in
 * another basic block, after which instruction selection (per block) no longer knows the value fits
 * 24 bits and falls back to the quarter-rate v_mul_lo_u32; pinning the instruction avoids that. */
template <bool FAST> __device__ __forceinline__ int mulq_vs(int v, int s_uniform)
{
    if (FAST) {
        int r;
        asm("v_mul_i32_i24 %0, %2, %1" : "=v"(r) : "v"(v), "s"(s_uniform));
        return r;
    }
    return v * s_uniform;
}

/* vgpr * sgpr + vgpr, 24-bit operands */
__device__ __forceinline__ int mad24_vs(int v, int s_uniform, int acc)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(s_uniform), "v"(acc));
    return r;
}

/* the low bytes of four ints -> one dword (3 permutes instead of 4 x (mask, shift, or)) */
__device__ __forceinline__ unsigned pack4(int v0, int v1, int v2, int v3)
{
    const unsigned lo = __builtin_amdgcn_perm((unsigned) v1, (unsigned) v0, 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm((unsigned) v3, (unsigned) v2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

/* noise LCG, crt_core.c:359-364 */
__device__ __forceinline__ unsigned lcg_step(unsigned rn) { return LCG_MUL * rn + LCG_ADD; }
__device__ __forceinline__ int noisy(int sample, unsigned rn, int noise)
{
    int s = sample + (((int) ((rn >> 16) & 0xffu) - 0x7f) * noise >> 8);
    return clampi(s, -127, 127);
}
/* LCG state after `idx` steps from rn0; jump16[q] = affine map of 16*q steps */
__device__ __forceinline__ unsigned lcg_at(const uint2 *__restrict__ jump16, unsigned rn0, int idx)
{
    uint2 j = jump16[idx >> 4];
    unsigned rn = j.x * rn0 + j.y;
    for (int k = idx & 15; k > 0; k--) rn = lcg_step(rn);
    return rn;
}

/* ------------------------------------------------------------------------- */
/* M4: blanking / sync / burst skeleton                                        */
/* ------------------------------------------------------------------------- */
/* value of skeleton sample (line n, column t) and whether crt_modulate writes it.
 * RGB systems: crt_ntsc.c:205-252 (+ crt_ntscvhs.c:234-238);  NES: crt_nes.c:81-104,173-178 */
template <class S>
__device__ __forceinline__ bool
skeleton(const crthip_params &P, int n, int t, int field, int inv_phase, int aux, bool nes_setup, int &val)
{
    if constexpr (S::IS_NES) {
        bool written = nes_setup;
        val = S::BLANK;
        if (t >= S::SYNC_BEG && t < (n >= 259 ? S::VS_SEP_END : S::BW_BEG)) val = S::SYNC;
        if (n >= P.yo && n < P.yo + S::LINES && t >= S::CB_BEG && t < S::CB_BEG + CB_SAMPLES) {
            int cb = P.burst[(n % 3 + aux) % 3][t & 3];
            val = (int) (signed char) ((S::BLANK + cb * S::BURST) >> 5);
            written = true;
        }
        return written;
    } else {
        if (n <= 3 || (n >= 7 && n <= 9)) {            /* equalising pulses */
            val = (t < 4 * S::HRES / 100 || (t >= 50 * S::HRES / 100 && t < 54 * S::HRES / 100)) ? S::SYNC : S::BLANK;
            return true;
        }
        if (n >= 4 && n <= 6) {                        /* vertical sync */
            int a = (field == 1 ? 4 : 46) * S::HRES / 100;
            val = (t < a || (t >= 50 * S::HRES / 100 && t < 96 * S::HRES / 100)) ? S::SYNC : S::BLANK;
            return true;
        }
        if (t >= S::AV_BEG) {                          /* active part: only cleared above CRT_TOP */
            val = S::BLANK;
            return n < S::TOP;
        }
        val = S::BLANK;
        if (t >= S::SYNC_BEG && t < S::BW_BEG && n < S::VRES - aux) val = S::SYNC;
        if (t >= S::CB_BEG && t < S::CB_BEG + CB_SAMPLES) {
            int cb = S::PATTERN == 1 ? P.burst[0][(t + inv_phase * 2) & 3] : P.burst[0][t & 3];
            val = (int) (signed char) ((S::BLANK + cb * S::BURST) >> 5);
        }
        return true;
    }
}

/* Drop-in (stage-level) path: write exactly the reference's write-set into analog[]; all other
 * samples keep their contents.  One lane per 16 consecutive samples. */
template <class S>
__global__ void __launch_bounds__(256)
k_template(const crthip_params P, int n_fields, signed char *__restrict__ dst, size_t fstride,
           const crthip_state *__restrict__ state, int nes_setup)
{
    constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_fields * CHUNKS) return;
    const int f = gid / CHUNKS;
    const int q = gid - f * CHUNKS;
    const int idx0 = q * 16;
    const crthip_state st = state[f];
    const int field = st.field & 1;
    const int inv_phase = (field == (st.frame & 1));
    int line = idx0 / S::HRES;
    int t = idx0 - line * S::HRES;
    signed char *out = dst + (size_t) f * fstride;

    /* chunks entirely inside the active rectangle belong to k_active */
    if (t >= P.xo && t + 15 < P.xo + P.destw && line >= P.yo && line < P.yo + P.desth) return;
    int vals[16];
    unsigned wmask = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int v = 0;
        const bool in_field = idx0 + k < S::INPUT_SIZE;
        const bool active = t >= P.xo && t < P.xo + P.destw && line >= P.yo && line < P.yo + P.desth;
        const bool wr = skeleton<S>(P, line, t, field, inv_phase, st.aux, nes_setup != 0, v);
        if (wr && !active && in_field) wmask |= 1u << k;
        vals[k] = v;
        if (++t == S::HRES) { t = 0; line++; }
    }
    if (wmask == 0xffffu) {
        v4i pk;
        pk.x = (int) pack4(vals[0], vals[1], vals[2], vals[3]);
        pk.y = (int) pack4(vals[4], vals[5], vals[6], vals[7]);
        pk.z = (int) pack4(vals[8], vals[9], vals[10], vals[11]);
        pk.w = (int) pack4(vals[12], vals[13], vals[14], vals[15]);
        store16u(out + idx0, pk);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (wmask >> k & 1u) out[idx0 + k] = (signed char) vals[k];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* M5: active video, one lane per destination row                              */
/* ------------------------------------------------------------------------- */
/* v_perm_b32 selector turning a loaded pixel (bytes in memory order, 3-byte formats zero-extended)
 * into 0x??RRGGBB for the 6 byte orders of crt_ntsc.c:278-305 */
__device__ __forceinline__ unsigned input_selector(int format)
{
    switch (format) {
    case CRTHIP_FMT_BGR: case CRTHIP_FMT_BGRA: return 0x03020100u;
    case CRTHIP_FMT_RGB: case CRTHIP_FMT_RGBA: return 0x03000102u;
    case CRTHIP_FMT_ARGB: return 0x00010203u;
    default /* ABGR */:   return 0x00030201u;
    }
}

/* NES PPU square wave, crt_nes.c:21-61 */
__device__ __forceinline__ int ppu_level(int p, int phase)
{
    const int hue = p & 15;
    if (hue >= 14) return 0;
    int high = ((hue + phase) % 12) < 6;
    if (hue == 0) high = 1;
    if (hue == 13) high = 0;
    /* active[] = {0300,0100,0500,0400,0600,0200}: emphasis bits attenuating this phase */
    const int slot = (phase >> 1) % 6;
    const int mask = slot == 0 ? 0300 : slot == 1 ? 0100 : slot == 2 ? 0500 : slot == 3 ? 0400 : slot == 4 ? 0600 : 0200;
    const int emph = (p & 0700 & mask) != 0;
    const int lum = (p >> 4) & 3;
    /* IRE[(high<<3) + (emph<<2) + lum] */
    int v;
    if (high) {
        v = emph ? (lum == 0 ? 26951 : lum == 1 ? 52181 : 83721)
                 : (lum == 0 ? 43581 : lum == 1 ? 75693 : 112965);
    } else {
        v = emph ? (lum == 0 ? -17203 : lum == 1 ? -8028 : lum == 2 ? 19497 : 57342)
                 : (lum == 0 ? -12042 : lum == 1 ? 0 : lum == 2 ? 34406 : 81427);
    }
    return v;
}

/* FAST: 24-bit multiplies (always in range for the IIRs and the carrier products: 8-bit pixels
 * bound every state; `white` and `noise` are range-checked on the host).
 * IN4: 4-byte input pixels moved through a cooperative LDS tile (see the comment above k_decode);
 * otherwise (3-byte formats, tiny images) each lane reads its own pixels bytewise.
 * The produced samples always leave through a cooperative LDS tile. */
/* ACT = dwords per row and tile (ACT pixels in, 4*ACT samples out): 32 moves full 128-byte lines per row
 * piece group, 16 halves the LDS footprint (more waves per SIMD) -- chosen by input width at launch */
/* CLAMP: output is inp[] (fused path), i.e. the +-127 clamp of crt_core.c:363-364 applies even when
 * no noise is added (only matters for NES, whose samples can be -128) */
template <class S, bool NOISE, bool FAST, bool IN4, bool CLAMP, int ACT>
__global__ void __launch_bounds__(64)
k_active(const crthip_params P, int n_fields, const unsigned char *__restrict__ images, size_t istride,
         signed char *__restrict__ dst, size_t fstride, const crthip_state *__restrict__ state,
         const uint2 *__restrict__ jump16)
{
    constexpr int AC_TILE = ACT, AC_STRIDE = ACT + 1, AC_PIECES = ACT / 4;   /* 16-byte pieces per tile row */
    constexpr int AC_ROWS = 64 / AC_PIECES, AC_SHIFT = ACT == 32 ? 5 : 4;      /* rows per load instruction */
    __shared__ unsigned s_pix[64 * AC_STRIDE];
    __shared__ unsigned s_out[64 * AC_STRIDE];
    __shared__ unsigned long long s_src[64], s_dst[64];

    const int lane = threadIdx.x;
    const int gid = blockIdx.x * 64 + lane;
    const int rows = P.desth;
    const bool live = gid < n_fields * rows;
    const int f = live ? gid / rows : 0;
    const int y = live ? gid - f * rows : 0;
    const crthip_state st = state[f];
    const unsigned char *img = images + (size_t) f * istride;
    const int start = (y + P.yo) * S::HRES + P.xo;
    unsigned rn = 0;
    if (NOISE) rn = lcg_at(jump16, (unsigned) st.rn, start);

    const int w = P.w, destw = P.destw;
    const int qstep = w / destw, rstep = w - qstep * destw;   /* column = floor(x*w/destw), incrementally */
    int col = 0, err = 0;                                     /* wave-uniform */
    const int ngroups = (destw + 3) >> 2;

    /* per-row source / destination, published to the whole wave */
    int sy;
    if constexpr (S::IS_NES) {
        sy = (y * P.h) / S::LINES;                            /* crt_nes.c:165-168 */
        if (sy >= P.h) sy = P.h;
        if (sy < 0) sy = 0;
    } else {
        const int field = st.field & 1;
        const int field_offset = (field * P.h + P.desth) / P.desth / 2;
        sy = (y * P.h) / P.desth + field_offset;              /* crt_ntsc.c:258-263 */
        if (sy >= P.h) sy = P.h;                              /* (sic) */
    }
    const int in_bpp = S::IS_NES ? 2 : P.in_bpp;
    const unsigned char *row = img + (size_t) sy * w * in_bpp;
    s_src[lane] = (unsigned long long) row;
    s_dst[lane] = live ? (unsigned long long) (dst + (size_t) f * fstride + start) : 0ull;
    __syncthreads();

    /* drain the sample tile: dwords [g0, g0+ng) of every row = samples [4*g0, ...) clipped to destw */
    auto drain = [&](int g0, int ng) {
        __syncthreads();
        const int orow = lane / AC_PIECES, piece = lane % AC_PIECES;   /* 16 bytes per piece */
        const int first = (g0 + piece * 4) * 4;               /* first sample of my piece */
        const int nbytes = destw - first < 16 ? destw - first : 16;
#pragma unroll 2
        for (int i = 0; i < AC_PIECES; i++) {
            const int r = i * AC_ROWS + orow;
            const unsigned long long d = s_dst[r];
            if (d != 0 && piece * 4 < ng && nbytes > 0) {
                const unsigned *sp = s_out + r * AC_STRIDE + piece * 4;
                v4i o; o.x = (int) sp[0]; o.y = (int) sp[1]; o.z = (int) sp[2]; o.w = (int) sp[3];
                if (nbytes == 16) {
                    gstore16u(d + first, o);
                } else {
                    const int wds[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        if (k < nbytes) gstore8(d + first + k, (unsigned) (wds[k >> 2] >> (8 * (k & 3))));
                    }
                }
            }
        }
        __syncthreads();
    };

    if constexpr (S::IS_NES) {
        /* crt_nes.c:162-193 */
        const unsigned short *prow = (const unsigned short *) row;
        int phase = 4 * ((y + P.yo + st.aux) % 3);           /* phasetab {0,4,8} */
        for (int g = 0; g < ngroups; g++) {
            unsigned pack = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = 4 * g + k;
                if (x < destw) {
                    int p = prow[col];
                    int ire = S::BLACK + P.black_point;
                    ire += ppu_level(p, phase + 0);
                    ire += ppu_level(p, phase + 1);
                    ire += ppu_level(p, phase + 2);
                    ire += ppu_level(p, phase + 3);
                    ire = (ire * P.white_point / 100) >> 12;
                    ire = (int) (signed char) ire;
                    if (NOISE) { rn = lcg_step(rn); ire = noisy(ire, rn, P.noise); }
                    else if (CLAMP) ire = clampi(ire, -127, 127);
                    pack |= (unsigned) (ire & 255) << (8 * k);
                    phase += 3;
                    col += qstep; err += rstep;
                    if (err >= destw) { err -= destw; col++; }
                }
            }
            s_out[lane * AC_STRIDE + (g & (AC_TILE - 1))] = pack;
            if ((g & (AC_TILE - 1)) == AC_TILE - 1 || g == ngroups - 1) drain(g & ~(AC_TILE - 1), (g & (AC_TILE - 1)) + 1);
        }
    } else {
        /* crt_ntsc.c:254-324 */
        const int field = st.field & 1;
        const int inv_phase = (field == (st.frame & 1));
        const int ph = (S::PATTERN == 1 && (inv_phase & 1)) ? -1 : 1;
        /* (h * ph) * cc == h * (ph * cc) in wrapping arithmetic; xo is a multiple of 4 (crt_ntsc.c:203)
         * so the carrier phase (x + xo) % 4 is x & 3 */
        const int cI0 = ph * P.modI[0], cI1 = ph * P.modI[1], cI2 = ph * P.modI[2], cI3 = ph * P.modI[3];
        const int cQ0 = ph * P.modQ[0], cQ1 = ph * P.modQ[1], cQ2 = ph * P.modQ[2], cQ3 = ph * P.modQ[3];
        const int cy_ = P.iir_c[0], ci_ = P.iir_c[1], cq_ = P.iir_c[2];
        const unsigned isel = input_selector(P.format);
        const int white = P.white, ire_base = P.ire_base, noise = P.noise;
        int hy = 0, hi = 0, hq = 0;

        /* IN4 pixel tiles: 32 pixels (128 bytes) per row; pieces of 16 bytes, 8 per row, 8 rows per
         * load instruction.  `have` = tile in LDS, `stage[]` = tile have+1 in flight / in registers.
         * A piece that would run past the row end is moved back to the row's last 16 bytes, so
         * nothing beyond the image is touched (w >= 4). */
        const int prow_ = lane / AC_PIECES, piece = lane % AC_PIECES;
        const int last_tile = (w - 1) >> AC_SHIFT;
        const int row_bytes = w * 4;
        v4i stage[AC_PIECES];
        auto piece_offset = [&](int tile) {
            int off = tile * (AC_TILE * 4) + piece * 16;
            return off > row_bytes - 16 ? row_bytes - 16 : off;
        };
        auto fetch = [&](int tile) {
            const int off = piece_offset(tile);
#pragma unroll
            for (int i = 0; i < AC_PIECES; i++) stage[i] = gload16u(s_src[i * AC_ROWS + prow_] + off);
        };
        auto stash = [&](int tile) {
            /* dword index inside the tile where my (possibly moved-back) piece belongs; moved-back
             * pieces of several lanes overlap and carry identical bytes */
            const int dw0 = (piece_offset(tile) - tile * (AC_TILE * 4)) >> 2;     /* may be negative for a moved-back piece */
            __syncthreads();
#pragma unroll
            for (int i = 0; i < AC_PIECES; i++) {
                unsigned *d = s_pix + (i * AC_ROWS + prow_) * AC_STRIDE;
                if (dw0 + 0 >= 0) d[dw0 + 0] = (unsigned) stage[i].x;
                if (dw0 + 1 >= 0) d[dw0 + 1] = (unsigned) stage[i].y;
                if (dw0 + 2 >= 0) d[dw0 + 2] = (unsigned) stage[i].z;
                if (dw0 + 3 >= 0) d[dw0 + 3] = (unsigned) stage[i].w;
            }
            __syncthreads();
        };
        int have = 0;
        if (IN4) {
            fetch(0);
            stash(0);
            if (last_tile > 0) fetch(1);
        }
        const int noise127 = 0x7f * noise;
        for (int g = 0; g < ngroups; g++) {
            int smp[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = 4 * g + k;
                if (x < destw) {
                    unsigned pixel;
                    if (IN4) {
                        const int need = col >> AC_SHIFT;          /* wave-uniform */
                        if (need != have) {
                            if (need != have + 1) fetch(need);      /* only when w > 32*destw */
                            stash(need);
                            have = need;
                            if (need < last_tile) fetch(need + 1);
                        }
                        pixel = s_pix[lane * AC_STRIDE + (col & (AC_TILE - 1))];
                    } else {
                        const unsigned char *pp = row + (size_t) col * in_bpp;
                        pixel = (unsigned) pp[0] | (unsigned) pp[1] << 8 | (unsigned) pp[2] << 16;
                        if (in_bpp == 4) pixel |= (unsigned) pp[3] << 24;
                    }
                    const unsigned rgb = __builtin_amdgcn_perm(pixel, pixel, isel);
                    const int r = (rgb >> 16) & 255, gg = (rgb >> 8) & 255, b = rgb & 255;
                    const int fy = (19595 * r + 38470 * gg + 7471 * b) >> 14;
                    const int fi = (39059 * r - 18022 * gg - 21103 * b) >> 14;
                    const int fq = (13894 * r - 34275 * gg + 20382 * b) >> 14;
                    hy += mulq<FAST>(fy - hy, cy_) >> 11;           /* iirf, crt_ntsc.c:117-126 */
                    hi += mulq<FAST>(fi - hi, ci_) >> 11;
                    hq += mulq<FAST>(fq - hq, cq_) >> 11;
                    const int mi = mulq<FAST>(hi, k == 0 ? cI0 : k == 1 ? cI1 : k == 2 ? cI2 : cI3) >> 4;
                    const int mq = mulq<FAST>(hq, k == 0 ? cQ0 : k == 1 ? cQ1 : k == 2 ? cQ2 : cQ3) >> 4;
                    int ire = ire_base + (mulq<FAST>(hy + mi + mq, white) >> 10);
                    ire = clampi(ire, 0, 110);
                    if (NOISE) {
                        rn = lcg_step(rn);
                        /* (byte - 0x7f) * noise, distributed: the byte select rides on the multiply (SDWA) */
                        ire = clampi(ire + ((mulq<FAST>((int) ((rn >> 16) & 0xffu), noise) - noise127) >> 8), -127, 127);
                    }
                    smp[k] = ire;
                    col += qstep; err += rstep;
                    if (err >= destw) { err -= destw; col++; }
                }
            }
            s_out[lane * AC_STRIDE + (g & (AC_TILE - 1))] = pack4(smp[0], smp[1], smp[2], smp[3]);
            if ((g & (AC_TILE - 1)) == AC_TILE - 1 || g == ngroups - 1) drain(g & ~(AC_TILE - 1), (g & (AC_TILE - 1)) + 1);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* M5, NES flavour (crt_nes.c:162-193) with a lookup table                     */
/* ------------------------------------------------------------------------- */
/* A composite sample of the NES is  ((BLACK + black_point + sum_{k<4} square(p, phase+k)) * white_point / 100) >> 12
 * truncated to a signed char.  square() depends on the phase only through (hue+phase)%12 and (phase>>1)%6,
 * both 12-periodic, so the sample is a function of (9-bit pixel, phase mod 12): 512 x 12 bytes, rebuilt per
 * launch because it contains the black / white point knobs. */
#define NES_TAB_SIZE (512 * 12)
template <class S>
__global__ void k_nes_table(const crthip_params P, signed char *tab)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NES_TAB_SIZE) return;
    const int p = idx / 12, ph = idx - p * 12;
    int ire = S::BLACK + P.black_point;
    ire += ppu_level(p, ph + 0);
    ire += ppu_level(p, ph + 1);
    ire += ppu_level(p, ph + 2);
    ire += ppu_level(p, ph + 3);
    ire = (ire * P.white_point / 100) >> 12;
    tab[idx] = (signed char) ire;
}

template <class S, bool NOISE, bool CLAMP, int ACT>
__global__ void __launch_bounds__(64)
k_active_nes(const crthip_params P, int n_fields, const unsigned char *__restrict__ images, size_t istride,
             signed char *__restrict__ dst, size_t fstride, const crthip_state *__restrict__ state,
             const uint2 *__restrict__ jump16, const signed char *__restrict__ tab)
{
    constexpr int AC_TILE = ACT, AC_STRIDE = ACT + 1, AC_PIECES = ACT / 4;
    constexpr int AC_ROWS = 64 / AC_PIECES, AC_SHIFT = ACT == 32 ? 5 : 4;
    __shared__ unsigned s_pix[64 * AC_STRIDE];
    __shared__ unsigned s_out[64 * AC_STRIDE];
    __shared__ unsigned long long s_src[64], s_dst[64];
    __shared__ unsigned s_tab[NES_TAB_SIZE / 4];

    const int lane = threadIdx.x;
    const int gid = blockIdx.x * 64 + lane;
    const int rows = S::LINES;
    const bool live = gid < n_fields * rows;
    const int f = live ? gid / rows : 0;
    const int y = live ? gid - f * rows : 0;
    const crthip_state st = state[f];
    const unsigned char *img = images + (size_t) f * istride;
    const int start = (y + P.yo) * S::HRES + P.xo;
    unsigned rn = 0;
    if (NOISE) rn = lcg_at(jump16, (unsigned) st.rn, start);
    for (int i = lane; i < NES_TAB_SIZE / 4; i += 64) s_tab[i] = ((const unsigned *) tab)[i];

    const int w = P.w, destw = P.destw;
    const int qstep = w / destw, rstep = w - qstep * destw;
    int col = 0, err = 0;
    const int ngroups = (destw + 3) >> 2;
    int sy = (y * P.h) / S::LINES;                              /* crt_nes.c:165-168 */
    if (sy >= P.h) sy = P.h;
    if (sy < 0) sy = 0;
    s_src[lane] = (unsigned long long) (img + (size_t) sy * w * 2);
    s_dst[lane] = live ? (unsigned long long) (dst + (size_t) f * fstride + start) : 0ull;
    __syncthreads();

    auto drain = [&](int g0, int ng) {
        __syncthreads();
        const int orow = lane / AC_PIECES, piece = lane % AC_PIECES;
        const int first = (g0 + piece * 4) * 4;
        const int nbytes = destw - first < 16 ? destw - first : 16;
#pragma unroll 2
        for (int i = 0; i < AC_PIECES; i++) {
            const int r = i * AC_ROWS + orow;
            const unsigned long long d = s_dst[r];
            if (d != 0 && piece * 4 < ng && nbytes > 0) {
                const unsigned *sp = s_out + r * AC_STRIDE + piece * 4;
                v4i o; o.x = (int) sp[0]; o.y = (int) sp[1]; o.z = (int) sp[2]; o.w = (int) sp[3];
                if (nbytes == 16) {
                    gstore16u(d + first, o);
                } else {
                    const int wds[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        if (k < nbytes) gstore8(d + first + k, (unsigned) (wds[k >> 2] >> (8 * (k & 3))));
                    }
                }
            }
        }
        __syncthreads();
    };
    /* pixel tiles: AC_TILE dwords = 2*AC_TILE PPU pixels per row (see k_active) */
    const int prow_ = lane / AC_PIECES, piece = lane % AC_PIECES;
    const int row_bytes = w * 2;
    const int last_tile = (row_bytes - 1) / (AC_TILE * 4);
    v4i stage[AC_PIECES];
    auto piece_offset = [&](int tile) {
        int off = tile * (AC_TILE * 4) + piece * 16;
        return off > row_bytes - 16 ? row_bytes - 16 : off;
    };
    auto fetch = [&](int tile) {
        const int off = piece_offset(tile);
#pragma unroll
        for (int i = 0; i < AC_PIECES; i++) stage[i] = gload16u(s_src[i * AC_ROWS + prow_] + off);
    };
    auto stash = [&](int tile) {
        const int dw0 = (piece_offset(tile) - tile * (AC_TILE * 4)) >> 2;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AC_PIECES; i++) {
            unsigned *d = s_pix + (i * AC_ROWS + prow_) * AC_STRIDE;
            if (dw0 + 0 >= 0) d[dw0 + 0] = (unsigned) stage[i].x;
            if (dw0 + 1 >= 0) d[dw0 + 1] = (unsigned) stage[i].y;
            if (dw0 + 2 >= 0) d[dw0 + 2] = (unsigned) stage[i].z;
            if (dw0 + 3 >= 0) d[dw0 + 3] = (unsigned) stage[i].w;
        }
        __syncthreads();
    };
    int have = 0;
    fetch(0);
    stash(0);
    if (last_tile > 0) fetch(1);

    int ph = 4 * ((y + P.yo + st.aux) % 3);                    /* phasetab {0,4,8}; advances by 3 per sample, mod 12 */
    const signed char *tb = (const signed char *) s_tab;
    for (int g = 0; g < ngroups; g++) {
        int smp[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = 4 * g + k;
            if (x < destw) {
                const int need = col >> (AC_SHIFT + 1);         /* wave-uniform */
                if (need != have) {
                    if (need != have + 1) fetch(need);
                    stash(need);
                    have = need;
                    if (need < last_tile) fetch(need + 1);
                }
                const unsigned dw = s_pix[lane * AC_STRIDE + ((col >> 1) & (AC_TILE - 1))];
                const int p = (int) ((col & 1) ? dw >> 16 : dw & 0xffffu);
                /* data[] is unsigned short (crt_nes.h:133): the reference's table walks only look at bits 0-8 */
                int ire = tb[(p & 511) * 12 + ph];
                if (NOISE) { rn = lcg_step(rn); ire = noisy(ire, rn, P.noise); }
                else if (CLAMP) ire = clampi(ire, -127, 127);
                smp[k] = ire;
                ph += 3;
                if (ph >= 12) ph -= 12;
                col += qstep; err += rstep;
                if (err >= destw) { err -= destw; col++; }
            }
        }
        s_out[lane * AC_STRIDE + (g & (AC_TILE - 1))] = pack4(smp[0], smp[1], smp[2], smp[3]);
        if ((g & (AC_TILE - 1)) == AC_TILE - 1 || g == ngroups - 1) drain(g & ~(AC_TILE - 1), (g & (AC_TILE - 1)) + 1);
    }
}

/* The clean skeleton (blanking / sync / burst, 0 where crt_modulate writes nothing) of a whole field depends
 * only on a handful of per-field inputs: RGB systems (field, frame parity) -> 4 variants; NES the dot crawl
 * offset mod 3 -> 3 variants.  k_skeleton writes the variants once per launch (a few fields' worth of work),
 * k_margin then only copies 16 bytes and adds the channel noise. */
#define SKEL_VARIANTS 4
template <class S>
__global__ void __launch_bounds__(256)
k_skeleton(const crthip_params P, signed char *__restrict__ skel, size_t fstride)
{
    constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= SKEL_VARIANTS * CHUNKS) return;
    const int var = gid / CHUNKS;
    const int idx0 = (gid - var * CHUNKS) * 16;
    const int field = S::IS_NES ? 0 : var >> 1;
    const int inv_phase = S::IS_NES ? 0 : (field == (var & 1));
    const int aux = S::IS_NES ? var : 0;                       /* VHS: the aberration band is patched in by k_margin */
    int line = idx0 / S::HRES;
    int t = idx0 - line * S::HRES;
    int vals[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int v = 0;
        if (!skeleton<S>(P, line, t, field, inv_phase, aux, true, v)) v = 0;
        vals[k] = v;
        if (++t == S::HRES) { t = 0; line++; }
    }
    v4i pk;
    pk.x = (int) pack4(vals[0], vals[1], vals[2], vals[3]);
    pk.y = (int) pack4(vals[4], vals[5], vals[6], vals[7]);
    pk.z = (int) pack4(vals[8], vals[9], vals[10], vals[11]);
    pk.w = (int) pack4(vals[12], vals[13], vals[14], vals[15]);
    store16u(skel + (size_t) var * fstride + idx0, pk);        /* the last chunk runs into the field's slack */
}

/* Fused path only: everything OUTSIDE the active rectangle of a field that started from a clean
 * analog[] -- skeleton value (or 0) plus channel noise, written straight into inp[].  The complement
 * of the rectangle in flat sample order is
 *     head   [0, S0)                                   S0 = yo*HRES + xo
 *     gap y  [S0 + y*HRES + destw, S0 + (y+1)*HRES)    y = 0 .. desth-2
 *     tail   [S0 + (desth-1)*HRES + destw, INPUT_SIZE)
 * and each lane takes one run of up to 16 samples of it: 16 bytes of the cached skeleton variant
 * (k_skeleton), + noise (LCG state by the 16-step jump table and a 16-entry table for the remainder). */
template <class S, bool NOISE>
__global__ void __launch_bounds__(256)
k_margin(const crthip_params P, int n_fields, signed char *__restrict__ dst, size_t fstride,
         const crthip_state *__restrict__ state, const uint2 *__restrict__ jump16, const uint2 *__restrict__ jump1,
         const signed char *__restrict__ skel, int head_chunks, int gap_chunks, int tail_chunks)
{
    const int per_field = head_chunks + (P.desth - 1) * gap_chunks + tail_chunks;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_fields * per_field) return;
    const int f = gid / per_field;
    int q = gid - f * per_field;
    const int s0 = P.yo * S::HRES + P.xo;
    const int gap_len = S::HRES - P.destw;
    int idx0, len;
    if (q < head_chunks) {
        idx0 = q * 16;
        len = s0 - idx0;
    } else if (q < head_chunks + (P.desth - 1) * gap_chunks) {
        q -= head_chunks;
        const int yy = q / gap_chunks, c = q - yy * gap_chunks;
        idx0 = s0 + yy * S::HRES + P.destw + c * 16;
        len = gap_len - c * 16;
    } else {
        q -= head_chunks + (P.desth - 1) * gap_chunks;
        idx0 = s0 + (P.desth - 1) * S::HRES + P.destw + q * 16;
        len = S::INPUT_SIZE - idx0;
    }
    if (len > 16) len = 16;
    const crthip_state *st = state + f;
    const int aux = st->aux;
    const int var = S::IS_NES ? aux % 3 : ((st->field & 1) << 1) | (st->frame & 1);
    signed char *out = dst + (size_t) f * fstride;
    const v4i sk = load16u(skel + (size_t) var * fstride + idx0);
    int wds[4] = { sk.x, sk.y, sk.z, sk.w };
    if constexpr (S::IS_VHS) {
        /* no sync pulse inside the aberration band (crt_ntscvhs.c:234-238): lines n >= VRES - aux */
        const int line0 = idx0 / S::HRES, t0 = idx0 - line0 * S::HRES;
        if (aux > 0 && line0 + 1 >= S::VRES - aux) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                int t = t0 + k, n = line0;
                if (t >= S::HRES) { t -= S::HRES; n++; }
                if (n >= S::VRES - aux && n >= 10 && t >= S::SYNC_BEG && t < S::BW_BEG)
                    wds[k >> 2] = (wds[k >> 2] & ~(0xff << (8 * (k & 3)))) | ((S::BLANK & 0xff) << (8 * (k & 3)));
            }
        }
    }
    v4i pk;
    if (NOISE) {
        unsigned rn;
        {
            const uint2 j = jump16[idx0 >> 4], r = jump1[idx0 & 15];
            rn = r.x * (j.x * (unsigned) st->rn + j.y) + r.y;
        }
        int vals[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            rn = lcg_step(rn);
            vals[k] = noisy((wds[k >> 2] << (24 - 8 * (k & 3))) >> 24, rn, P.noise);
        }
        pk.x = (int) pack4(vals[0], vals[1], vals[2], vals[3]);
        pk.y = (int) pack4(vals[4], vals[5], vals[6], vals[7]);
        pk.z = (int) pack4(vals[8], vals[9], vals[10], vals[11]);
        pk.w = (int) pack4(vals[12], vals[13], vals[14], vals[15]);
    } else {
        /* noise 0: crt_core.c:362-364 still clamps to +-127, which no skeleton value exceeds */
        pk.x = wds[0]; pk.y = wds[1]; pk.z = wds[2]; pk.w = wds[3];
    }
    if (len == 16) {
        store16u(out + idx0, pk);
    } else {
        const int o4[4] = { pk.x, pk.y, pk.z, pk.w };
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < len) out[idx0 + k] = (signed char) (o4[k >> 2] >> (8 * (k & 3)));
        }
    }
    if (gid - f * per_field == 0) {
        /* mirror of the struct members behind inp[] (see CRTHIP_TAIL) */
        signed char *tail = out + S::INPUT_SIZE;
        store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
    }
}

/* ------------------------------------------------------------------------- */
/* D1: channel noise (elementwise, 16 samples per lane)                        */
/* ------------------------------------------------------------------------- */
template <class S>
__global__ void __launch_bounds__(256)
k_noise(const crthip_params P, int n_fields, const signed char *__restrict__ analog,
        signed char *__restrict__ inp, size_t fstride, const crthip_state *__restrict__ state,
        const uint2 *__restrict__ jump16)
{
    constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_fields * CHUNKS) return;
    const int f = gid / CHUNKS;
    const int q = gid - f * CHUNKS;
    const signed char *src = analog + (size_t) f * fstride + q * 16;
    signed char *dst = inp + (size_t) f * fstride + q * 16;
    const uint2 j = jump16[q];
    unsigned rn = j.x * (unsigned) state[f].rn + j.y;
    const v4i in = load16u(src);
    const int wds[4] = { in.x, in.y, in.z, in.w };
    int outw[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        unsigned o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int s = (wds[d] << (24 - 8 * k)) >> 24;
            rn = lcg_step(rn);
            o |= (unsigned) (noisy(s, rn, P.noise) & 255) << (8 * k);
        }
        outw[d] = (int) o;
    }
    if (q * 16 + 16 <= S::INPUT_SIZE) {
        v4i o4; o4.x = outw[0]; o4.y = outw[1]; o4.z = outw[2]; o4.w = outw[3];
        store16u(dst, o4);
    } else {
        for (int k = 0; q * 16 + k < S::INPUT_SIZE; k++) dst[k] = (signed char) (outw[k >> 2] >> (8 * (k & 3)));
    }
    if (q == 0) {
        signed char *tail = inp + (size_t) f * fstride + S::INPUT_SIZE;
        store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
    }
}

/* ------------------------------------------------------------------------- */
/* D1, VHS flavour: noise from the C library's rand() stream                   */
/* ------------------------------------------------------------------------- */
/*
 * crt_core.c:343-357.  rand() is modelled as glibc's y[n] = y[n-31] + y[n-3] (crt_setup.c).  Calls:
 * #0 picks the band's phase (`line`); sample i then makes call A_i (its noise value) and call B_i
 * (always), plus call C_i only when the first half of the && is true:
 *     c1(i, B) = i > INPUT_SIZE - HRES*(6 + B%20)   <=>   6 + B%20 > floor((INPUT_SIZE - i) / HRES)
 * which can only happen for i > I0 = INPUT_SIZE - 25*HRES.  So
 *   - samples [0, T0), T0 = a multiple of VHS_CHUNK just below I0, use calls 1+2i, 2+2i: k_vhs_noise,
 *     PARALLEL, one lane per VHS_CHUNK samples; the lane's 31-value history at its first call K comes
 *     from the field's base history by the jump  y[K+j] = sum_m c_K[m] * y[m+j]
 *     (c_K = x^K mod x^31-x^28-1, host-made table `rows`);
 *   - samples [T0, INPUT_SIZE) have a data-dependent call count: k_vhs_tail, one wave per field,
 *     speculative block walk (see there); it hands back the final history and rn.
 */
#define VHS_CHUNK 124                      /* samples per lane in the parallel region = 248 calls = 8 * 31 (248: measured slower) */
#define VHS_BLK   43                       /* calls per lane in the tail's window: 64 * 43 >= 3 * HRES + 3 */

/* first sample of the tail: a chunk boundary with at least 16 samples (>= 31 calls) before I0 + 1 */
__host__ __device__ constexpr int vhs_tail_start(int input_size, int hres)
{
    return (input_size - 25 * hres + 1 - 16) / VHS_CHUNK * VHS_CHUNK;
}

__device__ __forceinline__ int dev_sine_q1(int a)
{
    /* crt_core.c:19-39 */
    const int knots[18] = { 0, 3208, 6392, 9512, 12536, 15440, 18200, 20784, 23168,
                            25328, 27240, 28896, 30272, 31352, 32136, 32608, 32768, 32608 };
    const int k = (a >> 8) & 255, t = a & 255;
    int lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < 17; q++) {
        if (k == q) { lo = knots[q]; hi = knots[q + 1]; }
    }
    return lo + (((hi - lo) * t) >> 8);
}
/* cosine only (crt_core.c:42-61), 14-bit angle */
__device__ __forceinline__ int dev_cos14(int n)
{
    n &= 16383;
    const int a = n & 8191;
    int cs = a < 4096 ? dev_sine_q1(4096 - a) : -dev_sine_q1(a - 4096);
    if (n & 8192) cs = -cs;
    return cs;
}

/* Parallel region.  A wave's 64 chunks are (mostly) one contiguous 15872-byte run of the field: it is moved
 * through an LDS tile of 64 x 62 dwords with coalesced 256-byte requests (lane-per-chunk byte accesses cost
 * 12x the algorithmic HBM write traffic); the lane's own dwords sit at an odd stride = conflict-free. */
template <class S>
__global__ void __launch_bounds__(64)
k_vhs_noise(const crthip_params P, int n_fields, const signed char *__restrict__ analog,
            signed char *__restrict__ inp, size_t fstride,
            const unsigned *__restrict__ hist, const unsigned *__restrict__ rows, int chunks_a)
{
    constexpr int DW = VHS_CHUNK / 4;                            /* dwords per chunk */
    constexpr int DWS = DW | 1;                                  /* odd LDS stride: conflict-free lane-per-chunk access */
    static_assert((2 * VHS_CHUNK) % 31 == 0 && VHS_CHUNK % 4 == 0, "static ring index / dword packing");
    __shared__ unsigned s_t[64 * DWS];
    __shared__ unsigned long long s_off[64];
    const int lane = threadIdx.x;
    const int gid = blockIdx.x * 64 + lane;
    const bool live = gid < n_fields * chunks_a;
    const int f = live ? gid / chunks_a : 0;
    const int q = live ? gid - f * chunks_a : 0;
    s_off[lane] = live ? (unsigned long long) f * fstride + (unsigned long long) q * VHS_CHUNK : ~0ull;
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < DW; it++) {
        const int n = it * 64 + lane, owner = n / DW, d = n - owner * DW;
        const unsigned long long off = s_off[owner];
        s_t[owner * DWS + d] = off != ~0ull ? *(const unsigned *) (analog + off + 4 * d) : 0u;
    }
    const unsigned *h = hist + (size_t) f * 32;
    /* base sequence z[0..60]: the history and the next 30 values */
    unsigned z[61];
#pragma unroll
    for (int j = 0; j < 31; j++) z[j] = h[j];
#pragma unroll
    for (int j = 31; j < 61; j++) z[j] = z[j - 31] + z[j - 3];
    /* history of call K = 1 + 2 * VHS_CHUNK * q */
    const unsigned *c = rows + (size_t) q * 31;
    unsigned w[31];
#pragma unroll
    for (int j = 0; j < 31; j++) w[j] = 0;
#pragma unroll
    for (int m = 0; m < 31; m++) {
        const unsigned cm = c[m];
#pragma unroll
        for (int j = 0; j < 31; j++) w[j] += cm * z[m + j];
    }
    const int noise = P.noise;
    __syncthreads();
    /* 2 calls per sample; ring index = call % 31 is static */
    unsigned *mine = s_t + lane * DWS;
    unsigned in4 = 0, out4 = 0;
#pragma unroll
    for (int t = 0; t < 2 * VHS_CHUNK; t++) {
        const unsigned v = w[t % 31] + w[(t + 28) % 31];
        w[t % 31] = v;
        if ((t & 1) == 0) {                                      /* call A of sample t/2 */
            const int k = t / 2;
            if ((k & 3) == 0) in4 = mine[k >> 2];
            const int rn = (int) (v >> 1);
            const int a = (int) (in4 << (24 - 8 * (k & 3))) >> 24;
            const int sv = clampi(a + ((((rn >> 16) & 0xff) - 0x7f) * noise >> 8), -127, 127);
            out4 = (k & 3) == 0 ? (unsigned) (sv & 255) : out4 | (unsigned) (sv & 255) << (8 * (k & 3));
            if ((k & 3) == 3) mine[k >> 2] = out4;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < DW; it++) {
        const int n = it * 64 + lane, owner = n / DW, d = n - owner * DW;
        const unsigned long long off = s_off[owner];
        if (off != ~0ull) *(unsigned *) (inp + off + 4 * d) = s_t[owner * DWS + d];
    }
}

/*
 * The tail: samples [T0, INPUT_SIZE), ONE WAVE PER FIELD.  Sample i starts at call position pos_i and
 * pos_{i+1} = pos_i + 2 + c1(i, call[pos_i + 1]) -- a serial chain, but c1 depends on i only through
 * k = floor((INPUT_SIZE - i) / HRES), constant over a SEGMENT of at most HRES samples.  Per segment:
 *   1. the next 64*43 calls (more than a segment can consume) are generated in parallel: lane b jumps
 *      to call 43*b of the window (x^(43b), 961 multiply-adds) and produces its block of 43 -> LDS;
 *   2. every lane walks its own block for each of the three possible entry offsets (the first call of
 *      a sample inside a block is its call 0, 1 or 2) -> exit offset + sample count;
 *   3. the 64 results are chained on the scalar unit (v_readlane), giving each block its
 *      real entry offset and the index of its first sample;
 *   4. every lane walks its block once more, now producing samples (through LDS byte staging); the
 *      lane that meets the segment's last sample publishes the next window's start and `rn`.
 */
template <class S>
__global__ void __launch_bounds__(64)
k_vhs_tail(const crthip_params P, int n_fields, const signed char *__restrict__ analog,
           signed char *__restrict__ inp, size_t fstride, crthip_state *__restrict__ state,
           unsigned *__restrict__ hist, const unsigned *__restrict__ tail_row, const unsigned *__restrict__ blk_rows)
{
    constexpr int N = S::INPUT_SIZE, H = S::HRES, B = VHS_BLK;
    constexpr int T0 = vhs_tail_start(N, H);
    constexpr int NB = (H + 63) / 64;                              /* bytes per lane and segment */
    static_assert(64 * B >= 3 * H + 3, "window too small for a segment");
    __shared__ unsigned s_y[64 * B + 8];                           /* the window's raw generator values */
    __shared__ unsigned s_h[64];                                   /* 31-value history in front of the window; base sequence */
    __shared__ unsigned s_misc[2];
    __shared__ signed char s_a[NB * 64], s_o[NB * 64];
    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= n_fields) return;
    unsigned *h = hist + (size_t) f * 32;
    const signed char *src = analog + (size_t) f * fstride;
    signed char *dst = inp + (size_t) f * fstride;
    const int noise = P.noise;

    /* the field's base sequence -> the history in front of call 1 + 2*T0 (lane j computes element j) */
    int vhs_line;
    {
        unsigned zf[61];
#pragma unroll
        for (int j = 0; j < 31; j++) zf[j] = h[j];
#pragma unroll
        for (int j = 31; j < 61; j++) zf[j] = zf[j - 31] + zf[j - 3];
        vhs_line = (int) ((zf[31] >> 1) & 7u) - 4 + 14;            /* call #0, crt_core.c:344 */
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 61; j++) s_y[j] = zf[j];
        }
        __syncthreads();
        unsigned acc = 0;
        const int j = lane < 31 ? lane : 0;
        for (int m = 0; m < 31; m++) acc += tail_row[m] * s_y[m + j];
        __syncthreads();
        if (lane < 31) s_h[lane] = acc;
    }
    unsigned cb[31];                                               /* x^(43*lane) */
#pragma unroll
    for (int m = 0; m < 31; m++) cb[m] = blk_rows[m * 64 + lane];
    __syncthreads();

    int seg_start = T0;
    while (seg_start < N) {
        const int kseg = (N - seg_start) / H;
        int seg_end = N - H * kseg;                                /* last sample with floor((N - i) / H) == kseg */
        if (seg_end > N - 1) seg_end = N - 1;
        const int n_s = seg_end - seg_start + 1;

        int abytes[NB];
#pragma unroll
        for (int r = 0; r < NB; r++) {
            const int idx = r * 64 + lane;
            abytes[r] = idx < n_s ? src[seg_start + idx] : 0;
        }

        /* 1. my block of the window */
        unsigned z[61];
#pragma unroll
        for (int j = 0; j < 31; j++) z[j] = s_h[j];
#pragma unroll
        for (int j = 31; j < 61; j++) z[j] = z[j - 31] + z[j - 3];
        unsigned w[31];
#pragma unroll
        for (int j = 0; j < 31; j++) w[j] = 0;
#pragma unroll
        for (int m = 0; m < 31; m++) {
#pragma unroll
            for (int j = 0; j < 31; j++) w[j] += cb[m] * z[m + j];
        }
        unsigned glo = 0, ghi = 0;                                 /* c1 flags of my 43 calls (as B calls of this segment) */
#pragma unroll
        for (int t = 0; t < B; t++) {
            const unsigned v = w[t % 31] + w[(t + 28) % 31];
            w[t % 31] = v;
            s_y[lane * B + t] = v;
            const unsigned flag = (6 + (int) ((v >> 1) % 20u) > kseg) ? 1u : 0u;
            if (t < 32) glo |= flag << t; else ghi |= flag << (t - 32);
        }
        {
            const unsigned nb = (unsigned) __shfl_down((int) (glo & 1u), 1);   /* call 43 = the next block's call 0 */
            if (lane < 63) ghi |= nb << (B - 32);
        }
#pragma unroll
        for (int r = 0; r < NB; r++) s_a[r * 64 + lane] = (signed char) abytes[r];
        const unsigned long long G = ((unsigned long long) ghi << 32) | glo;

        /* 2. speculative walks: entry offset e -> (samples, exit offset) */
        unsigned res = 0;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            int pos = e, cnt = 0;
#pragma unroll
            for (int it = 0; it < (B + 1) / 2; it++) {
                const bool in = pos < B;
                const int step = 2 + (int) ((G >> (pos + 1)) & 1ull);
                pos += in ? step : 0;
                cnt += in ? 1 : 0;
            }
            res |= (unsigned) (cnt | (pos - B) << 5) << (8 * e);
        }

        /* 3. chain the blocks */
        int e_in = 0, first = n_s;
        {
            int e = 0, cum = 0;
            for (int b = 0; b < 64 && cum < n_s; b++) {
                const unsigned r = (unsigned) __builtin_amdgcn_readlane((int) res, b) >> (8 * e);
                if (lane == b) { e_in = e; first = cum; }
                cum += (int) (r & 31u);
                e = (int) ((r >> 5) & 3u);
            }
        }
        __syncthreads();

        /* 4. samples (crt_core.c:347-365) */
        {
            int pos = e_in;
            for (int k = 0; k < (B + 1) / 2; k++) {
                const int s = first + k;
                const bool valid = pos < B && s < n_s;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) break;
                if (valid) {
                    const unsigned *yp = s_y + lane * B + pos;
                    const unsigned rnv = yp[0] >> 1, r2 = yp[1] >> 1;
                    const int i = seg_start + s;
                    const int c1 = 6 + (int) (r2 % 20u) > kseg ? 1 : 0;
                    int nn = noise;
                    if (c1) {
                        const unsigned r3 = yp[2] >> 1;
                        if (i < N - H * (5 + ((int) (r3 & 7u) - 4))) {
                            const int ln = (i * vhs_line) / H;
                            nn = dev_cos14(ln * 8192 / 180) >> 8;
                        }
                    }
                    const int sv = (int) s_a[s] + (((int) ((rnv >> 16) & 0xffu) - 0x7f) * nn >> 8);
                    s_o[s] = (signed char) clampi(sv, -127, 127);
                    pos += 2 + c1;
                    if (s == n_s - 1) { s_misc[0] = (unsigned) (lane * B + pos); s_misc[1] = rnv; }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NB; r++) {
            const int idx = r * 64 + lane;
            if (idx < n_s) dst[seg_start + idx] = s_o[idx];
        }
        /* the next window starts at the first call after this segment's last sample */
        const int pos_end = (int) s_misc[0];
        unsigned hv = 0;
        if (lane < 31) hv = s_y[pos_end - 31 + lane];
        __syncthreads();
        if (lane < 31) s_h[lane] = hv;
        __syncthreads();
        seg_start += n_s;
    }
    if (lane < 31) h[lane] = s_h[lane];                            /* the generator's state after the field */
    if (lane == 0) {
        state[f].rn = (int) s_misc[1];                             /* crt_core.c:367 */
        signed char *tail = dst + N;
        store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
    }
}

/* rn <- rn after INPUT_SIZE steps (crt_core.c:367) */
__global__ void k_advance_rn(int n_fields, crthip_state *state, uint2 whole_field)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_fields) state[f].rn = (int) (whole_field.x * (unsigned) state[f].rn + whole_field.y);
}

/* ------------------------------------------------------------------------- */
/* D2-D7: the serial sync chain                                                */
/* ------------------------------------------------------------------------- */
/* The chain is serial in the line index (hsync and the burst integrators carry over,
 * crt_core.c:447,456-467).  Two kernels:
 *   k_vsync  D2      one wave per field: all 2*VWIN candidate lines are fetched up front, then
 *                    scanned with a wave-wide prefix sum; runs once per field.
 *   k_hsync  D4-D7   ONE DPP ROW (16 lanes) PER FIELD, four fields per wave: the hsync window is
 *                    16 samples (= one row, prefix sum by DPP row shifts), the burst integrators
 *                    are 4 chains (= one quad).  Row-uniform values live redundantly in the 16 lanes.
 * Latency: the bytes a line needs lie in [ln+hsync+SYNC_BEG-HWIN, ln+(hsync'&~3)+CB_BEG+40) with
 * |hsync'-hsync| <= HWIN; a 256-byte window [ln+hsync-40, ln+hsync+216) of line L+2 is fetched
 * while line L is processed (speculating that hsync moves by at most 3*HWIN until then) and parked
 * in a 3-slot LDS ring one iteration later, so a fetch has a whole iteration to land.  Whenever the bytes actually needed are not inside the parked window (hsync
 * wrapped around, ...) they are loaded directly -- same result, only slower. */
#define DPP_ROW_SHR(n)   (0x110 + (n))
#define DPP_ROW_BCAST15  0x142
#define DPP_ROW_BCAST31  0x143
#define DPP_QUAD_BCAST(k) ((k) * 0x55)            /* quad_perm:[k,k,k,k] */

/* inclusive prefix sum inside each row of 16 lanes */
__device__ __forceinline__ int row_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true);
    return v;
}
/* inclusive prefix sum over the 64 lanes of the wave */
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v = row_incl_scan(v);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xa, 0xf, false);   /* rows 1,3 += last of rows 0,2 */
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xc, 0xf, false);   /* rows 2,3 += lane 31 */
    return v;
}

/* D2 vsync, crt_core.c:379-396: first (line, j) whose running line sum <= VTHR */
template <class S>
__global__ void __launch_bounds__(64)
k_vsync(int n_fields, const signed char *__restrict__ inp, size_t fstride, crthip_state *__restrict__ state,
        uint2 whole_field, int advance_rn)
{
    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= n_fields) return;
    const signed char *in = inp + (size_t) f * fstride;
    crthip_state *st = state + f;
    const int vsync = st->vsync;
    int vline = 0, vj = S::HRES;
    v4i cand[2 * S::VWIN];
#pragma unroll
    for (int i = 0; i < 2 * S::VWIN; i++) {
        const int l = posmod(vsync + i - S::VWIN, S::VRES);
        cand[i] = load16u(in + l * S::HRES + lane * 16);
    }
    bool found = false;
#pragma unroll
    for (int i = 0; i < 2 * S::VWIN; i++) {
        if (!found) {
            vline = posmod(vsync + i - S::VWIN, S::VRES);
            const int wds[4] = { cand[i].x, cand[i].y, cand[i].z, cand[i].w };
            int pre[16];
            int run = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                int s = (wds[k >> 2] << (24 - 8 * (k & 3))) >> 24;
                if (lane * 16 + k >= S::HRES) s = 0;
                run += s;
                pre[k] = run;
            }
            const int excl = wave_incl_scan(run) - run;
            int first = 16;
#pragma unroll
            for (int k = 15; k >= 0; k--) {
                if (lane * 16 + k < S::HRES && excl + pre[k] <= S::VTHR) first = k;
            }
            const unsigned long long m = __ballot(first < 16);
            if (m) {
                const int L = __ffsll((long long) m) - 1;
                vj = L * 16 + __builtin_amdgcn_readlane(first, L);
                found = true;
            }
        }
    }
    if (!found) vj = S::HRES;
    if (lane == 0) {
        st->vsync = vline;
        st->odd_field = vj > S::HRES / 2;
        if (advance_rn) st->rn = (int) (whole_field.x * (unsigned) st->rn + whole_field.y);
    }
}

#define SYNC_WIN      256      /* bytes of a line's parked sync/burst window (16 lanes x 16 bytes) */
#define SYNC_WIN_BACK 40       /* window starts this far before ln + hsync                         */

/* D4-D7, crt_core.c:428-479.  Needs state.vsync / state.odd_field from k_vsync. */
template <class S>
__global__ void __launch_bounds__(64)
k_hsync(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
        crthip_state *__restrict__ state, crthip_line *__restrict__ lines)
{
    __shared__ int s_win[4][3][SYNC_WIN / 4];
    __shared__ int s_fb[4][16];                          /* fallback scratch: 16 hsync + 40 burst bytes per row */
    __shared__ int s_lines[4][S::LINES * 6];              /* the rows' line tables; written to memory after the loop so
                                                             that no store sits between the window prefetches */
    const int lane = threadIdx.x;
    const int row = lane >> 4, j = lane & 15;            /* field slot in the wave, lane in the row */
    const int f = blockIdx.x * 4 + row;
    const bool live = f < n_fields;
    const int fc = live ? f : n_fields - 1;              /* dead rows shadow the last field, never store */
    const signed char *in = inp + (size_t) fc * fstride;
    crthip_state *st = state + fc;
    int hsync = st->hsync;
    const int vsync = st->vsync;
    const int field_rows = st->odd_field * (P.ratio / 2);             /* crt_core.c:407 */
    int ccr[S::VPER];                                    /* lane holds ccf[r][j & 3] */
#pragma unroll
    for (int r = 0; r < S::VPER; r++) ccr[r] = st->ccf[r][j & 3];
    crthip_line *out_lines = lines + (size_t) fc * S::LINES;

    /* flat base of the window of line `line` assuming hsync h (row-uniform) */
    auto window_base = [&](int line, int h) {
        int l = line + vsync;                          /* < 2*VRES: BOT + 1 + VRES - 1 */
        if (l >= S::VRES) l -= S::VRES;
        const int b = l * S::HRES + h - SYNC_WIN_BACK;
        return b < 0 ? 0 : b;
    };
    /* each of the 16 lanes of a row moves 16 bytes of its field's window */
    int base_cur = window_base(S::TOP, hsync);           /* window parked for the current line */
    {
        const v4i w = load16u(in + base_cur + j * 16);
        int *d = s_win[row][S::TOP % 3] + j * 4;
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
    }
    int base_p = window_base(S::TOP + 1, hsync);         /* window in flight for line + 1 */
    v4i wp = load16u(in + base_p + j * 16);
    __syncthreads();

    const unsigned span = (unsigned) P.outh + P.v_fac;
    int prev_beg = -1, rank = 0;                          /* row collisions when outh + v_fac < LINES */
    for (int line = S::TOP; line < S::BOT; line++) {
        /* speculative fetch of the window of line + 2 (see the comment above) */
        const int base_n = window_base(line + 2, hsync);
        v4i wn = wp;
        if (line + 2 < S::BOT) wn = load16u(in + base_n + j * 16);
        const signed char *win = (const signed char *) s_win[row][line % 3];

        /* D4, crt_core.c:428-432 (unsigned arithmetic: v_fac is unsigned) */
        int beg = (int) ((unsigned) (line - S::TOP + 0) * span / (unsigned) S::LINES + (unsigned) field_rows);
        int end = (int) ((unsigned) (line - S::TOP + 1) * span / (unsigned) S::LINES + (unsigned) field_rows);
        const bool skip = beg >= P.outh;                               /* :431, row-uniform */
        if (end > P.outh) end = P.outh;
        if (!skip) {
            /* several lines can start on the same output row (outh + v_fac < LINES); the reference
             * handles them one after the other, so they are decoded in rank order by separate passes */
            rank = beg == prev_beg ? rank + 1 : 0;
            prev_beg = beg;
        }

        /* D5 hsync, crt_core.c:437-450.  0 <= vsync < VRES (k_vsync), so one conditional subtract wraps */
        int lidx = line + vsync;
        if (lidx >= S::VRES) lidx -= S::VRES;
        const int ln = lidx * S::HRES;
        const int a_off = ln + hsync + S::SYNC_BEG - S::HWIN - base_cur;          /* window-relative */
        /* the fast path reads LDS only; if the bytes are not in the parked window (rare) the fallback fetches
         * them into an LDS scratch row INSIDE its own branch, so no memory wait leaks into the common path */
        signed char *fb = (signed char *) s_fb[row];
        const bool a_in = a_off >= 0 && a_off + 2 * S::HWIN <= SYNC_WIN;
        if (!a_in) {
            if (j < 2 * S::HWIN) fb[j] = in[ln + hsync + S::SYNC_BEG - S::HWIN + j];
            __builtin_amdgcn_s_waitcnt(0);
        }
        int sv = 0;
        if (j < 2 * S::HWIN) sv = a_in ? win[a_off + j] : fb[j];
        const int pref = row_incl_scan(sv);
        const unsigned long long hm = __ballot(j < 2 * S::HWIN && pref <= S::HTHR);
        const unsigned m16 = (unsigned) (hm >> (row * 16)) & 0xffffu;
        const int hi = m16 ? (__ffs((int) m16) - 1 - S::HWIN) : S::HWIN;
        int hsync_new = hi + hsync;                                      /* POSMOD(i + hsync, HRES), :447 */
        if (hsync >= 0 && hsync < S::HRES) {                             /* |hi| <= HWIN: one wrap either way */
            if (hsync_new < 0) hsync_new += S::HRES;
            if (hsync_new >= S::HRES) hsync_new -= S::HRES;
        } else {
            hsync_new = posmod(hsync_new, S::HRES);                      /* caller-supplied out-of-range hsync */
        }
        if (!skip) hsync = hsync_new;

        int xpos, ypos;                                                  /* :452-454 */
        if (hsync >= 0 && hsync < S::HRES) {
            xpos = S::AV_BEG + hsync - 3;
            if (xpos >= S::HRES) xpos -= S::HRES;
        } else {
            xpos = posmod(S::AV_BEG + hsync - 3, S::HRES);
        }
        ypos = lidx + 3;
        if (ypos >= S::VRES) ypos -= S::VRES;
        const int pos = xpos + ypos * S::HRES;

        /* D6 burst lock, crt_core.c:456-467.  Lane j integrates phase (j & 3): its samples are burst
         * bytes k0, k0+4, ... with (CB_BEG + k0) & 3 == (j & 3) */
        const int b_off = ln + (hsync & ~3) + S::CB_BEG - base_cur;
        const bool b_in = b_off >= 0 && b_off + CB_SAMPLES <= SYNC_WIN;
        const int k0 = ((j & 3) - S::CB_BEG) & 3;
        if (!b_in) {
            const signed char *g = in + ln + (hsync & ~3) + S::CB_BEG;
            if (j < 10) { fb[16 + 4 * j + 0] = g[4 * j + 0]; fb[16 + 4 * j + 1] = g[4 * j + 1];
                          fb[16 + 4 * j + 2] = g[4 * j + 2]; fb[16 + 4 * j + 3] = g[4 * j + 3]; }
            __builtin_amdgcn_s_waitcnt(0);
        }
        const signed char *bsrc = b_in ? win + b_off : fb + 16;
        int smp[CB_SAMPLES / 4];
#pragma unroll
        for (int q = 0; q < CB_SAMPLES / 4; q++) smp[q] = bsrc[k0 + 4 * q];
        const int r = S::VPER == 1 ? 0 : ypos % S::VPER;
        int acc = ccr[0];
#pragma unroll
        for (int k = 1; k < S::VPER; k++) if (r == k) acc = ccr[k];
#pragma unroll
        for (int q = 0; q < CB_SAMPLES / 4; q++) {
            const int t127 = (int) (((unsigned) acc << 7) - (unsigned) acc);   /* acc * 127 with wrap, no slow multiply */
            acc = ((t127 + ((t127 >> 31) & 127)) >> 7) + smp[q];          /* C's truncating /128 */
        }
        if (!skip) {
#pragma unroll
            for (int k = 0; k < S::VPER; k++) if (r == k) ccr[k] = acc;
        }

        /* D7 carrier table, crt_core.c:469-479: quad lanes 0..3 hold ccr[0..3] */
        const int q0 = __builtin_amdgcn_update_dpp(0, acc, DPP_QUAD_BCAST(0), 0xf, 0xf, false);
        const int q1 = __builtin_amdgcn_update_dpp(0, acc, DPP_QUAD_BCAST(1), 0xf, 0xf, false);
        const int q2 = __builtin_amdgcn_update_dpp(0, acc, DPP_QUAD_BCAST(2), 0xf, 0xf, false);
        const int q3 = __builtin_amdgcn_update_dpp(0, acc, DPP_QUAD_BCAST(3), 0xf, 0xf, false);
        const int pa = hsync & 3;
        const int c0 = pa == 0 ? q0 : pa == 1 ? q1 : pa == 2 ? q2 : q3;
        const int c1 = pa == 0 ? q1 : pa == 1 ? q2 : pa == 2 ? q3 : q0;
        const int c2 = pa == 0 ? q2 : pa == 1 ? q3 : pa == 2 ? q0 : q1;
        const int c3 = pa == 0 ? q3 : pa == 1 ? q0 : pa == 2 ? q1 : q2;
        const int dci = c1 - c3, dcq = c2 - c0;
        if (j == 0 && live) {
            crthip_line lp;
            if (skip) {
                lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0; lp.hsync = hsync;
            } else {
                lp.pos = pos;
                lp.wave0 = ((dci * P.huecs - dcq * P.huesn) >> 4) * P.saturation;
                lp.wave1 = ((dcq * P.huecs + dci * P.huesn) >> 4) * P.saturation;
                lp.beg = beg;
                int nrows = end - P.scanlines - beg;                        /* rows beg .. end-scanlines-1, :662 */
                nrows = nrows < 1 ? 1 : nrows;
                /* carrier amplitude outside the 24-bit-multiply envelope of the fast decoder? */
                if (nrows > CRTHIP_LINE_NROWS_MASK) nrows = CRTHIP_LINE_NROWS_MASK;
                nrows |= (rank & CRTHIP_LINE_RANK_MASK) << CRTHIP_LINE_RANK_SHIFT;
                if (lp.wave0 > FAST_WAVE_MAX || lp.wave0 < -FAST_WAVE_MAX || lp.wave1 > FAST_WAVE_MAX || lp.wave1 < -FAST_WAVE_MAX)
                    nrows |= CRTHIP_LINE_EXACT;
                else if (lp.wave0 > T0_WAVE_MAX || lp.wave0 < -T0_WAVE_MAX || lp.wave1 > T0_WAVE_MAX || lp.wave1 < -T0_WAVE_MAX)
                    nrows |= CRTHIP_LINE_NOT64;
                else if (lp.wave0 > LOSKIP_WAVE_MAX || lp.wave0 < -LOSKIP_WAVE_MAX || lp.wave1 > LOSKIP_WAVE_MAX || lp.wave1 < -LOSKIP_WAVE_MAX)
                    nrows |= CRTHIP_LINE_WIDE;
                lp.nrows = nrows;
                lp.hsync = hsync;
            }
            int *d = s_lines[row] + (line - S::TOP) * 6;
            d[0] = lp.pos; d[1] = lp.wave0; d[2] = lp.wave1; d[3] = lp.beg; d[4] = lp.nrows; d[5] = lp.hsync;
        }
        /* park the window of line + 1 (fetched one iteration ago), keep line + 2's in flight */
        if (line + 1 < S::BOT) {
            int *d = s_win[row][(line + 1) % 3] + j * 4;
            d[0] = wp.x; d[1] = wp.y; d[2] = wp.z; d[3] = wp.w;
        }
        base_cur = base_p;
        base_p = base_n;
        wp = wn;
        __syncthreads();
    }
    /* line tables: LINES * 24 bytes per row, copied out 16 bytes per lane and pass */
    if (live) {
        int *dst = (int *) out_lines;
        for (int i = j * 4; i < S::LINES * 6; i += 64) {
            v4i v; v.x = s_lines[row][i]; v.y = s_lines[row][i + 1]; v.z = s_lines[row][i + 2]; v.w = s_lines[row][i + 3];
            store16u(dst + i, v);
        }
    }
    if (j < 4 && live) {
#pragma unroll
        for (int r = 0; r < S::VPER; r++) st->ccf[r][j] = ccr[r];
    }
    if (j == 0 && live) st->hsync = hsync;
}

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
/* lfm / hfm: pre-shifted multipliers (see above); NEAR1: coefficients are >= 2^15.
 * The band gains (crt_core.c:203-206) are applied as (r * g) >> 16 per band IN 32-BIT WRAPPING ARITHMETIC, i.e.
 * a gain of 65536 is "sign-extend the low 16 bits".  Inside the envelopes that is the identity:
 *   luma   |lo3|, |hi3| <= 2727                     -> low band = lo3, mid band (gain 8192) = (hi3 - lo3) >> 3
 *   chroma gains (65536, 65536, g2): low + mid = lo3 + (hi3 - lo3) = hi3 whenever |lo3|, |hi3 - lo3| < 2^15.
 *          LOSKIP (tier 0, |wave| <= LOSKIP_WAVE_MAX): every stage output stays inside the hull of its inputs
 *          (0 < c < 2^16, round-to-nearest never overshoots), the input is |s * wave >> 9| <= 16383, hence
 *          |lo3| <= 16383 and |hi3 - lo3| <= 32766: the four low stages feed nothing and are not computed. */
template <bool NEAR1, int G1, int G2, bool LOSKIP>
__device__ __forceinline__ int eq_step64(Eq64 &f, const int lfm, const int hfm, const long sp)
{
#define EQ64_STAGE(X, UPAIR, M) X = rearm(mad64(hi32(UPAIR) - hi32(X), M, NEAR1 ? UPAIR : X))
    static_assert(!LOSKIP || G1 == 65536, "dropping the low cascade needs low gain == mid gain == 65536");
    if (!LOSKIP) {
        EQ64_STAGE(f.lo0, sp, lfm);
        EQ64_STAGE(f.lo1, f.lo0, lfm);
        EQ64_STAGE(f.lo2, f.lo1, lfm);
        EQ64_STAGE(f.lo3, f.lo2, lfm);
    }
    EQ64_STAGE(f.hi0, sp, hfm);
    EQ64_STAGE(f.hi1, f.hi0, hfm);
    EQ64_STAGE(f.hi2, f.hi1, hfm);
    EQ64_STAGE(f.hi3, f.hi2, hfm);
#undef EQ64_STAGE
    const int lo3 = hi32(f.lo3), hi3 = hi32(f.hi3);
    int r;
    if (LOSKIP) r = hi3;
    else if (NEAR1 && G1 == 8192) r = lo3 + ((hi3 - lo3) >> 3);        /* luma envelope, see above */
    else {
        r = (lo3 * 65536) >> 16;
        if (G1 == 65536 || G1 == 8192) r += ((hi3 - lo3) * G1) >> 16;
        else r += __mul24(hi3 - lo3, G1) >> 16;
    }
    if (G2 != 0) {
        r += __mul24(f.h2 - hi3, G2) >> 16;
        f.h2 = f.h1; f.h1 = f.h0; f.h0 = hi32(sp);
    }
    return r;
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
 * while each lane reads / writes only its own row of the tile.  Row stride = TILE+1 dwords, all LDS
 * accesses are 32-bit: bank = (row + column) % 32, conflict-free for both access directions.
 * Workgroup = one wave, so __syncthreads() is only an ordering fence between the two phases.
 */
#define IN_TILE_DW   16                    /* decoder input tile: 64 samples per row            */
#define IN_STRIDE    (IN_TILE_DW + 1)
/* decoder output tile: PXT pixels per row (16: 64-byte store pieces, less LDS -> more waves, best for
 * narrow pictures that are ALU bound; 32: full 128-byte lines per store piece group, best for wide
 * pictures that lean on HBM write bandwidth) */
/* TIER: 0 = 64-bit-mad stages without the I/Q low cascades, 1 = 64-bit-mad stages, 2 = 24-bit mads,
 * 3 = exact 32-bit multiplies; a wave of 64 lines is decoded by the kernel
 * of its tier = max(tier flagged by k_hsync from its carrier amplitude, min_tier of the batch);
 * want_rank: only lines of this collision rank (always 0 unless outh + v_fac < LINES) */
template <class S, int TIER, bool BPP3, int PXT>
__global__ void __launch_bounds__(64)
k_decode(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
         const crthip_line *__restrict__ lines, unsigned char *__restrict__ outp, size_t ostride, int min_tier,
         int want_rank)
{
    constexpr bool FAST = TIER <= 2;        /* tiers 0-2 use 24-bit multiplies outside the filter stages */
    constexpr bool LOSKIP = TIER == 0;
    __shared__ unsigned s_in[64 * IN_STRIDE];
    constexpr int PX_TILE = PXT, PX_STRIDE = PXT + 1, PX_PIECES = PXT / 4;
    __shared__ unsigned s_px[64 * PX_STRIDE];
    __shared__ unsigned long long s_src[64], s_dst[64];
    __shared__ int s_nrows[64];

    const int lane = threadIdx.x;
    const int gid = blockIdx.x * 64 + lane;
    const bool live = gid < n_fields * S::LINES;
    crthip_line lp;
    lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0; lp.hsync = 0;
    const int f = live ? gid / S::LINES : 0;
    if (live) lp = lines[gid];
    /* the WAVE's tier = the highest one any of its lines needs (a higher tier decodes lower-tier lines just
     * as exactly), at least the batch-wide floor from the host (brightness, contrast): a wave with mixed lines
     * runs once, not once per tier */
    int tier = __ballot(lp.nrows & CRTHIP_LINE_EXACT) ? 3 : __ballot(lp.nrows & CRTHIP_LINE_NOT64) ? 2
             : __ballot(lp.nrows & CRTHIP_LINE_WIDE) ? 1 : 0;
    if (tier < min_tier) tier = min_tier;
    if (tier != TIER) return;
    int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    const int rank = (lp.nrows >> CRTHIP_LINE_RANK_SHIFT) & CRTHIP_LINE_RANK_MASK;
    if (!live || rank != want_rank) nrows = 0;
    if (__ballot(nrows > 0) == 0ull) return;          /* whole wave has nothing to do */
    const bool act = nrows > 0;
    constexpr int bpp = BPP3 ? 3 : 4;
    const size_t pitch = (size_t) P.outw * bpp;
    s_src[lane] = (unsigned long long) (inp + (size_t) f * fstride + (act ? lp.pos : 0));
    s_dst[lane] = (unsigned long long) (outp + (size_t) f * ostride + (size_t) (act ? lp.beg : 0) * pitch);
    s_nrows[lane] = nrows;
    __syncthreads();

    const int w0 = lp.wave0, w1 = lp.wave1, nw0 = -lp.wave0, nw1 = -lp.wave1;
    const int bright = P.bright, contrast = P.contrast;
    const int ylf = P.eq_lf[0], yhf = P.eq_hf[0], ilf = P.eq_lf[1], ihf = P.eq_hf[1], qlf = P.eq_lf[2], qhf = P.eq_hf[2];
    const unsigned psel = pack_selector(P.out_format), usel = unpack_selector(P.out_format);
    const bool rgb_order = P.out_format == CRTHIP_FMT_RGB;
    const bool blend = P.blend != 0;
    Eq3 ey = {}, ei = {}, eq = {};
    Eq64 wy, wi_, wq_;                             /* tier 0 state (register pairs) */
    eq64_reset(wy); eq64_reset(wi_); eq64_reset(wq_);
    /* tier 0 multipliers: luma coefficients are 2^16 + c', chroma ones < 2^15 (host-checked) */
    const int ylfm = (ylf - 65536) << 16, yhfm = (yhf - 65536) << 16;
    const int ilfm = ilf << 16, ihfm = ihf << 16, qlfm = qlf << 16, qhfm = qhf << 16;
    int py = 0, pi = 0, pq = 0;                    /* yiq of the previous sample */

    /* wave-uniform output pixel schedule, crt_core.c:528-531,555-562 */
    const unsigned scan_r = (unsigned) (S::AV_LEN - 1) << 12;
    const unsigned dx = (unsigned) P.dx;
    unsigned ppos = 0;
    int px = 0;
    const int outw = P.outw;

    /* cooperative input tile: piece = 16 bytes, 4 pieces per row, 16 rows per load instruction */
    const int in_row = lane >> 2, in_piece = lane & 3;
    v4i stage[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        stage[i] = gload16u(s_src[i * 16 + in_row] + in_piece * 16);
    }
    constexpr int NQ = (S::AV_LEN + 3) / 4;        /* dwords per line window; the last one may run past
                                                      AV_LEN: the filters are causal, the extra samples feed nothing */
    constexpr int NT = (NQ + IN_TILE_DW - 1) / IN_TILE_DW;
    for (int t = 0; t < NT; t++) {
        /* stash tile t (already in registers), then start fetching tile t+1 */
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned *d = s_in + (i * 16 + in_row) * IN_STRIDE + in_piece * 4;
            d[0] = (unsigned) stage[i].x; d[1] = (unsigned) stage[i].y; d[2] = (unsigned) stage[i].z; d[3] = (unsigned) stage[i].w;
        }
        __syncthreads();
        if (t + 1 < NT) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                stage[i] = gload16u(s_src[i * 16 + in_row] + (t + 1) * (IN_TILE_DW * 4) + in_piece * 16);
            }
        }
        const int xq_end = (t + 1) * IN_TILE_DW < NQ ? (t + 1) * IN_TILE_DW : NQ;
        for (int xq = t * IN_TILE_DW; xq < xq_end; xq++) {
            const int word = (int) s_in[lane * IN_STRIDE + (xq - t * IN_TILE_DW)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = xq * 4 + k;
                const int s = (word << (24 - 8 * k)) >> 24;
                /* D8, crt_core.c:539-543; wave[] = {w0, w1, -w0, -w1}: I uses wave[x&3], Q wave[(x+3)&3] */
                const int wi = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? nw0 : nw1;
                const int wq = k == 0 ? nw1 : k == 1 ? w0 : k == 2 ? w1 : nw0;
                int cy, ci, cq;
                if (TIER <= 1) {
                    /* luma stays unshifted here: (y << 4) * w >> 2 == (y * w) << 2 while nothing wraps, see D9 */
                    cy = eq_step64<true, 8192, 9175, false>(wy, ylfm, yhfm, pair_of(s + bright));
                    ci = eq_step64<false, 65536, 1311, LOSKIP>(wi_, ilfm, ihfm, pair_of(mulq<true>(s, wi) >> 9)) >> 3;
                    cq = eq_step64<false, 65536, 0, LOSKIP>(wq_, qlfm, qhfm, pair_of(mulq<true>(s, wq) >> 9)) >> 3;
                } else {
                    cy = eq_step<FAST, 8192, 9175>(ey, ylf, yhf, s + bright) << 4;
                    ci = eq_step<FAST, 65536, 1311>(ei, ilf, ihf, mulq<FAST>(s, wi) >> 9) >> 3;
                    cq = eq_step<FAST, 65536, 0>(eq, qlf, qhf, mulq<FAST>(s, wq) >> 9) >> 3;
                }
                /* D9: every output pixel whose left tap is sample x-1 is now computable */
                while (px < outw && ppos < scan_r && (int) (ppos >> 12) == x - 1) {
                    const int R = (int) (ppos & 0xfffu), L = 0xfff - R;
                    int yy;
                    if (TIER <= 1) {
                        /* crt_core.c:556: (py * L >> 2) + (cy * R >> 2) with py, cy = luma << 4.  |luma| <= 4173
                         * inside the tier's envelope, so no product wraps and both shifts are exact */
                        yy = mad24_vs(cy, R << 2, mulq_vs<true>(py, L << 2));
                    } else {
                        yy = (mulq_vs<FAST>(py, L) >> 2) + (mulq_vs<FAST>(cy, R) >> 2);
                    }
                    const int ii = (mulq_vs<FAST>(pi, L) >> 14) + (mulq_vs<FAST>(ci, R) >> 14);
                    const int qq = (mulq_vs<FAST>(pq, L) >> 14) + (mulq_vs<FAST>(cq, R) >> 14);
                    int r = mulq<FAST>((yy + mulq<FAST>(3879, ii) + mulq<FAST>(2556, qq)) >> 12, contrast) >> 8;
                    int g = mulq<FAST>((yy - mulq<FAST>(1126, ii) - mulq<FAST>(2605, qq)) >> 12, contrast) >> 8;
                    int b = mulq<FAST>((yy - mulq<FAST>(4530, ii) + mulq<FAST>(7021, qq)) >> 12, contrast) >> 8;
                    r = clampi(r, 0, 255); g = clampi(g, 0, 255); b = clampi(b, 0, 255);
                    s_px[lane * PX_STRIDE + (px & (PX_TILE - 1))] = (unsigned) (((r << 8 | g) << 8) | b);
                    if ((px & (PX_TILE - 1)) == PX_TILE - 1 || px == outw - 1) {
                        /* drain the pixel tile: pixels [px0, px0+cnt) of every row, + D10 duplicates (:661-664) */
                        const int px0 = px & ~(PX_TILE - 1);
                        const int cnt = px - px0 + 1;
                        __syncthreads();
                        if (!BPP3) {
                            const int orow_ = lane / PX_PIECES, piece = lane % PX_PIECES;  /* 4 pixels = 16 bytes per piece */
                            const int have = cnt - piece * 4;                      /* pixels of this piece that exist */
#pragma unroll 2
                            for (int i = 0; i < PX_PIECES; i++) {
                                const int rr_ = i * (64 / PX_PIECES) + orow_;
                                const int nr = s_nrows[rr_];
                                if (nr > 0 && have > 0) {
                                    const unsigned long long d = s_dst[rr_] + (size_t) (px0 + piece * 4) * 4;
                                    const unsigned *sp = s_px + rr_ * PX_STRIDE + piece * 4;
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
#pragma unroll
                                    for (int c = 0; c < 4; c++) {
                                        const unsigned full = 0xff000000u | v[c];
                                        v[c] = __builtin_amdgcn_perm(full, full, psel);
                                    }
                                    for (int dup = 0; dup < nr; dup++) {
                                        const unsigned long long dd = d + (size_t) dup * pitch;
                                        if (have >= 4) {
                                            v4i o; o.x = (int) v[0]; o.y = (int) v[1]; o.z = (int) v[2]; o.w = (int) v[3];
                                            gstore16u(dd, o);
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
                                    int rgb = (int) s_px[rr_ * PX_STRIDE + c];
                                    if (blend) {
                                        const int o0 = (int) gload8(d), o1 = (int) gload8(d + 1), o2 = (int) gload8(d + 2);
                                        const int old = rgb_order ? (o0 << 16 | o1 << 8 | o2) : (o2 << 16 | o1 << 8 | o0);
                                        rgb = ((rgb & 0xfefeff) >> 1) + ((old & 0xfefeff) >> 1);
                                    }
                                    const unsigned char c0 = (unsigned char) (rgb_order ? rgb >> 16 : rgb);
                                    const unsigned char c2 = (unsigned char) (rgb_order ? rgb : rgb >> 16);
                                    for (int dup = 0; dup < nr; dup++) {
                                        const unsigned long long dd = d + (size_t) dup * pitch;
                                        gstore8(dd, c0); gstore8(dd + 1, (unsigned) (rgb >> 8)); gstore8(dd + 2, c2);
                                    }
                                }
                            }
                        }
                        __syncthreads();
                    }
                    ppos += dx;
                    px++;
                }
                py = cy; pi = ci; pq = cq;
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Sequence mode: n consecutive fields of ONE television set  */
/* ------------------------------------------------------------------------- */
/* Sequential semantics of   for k: crt_modulate(field k); crt_demodulate(); save(out)   (the loop of
 * extra/video_convert.c:246-277) reproduced with parallel kernels:
 *   rn      : closed form, rn_k = J^k(rn_0), J = INPUT_SIZE steps of the LCG;
 *   encoder : independent per field (fused, writes the noisy field);
 *   sync    : field k starts from field k-1's final (hsync, vsync).  Solved as a fixed point: every
 *             pass runs k_vsync/k_hsync for ALL fields in parallel with init_k = final_{k-1} of the
 *             previous pass; after pass j fields 0..j-1 are final for good, and because a field's final
 *             state hardly ever depends on its initial one the iteration normally stops after 2-3 passes;
 *   decoder : independent per field given its line table (blend must be 0: with blend the output is a
 *             recurrence over fields);
 *   weave   : output image k = the single output buffer after field k: rows field k does not write come
 *             from the latest earlier field that wrote them (or the initial buffer). */
__global__ void k_seq_rn(int n_fields, crthip_state *state, uint2 whole_field)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_fields) return;
    /* (m, a) = J^k by square and multiply */
    unsigned pm = whole_field.x, pa = whole_field.y, m = 1u, a = 0u;
    for (unsigned e = (unsigned) k; e; e >>= 1) {
        if (e & 1u) { m = pm * m; a = pm * a + pa; }
        pa = pm * pa + pa;
        pm = pm * pm;
    }
    const unsigned rn0 = (unsigned) state[0].rn;       /* entry 0 is never written here */
    if (k > 0) state[k].rn = (int) (m * rn0 + a);
}

/* init_k = (k ? guess[k-1] : first); also remembers nothing else */
__global__ void k_seq_load(int n_fields, crthip_state *state, const int2 *guess, int2 first)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_fields) return;
    const int2 v = k ? guess[k - 1] : first;
    state[k].hsync = v.x;
    state[k].vsync = v.y;
}

/* guess <- finals; *changed |= any difference */
__global__ void k_seq_compare(int n_fields, const crthip_state *state, int2 *guess, int *changed)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_fields) return;
    const int2 f = make_int2(state[k].hsync, state[k].vsync);
    const int2 g = guess[k];
    if (f.x != g.x || f.y != g.y) {
        guess[k] = f;
        atomicOr(changed, 1);
    }
}

/* rows written by field k (crt_core.c:552,661-664) -> owner[k][row] = 1 */
__global__ void k_seq_rows(int n_fields, int lines_per_field, int outh, const crthip_line *lines, unsigned char *owner)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_fields * lines_per_field) return;
    const int k = gid / lines_per_field;
    const crthip_line lp = lines[gid];
    const int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    for (int r = 0; r < nrows; r++) {
        if (lp.beg + r < outh) owner[(size_t) k * outh + lp.beg + r] = 1;
    }
}

/* latest[k][row] = last field <= k that wrote the row, -1 = none (serial in k, one lane per row) */
__global__ void k_seq_latest(int n_fields, int outh, const unsigned char *owner, int *latest)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= outh) return;
    int last = -1;
    for (int k = 0; k < n_fields; k++) {
        if (owner[(size_t) k * outh + r]) last = k;
        latest[(size_t) k * outh + r] = last;
    }
}

/* rows of image k that field k did not write <- the same row of image latest[k][row] (or the initial image) */
__global__ void __launch_bounds__(256)
k_seq_weave(int n_fields, int outh, size_t pitch, unsigned char *out, size_t ostride, const unsigned char *init,
            const int *latest)
{
    const int row = blockIdx.x % outh, k = blockIdx.x / outh;
    if (k >= n_fields) return;
    const int src_k = latest[(size_t) k * outh + row];
    if (src_k == k) return;
    unsigned char *dst = out + (size_t) k * ostride + (size_t) row * pitch;
    const unsigned char *src = src_k >= 0 ? out + (size_t) src_k * ostride + (size_t) row * pitch
                                          : (init ? init + (size_t) row * pitch : nullptr);
    for (size_t b = (size_t) threadIdx.x * 16; b < pitch; b += 256 * 16) {
        const size_t nb = pitch - b < 16 ? pitch - b : 16;
        if (nb == 16) {
            v4i v = { 0, 0, 0, 0 };
            if (src) v = load16u(src + b);
            store16u(dst + b, v);
        } else {
            for (size_t c = 0; c < nb; c++) dst[b + c] = src ? src[b + c] : (unsigned char) 0;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* host side: context, dispatch by system, C ABI                               */
/* ------------------------------------------------------------------------- */
struct crthip_ctx {
    int device;
    int system, pattern;
    struct crt_sysdef sd;
    hipStream_t stream;
    bool own_stream;
    uint2 *d_jump16;
    uint2 whole_field;          /* affine map of INPUT_SIZE LCG steps */
    size_t fstride;
    /* workspace for crthip_fieldpass */
    int cap_fields;
    signed char *d_analog, *d_inp;
    crthip_line *d_lines;
    /* profiling */
    bool force_exact;           /* debug/test: never use the 24-bit fast kernels */
    bool no_tier0;              /* debug/test: never use the 64-bit-mad decoder tiers */
    bool no_loskip;             /* debug/test: never drop the I/Q low cascades */
    signed char *d_nes_tab;     /* NES: 512 x 12 composite-sample table, rebuilt per encoder launch */
    signed char *d_skel;        /* SKEL_VARIANTS clean skeleton fields, rebuilt per fused encoder launch */
    uint2 *d_jump1;             /* LCG affine maps of 0..15 steps */
    unsigned char *d_seq;       /* crthip_sequence scratch */
    size_t seq_cap;
    unsigned *d_vhs_rows;       /* VHS: jump coefficients, (vhs_chunks + 1) x 31 words, then 31 x 64 (tail blocks) */
    int vhs_chunks;
    unsigned *d_vhs_hist;       /* VHS: bound per-field generator histories (caller's memory) */
    int px_tile;                /* 0 = by output width, else 16 / 32 (tuning / tests) */
    int ac_tile;                /* encoder tile, same convention, by input width */
    int overlap_chunks;         /* crthip_fieldpass: chunks alternating between two streams (1 = off) */
    hipStream_t aux_stream;
    hipEvent_t ev_fork, ev_join;
    bool prof;
    double prof_ms[CRTHIP_K_COUNT];
    int prof_n[CRTHIP_K_COUNT];
    struct Pending { int k; hipEvent_t a, b; } *pend;
    int npend, cappend;
    char err[256];
};

static int set_err(crthip_ctx *c, int code, const char *what, hipError_t e)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(ctx, CRTHIP_E_HIP, #call, e_); } while (0)

static void lcg_jump_host(unsigned k, unsigned *mul, unsigned *add)
{
    unsigned am = LCG_MUL, ac = LCG_ADD, rm = 1u, rc = 0u;
    while (k) {
        if (k & 1u) { rm = am * rm; rc = am * rc + ac; }
        ac = am * ac + ac;
        am = am * am;
        k >>= 1;
    }
    *mul = rm;
    *add = rc;
}

template <class S> static bool sysdef_matches(const struct crt_sysdef &d)
{
    return d.hres == S::HRES && d.vres == S::VRES && d.input_size == S::INPUT_SIZE && d.top == S::TOP &&
           d.bot == S::BOT && d.vper == S::VPER && d.hsync_window == S::HWIN && d.vsync_window == S::VWIN &&
           d.hsync_thresh == S::HTHR && d.vsync_thresh == S::VTHR && d.sync_beg == S::SYNC_BEG &&
           d.bw_beg == S::BW_BEG && d.cb_beg == S::CB_BEG && d.av_beg == S::AV_BEG && d.av_len == S::AV_LEN &&
           d.vs_sep_end == S::VS_SEP_END && d.white_level == S::WHITE && d.burst_level == S::BURST &&
           d.black_level == S::BLACK && d.blank_level == S::BLANK && d.sync_level == S::SYNC;
}

/* call fn(S{}) with the system table type S of (system, pattern); fn is a generic lambda */
template <class F> static int dispatch_system(int system, int pattern, F &&fn)
{
    if (system == CRTHIP_SYSTEM_NTSC) return pattern == 1 ? fn(SysNTSC{}) : fn(SysNTSC0{});
    if (system == CRTHIP_SYSTEM_NTSCVHS) return pattern == 1 ? fn(SysVHS{}) : fn(SysVHS0{});
    if (system == CRTHIP_SYSTEM_NES) {
        if (pattern == 2) return fn(SysNES2{});
        if (pattern == 1) return fn(SysNES1{});
        return fn(SysNES0{});
    }
    return CRTHIP_E_ARG;
}

struct ProfScope {
    crthip_ctx *c; int k; hipEvent_t a, b; bool on;
    ProfScope(crthip_ctx *ctx, int kernel) : c(ctx), k(kernel), on(ctx->prof)
    {
        if (on) {
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope()
    {
        if (!on) return;
        hipEventRecord(b, c->stream);
        if (c->npend == c->cappend) {
            int ncap = c->cappend ? c->cappend * 2 : 64;
            c->pend = (crthip_ctx::Pending *) realloc(c->pend, sizeof(*c->pend) * (size_t) ncap);
            c->cappend = ncap;
        }
        c->pend[c->npend].k = k; c->pend[c->npend].a = a; c->pend[c->npend].b = b;
        c->npend++;
    }
};

static bool encoder_fast_ok(const crthip_params *p)
{
    const int wh = p->white < 0 ? -p->white : p->white;
    const int nz = p->noise < 0 ? -p->noise : p->noise;
    return wh < (1 << 23) && nz < (1 << 23);
}

template <class S, bool FULL, bool FAST>
static void launch_active(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                          signed char *dst, const crthip_state *d_state)
{
    ProfScope ps(c, CRTHIP_K_ACTIVE);
    const int total = n * p->desth;
    const dim3 grid((total + 63) / 64), block(64);
    const unsigned char *img = (const unsigned char *) d_images;
    if constexpr (S::IS_NES) {
        if (p->w >= 8) {
            hipLaunchKernelGGL((k_nes_table<S>), dim3((NES_TAB_SIZE + 255) / 256), dim3(256), 0, c->stream, *p, c->d_nes_tab);
            if (FULL && p->noise != 0)
                hipLaunchKernelGGL((k_active_nes<S, true, FULL, 16>), grid, block, 0, c->stream, *p, n, img, istride, dst, c->fstride, d_state, c->d_jump16, c->d_nes_tab);
            else
                hipLaunchKernelGGL((k_active_nes<S, false, FULL, 16>), grid, block, 0, c->stream, *p, n, img, istride, dst, c->fstride, d_state, c->d_jump16, c->d_nes_tab);
            return;
        }
    }
    const bool in4 = S::IS_NES || (p->in_bpp == 4 && p->w >= 4);
    const bool noise = FULL && p->noise != 0;
    const bool wide_in = c->ac_tile ? c->ac_tile == 32 : p->w >= 1280;
#define CRTHIP_LAUNCH_ACTIVE(NZ, I4) \
    do { if (wide_in) hipLaunchKernelGGL((k_active<S, NZ, FAST, I4, FULL, 32>), grid, block, 0, c->stream, *p, n, img, istride, dst, c->fstride, d_state, c->d_jump16); \
         else hipLaunchKernelGGL((k_active<S, NZ, FAST, I4, FULL, 16>), grid, block, 0, c->stream, *p, n, img, istride, dst, c->fstride, d_state, c->d_jump16); } while (0)
    if (noise) { if (in4) CRTHIP_LAUNCH_ACTIVE(true, true); else CRTHIP_LAUNCH_ACTIVE(true, false); }
    else       { if (in4) CRTHIP_LAUNCH_ACTIVE(false, true); else CRTHIP_LAUNCH_ACTIVE(false, false); }
#undef CRTHIP_LAUNCH_ACTIVE
}

template <class S, bool FULL>
static int launch_encoder(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                          signed char *dst, const crthip_state *d_state, int nes_setup)
{
    if (FULL) {
        /* fused path: skeleton + noise for everything outside the active rectangle */
        ProfScope ps(c, CRTHIP_K_TEMPLATE);
        const int s0 = p->yo * S::HRES + p->xo;
        const int head = (s0 + 15) / 16;
        const int gap = (S::HRES - p->destw + 15) / 16;
        const int tail_len = S::INPUT_SIZE - (s0 + (p->desth - 1) * S::HRES + p->destw);
        const int tail = (tail_len + 15) / 16;
        const int total = n * (head + (p->desth - 1) * gap + tail);
        constexpr int SK_LANES = SKEL_VARIANTS * ((S::INPUT_SIZE + 15) / 16);
        hipLaunchKernelGGL((k_skeleton<S>), dim3((SK_LANES + 255) / 256), dim3(256), 0, c->stream, *p, c->d_skel, c->fstride);
        if (p->noise != 0)
            hipLaunchKernelGGL((k_margin<S, true>), dim3((total + 255) / 256), dim3(256), 0, c->stream,
                               *p, n, dst, c->fstride, d_state, c->d_jump16, c->d_jump1, c->d_skel, head, gap, tail);
        else
            hipLaunchKernelGGL((k_margin<S, false>), dim3((total + 255) / 256), dim3(256), 0, c->stream,
                               *p, n, dst, c->fstride, d_state, c->d_jump16, c->d_jump1, c->d_skel, head, gap, tail);
    } else {
        constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
        ProfScope ps(c, CRTHIP_K_TEMPLATE);
        const int total = n * CHUNKS;
        hipLaunchKernelGGL((k_template<S>), dim3((total + 255) / 256), dim3(256), 0, c->stream,
                           *p, n, dst, c->fstride, d_state, nes_setup);
    }
    if (encoder_fast_ok(p) && !c->force_exact) launch_active<S, FULL, true>(c, p, n, d_images, istride, dst, d_state);
    else launch_active<S, FULL, false>(c, p, n, d_images, istride, dst, d_state);
    return CRTHIP_OK;
}

/* after crt_modulate: ccf preset (crt_ntsc.c:325-329, crt_nes.c:196-200), VHS resets (crt_ntscvhs.c:259,332-336) */
template <class S>
__global__ void k_encoder_state(const crthip_params P, int n_fields, crthip_state *state)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fields) return;
    crthip_state *st = state + f;
    if constexpr (S::IS_NES) {
        for (int r = 0; r < 3; r++) {
            /* iccf[n % 3] is last written by the bottom-most line with that residue; all lines of a
             * residue class write the same burst, so any line n of the class will do */
            for (int k = 0; k < 4; k++) {
                int cb = P.burst[(r + st->aux) % 3][k];
                st->ccf[r][k] = ((int) (signed char) ((S::BLANK + cb * S::BURST) >> 5)) << 7;
            }
        }
    } else {
        const int inv_phase = ((st->field & 1) == (st->frame & 1));
        for (int k = 0; k < 4; k++) {
            int cb = S::PATTERN == 1 ? P.burst[0][(k + inv_phase * 2) & 3] : P.burst[0][k];
            int v = ((int) (signed char) ((S::BLANK + cb * S::BURST) >> 5)) << 7;
            st->ccf[0][k] = S::IS_VHS ? 0 : v;
        }
        if (S::IS_VHS) st->hsync = 0;
        st->field &= 1;
        st->frame &= 1;
    }
}

extern "C" {

int crthip_abi_version(void) { return CRTHIP_ABI_VERSION; }

int crthip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int crthip_create(crthip_ctx **out, int device, int system, int chroma_pattern)
{
    if (!out) return CRTHIP_E_ARG;
    *out = 0;
    struct crt_sysdef sd;
    if (crt_sysdef_get(&sd, system, chroma_pattern) != CRTHIP_OK) return CRTHIP_E_ARG;
    if (dispatch_system(system, chroma_pattern, [&](auto tag) {
            return sysdef_matches<decltype(tag)>(sd) ? CRTHIP_OK : CRTHIP_E_ARG; }) != CRTHIP_OK) {
        fprintf(stderr, "crthip: host/device system tables disagree (system %d pattern %d)\n", system, chroma_pattern);
        return CRTHIP_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CRTHIP_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CRTHIP_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "crthip: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
        return CRTHIP_E_NODEVICE;
    }
    crthip_ctx *c = new (std::nothrow) crthip_ctx();
    if (!c) return CRTHIP_E_NOMEM;
    memset(c, 0, sizeof(*c));
    c->device = device; c->system = system; c->pattern = chroma_pattern; c->sd = sd;
    c->fstride = crthip_field_stride(system, chroma_pattern);
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return CRTHIP_E_HIP;
    }
    c->stream = 0;              /* the device's default stream until crthip_set_stream() */
    c->overlap_chunks = 1;
    c->own_stream = false;
    /* noise LCG jump tables: state after 16*q steps, q = 0 .. INPUT_SIZE/16 */
    const int nq = sd.input_size / 16 + 2;
    uint2 *h = (uint2 *) malloc(sizeof(uint2) * (size_t) nq);
    if (!h) { crthip_destroy(c); return CRTHIP_E_NOMEM; }
    unsigned m16, a16;
    lcg_jump_host(16, &m16, &a16);
    h[0].x = 1u; h[0].y = 0u;
    for (int q = 1; q < nq; q++) { h[q].x = m16 * h[q - 1].x; h[q].y = m16 * h[q - 1].y + a16; }
    lcg_jump_host((unsigned) sd.input_size, &c->whole_field.x, &c->whole_field.y);
    if (hipMalloc((void **) &c->d_jump16, sizeof(uint2) * (size_t) nq) != hipSuccess ||
        hipMemcpy(c->d_jump16, h, sizeof(uint2) * (size_t) nq, hipMemcpyHostToDevice) != hipSuccess) {
        free(h);
        crthip_destroy(c);
        return CRTHIP_E_HIP;
    }
    free(h);
    if (system == CRTHIP_SYSTEM_NES && hipMalloc((void **) &c->d_nes_tab, NES_TAB_SIZE) != hipSuccess) {
        crthip_destroy(c);
        return CRTHIP_E_NOMEM;
    }
    {
        uint2 j1[16];
        j1[0] = make_uint2(1u, 0u);
        for (int k = 1; k < 16; k++) j1[k] = make_uint2(LCG_MUL * j1[k - 1].x, LCG_MUL * j1[k - 1].y + LCG_ADD);
        if (hipMalloc((void **) &c->d_skel, (size_t) SKEL_VARIANTS * c->fstride) != hipSuccess ||
            hipMalloc((void **) &c->d_jump1, sizeof(j1)) != hipSuccess ||
            hipMemcpy(c->d_jump1, j1, sizeof(j1), hipMemcpyHostToDevice) != hipSuccess) {
            crthip_destroy(c);
            return CRTHIP_E_NOMEM;
        }
    }
    if (system == CRTHIP_SYSTEM_NTSCVHS) {
        /* jump coefficients of the rand() recurrence: one row per parallel chunk, one for the first call of
         * the tail, and the tail's 64 block offsets (transposed: [m][block]) */
        const int chunks = vhs_tail_start(sd.input_size, sd.hres) / VHS_CHUNK;
        const size_t words = 31 * (size_t) (chunks + 1) + 31 * 64;
        unsigned *rows = (unsigned *) malloc(sizeof(unsigned) * words);
        if (!rows) { crthip_destroy(c); return CRTHIP_E_NOMEM; }
        crt_setup_vhs_power_table(1ul, 2ul * VHS_CHUNK, chunks, rows);
        crt_setup_vhs_power(1ul + 2ul * (unsigned long) chunks * VHS_CHUNK, rows + 31 * (size_t) chunks);
        {
            unsigned blk[31 * 64];
            unsigned *t = rows + 31 * (size_t) (chunks + 1);
            crt_setup_vhs_power_table(0ul, (unsigned long) VHS_BLK, 64, blk);
            for (int b = 0; b < 64; b++) {
                for (int m = 0; m < 31; m++) t[m * 64 + b] = blk[b * 31 + m];
            }
        }
        c->vhs_chunks = chunks;
        if (hipMalloc((void **) &c->d_vhs_rows, sizeof(unsigned) * words) != hipSuccess ||
            hipMemcpy(c->d_vhs_rows, rows, sizeof(unsigned) * words, hipMemcpyHostToDevice) != hipSuccess) {
            free(rows);
            crthip_destroy(c);
            return CRTHIP_E_HIP;
        }
        free(rows);
    }
    *out = c;
    return CRTHIP_OK;
}

void crthip_destroy(crthip_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->npend; i++) { hipEventDestroy(c->pend[i].a); hipEventDestroy(c->pend[i].b); }
    free(c->pend);
    if (c->aux_stream) { hipStreamSynchronize(c->aux_stream); hipStreamDestroy(c->aux_stream); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
    if (c->d_jump16) hipFree(c->d_jump16);
    if (c->d_vhs_rows) hipFree(c->d_vhs_rows);
    if (c->d_seq) hipFree(c->d_seq);
    if (c->d_nes_tab) hipFree(c->d_nes_tab);
    if (c->d_skel) hipFree(c->d_skel);
    if (c->d_jump1) hipFree(c->d_jump1);
    if (c->d_analog) hipFree(c->d_analog);
    if (c->d_inp) hipFree(c->d_inp);
    if (c->d_lines) hipFree(c->d_lines);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int crthip_set_stream(crthip_ctx *c, void *hip_stream)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t) hip_stream;       /* NULL = the default stream */
    return CRTHIP_OK;
}

int crthip_synchronize(crthip_ctx *c)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CRTHIP_OK;
}

const char *crthip_error_string(const crthip_ctx *c) { return c ? c->err : "null context"; }

int crthip_reserve(crthip_ctx *c, int n)
{
    if (!c || n <= 0) return CRTHIP_E_ARG;
    if (n <= c->cap_fields) return CRTHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_analog) hipFree(c->d_analog);
    if (c->d_inp) hipFree(c->d_inp);
    if (c->d_lines) hipFree(c->d_lines);
    c->d_analog = 0; c->d_inp = 0; c->d_lines = 0; c->cap_fields = 0;
    const size_t bytes = c->fstride * (size_t) n + 4096;
    if (hipMalloc((void **) &c->d_inp, bytes) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc inp", hipSuccess);
    if (hipMalloc((void **) &c->d_analog, bytes) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc analog", hipSuccess);
    if (hipMalloc((void **) &c->d_lines, sizeof(crthip_line) * (size_t) n * c->sd.lines) != hipSuccess)
        return set_err(c, CRTHIP_E_NOMEM, "hipMalloc lines", hipSuccess);
    HIPCHK(c, hipMemsetAsync(c->d_inp, 0, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_analog, 0, bytes, c->stream));
    c->cap_fields = n;
    return CRTHIP_OK;
}

static int check_params(crthip_ctx *c, const crthip_params *p, int n)
{
    if (!c || !p || n <= 0) return CRTHIP_E_ARG;
    if (p->finalized != CRTHIP_PARAMS_MAGIC) return set_err(c, CRTHIP_E_ARG, "params not finalized", hipSuccess);
    if (p->system != c->system || p->chroma_pattern != c->pattern) return set_err(c, CRTHIP_E_ARG, "params are for another system", hipSuccess);
    return CRTHIP_OK;
}

/* the encoder contract: the active rectangle lies inside the field (the reference would
 * scribble over neighbouring lines / out of bounds otherwise, crt_ntsc.c:322) */
static int check_encoder(crthip_ctx *c, const crthip_params *p)
{
    if (c->system != CRTHIP_SYSTEM_NES && p->in_bpp == 0) return 1;   /* silent no-op, crt_ntsc.c:190-193 */
    if (p->xo < 0 || p->yo < 0 || p->xo + p->destw > c->sd.hres || p->yo + p->desth > c->sd.vres || p->destw <= 0 || p->desth <= 0)
        return set_err(c, CRTHIP_E_ARG, "active rectangle leaves the field (xoffset/yoffset out of contract)", hipSuccess);
    return CRTHIP_OK;
}

int crthip_modulate(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                    signed char *d_analog, crthip_state *d_state)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    rc = check_encoder(c, p);
    if (rc) return rc < 0 ? rc : CRTHIP_OK;
    if (!d_images || !d_analog || !d_state) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        int r = launch_encoder<S, false>(c, p, n, d_images, istride, d_analog, d_state, (p->flags & CRTHIP_F_NES_SETUP) != 0);
        hipLaunchKernelGGL((k_encoder_state<S>), dim3((n + 63) / 64), dim3(64), 0, c->stream, *p, n, d_state);
        return r;
    });
    HIPCHK(c, hipGetLastError());
    return rc;
}

int crthip_noise(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_analog, signed char *d_inp,
                 crthip_state *d_state)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (p->out_bpp == 0) return CRTHIP_OK;                       /* crt_core.c:312-315 */
    if (!d_analog || !d_inp || !d_state) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->system == CRTHIP_SYSTEM_NTSCVHS) {
        if (!c->d_vhs_hist) return set_err(c, CRTHIP_E_ARG, "VHS: no generator histories bound (crthip_vhs_bind_history)", hipSuccess);
        rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
            using S = decltype(tag);
            ProfScope ps(c, CRTHIP_K_NOISE);
            if constexpr (S::IS_VHS) {
                /* the tail kernel rewrites the histories the parallel region reads: stream order keeps them apart */
                hipLaunchKernelGGL((k_vhs_noise<S>), dim3((n * c->vhs_chunks + 63) / 64), dim3(64), 0, c->stream,
                                   *p, n, d_analog, d_inp, c->fstride, c->d_vhs_hist, c->d_vhs_rows, c->vhs_chunks);
                hipLaunchKernelGGL((k_vhs_tail<S>), dim3(n), dim3(64), 0, c->stream,
                                   *p, n, d_analog, d_inp, c->fstride, d_state, c->d_vhs_hist,
                                   c->d_vhs_rows + (size_t) c->vhs_chunks * 31, c->d_vhs_rows + (size_t) (c->vhs_chunks + 1) * 31);
            }
            return CRTHIP_OK;
        });
        HIPCHK(c, hipGetLastError());
        return rc;
    }
    rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
        {
            ProfScope ps(c, CRTHIP_K_NOISE);
            hipLaunchKernelGGL((k_noise<S>), dim3((n * CHUNKS + 255) / 256), dim3(256), 0, c->stream,
                               *p, n, d_analog, d_inp, c->fstride, d_state, c->d_jump16);
        }
        hipLaunchKernelGGL(k_advance_rn, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, d_state, c->whole_field);
        return CRTHIP_OK;
    });
    HIPCHK(c, hipGetLastError());
    return rc;
}

static int launch_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
                       crthip_line *d_lines, int advance_rn)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        ProfScope ps(c, CRTHIP_K_SYNC);
        hipLaunchKernelGGL((k_vsync<S>), dim3(n), dim3(64), 0, c->stream, n, d_inp, c->fstride, d_state, c->whole_field, advance_rn);
        hipLaunchKernelGGL((k_hsync<S>), dim3((n + 3) / 4), dim3(64), 0, c->stream, *p, n, d_inp, c->fstride, d_state, d_lines);
        return CRTHIP_OK;
    });
}

int crthip_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
                crthip_line *d_lines)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (p->out_bpp == 0) return CRTHIP_OK;
    if (!d_inp || !d_state || !d_lines) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rc = launch_sync(c, p, n, d_inp, d_state, d_lines, 0);
    HIPCHK(c, hipGetLastError());
    return rc;
}

/* host half of the decoder envelopes: the batch-wide floor of the decoder tier.
 * tier 0 additionally needs the luma coefficients in [2^15, 1.5*2^16) and the chroma ones below 2^15 */
static int decoder_min_tier(const crthip_ctx *c, const crthip_params *p)
{
    const int b = p->bright < 0 ? -p->bright : p->bright;
    const int ct = p->contrast < 0 ? -p->contrast : p->contrast;
    if (c->force_exact || b > FAST_BRIGHT_MAX || ct >= (1 << 23)) return 3;
    const bool coef_ok = p->eq_lf[0] >= 32768 && p->eq_lf[0] < 98304 && p->eq_hf[0] >= 32768 && p->eq_hf[0] < 98304 &&
                         p->eq_lf[1] > 0 && p->eq_lf[1] < 32768 && p->eq_hf[1] > 0 && p->eq_hf[1] < 32768 &&
                         p->eq_lf[2] > 0 && p->eq_lf[2] < 32768 && p->eq_hf[2] > 0 && p->eq_hf[2] < 32768;
    if (c->no_tier0 || !coef_ok || b > T0_BRIGHT_MAX) return 2;
    return c->no_loskip ? 1 : 0;
}

static int launch_decode(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                         const crthip_line *d_lines, void *d_out, size_t ostride)
{
    const int min_tier = decoder_min_tier(c, p);
    const bool wide = c->px_tile ? c->px_tile == 32 : p->outw >= 1280;
    /* lines per output row when the picture is shorter than the raster: one pass per rank */
    const unsigned span = (unsigned) p->outh + p->v_fac;
    const int passes = span >= (unsigned) c->sd.lines ? 1 : (int) (((unsigned) c->sd.lines + span - 1) / (span ? span : 1));
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        const int total = n * S::LINES;
        const dim3 grid((total + 63) / 64), block(64);
        unsigned char *o = (unsigned char *) d_out;
        ProfScope ps(c, CRTHIP_K_DECODE);
        for (int rank = 0; rank < passes; rank++) {
#define CRTHIP_LAUNCH_DECODE(T, B3) \
    do { if (wide) hipLaunchKernelGGL((k_decode<S, T, B3, 32>), grid, block, 0, c->stream, *p, n, d_inp, c->fstride, d_lines, o, ostride, min_tier, rank); \
         else hipLaunchKernelGGL((k_decode<S, T, B3, 16>), grid, block, 0, c->stream, *p, n, d_inp, c->fstride, d_lines, o, ostride, min_tier, rank); } while (0)
            /* every tier >= min_tier gets its pass; waves without lines of that tier leave at once */
            if (p->out_bpp == 3) {
                if (min_tier <= 0) CRTHIP_LAUNCH_DECODE(0, true);
                if (min_tier <= 1) CRTHIP_LAUNCH_DECODE(1, true);
                if (min_tier <= 2) CRTHIP_LAUNCH_DECODE(2, true);
                CRTHIP_LAUNCH_DECODE(3, true);
            } else {
                if (min_tier <= 0) CRTHIP_LAUNCH_DECODE(0, false);
                if (min_tier <= 1) CRTHIP_LAUNCH_DECODE(1, false);
                if (min_tier <= 2) CRTHIP_LAUNCH_DECODE(2, false);
                CRTHIP_LAUNCH_DECODE(3, false);
            }
#undef CRTHIP_LAUNCH_DECODE
        }
        return CRTHIP_OK;
    });
}

int crthip_decode(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, const crthip_line *d_lines,
                  void *d_out, size_t ostride)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (p->out_bpp == 0) return CRTHIP_OK;
    if (!d_inp || !d_lines || !d_out) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rc = launch_decode(c, p, n, d_inp, d_lines, d_out, ostride);
    HIPCHK(c, hipGetLastError());
    return rc;
}

/* one chunk of a batch: fields [first, first+n) on the context's current stream */
static int fieldpass_chunk(crthip_ctx *c, const crthip_params *p, int enc, int first, int n,
                           const void *d_images, size_t istride, void *d_out, size_t ostride, crthip_state *d_state)
{
    const unsigned char *img = (const unsigned char *) d_images + (size_t) first * istride;
    unsigned char *out = (unsigned char *) d_out + (size_t) first * ostride;
    crthip_state *st = d_state + first;
    signed char *inp = c->d_inp + (size_t) first * c->fstride;
    signed char *analog = c->d_analog + (size_t) first * c->fstride;
    crthip_line *ln = c->d_lines + (size_t) first * c->sd.lines;
    if (c->system == CRTHIP_SYSTEM_NTSCVHS) {
        /* VHS noise follows the C library's rand() stream, not the LCG: the fused encoder (margins + active
         * rectangle = every sample of the field) runs with noise 0 into analog[], then the dedicated noise
         * kernels (which also produce rn) */
        int r = CRTHIP_OK;
        if (enc == 0) {
            crthip_params clean = *p;
            clean.noise = 0;
            r = dispatch_system(c->system, c->pattern, [&](auto tag) {
                using S = decltype(tag);
                launch_encoder<S, true>(c, &clean, n, img, istride, analog, st, 1);
                hipLaunchKernelGGL((k_encoder_state<S>), dim3((n + 63) / 64), dim3(64), 0, c->stream, *p, n, st);
                return CRTHIP_OK;
            });
        } else {
            hipMemsetAsync(analog, 0, c->fstride * (size_t) n, c->stream);   /* crt_modulate refused the format */
        }
        if (r) return r;
        if (p->out_bpp == 0) return CRTHIP_OK;
        unsigned *saved = c->d_vhs_hist;
        c->d_vhs_hist = saved + (size_t) first * 32;
        r = crthip_noise(c, p, n, analog, inp, st);
        c->d_vhs_hist = saved;
        if (r) return r;
        r = launch_sync(c, p, n, inp, st, ln, 0);
        if (r) return r;
        return launch_decode(c, p, n, inp, ln, out, ostride);
    }
    int rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        if (enc == 0) {
            /* the encoder writes the noisy field straight into inp[]; analog[] is never materialised */
            launch_encoder<S, true>(c, p, n, img, istride, inp, st, 1);
            hipLaunchKernelGGL((k_encoder_state<S>), dim3((n + 63) / 64), dim3(64), 0, c->stream, *p, n, st);
        } else {
            /* invalid input format: crt_modulate is a no-op, the decoder sees a clean field + noise */
            hipMemsetAsync(analog, 0, c->fstride * (size_t) n, c->stream);
            constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
            hipLaunchKernelGGL((k_noise<S>), dim3((n * CHUNKS + 255) / 256), dim3(256), 0, c->stream,
                               *p, n, analog, inp, c->fstride, st, c->d_jump16);
        }
        return CRTHIP_OK;
    });
    if (rc) return rc;
    if (p->out_bpp != 0) {
        rc = launch_sync(c, p, n, inp, st, ln, 1);
        if (rc) return rc;
        rc = launch_decode(c, p, n, inp, ln, out, ostride);
        if (rc) return rc;
    }
    return CRTHIP_OK;
}

int crthip_fieldpass(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                     void *d_out, size_t ostride, crthip_state *d_state)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (!d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    if (c->system == CRTHIP_SYSTEM_NTSCVHS && !c->d_vhs_hist)
        return set_err(c, CRTHIP_E_ARG, "VHS: no generator histories bound (crthip_vhs_bind_history)", hipSuccess);
    int enc = check_encoder(c, p);
    if (enc < 0) return enc;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->cap_fields) {
        rc = crthip_reserve(c, n);
        if (rc) return rc;
    }
    /* Fields are independent, so a large batch is cut into chunks that alternate between the
     * caller's stream and an internal one: the latency-bound kernels of one chunk (sync chain,
     * margins, launch gaps) then overlap the VALU-bound kernels of the other.  The internal stream
     * is fenced by events on both sides, so to the caller everything is still ordered on ITS stream. */
    const int nchunks = (c->overlap_chunks > 1 && n >= 256 * c->overlap_chunks && !c->prof) ? c->overlap_chunks : 1;
    if (nchunks == 1) {
        rc = fieldpass_chunk(c, p, enc, 0, n, d_images, istride, d_out, ostride, d_state);
    } else {
        if (!c->aux_stream) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        hipStream_t main_stream = c->stream;
        HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        const int per = ((n + nchunks - 1) / nchunks + 3) & ~3;
        for (int k = 0, first = 0; first < n && rc == CRTHIP_OK; k++, first += per) {
            const int cnt = n - first < per ? n - first : per;
            c->stream = (k & 1) ? c->aux_stream : main_stream;
            rc = fieldpass_chunk(c, p, enc, first, cnt, d_images, istride, d_out, ostride, d_state);
        }
        c->stream = main_stream;
        HIPCHK(c, hipEventRecord(c->ev_join, c->aux_stream));
        HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
    }
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

int crthip_set_pixel_tile(crthip_ctx *c, int px)
{
    if (!c || (px != 0 && px != 16 && px != 32)) return CRTHIP_E_ARG;
    c->px_tile = px;
    c->ac_tile = px;
    return CRTHIP_OK;
}

int crthip_sequence(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                    void *d_out, size_t ostride, const void *d_out_init, crthip_state *d_state, int *passes_out)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (!d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    if (c->system == CRTHIP_SYSTEM_NTSCVHS)
        return set_err(c, CRTHIP_E_ARG, "sequence mode: the VHS rand() stream is a chain over fields, use crthip_fieldpass per field", hipSuccess);
    if (p->blend) return set_err(c, CRTHIP_E_ARG, "sequence mode needs blend == 0 (blend is a recurrence over fields)", hipSuccess);
    if (p->out_bpp == 0) return CRTHIP_OK;
    int enc = check_encoder(c, p);
    if (enc != 0) return enc < 0 ? enc : set_err(c, CRTHIP_E_ARG, "sequence mode: unknown input pixel format", hipSuccess);
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->cap_fields) {
        rc = crthip_reserve(c, n);
        if (rc) return rc;
    }
    const int outh = p->outh;
    const size_t pitch = (size_t) p->outw * p->out_bpp;
    /* scratch: guess[n] (int2), changed flag, owner[n][outh] (u8), latest[n][outh] (int) */
    const size_t need = sizeof(int2) * (size_t) n + 256 + (size_t) n * outh + 256 + sizeof(int) * (size_t) n * outh;
    if (need > c->seq_cap) {
        if (c->d_seq) hipFree(c->d_seq);
        c->d_seq = 0; c->seq_cap = 0;
        if (hipMalloc((void **) &c->d_seq, need) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc sequence scratch", hipSuccess);
        c->seq_cap = need;
    }
    int2 *guess = (int2 *) c->d_seq;
    int *changed = (int *) (c->d_seq + sizeof(int2) * (size_t) n);
    unsigned char *owner = c->d_seq + sizeof(int2) * (size_t) n + 256;
    int *latest = (int *) (owner + (((size_t) n * outh + 255) & ~(size_t) 255));
    const dim3 gn((n + 63) / 64), b64(64);

    crthip_state first;
    HIPCHK(c, hipMemcpyAsync(&first, d_state, sizeof(first), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(k_seq_rn, gn, b64, 0, c->stream, n, d_state, c->whole_field);
    /* encode every field (noise fused), ccf presets */
    rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        launch_encoder<S, true>(c, p, n, d_images, istride, c->d_inp, d_state, 1);
        return CRTHIP_OK;
    });
    if (rc) return rc;
    /* first guess: nobody's sync state moves */
    {
        int2 *h = (int2 *) malloc(sizeof(int2) * (size_t) n);
        if (!h) return CRTHIP_E_NOMEM;
        for (int k = 0; k < n; k++) h[k] = make_int2(first.hsync, first.vsync);
        hipError_t e = hipMemcpyAsync(guess, h, sizeof(int2) * (size_t) n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        free(h);
        HIPCHK(c, e);
    }
    int passes = 0;
    for (;;) {
        passes++;
        HIPCHK(c, hipMemsetAsync(changed, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_seq_load, gn, b64, 0, c->stream, n, d_state, guess, make_int2(first.hsync, first.vsync));
        rc = dispatch_system(c->system, c->pattern, [&](auto tag) {
            using S = decltype(tag);
            hipLaunchKernelGGL((k_encoder_state<S>), gn, b64, 0, c->stream, *p, n, d_state);   /* ccf preset, crt_ntsc.c:325-329 */
            return CRTHIP_OK;
        });
        if (rc) return rc;
        rc = launch_sync(c, p, n, c->d_inp, d_state, c->d_lines, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_seq_compare, gn, b64, 0, c->stream, n, d_state, guess, changed);
        int flag = 0;
        HIPCHK(c, hipMemcpyAsync(&flag, changed, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!flag || passes > n + 1) break;
    }
    if (passes_out) *passes_out = passes;
    /* rn after each field, decode, weave */
    hipLaunchKernelGGL(k_advance_rn, gn, b64, 0, c->stream, n, d_state, c->whole_field);
    rc = launch_decode(c, p, n, c->d_inp, c->d_lines, d_out, ostride);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(owner, 0, (size_t) n * outh, c->stream));
    hipLaunchKernelGGL(k_seq_rows, dim3((n * c->sd.lines + 255) / 256), dim3(256), 0, c->stream, n, c->sd.lines, outh, c->d_lines, owner);
    hipLaunchKernelGGL(k_seq_latest, dim3((outh + 63) / 64), b64, 0, c->stream, n, outh, owner, latest);
    hipLaunchKernelGGL(k_seq_weave, dim3((unsigned) n * (unsigned) outh), dim3(256), 0, c->stream, n, outh, pitch,
                       (unsigned char *) d_out, ostride, (const unsigned char *) d_out_init, latest);
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

int crthip_set_overlap(crthip_ctx *c, int chunks)
{
    if (!c || chunks < 1 || chunks > 64) return CRTHIP_E_ARG;
    c->overlap_chunks = chunks;
    return CRTHIP_OK;
}

int crthip_vhs_bind_history(crthip_ctx *c, unsigned *d_hist)
{
    if (!c || c->system != CRTHIP_SYSTEM_NTSCVHS) return CRTHIP_E_ARG;
    c->d_vhs_hist = d_hist;
    return CRTHIP_OK;
}

int crthip_set_exact(crthip_ctx *c, int on)
{
    if (!c) return CRTHIP_E_ARG;
    c->force_exact = on == 1;
    c->no_tier0 = on == 2;      /* 2: allow the 24-bit tier but not the 64-bit-mad ones */
    c->no_loskip = on == 3;     /* 3: allow the 64-bit-mad tier but keep the I/Q low cascades */
    return CRTHIP_OK;
}

int crthip_profile_enable(crthip_ctx *c, int on)
{
    if (!c) return CRTHIP_E_ARG;
    c->prof = on != 0;
    return CRTHIP_OK;
}

int crthip_profile_read(crthip_ctx *c, double total_ms[CRTHIP_K_COUNT], int launches[CRTHIP_K_COUNT])
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->npend; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->pend[i].a, c->pend[i].b) == hipSuccess) {
            c->prof_ms[c->pend[i].k] += ms;
            c->prof_n[c->pend[i].k]++;
        }
        hipEventDestroy(c->pend[i].a);
        hipEventDestroy(c->pend[i].b);
    }
    c->npend = 0;
    for (int k = 0; k < CRTHIP_K_COUNT; k++) {
        if (total_ms) total_ms[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_n[k];
        c->prof_ms[k] = 0.0;
        c->prof_n[k] = 0;
    }
    return CRTHIP_OK;
}

void *crthip_malloc(crthip_ctx *c, size_t bytes)
{
    void *p = 0;
    if (!c || hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return 0;
    return p;
}

void crthip_free(crthip_ctx *c, void *d)
{
    if (c && d) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); hipFree(d); }
}

int crthip_upload(crthip_ctx *c, void *d, const void *h, size_t bytes)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CRTHIP_OK;
}

int crthip_download(crthip_ctx *c, void *h, const void *d, size_t bytes)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CRTHIP_OK;
}

int crthip_memset(crthip_ctx *c, void *d, int value, size_t bytes)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipMemsetAsync(d, value, bytes, c->stream));
    return CRTHIP_OK;
}

}  /* extern "C" */
