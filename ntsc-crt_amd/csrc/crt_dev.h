/*
 * crt_dev.h -- shared by the HIP translation units of libcrthip (gfx950 / MI355X only).
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
 * Translation units (stage names M0-M6 / D0-D10 as in DESIGN.md):
 *   crt_encode.hip  M4-M6   k_template / k_skeleton + k_margin (skeleton), k_active, k_nes_table + k_active_nes
 *   crt_noise.hip   D1      k_noise (LCG, jump-ahead tables), k_vhs_noise + k_vhs_tail (libc rand() model)
 *   crt_sync.hip    D2-D7   k_hsync_wave (a wave per field: vertical search, hsync fixed point, burst chains, line table), k_vsync, k_bloom
 *   crt_decode.hip  D8-D10  k_decode (3x 3-band IIR equaliser, resample, YIQ->RGB, row duplication)
 *   crt_host.hip            context, the crthip_* C ABI, sequence mode
 * Integer-only; signed overflow wraps (-fwrapv), >> of negatives is arithmetic, / truncates --
 * identical to the reference on x86-64.
 */
#ifndef CRT_DEV_H
#define CRT_DEV_H

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
/* defaults shared by every system; the per-system structs below override what differs */
struct SysCommon {
    static constexpr int VRES = 262;
    static constexpr int CCS = 4, CB_LEN = 40;          /* CRT_CC_SAMPLES; CB_CYCLES * CRT_CB_FREQ */
    static constexpr int VS_SEP_END = 0;
    static constexpr bool IS_NES = false;               /* PPU-pixel input (crt_nes.c) */
    static constexpr bool NES_TIMING = false;           /* setup_field skeleton, burst on the active lines only */
    static constexpr bool IS_VHS = false;
    static constexpr bool LINE_ROWS = false;            /* carrier row = line class + dot_crawl_offset */
    static constexpr bool BANDLIMIT = true;             /* CRT_DO_BANDLIMITING */
    static constexpr bool IIR_Y_NEAR = true;            /* luma low-pass coefficient >= 1024 (Q11): every system but VHS */
    static constexpr bool FIELD_ROWS = true;            /* source row offset by field parity, crt_ntsc.c:258 */
    static constexpr int EQU_A_LO = 0, EQU_A_HI = 3, EQU_B_LO = 7, EQU_B_HI = 9;   /* crt_ntsc.c:211 */
    static constexpr int VS_LO = 4, VS_HI = 6;          /* crt_ntsc.c:217 */
    static constexpr bool VS_BY_FIELD = true;
    static constexpr int CCF_SHIFT = 0;                 /* ccf preset row = (line + CCF_SHIFT) % VPER */
};
template <int CC_LINE>
struct RgbTiming : SysCommon {   /* crt_ntsc.h:25-109 / crt_ntscvhs.h / crt_template.h */
    static constexpr int HRES = CC_LINE * 4 / 10;
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
};
template <int CC_LINE>
struct NesTiming : SysCommon {   /* crt_nes.h:30-126, crt_nesrgb.h, crt_snes.h (PPU pixels on a 341 px line) */
    static constexpr int HRES = CC_LINE * 4 / 10;
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
    static constexpr int LAV_BEG = 58 * HRES / 341;     /* crt_nes.h:114: where the border colour starts (NES_BORDER builds) */
    static constexpr bool IS_NES = true, NES_TIMING = true, LINE_ROWS = true, BANDLIMIT = false, FIELD_ROWS = false;
};
struct SysNTSC : RgbTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSC, PATTERN = 1; };
struct SysNTSC0 : RgbTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSC, PATTERN = 0; };
struct SysVHS : RgbTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSCVHS, PATTERN = 1; static constexpr bool IS_VHS = true, IIR_Y_NEAR = false; };
struct SysVHS0 : RgbTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NTSCVHS, PATTERN = 0; static constexpr bool IS_VHS = true, IIR_Y_NEAR = false; };
struct SysNES2 : NesTiming<2273> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 2; };
struct SysNES1 : NesTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 1; };
struct SysNES0 : NesTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NES, PATTERN = 0; };
/* SURVEY.md 8(f4).  NES-RGB (crt_nesrgb.c): the NES's timing and skeleton around an RGB image, WHITE_LEVEL 100 */
template <int CC_LINE> struct NesRgbTiming : NesTiming<CC_LINE> { static constexpr int WHITE = 100; static constexpr bool IS_NES = false; };
struct SysNESRGB2 : NesRgbTiming<2273> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NESRGB, PATTERN = 2; };
struct SysNESRGB1 : NesRgbTiming<2275> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NESRGB, PATTERN = 1; };
struct SysNESRGB0 : NesRgbTiming<2280> { static constexpr int SYSTEM = CRTHIP_SYSTEM_NESRGB, PATTERN = 0; };
/* SNES (crt_snes.c/h): PPU-pixel line timing, NTSC levels and the NTSC-style skeleton, carriers per line class */
struct SysSNES : NesTiming<2273> {
    static constexpr int SYSTEM = CRTHIP_SYSTEM_SNES, PATTERN = 1;
    static constexpr int WHITE = 100, BURST = 20, BLACK = 7, BLANK = 0, SYNC = -40;
    static constexpr int HTHR = 4 * SYNC, VTHR = 94 * SYNC;
    static constexpr bool IS_NES = false, NES_TIMING = false;
    static constexpr int EQU_A_HI = 2, VS_LO = 3;       /* crt_snes.h:137-146 */
    static constexpr bool VS_BY_FIELD = false;          /* crt_snes.c:216-218 */
    static constexpr int CCF_SHIFT = 3;                 /* crt_snes.c:240 */
};
/* template system (crt_template.c/h): NTSC timing, two line classes, band limit on */
struct SysTEMP : RgbTiming<2275> {
    static constexpr int SYSTEM = CRTHIP_SYSTEM_TEMP, PATTERN = 1;
    static constexpr int VPER = 2;
    static constexpr bool LINE_ROWS = true;
    static constexpr int EQU_A_HI = 2, VS_LO = 3;
    static constexpr int CCF_SHIFT = 3;
};
/* Casio PV-1000 (crt_pv1k.c/h): 5 samples per chroma cycle, 1920 samples per line, 5 line classes */
struct SysPV1K : SysCommon {
    static constexpr int SYSTEM = CRTHIP_SYSTEM_PV1K, PATTERN = 1;
    static constexpr int HRES = 2304 * 5 / 6;
    static constexpr int INPUT_SIZE = HRES * VRES;
    static constexpr int TOP = 21, BOT = 261, LINES = BOT - TOP;
    static constexpr int VPER = 5, CCS = 5, CB_LEN = 50;
    static constexpr int HWIN = 8, VWIN = 8;
    static constexpr int WHITE = 100, BURST = 20, BLACK = 7, BLANK = 0, SYNC = -40;
    static constexpr int HTHR = 4 * SYNC, VTHR = 94 * SYNC;
    static constexpr int LINE_NS = 71 * 892;
    static constexpr int SYNC_BEG = 3 * 892 * HRES / LINE_NS;
    static constexpr int BW_BEG = 6 * 892 * HRES / LINE_NS;
    static constexpr int CB_BEG = 8 * 892 * HRES / LINE_NS;
    static constexpr int AV_BEG = 16 * 892 * HRES / LINE_NS;
    static constexpr int AV_LEN = 55 * 892 * HRES / LINE_NS;
    static constexpr bool LINE_ROWS = true;
    static constexpr int EQU_A_HI = -1;                 /* crt_pv1k.c:197: only lines 7..9 */
    static constexpr int VS_LO = 258, VS_HI = 260;      /* crt_pv1k.c:204 */
    static constexpr int CCF_SHIFT = 3;
};

static_assert(SysNTSC::HRES == 910 && SysNTSC::AV_BEG == 156 && SysNTSC::AV_LEN == 753 &&
              SysNTSC::SYNC_BEG == 21 && SysNTSC::CB_BEG == 97, "NTSC timing (SURVEY.md section 8)");
static_assert(SysNES2::HRES == 909 && SysNES2::AV_LEN == 682 && SysNES0::HRES == 912 &&
              SysNES0::AV_LEN == 684, "NES timing (SURVEY.md section 8)");
static_assert(SysPV1K::HRES == 1920 && SysPV1K::AV_LEN == 1487, "PV-1000 timing");

/* carrier-table row of analog line n of a field (crthip_params.burst / modI / modQ):
 * line class + dot_crawl_offset (kept unreduced inside 0..CRTHIP_DCO_MAX, else reduced mod VPER), or the
 * field==frame flag of the NTSC / VHS encoder */
template <class S> __device__ __forceinline__ int carrier_row(int n, int field, int frame, int aux)
{
    if constexpr (S::LINE_ROWS) {
        int dco = aux;
        if (dco < 0 || dco > CRTHIP_DCO_MAX) dco = ((dco % S::VPER) + S::VPER) % S::VPER;
        return n % S::VPER + dco;
    } else {
        return (field & 1) == (frame & 1);
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* (r6) The fused path's own signal layout.  The reference addresses analog[] / inp[] FLAT: line n starts at n * HRES (910:   */
/* no line of a field starts on a cache line, and the encoder's 753-byte row stores straddle the memory's sectors at both     */
/* ends -- a third of k_active's time, profiles/r05_experiments.txt section 1).  That layout is a property of `struct CRT`,   */
/* i.e. of the stage-level entry points and the drop-in mirrors; the signal crthip_fieldpass keeps in its own workspace       */
/* between its encoder and its decoder is nobody's business, and lives in PADDED lines:                                        */
/*     sample (line n, column c) of the field  ->  shift + n * PITCH + c            PITCH = 1024 (1920-sample lines: 2048)    */
/*     and the first `padv` samples of line n + 1 a SECOND time behind line n (columns HRES .. HRES + padv)                   */
/* `shift` puts the active rows on 128-byte boundaries.  Because of the copies, every window of the flat signal that starts in */
/* line n at column c and is no longer than HRES + padv - c bytes is contiguous here too, at shift + n * PITCH + c -- that is   */
/* every sync / burst window (80 / 48 bytes) wherever it starts, and every decoder window (DECWIN bytes from crthip_line.pos)  */
/* unless the line's hsync is far from lock (xpos > HRES + padv - DECWIN: hsync beyond about +88 for NTSC).  Those few lines   */
/* are copied by the sync kernel into a scratch row of their own behind the field (lines VRES + 1 ...), and their table entry  */
/* points there: the decoders never know.  Line VRES holds the mirrored struct tail (CRTHIP_TAIL) like the flat layout does.   */
/* ------------------------------------------------------------------------------------------------------------ */
template <class S> struct PadGeom {
    static constexpr int PITCH = (S::HRES + 64 + 127) / 128 * 128;
    static constexpr int PADW = PITCH - S::HRES;                       /* bytes behind a line (114 for NTSC) */
    static constexpr int PADC = PADW < 96 ? PADW : 96;                 /* ... of which the margin kernel fills at most this many with the copy (what the
                                                                          windows need is 80: six 16-byte chunks per line, every store counts there) */
    static constexpr int DECWIN = (((S::AV_LEN + 3) / 4 + 15) / 16) * 64;   /* bytes every decoder shape reads from crthip_line.pos at most */
    static constexpr int SCR_LINE0 = S::VRES + 1;                      /* first scratch row (one per decoded line) */
    static constexpr size_t FSTRIDE = (size_t) (S::VRES + 1 + S::LINES) * PITCH;
    static_assert(DECWIN + 16 <= PITCH && PADW >= 96, "a decoder window fits a scratch row; the pad holds the sync windows");
};
/* what the host decides per field-pass (crt_fused_layout, crt_host.hip) */
struct sig_layout {
    int pitch;        /* bytes between line starts: HRES = the reference's flat layout, PadGeom::PITCH = padded */
    int shift;        /* padded: bytes in front of line 0 */
    int padv;         /* padded: valid copy bytes behind every line */
    int wrap;         /* samples of an active row beyond its line's end: xo + destw - HRES, >= 0 */
    size_t fstride;   /* bytes between fields */
};
/* offset of flat sample index `flat` (>= 0) in a padded field, without the shift */
template <class S> __device__ __forceinline__ int sig_phys(int flat)
{
    return flat + (int) ((unsigned) flat / (unsigned) S::HRES) * PadGeom<S>::PADW;
}

#define CRTHIP_LINE_EXACT 0x40000000      /* bit in crthip_line.nrows: outside the 24-bit envelope */
#define CRTHIP_LINE_NROWS_MASK 0xffff     /* crthip_line.nrows bits 0-15: rows written              */
#define CRTHIP_LINE_RANK_SHIFT 16         /* bits 16-27: rank among lines starting on the same row   */
#define CRTHIP_LINE_RANK_MASK  0xfff
#define CRTHIP_LINE_WIDE  0x10000000      /* bit 28: carrier << 7 is no 24-bit multiplier any more (decoder tier 0 -> 1) */
#define CRTHIP_LINE_NOT64 0x20000000      /* bit 29: outside the no-wrap envelope of the 64-bit-mad decoder */
#define CRTHIP_LINE_KEEPLO 0x80000000     /* bit 31: chroma too strong to drop the I/Q low cascades (crthip_params.loskip_wave_max) */
#define LOSKIP_WAVE_MAX   65532           /* |wave[k]| bound of decoder tier 0's products: (wave << 7) fits 24 bits; also the
                                             no-low-cascade bound for arbitrary inp[] (|s * wave >> 9| <= 16383) */
#define T0_WAVE_MAX       120000          /* |wave[k]| bound of decoder tiers 0 and 1 */
#define T0_BRIGHT_MAX     2600            /* |bright| bound of decoder tiers 0 and 1  */
#define FAST_WAVE_MAX     524288          /* |wave[k]| bound of the fast decoder, 2^19 */
#define FAST_BRIGHT_MAX   130000          /* |bright| bound of the fast decoder        */
#define LCG_MUL 214019u          /* crt_core.c:359 */
#define LCG_ADD 140327895u

/* decoder tier flags of a line (crthip_line.nrows) from its carrier amplitude.  4 samples per chroma cycle: wave0 / wave1
 * are the carriers themselves.  5 samples (PV-1000): they are dci / dcq and the five carriers are
 * ((dci * cos_i + dcq * sin_i) >> 15) * saturation with |cos|, |sin| <= 32767 (crt_core.c:497-505), so
 * |carrier| <= (|dci| + |dcq| + 1) * |saturation|. */
template <int CCS>
__device__ __forceinline__ int line_tier_flags(int wave0, int wave1, int saturation, int loskip_wave_max, bool reads_tail)
{
    /* a window that runs past the end of the field reads the struct members mirrored there (CRTHIP_TAIL): any byte value */
    if (reads_tail && loskip_wave_max > LOSKIP_WAVE_MAX) loskip_wave_max = LOSKIP_WAVE_MAX;
    long a;
    const long a0 = wave0 < 0 ? -(long) wave0 : wave0, a1 = wave1 < 0 ? -(long) wave1 : wave1;
    if (CCS == 4) a = a0 > a1 ? a0 : a1;
    else a = (a0 + a1 + 1) * (saturation < 0 ? -(long) saturation : (long) saturation);
    if (a > FAST_WAVE_MAX) return CRTHIP_LINE_EXACT;
    if (a > T0_WAVE_MAX) return CRTHIP_LINE_NOT64;
    int f = 0;
    if (a > LOSKIP_WAVE_MAX) f |= CRTHIP_LINE_WIDE;
    if (a > loskip_wave_max) f |= (int) CRTHIP_LINE_KEEPLO;
    return f;
}

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
/* streaming variants for data nobody touches again on the device (the input image, the decoded picture) */
/* the streamed rows are only 4-byte aligned (odd output widths, row pitch): say so in the type */
typedef v4i v4i_a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) v4i_a4 g_v4i;
__device__ __forceinline__ void gstore16u_nt(unsigned long long a, v4i v) { __builtin_nontemporal_store(v, (g_v4i *) a); }
__device__ __forceinline__ v4i gload16u_nt(unsigned long long a) { return __builtin_nontemporal_load((const g_v4i *) a); }
__device__ __forceinline__ unsigned gload32(unsigned long long a) { return *(const g_u32 *) a; }
__device__ __forceinline__ void gstore32(unsigned long long a, unsigned v) { *(g_u32 *) a = v; }
__device__ __forceinline__ unsigned gload8(unsigned long long a) { return *(const g_u8 *) a; }
__device__ __forceinline__ void gstore8(unsigned long long a, unsigned v) { *(g_u8 *) a = (unsigned char) v; }

__device__ __forceinline__ v4i load16u(const void *p) { return ((const unaligned16 *) p)->v; }
__device__ __forceinline__ void store16u(void *p, v4i v) { ((unaligned16 *) p)->v = v; }
__device__ __forceinline__ int load4u(const void *p) { return ((const unaligned4 *) p)->v; }
__device__ __forceinline__ void store4u(void *p, int v) { ((unaligned4 *) p)->v = v; }

/* Ordering point between the two phases of an LDS tile in a kernel whose workgroup is ONE wave.  The LDS unit
 * executes a wave's ds_* instructions in order, so lane A's write is visible to lane B's later read without any
 * barrier; what is needed is only that the compiler keeps the program order.  __syncthreads() would do, but it
 * also emits s_waitcnt vmcnt(0): every tile hand-over then waits for ALL outstanding global memory operations,
 * in the decoder the stores of the previous drain. */
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
/* the same in ONE vector instruction: the low half of v_mad_u64_u32(rn, MUL, {ADD, 0}) is the wrapped 32-bit result
 * (v_mul_lo_u32 + v_add_u32 otherwise).  `add_pair` = {LCG_ADD, 0} kept in a register pair by the caller. */
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lcg_step_mad64(unsigned rn, v2u add_pair)
{
    v2u r;
    unsigned long long carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(rn), "s"(LCG_MUL), "v"(add_pair));
    return r.x;
}
/* vgpr * sgpr + vgpr with the addend already in a VGPR (24-bit operands) */
__device__ __forceinline__ int mad24_vv(int v, int s_uniform, int acc_v)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(s_uniform), "v"(acc_v));
    return r;
}
/* v_mad_i64_i32: vgpr * sgpr + 64-bit register pair (used as "multiply, shift and accumulate in one instruction": the
 * state of a one-pole filter lives in the HIGH half of a pair, the multiplier is pre-shifted so that the wanted
 * quotient bits land there; see eq_step64_yiq in crt_decode_lane.h and the encoder's low-passes in crt_encode.hip) */
__device__ __forceinline__ long mad64_vs(int d, int m_uniform, long acc)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(d), "s"(m_uniform), "v"(acc));
    return r;
}
__device__ __forceinline__ int pair_hi(long v) { return (int) (v >> 32); }
/* a + (b >> 16), b taken as a signed 32-bit value: one SDWA add (the high word of b, sign-extended) */
__device__ __forceinline__ int add_hiword(int a, int b)
{
    int r;
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* (a >> 16) - b, a taken as a signed 32-bit value */
__device__ __forceinline__ int sub_hiword(int a, int b)
{
    int r;
    asm("v_sub_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* (a >> 16) + (b >> 16), both taken as signed 32-bit values */
__device__ __forceinline__ int add_hiwords(int a, int b)
{
    int r;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* a with its HIGH half replaced by the low 16 bits of (b >> 16) + (c >> 16): the second member of a packed pair */
__device__ __forceinline__ int add_hiwords_to_hi(int a, int b, int c)
{
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a) : "v"(b), "v"(c));
    return a;
}
/* acc + lo16(v) * lo16(k) + hi16(v) * hi16(k), signed halves, k wave-uniform */
__device__ __forceinline__ int dot2_vs(int v, int k_uniform, int acc)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(k_uniform), "v"(acc));
    return r;
}
/* the low 32 bits of a * b through the (full rate) 64-bit multiply-add: the wrapped 32-bit product, vgpr * vgpr */
__device__ __forceinline__ int mul_lo_mad64(int a, int b)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
    return (int) r;
}
/* v_mad_i64_i32 with a zero addend */
__device__ __forceinline__ long mad64_vs0(int d, int m_uniform)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(d), "s"(m_uniform));
    return r;
}
/* (a << sh) | b in one instruction */
__device__ __forceinline__ unsigned lshl_or(unsigned a, int sh, unsigned b)
{
    unsigned r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(sh), "v"(b));
    return r;
}
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

/* fields per launch up to which the scanline-parallel kernel shapes are chosen automatically: lane-per-scanline needs
 * n * 240 / 64 wavefronts >= a few per SIMD (1024 SIMDs) to hide its serial chains.  Measured crossovers
 * (profiles/r02_shape_sweep.txt): decoders between 128 and 256 fields, encoders between 256 and 512 (a field-pass of
 * 512 fields: 0.672 ms with the lane-shaped encoder, 0.734 with the row-shaped one) */
#define ROWS_SHAPE_MAX_FIELDS 128          /* k_decode_row */
#define ROWS_SHAPE_MAX_FIELDS_ENC 256      /* k_active_row */
/* k_active's larger signal pieces cost LDS, i.e. occupancy: taken from this many waves (64 destination rows each) on -- about four
 * residency rounds at the occupancy they leave (1080p: 1638 fields; 640x480: 1792 fields).  Below: a second, half-empty round costs
 * more than the pieces give (profiles/r04_experiments.txt section 19, profiles/r05_experiments.txt sections 2-4) */
#define SIG_TILE64_MIN_WAVES_WIDE 6144     /* wide image tile (w >= 1280): 256-byte pieces */
/* narrow image tile, 128-byte pieces (12 waves per CU = 3072 at a time, against about 4096 with the 64-byte-piece tile): (r6, measured
 * by batch size, profiles/r06_ab_signal_tile_by_batch.txt) they pay where the wave count fills one round of theirs (2880 waves: -14 %)
 * and from where the small tile needs a second round (4800: -13 %, 5760: -10 %, 7680: -5 %); in between the large tile would run
 * a thin second round (3840 waves: +13 %), below the band both are equal */
#define SIG_TILE32_BAND_LO        2304
#define SIG_TILE32_ONE_ROUND      3072
#define SIG_TILE16_ONE_ROUND      4096
#define MARGIN_SIDE_MIN_FIELDS 512         /* launch_encoder: k_margin on the internal stream beside k_active from this many fields on */
#define SYNC_FPB4_MIN_FIELDS 768            /* k_hsync_wave: four fields per workgroup from here on, one below (crt_sync.hip) */
#define WIDE_LPW8_MAX_WAVES 1440           /* k_decode_wide: 8 scanlines per wave while 16 per wave would make fewer waves than this (1080p: < 96 fields; measured: 32 fields -10 %, 64 -4 %, 128 +4 %) */
#define WIDE_SHAPE_MIN_FIELDS 32           /* wide pictures: k_decode_wide instead of k_decode_row from here on (crt_decode.hip) */

/* LDS tile of 64 rows x T dwords that one wave accesses in two ways: LANE PER ROW (every lane its own row, all lanes the same
 * column) and COOPERATIVELY (row-contiguous 16-byte pieces: T / 4 lanes per row, each four consecutive dwords, as four dword --
 * or two ds_read2 / ds_write2 -- accesses).  A dword access is served in two groups of 32 lanes over 32 banks
 * (/opt/skills/guides/MI355X_MICROARCH.md, LDS).  The odd row stride T + 1 of rounds 1-5 makes the first pattern conflict-free
 * for every T, the second only for T >= 32: with T = 16 a group covers 8 rows x 4 pieces and rows r and r + 4 meet on three of
 * their four banks whatever the odd stride (2-way conflicts: SQ_LDS_BANK_CONFLICT 68 % of k_decode's LDS cycles in
 * profiles/r05_headline_sq_counters.json -- free for the stores, whose cycles are set by their data path, twice the cycles for
 * the drain's reads).  One pad dword per 32 / T ROWS instead of per row --
 *     dword (row, col)  ->  row * T + (row >> log2(32 / T)) + col             (T = 16: row pairs 33 apart; T = 8: row quads)
 * -- is conflict-free for BOTH patterns (exhaustive check over strides and pads: profiles/r06_lds_tile_layout.txt; T = 64, the
 * 256-byte signal pieces, stays 2-way cooperatively whatever the layout: one row's 16 pieces are 64 consecutive dwords), costs
 * nothing per access (the row term is a per-lane constant either way) and is a little smaller. */
template <int T> struct TileRows {
    static_assert(T == 8 || T == 16 || T == 32 || T == 64, "tile width in dwords");
    static constexpr int SH = T == 16 ? 1 : 2;                                        /* (T < 32) */
    static constexpr int DWORDS = T >= 32 ? 64 * (T + 1) : 64 * T + (64 >> SH);
    __host__ __device__ static constexpr int row(int r) { return T >= 32 ? r * (T + 1) : r * T + (r >> SH); }
};

/* (r6) Workgroup order.  Workgroup b of a launch takes work item block_item(b): itself (K = 0: consecutive workgroups -- which
 * the dispatcher deals round robin over the 8 XCDs and which are resident together -- work on neighbouring scanlines of the same
 * pictures), or the launch's `total` items dealt out in K strides of per = ceil(total / K): consecutive workgroups are then `per`
 * items apart, i.e. with K = the batch's fields on the SAME scanline group of consecutive pictures.  grid = K * per; items beyond
 * `total` leave at once.  Measured: profiles/r06_1080p_placement.txt, r06_block_order.txt */
__device__ __forceinline__ int block_item(unsigned b, int K, int per)
{
    return K ? (int) (b % (unsigned) K) * per + (int) (b / (unsigned) K) : (int) b;
}
struct block_order { int K, per; unsigned grid; };
static inline block_order make_block_order(int total, int K)
{
    block_order o;
    if (K <= 1 || K >= total) K = 0;
    o.K = K;
    o.per = K ? (total + K - 1) / K : 0;
    o.grid = K ? (unsigned) K * (unsigned) o.per : (unsigned) total;
    return o;
}

#define CRTHIP_MAX_CHUNKS 64               /* crthip_set_overlap */

/* sizes shared between kernels and the context */
#define NES_TAB_SIZE (512 * 12)            /* NES composite-sample table: 9-bit pixel x phase mod 12 */
#define SKEL_VARIANTS 12                   /* cached clean skeleton fields (k_skeleton): (field, frame) or field x dot_crawl_offset */
#define VHS_CHUNK 124                      /* samples per lane in the parallel region = 248 calls = 8 * 31 (248: measured slower) */
#define VHS_DIG_ROW 128                    /* bytes per coefficient row in digit form: 4 planes x 32 */
#define VHS_BLK   43                       /* calls per lane in the tail's window: 64 * 43 >= 3 * HRES + 3 */

/* first sample of the tail: a chunk boundary with at least 16 samples (>= 31 calls) before I0 + 1 */
__host__ __device__ constexpr int vhs_tail_start(int input_size, int hres)
{
    return (input_size - 25 * hres + 1 - 16) / VHS_CHUNK * VHS_CHUNK;
}


/* ------------------------------------------------------------------------- */
/* host side: context, dispatch by system, C ABI                               */
/* ------------------------------------------------------------------------- */
#define CRTHIP_MAX_RETIRED 64
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
    signed char *d_nes_tab;     /* NES: 512 x 12 composite-sample table (cached: black / white point) */
    bool nes_tab_valid;
    int nes_tab_black, nes_tab_white;
    signed char *d_skel;        /* SKEL_VARIANTS clean skeleton fields (cached: the burst table they were built from) */
    signed char *d_skel_alt, *d_nes_tab_alt;   /* second set: where the tables are built when the context's stream is being captured */
    hipStream_t table_stream;   /* ... and the (never captured) stream they are built on then; created on first use */
    bool skel_captured, nes_captured;   /* a captured graph may be reading the current skeleton set / NES table: that buffer is never written again
                                           (crt_run_encoder_prepare; tracked per table -- they are rebuilt independently of each other) */
    unsigned table_gen;         /* incremented by every table rebuild (crthip_table_generation) */
    signed char *retired[CRTHIP_MAX_RETIRED];   /* table sets that graphs may still read: freed by crthip_destroy */
    int n_retired;
    bool skel_valid;
    int skel_burst[CRTHIP_CARRIER_ROWS][CRTHIP_MAX_CCS];
    int skel_border[4];         /* ... NES_BORDER: flag, colour, black point, white point */
    int skel_yo;                /* ... and the first active line (NES timing: the burst is only on the active lines) */
    int shape;                  /* crthip_set_shape: 0 auto, 1 lane-per-scanline, 2 scanline-parallel */
    int row_tile;               /* CRTHIP_ROW_TILE: samples per tile of the scanline-parallel decoder, 32 (default) or 16 */
    int sync_kernel;            /* CRTHIP_SYNC_KERNEL: 0 by batch size, 2 / 3 = k_hsync_wave with 1 / 4 fields per workgroup (A/B measurements) */
    uint2 *d_jump1;             /* LCG affine maps of 0..15 steps */
    int *d_bloom;               /* bloom build, lane-per-scanline decoder: line_w histogram, cursors, slot -> line (crt_decode3.hip) */
    size_t bloom_cap;
    unsigned char *d_seq;       /* crthip_sequence scratch */
    size_t seq_cap;
    bool vhs_prechained;        /* crthip_seq_vhs_prechained: the bound histories already sit at the start of every field */
    int seq_guess_n;            /* crthip_seq_sync: the guess array holds the finals of a previous call for this many fields (warm restart) */
    unsigned *d_vhs_rows;       /* VHS: jump coefficients, (vhs_chunks + 1) x 31 words, then 31 x 64 (tail blocks) */
    int vhs_chunks;
    signed char *d_vhs_dig;     /* VHS: the same coefficients as signed byte digits, VHS_DIG_ROW bytes per row (matrix-core jump) */
    int vhs_mfma;               /* CRTHIP_VHS_MFMA: 1 (default) = the 31 x 31 jumps on the matrix cores, 0 = on the vector unit */
    int wide_decode;            /* CRTHIP_WIDE_DECODE: 1 (default) = wide pictures through k_decode_wide (crt_decode4.hip), 0 = A/B switch */
    unsigned *d_vhs_hist;       /* VHS: bound per-field generator histories (caller's memory) */
    unsigned *d_vhs_next;       /* VHS: where k_vhs_tail leaves the histories while k_vhs_noise still reads the old ones */
    int px_tile;                /* 0 = by output width, else 16 / 32 (tuning / tests) */
    int ac_tile;                /* encoder tile, same convention, by input width */
    int ac_tile_env;            /* CRTHIP_AC_TILE: overrides both (A/B measurements) */
    int wide_lpw_env;           /* CRTHIP_WIDE_LPW = 8 | 16: pins k_decode_wide's scanlines per wave (A/B measurements); 0 = by batch size */
    int sig_pad;                /* CRTHIP_SIG_PAD / crthip_set_signal_layout: 1 (default) = the fused path keeps its signal in padded lines, 0 = flat */
    size_t fstride_pad;         /* PadGeom::FSTRIDE of the context's system: d_inp is sized for it */
    sig_layout last_lay;        /* the layout of d_inp's current contents (crthip_fieldpass_signal) */
    int last_n;
    int wide_order_env, dec_order_env, act_order_env;   /* CRTHIP_WIDE_ORDER / _DEC_ORDER / _ACT_ORDER: workgroup order of k_decode_wide / k_decode / k_active
                                                           (block_item above): 0 = the default, K > 1 = that many strides, -1 = one stride per field, 1 = in order */
    int sig_tile_env;           /* CRTHIP_SIG_TILE = 16 | 32 | 64: pins k_active's small / large signal tile (A/B measurements); 0 = by batch size */
    int overlap_chunks;         /* crthip_fieldpass: chunks alternating between two streams (1 = off) */
    hipStream_t aux_stream;
    hipEvent_t ev_fork, ev_join, ev_chunk[CRTHIP_MAX_CHUNKS];
    hipEvent_t ev_mfork, ev_mjoin;   /* the margin kernel beside the active-video kernel (crt_encode.hip, launch_encoder) */
    int margin_side;            /* CRTHIP_MARGIN_SIDE: 1 (default) = large fused batches run k_margin on the internal stream beside k_active, 0 = in sequence (A/B) */
    bool prof;
    double prof_ms[CRTHIP_K_COUNT];
    int prof_n[CRTHIP_K_COUNT];
    struct Pending { int k; hipEvent_t a, b; } *pend;
    int npend, cappend;
    char err[256];
};

static inline int set_err(crthip_ctx *c, int code, const char *what, hipError_t e)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(ctx, CRTHIP_E_HIP, #call, e_); } while (0)

static inline void lcg_jump_host(unsigned k, unsigned *mul, unsigned *add)
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

template <class S> static inline bool sysdef_matches(const struct crt_sysdef &d)
{
    return d.hres == S::HRES && d.vres == S::VRES && d.input_size == S::INPUT_SIZE && d.top == S::TOP &&
           d.bot == S::BOT && d.vper == S::VPER && d.hsync_window == S::HWIN && d.vsync_window == S::VWIN &&
           d.hsync_thresh == S::HTHR && d.vsync_thresh == S::VTHR && d.sync_beg == S::SYNC_BEG &&
           d.bw_beg == S::BW_BEG && d.cb_beg == S::CB_BEG && d.av_beg == S::AV_BEG && d.av_len == S::AV_LEN &&
           d.vs_sep_end == S::VS_SEP_END && d.white_level == S::WHITE && d.burst_level == S::BURST &&
           d.black_level == S::BLACK && d.blank_level == S::BLANK && d.sync_level == S::SYNC &&
           d.cc_samples == S::CCS && d.cb_len == S::CB_LEN && (d.ppu_input != 0) == S::IS_NES &&
           (d.nes_timing != 0) == S::NES_TIMING && (d.field_rows != 0) == S::FIELD_ROWS &&
           (d.line_rows != 0) == S::LINE_ROWS && (d.y_freq != 0) == S::BANDLIMIT && d.equ_a_lo == S::EQU_A_LO &&
           d.equ_a_hi == S::EQU_A_HI && d.equ_b_lo == S::EQU_B_LO && d.equ_b_hi == S::EQU_B_HI && d.vs_lo == S::VS_LO &&
           d.vs_hi == S::VS_HI && (d.vs_by_field != 0) == S::VS_BY_FIELD && d.ccf_row_shift == S::CCF_SHIFT;
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
    if (system == CRTHIP_SYSTEM_NESRGB) {
        if (pattern == 2) return fn(SysNESRGB2{});
        if (pattern == 1) return fn(SysNESRGB1{});
        return fn(SysNESRGB0{});
    }
    if (system == CRTHIP_SYSTEM_SNES) return fn(SysSNES{});
    if (system == CRTHIP_SYSTEM_TEMP) return fn(SysTEMP{});
    if (system == CRTHIP_SYSTEM_PV1K) return fn(SysPV1K{});
    return CRTHIP_E_ARG;
}

struct ProfScope {
    crthip_ctx *c; int k; hipEvent_t a, b; bool on;
    ProfScope(crthip_ctx *ctx, int kernel) : c(ctx), k(kernel), on(ctx->prof)
    {
        if (on) {
            /* profiling is best effort: a failed event simply drops this sample (crthip_profile_read checks) */
            (void) hipEventCreate(&a); (void) hipEventCreate(&b);
            (void) hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope()
    {
        if (!on) return;
        (void) hipEventRecord(b, c->stream);
        if (c->npend == c->cappend) {
            int ncap = c->cappend ? c->cappend * 2 : 64;
            c->pend = (crthip_ctx::Pending *) realloc(c->pend, sizeof(*c->pend) * (size_t) ncap);
            c->cappend = ncap;
        }
        c->pend[c->npend].k = k; c->pend[c->npend].a = a; c->pend[c->npend].b = b;
        c->npend++;
    }
};


/* the internal second stream and its fork / join events (created on first use) */
static inline int crt_ensure_aux(crthip_ctx *c)
{
    if (c->aux_stream) return CRTHIP_OK;
    /* highest priority: what runs here are short latency-bound kernels beside a wide one on the caller's stream, and they
     * only help if their workgroups are dispatched at once instead of queueing behind the wide kernel's */
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { c->aux_stream = nullptr; return CRTHIP_E_HIP; }
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return CRTHIP_E_HIP;
    for (int k = 0; k < CRTHIP_MAX_CHUNKS; k++)
        if (hipEventCreateWithFlags(&c->ev_chunk[k], hipEventDisableTiming) != hipSuccess) return CRTHIP_E_HIP;
    if (hipEventCreateWithFlags(&c->ev_mfork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_mjoin, hipEventDisableTiming) != hipSuccess) return CRTHIP_E_HIP;
    return CRTHIP_OK;
}

/* launch entry points of the other translation units (enqueue on c->stream; no synchronisation) */
int crt_run_encoder(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                    signed char *dst, crthip_state *d_state, bool fused, int nes_setup, bool with_state, const sig_layout *lay = nullptr);
bool crt_fused_layout(const crthip_ctx *c, const crthip_params *p, int n, sig_layout *lay);
int crt_run_unpad(crthip_ctx *c, int n, const sig_layout *lay, const signed char *d_src, signed char *d_dst);
int crt_run_encoder_state(crthip_ctx *c, const crthip_params *p, int n, crthip_state *d_state);
int crt_run_encoder_prepare(crthip_ctx *c, const crthip_params *p, bool fused);
int crt_run_noise(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_analog, signed char *d_inp,
                  crthip_state *d_state, bool advance_rn);
int crt_run_advance_rn(crthip_ctx *c, int n, crthip_state *d_state);
int crt_run_vhs_chain(crthip_ctx *c, int n, crthip_state *d_state, int draw_aberration);
int crt_run_clean_vsync(crthip_ctx *c, int n, const signed char *d_analog, crthip_state *d_state);
int crt_run_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
                 crthip_line *d_lines, int advance_rn, int preset_ccf = 0, const sig_layout *lay = nullptr);
int crt_run_decode(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                   const crthip_line *d_lines, void *d_out, size_t ostride, size_t fstride = 0);   /* fstride 0: the flat layout's */
bool crt_decode_wide_ok(const crthip_ctx *c, const crthip_params *p, int min_tier, bool wide);
int crt_run_decode_wide(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, const crthip_line *d_lines,
                        void *d_out, size_t ostride, int min_tier, int rank, size_t fstride);
int crt_reserve_bloom(crthip_ctx *c, int n);
int crt_run_decode_bloom_lanes(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                               const crthip_line *d_lines, void *d_out, size_t ostride, int min_tier, size_t fstride);
int crt_run_decode_rows(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                        const crthip_line *d_lines, void *d_out, size_t ostride, size_t fstride);

#endif /* CRT_DEV_H */
