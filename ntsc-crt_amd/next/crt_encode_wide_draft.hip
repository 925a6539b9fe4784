/* crt_encode_wide_draft.hip -- DRAFT for the round after round 4.  NOT part of the product: not built by the Makefile, not loaded by
 * anything, never run on a GPU.  It compiles (tools: see the command at the end) and that is all that is claimed for it.
 *
 * What it is: M5 (active video, crt_ntsc.c:254-324) for WIDE 4-byte images in the work split of the wide-run decoder
 * (crt_decode4.hip): 16 destination rows per wavefront, a row's three one-pole low-passes (iirf, crt_ntsc.c:117-126: Y, I, Q --
 * three independent chains) on three lanes of a quad.  Why (lead 5; profiles/r04_experiments.txt sections
 * 18-19): at 1920x1080 k_active is bound by the memory system, not by anything inside a CU -- the same 0.95 ms at 11, 8 and 5
 * waves per CU; without its signal stores 0.60 ms, without its image loads 0.55 ms -- and a wave that owns 64 rows can only
 * afford 64-byte signal pieces and 128-byte image pieces per row.  A wave that owns 16 rows holds their WHOLE signal lines in LDS
 * (16 x 753 bytes) and stores them as 14.5 KB of nearly contiguous memory (the lines of a field follow each other in inp[]), and
 * fetches 512-byte image pieces.
 *
 * Arithmetic: the FAST + 64-bit-mad path of k_active, instruction for instruction per component -- lane c of a quad converts ONLY
 * its own component (3 multiplies instead of 9), runs its own low-pass, multiplies by its own carrier (the luma lane by 2^16:
 * "itself"), the quad sum is the sample.  The split itself is checked on the CPU against the oracle's crt_modulate, sample for
 * sample (next/check_split_arithmetic.py: NTSC / VHS, four pixel formats, both fields: 0 mismatches); the parity tests that cover
 * k_active cover the kernel once it is wired in.
 *
 * Scope of the draft: 4-samples-per-cycle systems with band-limiting (SysNTSC, SysNTSC0, SysVHS ...), 4-byte pixels, image rows
 * that are a multiple of 512 bytes (1280, 1920, 2560, 3840 pixels), destw <= 768.  Everything else stays with k_active.
 *
 * Open points for whoever builds it: (1) the host dispatch in crt_run_encoder_active (conditions above + FAST envelope);
 * (2) a CLAMP / non-noise variant check against k_active's tail (k_active clamps to 0..110 and, with noise, to +-127: same here);
 * (3) LDS bank behaviour of the byte stores (row stride 784 bytes: rows l and l + 8 share a bank -- measure SQ_LDS_BANK_CONFLICT
 *     first, crt_decode4.hip lost 30 % to exactly this); (4) whether the idle fourth lane of the quad should run the LCG.
 */
#ifdef EW_COMPILE_CHECK
#include "../csrc/crt_encode.hip"      /* for the check only: source_row, format_alpha_first, format_blue_low live there */
#else
#include "crt_dev.h"
#endif

#define EW_ROWS    16                  /* destination rows per wave */
#define EW_TILE    128                 /* image pixels per row and tile: 512 bytes */
#define EW_PSTRIDE 132                 /* dwords per tile row in LDS (16-byte aligned rows) */
#define EW_SSTRIDE 784                 /* bytes per signal line in LDS (16-byte aligned, >= destw) */

/* an 8-bit LDS store by the lanes of `lanes` only (see lds_store16_lanes in crt_decode4.hip) */
__device__ __forceinline__ void ew_store8_lanes(unsigned addr, int v, unsigned long long lanes)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tds_write_b8 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(addr), "v"(v), "s"(lanes) : "memory");
}
/* v_mad_i64_i32 with a per-lane multiplier (crt_decode4.hip) */
__device__ __forceinline__ long ew_mad64_vv(int d, int m, long acc)
{
    long r, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(d), "v"(m), "v"(acc));
    return r;
}

template <class S, bool NOISE>
__global__ void __launch_bounds__(64)
k_active_wide(const crthip_params P, int n_fields, const unsigned char *__restrict__ images, size_t istride,
              signed char *__restrict__ dst, size_t fstride, const crthip_state *__restrict__ state,
              const uint2 *__restrict__ jump16)
{
    static_assert(S::CCS == 4 && S::BANDLIMIT && !S::IS_NES, "4-samples-per-cycle systems with band-limiting");
    __shared__ __attribute__((aligned(16))) unsigned s_pix[EW_ROWS * EW_PSTRIDE];
    __shared__ __attribute__((aligned(16))) unsigned char s_sig[EW_ROWS * EW_SSTRIDE];
    __shared__ unsigned long long s_src[EW_ROWS], s_dst[EW_ROWS];

    const int lane = threadIdx.x, l = lane >> 2, c = lane & 3;      /* my row of the wave, my component: 0 Y, 1 I, 2 Q, 3 idle */
    const int rows = P.desth;
    const int gid = blockIdx.x * EW_ROWS + l;
    const bool live = gid < n_fields * rows;
    const int f = live ? gid / rows : 0;
    const int y = live ? gid - f * rows : 0;
    const crthip_state st = state[f];
    const int start = (y + P.yo) * S::HRES + P.xo;
    unsigned rn = 0;
    if (NOISE) rn = lcg_at(jump16, (unsigned) st.rn, start);

    const int w = P.w, destw = P.destw;
    const int row_bytes = w * 4;                                     /* host: a multiple of EW_TILE * 4 */
    const int last_tile = row_bytes / (EW_TILE * 4) - 1;
    {
        const int sy = source_row<S>(P, y, st.field & 1);
        if (c == 0) {
            s_src[l] = (unsigned long long) (images + (size_t) f * istride + (size_t) sy * row_bytes);
            s_dst[l] = live ? (unsigned long long) (dst + (size_t) f * fstride + start) : 0ull;
        }
    }
    __syncthreads();

    /* my component (crt_ntsc.c:307-309 by byte position, as in k_active) and my low-pass */
    const bool alpha_first = format_alpha_first(P.format), blue_low = format_blue_low(P.format);
    const int k0 = c == 0 ? (blue_low ? 7471 : 19595) : c == 1 ? (blue_low ? -21103 : 39059) : c == 2 ? (blue_low ? 20382 : 13894) : 0;
    const int k1 = c == 0 ? 38470 : c == 1 ? -18022 : c == 2 ? -34275 : 0;
    const int k2 = c == 0 ? (blue_low ? 19595 : 7471) : c == 1 ? (blue_low ? 39059 : -21103) : c == 2 ? (blue_low ? 13894 : 20382) : 0;
    const bool near1 = c == 0 && S::IIR_Y_NEAR;                       /* h' = s + ((c - 2048) * (s - h) >> 11) */
    const int M = c == 0 ? (S::IIR_Y_NEAR ? P.iir_c[0] - 2048 : P.iir_c[0]) * (1 << 21) : c == 1 ? P.iir_c[1] * (1 << 21) : c == 2 ? P.iir_c[2] * (1 << 21) : 0;
    /* my carrier by sample phase, pre-scaled by 2^12 like k_active's; the luma lane multiplies by 2^16, i.e. passes its value */
    const int crow = carrier_row<S>(y + P.yo, st.field, st.frame, st.aux);
    int wk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) wk[k] = c == 0 ? 65536 : c == 1 ? P.modI[crow][k] * 4096 : c == 2 ? P.modQ[crow][k] * 4096 : 0;
    const int white = P.white, noise256 = P.noise * 256;
    int ire_base_1024 = P.ire_base * 1024, neg_noise127_256 = -0x7f * P.noise * 256;
    asm volatile("" : "+v"(ire_base_1024));
    asm volatile("" : "+v"(neg_noise127_256));
    v2u lcg_add = { LCG_ADD, 0u };
    asm volatile("" : "+v"(lcg_add));

    /* image tiles: a load instruction moves 2 rows x 512 bytes (32 lanes x 16 bytes per row), 8 instructions per tile */
    const int prow = lane >> 5, piece = lane & 31;
    v4i stage[EW_ROWS / 2];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int i = 0; i < EW_ROWS / 2; i++)
            stage[i] = gload16u_nt(s_src[2 * i + prow] + (unsigned long long) (tile * (EW_TILE * 4) + piece * 16));
    };
    auto stash = [&]() {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EW_ROWS / 2; i++) {
            v4i v = stage[i];
            if (alpha_first) { v.x = (int) ((unsigned) v.x >> 8); v.y = (int) ((unsigned) v.y >> 8); v.z = (int) ((unsigned) v.z >> 8); v.w = (int) ((unsigned) v.w >> 8); }
            *(v4i *) (s_pix + (2 * i + prow) * EW_PSTRIDE + piece * 4) = v;
        }
        __syncthreads();
    };
    int have = 0;
    fetch(0);
    stash();
    if (last_tile > 0) fetch(1);

    long hp = 0;                                                    /* my low-pass state in the high half (k_active: hyp / hip / hqp) */
    const unsigned long long cstep = ((unsigned long long) P.col_step_hi << 32) | P.col_step_lo;
    unsigned long long cpos = 0;
    const int ngroups = (destw + 3) >> 2;
    for (int g = 0; g < ngroups; g++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = 4 * g + k;
            const int col = (int) (cpos >> 32);
            const int need = col >> 7;                              /* wave-uniform */
            if (need != have && need <= last_tile) {
                if (need != have + 1) fetch(need);                  /* source more than EW_TILE x wider than the line */
                stash();
                have = need;
                if (need < last_tile) fetch(need + 1);
            }
            const unsigned pixel = s_pix[l * EW_PSTRIDE + (col & (EW_TILE - 1))];
            const int c0 = pixel & 255, c1 = (pixel >> 8) & 255, c2 = (pixel >> 16) & 255;
            const int comp = (k0 * c0 + k1 * c1 + k2 * c2) >> 14;   /* crt_ntsc.c:307-309, my row of the matrix */
            /* iirf: h' = h + (c * (s - h) >> 11) as the high half of (c << 21) * (s - h) + { 0, near1 ? s : h } */
            const int h = pair_hi(hp);
            hp = ew_mad64_vv(comp - h, M, (long) ((unsigned long) (unsigned) (near1 ? comp : h) << 32));
            const int o = pair_hi(hp);
            /* (o * carrier) >> 16 per component, summed over the quad: oy + ((oi * ccI) >> 4 >> 12) + ((oq * ccQ) >> 4 >> 12) */
            int t = __mul24(o, wk[k]) >> 16;
            t += __builtin_amdgcn_mov_dpp(t, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
            t += __builtin_amdgcn_mov_dpp(t, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
            int ire = mad24_vv(t, white, ire_base_1024) >> 10;      /* crt_ntsc.c:316 */
            ire = clampi(ire, 0, 110);
            if (NOISE) {
                rn = lcg_step_mad64(rn, lcg_add);
                const int nb = (int) ((rn >> 16) & 0xffu);
                ire = add_hiword(ire, mad24_vv(nb, noise256, neg_noise127_256));
                ire = clampi(ire, -127, 127);
            }
            if (x < destw) ew_store8_lanes((unsigned) (l * EW_SSTRIDE + x), ire, 0x1111111111111111ull);
            cpos += cstep;
        }
    }
    /* the 16 signal lines leave as whole lines: 16-byte pieces, a row's pieces next to each other */
    __syncthreads();
    const int ppr = (destw + 15) >> 4;                              /* pieces per row */
    for (int q = lane; q < EW_ROWS * ppr; q += 64) {
        const int r = q / ppr, p = q - r * ppr;
        const unsigned long long d = s_dst[r];
        if (d == 0) continue;
        const v4i o = *(const v4i *) (s_sig + r * EW_SSTRIDE + p * 16);
        const int nbytes = destw - p * 16 < 16 ? destw - p * 16 : 16;
        if (nbytes == 16) {
            gstore16u(d + p * 16, o);
        } else {
            const int wds[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
            for (int b = 0; b < 16; b++)
                if (b < nbytes) gstore8(d + p * 16 + b, (unsigned) (wds[b >> 2] >> (8 * (b & 3))));
        }
    }
}

#ifdef EW_COMPILE_CHECK
/* cd ntsc-crt_amd/next && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -I../../include -I../csrc -DEW_COMPILE_CHECK \
 *     --cuda-device-only -S -o /tmp/ew.s crt_encode_wide_draft.hip */
template __global__ void k_active_wide<SysNTSC, true>(const crthip_params, int, const unsigned char *, size_t, signed char *, size_t,
                                                      const crthip_state *, const uint2 *);
template __global__ void k_active_wide<SysNTSC, false>(const crthip_params, int, const unsigned char *, size_t, signed char *, size_t,
                                                       const crthip_state *, const uint2 *);
#endif
