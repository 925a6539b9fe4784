/* crt_noise.hip -- D1, the noisy channel: LCG noise and the VHS build's libc rand() stream.  See crt_dev.h. */
#include "crt_dev.h"

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
    /* LCG step and noise product through the full-rate 64-bit multiply-add (same wrapped 32-bit arithmetic as lcg_step() /
     * noisy(); v_mul_lo_u32 runs at a quarter of its rate) */
    v2u lcg_add = { LCG_ADD, 0u };
    asm volatile("" : "+v"(lcg_add));
    const int noise = P.noise;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = (wds[d] << (24 - 8 * k)) >> 24;
            rn = lcg_step_mad64(rn, lcg_add);
            v[k] = clampi(s + (mul_lo_mad64((int) ((rn >> 16) & 0xffu) - 0x7f, noise) >> 8), -127, 127);
        }
        outw[d] = (int) pack4(v[0], v[1], v[2], v[3]);
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

/* ------------------------------------------------------------------------- */
/* the 31 x 31 jump of the rand() model on the matrix cores                     */
/* ------------------------------------------------------------------------- */
/*
 * Every lane of a wave needs the generator's 31-value history at ITS call position:  w_lane[j] = sum_m c_lane[m] * z[m + j]
 * (mod 2^32), c_lane = x^K_lane mod (x^31 - x^28 - 1) from the host's table, z = the field's base sequence (history + the
 * next 30 values).  Over the wave that is the matrix product  W^T (31 x 64) = H^T (31 x 31, Hankel: H[m][j] = z[m + j]) x
 * C^T (31 x 64) -- 961 multiply-adds per lane on the vector unit, with v_mul_lo_u32 at a quarter of the full rate: half of
 * everything the VHS noise kernels did.  Here: 32-bit words as four SIGNED byte digits
 * (x == d0 + d1 2^8 + d2 2^16 + d3 2^24 mod 2^32, |d| <= 128), the ten digit pairs (p, q), p + q <= 3, as
 * v_mfma_i32_32x32x32_i8 (exact: |sum| <= 4 * 31 * 128 * 128 < 2^21 per accumulator), pairs of equal weight accumulated in
 * one accumulator, then W = acc0 + (acc1 << 8) + (acc2 << 16) + (acc3 << 24) with 32-bit wrap-around -- pairs of weight
 * 2^32 and beyond vanish.  Two 32-lane tiles per wave.  Operand layouts: A = H^T rows j = lane & 31, B = C^T columns =
 * chunk lane & 31; both take k = m in the SAME slot order (16 * (lane >> 5) + byte), so the order itself does not matter;
 * D: column = lane & 31, rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cdna_hip_programming.md, dtype independent).
 */
typedef int v16i __attribute__((ext_vector_type(16)));
#define VHS_ZD_STRIDE 72                   /* bytes per digit plane of the base sequence in LDS: 61 values, zero padding */

/* element idx of the base sequence z[0..60] of the history h[0..30] (z[n] = z[n-31] + z[n-3]), without the recurrence:
 * z[31 + t] = h[28 + t % 3] + h[t] + h[t - 3] + h[t - 6] + ...   (idx >= 61: 0) */
__device__ __forceinline__ unsigned vhs_base_value(const unsigned *h, int idx)
{
    if (idx < 31) return h[idx];
    if (idx >= 61) return 0u;
    const int t = idx - 31;
    unsigned v = h[28 + t % 3];
#pragma unroll
    for (int i = 0; i <= 10; i++) {
        const int k = t - 3 * i;
        if (k >= 0) v += h[k];
    }
    return v;
}

/* lane idx publishes the four digits of z[idx] (s_zd: 4 planes of VHS_ZD_STRIDE bytes); the caller fences */
__device__ __forceinline__ void vhs_publish_digits(unsigned char *s_zd, int lane, unsigned zval)
{
    unsigned x = zval;
#pragma unroll
    for (int pl = 0; pl < 4; pl++) {
        const int d = (int) (signed char) (x & 255u);
        s_zd[pl * VHS_ZD_STRIDE + lane] = (unsigned char) d;
        if (lane < VHS_ZD_STRIDE - 64) s_zd[pl * VHS_ZD_STRIDE + 64 + lane] = 0;
        x = (x - (unsigned) d) >> 8;
    }
}

/* the coefficient fragments of the chunk rows row_t0 (tile 0: the row of lane & 31) and row_t1 (tile 1) */
__device__ __forceinline__ void vhs_load_bfrag(const signed char *__restrict__ dig, int row_t0, int row_t1, int lane, v4i bfr[2][4])
{
    const int half = lane >> 5;
#pragma unroll
    for (int pl = 0; pl < 4; pl++) {
        bfr[0][pl] = *(const v4i *) (dig + (size_t) row_t0 * VHS_DIG_ROW + pl * 32 + half * 16);
        bfr[1][pl] = *(const v4i *) (dig + (size_t) row_t1 * VHS_DIG_ROW + pl * 32 + half * 16);
    }
}

/* w[j] = sum_m c_lane[m] * z[m + j] mod 2^32 for the lanes of the wave with `mine` set (the others keep their w); s_zd = the
 * digit planes of z (published and fenced), bfr = the lanes' coefficient fragments (vhs_load_bfrag) */
__device__ __forceinline__ void vhs_jump_mfma(const unsigned char *s_zd, const v4i bfr[2][4], int lane, unsigned w[31], bool mine = true)
{
    /* A fragments: 16 consecutive digits of z from index (lane & 31) + 16 * (lane >> 5) */
    v4i afr[4];
    const int o = (lane & 31) + 16 * (lane >> 5);
    const unsigned sh = (unsigned) (o & 3);
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        const unsigned *wp = (const unsigned *) (s_zd + qd * VHS_ZD_STRIDE) + (o >> 2);
        const unsigned d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
        afr[qd].x = (int) __builtin_amdgcn_alignbyte(d1, d0, sh);
        afr[qd].y = (int) __builtin_amdgcn_alignbyte(d2, d1, sh);
        afr[qd].z = (int) __builtin_amdgcn_alignbyte(d3, d2, sh);
        afr[qd].w = (int) __builtin_amdgcn_alignbyte(d4, d3, sh);
    }
    const bool lo = lane < 32;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        unsigned W[16];
#pragma unroll
        for (int sdeg = 0; sdeg < 4; sdeg++) {                 /* digit pairs of weight 2^(8 * sdeg): coefficient digit p, sequence digit sdeg - p */
            v16i acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int pl = 0; pl <= sdeg; pl++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[sdeg - pl], bfr[t][pl], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) W[r] = sdeg == 0 ? (unsigned) acc[r] : W[r] + ((unsigned) acc[r] << (8 * sdeg));
            /* (one accumulator at a time: four of them side by side cost the kernel half its occupancy) */
#pragma unroll
            for (int r = 0; r < 16; r++) asm volatile("" : "+v"(W[r]));
        }
        /* Tile t holds the results of chunks 32 t .. 32 t + 31: column lane & 31, rows j0 = (r & 3) + 8 (r >> 2) in lanes 0..31 and
         * j0 + 4 in lanes 32..63.  The half-wave that owns those chunks (lanes 0..31 for tile 0, 32..63 for tile 1) keeps its
         * own rows and receives the others from the lane 32 away. */
        const bool own = lo == (t == 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const unsigned recv = (unsigned) __shfl_xor((int) W[r], 32);
            const int j0 = (r & 3) + 8 * (r >> 2);
            const unsigned v0 = lo ? W[r] : recv, v4 = lo ? recv : W[r];     /* rows j0 / j0 + 4 of the owner's chunk */
            if (own && mine) w[j0] = v0;
            if (j0 + 4 < 31 && own && mine) w[j0 + 4] = v4;
        }
    }
}

/* Parallel region.  A wave's 64 chunks (VHS_CHUNK = 124 samples each) are (mostly) one contiguous 7936-byte run of the field: it
 * is moved through an LDS tile of 64 x 31 dwords with coalesced requests (lane-per-chunk byte accesses cost 12x the algorithmic
 * HBM write traffic); a chunk's 31 dwords are an odd stride already: the lane-per-chunk accesses are conflict-free. */
template <class S, bool MFMA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_vhs_noise(const crthip_params P, int n_fields, const signed char *__restrict__ analog,
            signed char *__restrict__ inp, size_t fstride,
            const unsigned *__restrict__ hist, const unsigned *__restrict__ rows, int chunks_a, const signed char *__restrict__ dig)
{
    constexpr int DW = VHS_CHUNK / 4;                            /* dwords per chunk */
    constexpr int DWS = DW | 1;                                  /* odd LDS stride: conflict-free lane-per-chunk access */
    static_assert((2 * VHS_CHUNK) % 31 == 0 && VHS_CHUNK % 4 == 0, "static ring index / dword packing");
    __shared__ __attribute__((aligned(16))) unsigned s_t[64 * DWS];
    __shared__ unsigned long long s_off[64];
    const int lane = threadIdx.x;
    const int gid = blockIdx.x * 64 + lane;
    const bool live = gid < n_fields * chunks_a;
    const int f = live ? gid / chunks_a : 0;
    const int q = live ? gid - f * chunks_a : 0;
    s_off[lane] = live ? (unsigned long long) f * fstride + (unsigned long long) q * VHS_CHUNK : ~0ull;
    __syncthreads();
    /* what the jump needs -- the field's history and the chunk's coefficient row -- is fetched FIRST and waited for (the
     * memory counter is in-order: a wait for these behind the tile requests would wait for the tile too) */
    const unsigned *h = hist + (size_t) f * 32;
    const unsigned *c = rows + (size_t) q * 31;
    unsigned z[MFMA ? 1 : 61], cm[MFMA ? 1 : 31];
    __shared__ unsigned s_zd_[MFMA ? 4 * VHS_ZD_STRIDE / 4 : 1];
    unsigned char *const s_zd = (unsigned char *) s_zd_;
    v4i bfr[2][4];
    /* the fields this wave's chunks belong to: one, or two where the wave straddles a field boundary (1 wave in 27) */
    const int total = n_fields * chunks_a;
    const int gid_first = blockIdx.x * 64, gid_last = gid_first + 63 < total ? gid_first + 63 : total - 1;
    const int f_first = gid_first / chunks_a, f_last = gid_last / chunks_a;
    unsigned zmine[2] = { 0u, 0u };
    if (MFMA) {
        /* coefficient fragments: tile t column lane & 31 is the chunk of lane 32 t + (lane & 31) */
        int row_t[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int g = gid_first + 32 * t + (lane & 31);
            row_t[t] = g < total ? g - (g / chunks_a) * chunks_a : 0;
        }
        vhs_load_bfrag(dig, row_t[0], row_t[1], lane, bfr);
        /* element `lane` of the base sequence of the wave's field(s) */
        zmine[0] = vhs_base_value(hist + (size_t) f_first * 32, lane);
        if (f_last != f_first) zmine[1] = vhs_base_value(hist + (size_t) f_last * 32, lane);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int pl = 0; pl < 4; pl++) asm volatile("" : "+v"(bfr[t][pl]));
        asm volatile("" : "+v"(zmine[0]), "+v"(zmine[1]));
    } else {
#pragma unroll
        for (int j = 0; j < 31; j++) z[j] = h[j];
#pragma unroll
        for (int m = 0; m < 31; m++) cm[m] = c[m];
#pragma unroll
        for (int j = 0; j < 31; j++) asm volatile("" : "+v"(z[j]), "+v"(cm[j]));
    }
    /* the tile straight into LDS (global_load_lds_dword: dword n of the tile belongs to lane n % 64 of request n / 64, which
     * is exactly where the hardware puts it), all 31 requests in flight at once and none of them waited for before the
     * jump below is done.  Through registers, four loads at a time, a wave spent 54 % of its life in s_waitcnt (SQ_WAIT_ANY,
     * eight memory round trips per wave): 0.98 -> 0.845 ms per 2048 fields; 31 staging registers instead sent the
     * allocator from 102 to 175 VGPRs (0.93 ms).  Lanes without a chunk request nothing: their tile rows are never stored.
     * (Several tiles per jump -- 37 % fewer vector instructions, but a loop the register allocator answers with 162
     * VGPRs -- measured 0.824 ms, the field-pass the same: not kept.) */
    static_assert(DWS == DW, "tile rows are packed: LDS dword index == request * 64 + lane");
    /* (r5) A wave whose 64 chunks are ONE contiguous run of one field (26 waves in 27) moves its tile in 16-byte pieces -- 8 requests
     * of 1 KB (global_load_lds_dwordx4, gfx950: lane l's 16 bytes land at LDS base + 16 l, tools/probe_lds128.hip) instead of 31
     * of 256 bytes, and 8 stores instead of 31 on the way out; the waves that straddle a field boundary or the batch's end keep
     * the dword path */
    static_assert((64 * DW) % 4 == 0, "the tile is a whole number of 16-byte pieces");
    constexpr int P16 = (64 * DW / 4 + 63) / 64;                 /* 16-byte requests per tile */
    const bool one_run = f_first == f_last && gid_first + 63 < total;
    const unsigned long long run_base = (unsigned long long) f_first * fstride + (unsigned long long) (gid_first - f_first * chunks_a) * VHS_CHUNK;
    if (one_run) {
#pragma unroll
        for (int it = 0; it < P16; it++) {
            const int n = (it * 64 + lane) * 4;                  /* first dword of my piece */
            if (n < 64 * DW)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (analog + run_base + 4 * (size_t) n),
                                                 (void __attribute__((address_space(3))) *) (s_t + it * 256), 16, 0, 0);
        }
    } else {
#pragma unroll
    for (int it = 0; it < DW; it++) {
        const int n = it * 64 + lane, owner = n / DW, d = n - owner * DW;
        const unsigned long long off = s_off[owner];
        if (off != ~0ull)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (analog + off + 4 * d),
                                             (void __attribute__((address_space(3))) *) (s_t + it * 64), 4, 0, 0);
    }
    }
    /* history of call K = 1 + 2 * VHS_CHUNK * q */
    unsigned w[31];
#pragma unroll
    for (int j = 0; j < 31; j++) w[j] = 0;
    if (MFMA) {
        for (int pass = 0; pass < (f_last != f_first ? 2 : 1); pass++) {
            const int fp = pass ? f_last : f_first;
            wave_lds_fence();                                  /* (not __syncthreads(): that would wait for the tile requests too) */
            vhs_publish_digits(s_zd, lane, pass ? zmine[1] : zmine[0]);
            wave_lds_fence();
            vhs_jump_mfma(s_zd, bfr, lane, w, f == fp);
        }
    } else {
        /* base sequence z[0..60]: the history and the next 30 values */
#pragma unroll
        for (int j = 31; j < 61; j++) z[j] = z[j - 31] + z[j - 3];
#pragma unroll
        for (int m = 0; m < 31; m++) {
#pragma unroll
            for (int j = 0; j < 31; j++) w[j] += cm[m] * z[m + j];
        }
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
            /* (the wrapped 32-bit product through the full-rate 64-bit multiply-add: v_mul_lo_u32 runs at a quarter of it) */
            const int sv = clampi(a + (mul_lo_mad64(((rn >> 16) & 0xff) - 0x7f, noise) >> 8), -127, 127);
            out4 = (k & 3) == 0 ? (unsigned) (sv & 255) : out4 | (unsigned) (sv & 255) << (8 * (k & 3));
            if ((k & 3) == 3) mine[k >> 2] = out4;
        }
    }
    __syncthreads();
    if (one_run) {
#pragma unroll
        for (int it = 0; it < P16; it++) {
            const int n = (it * 64 + lane) * 4;
            if (n < 64 * DW) {
                const v4i o = *(const v4i *) (s_t + n);
                gstore16u((unsigned long long) (inp + run_base + 4 * (size_t) n), o);
            }
        }
        return;
    }
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
/* (three waves per SIMD: 168 registers and a few spilled dwords instead of 242.  The kernel runs BESIDE k_vhs_noise, and two of its
 * waves per SIMD at 242 registers left that kernel none: the two ran one after the other -- 0.41 + 0.46 ms -- whatever the
 * streams said; at 168 one wave of k_vhs_noise fits next to them: 0.845 -> 0.785 ms for the pair, profiles/r04_experiments.txt) */
#define VHS_COS_TAB 24                                         /* >= (HRES * 17) / HRES + 2 distinct values of (i * line) / HRES per segment */
#define VHS_DIV_MAGIC(h) ((unsigned long long) ((((1ull << 40) + (h) - 1) / (h))))
template <class S, bool MFMA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_vhs_tail(const crthip_params P, int n_fields, const signed char *__restrict__ analog,
           signed char *__restrict__ inp, size_t fstride, crthip_state *__restrict__ state,
           const unsigned *hist, unsigned *hist_out, const unsigned *__restrict__ tail_row, const unsigned *__restrict__ blk_rows,
           const signed char *__restrict__ dig_blocks)
{
    constexpr int N = S::INPUT_SIZE, H = S::HRES, B = VHS_BLK;
    constexpr int T0 = vhs_tail_start(N, H);
    constexpr int NB = (H + 63) / 64;                              /* bytes per lane and segment */
    static_assert(64 * B >= 3 * H + 3, "window too small for a segment");
    __shared__ unsigned s_y[64 * B + 8];                           /* the window's raw generator values */
    __shared__ unsigned s_h[64];                                   /* 31-value history in front of the window; base sequence */
    __shared__ unsigned s_misc[2];
    __shared__ signed char s_a[NB * 64], s_o[NB * 64];
    __shared__ int s_cos[VHS_COS_TAB];                             /* the band's noise amplitudes of the current segment, by (i * line) / HRES */
    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= n_fields) return;
    /* one wave per field and a long dependent chain: when it shares a SIMD with the parallel region's waves (see
     * crt_run_noise) it should win the instruction arbitration, or its latency is multiplied by the occupancy */
    __builtin_amdgcn_s_setprio(3);
    const unsigned *h = hist + (size_t) f * 32;
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
    unsigned cb[MFMA ? 1 : 31];                                    /* x^(43*lane) */
    __shared__ unsigned s_zd_[MFMA ? 4 * VHS_ZD_STRIDE / 4 : 1];
    unsigned char *const s_zd = (unsigned char *) s_zd_;
    v4i bfr[2][4];                                                 /* ... as digit fragments for the matrix cores (rows = blocks) */
    if (MFMA) {
        vhs_load_bfrag(dig_blocks, lane & 31, 32 + (lane & 31), lane, bfr);
    } else {
#pragma unroll
        for (int m = 0; m < 31; m++) cb[m] = blk_rows[m * 64 + lane];
    }
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
        unsigned w[31];
        if (MFMA) {
            vhs_publish_digits(s_zd, lane, vhs_base_value(s_h, lane));
            __syncthreads();
            vhs_jump_mfma(s_zd, bfr, lane, w);
        } else {
            unsigned z[61];
#pragma unroll
            for (int j = 0; j < 31; j++) z[j] = s_h[j];
#pragma unroll
            for (int j = 31; j < 61; j++) z[j] = z[j - 31] + z[j - 3];
#pragma unroll
            for (int j = 0; j < 31; j++) w[j] = 0;
#pragma unroll
            for (int m = 0; m < 31; m++) {
#pragma unroll
                for (int j = 0; j < 31; j++) w[j] += cb[m] * z[m + j];
            }
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
        /* (r5) The amplitude inside the aberration band, nn = cos14(((i * line) / HRES) * 8192 / 180) >> 8 (crt_core.c:353-355),
         * used to be evaluated per sample inside the walk below -- two divisions and the sine polynomial behind a branch that some
         * lane of the wave nearly always takes: two thirds of the walk's instructions, on the one chain that bounds the VHS noise
         * pair (25 segments x 19 us per field).  (i * line) / HRES takes at most 18 consecutive values
         * over a segment of <= HRES samples (line <= 17): one lane per value evaluates the cosine ONCE per segment into a table,
         * the walk divides by HRES with an exact multiply-shift and looks it up.  The per-sample `r2 % 20 > kseg - 6` is the flag
         * bit the speculative walks already use (G), not a second modulo. */
        const int ln_first = (seg_start * vhs_line) / H;
        if (lane < VHS_COS_TAB) s_cos[lane] = dev_cos14((ln_first + lane) * 8192 / 180) >> 8;
        __syncthreads();
        {
            int pos = e_in;
            for (int k = 0; k < (B + 1) / 2; k++) {
                const int s = first + k;
                const bool valid = pos < B && s < n_s;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) break;
                if (valid) {
                    const unsigned *yp = s_y + lane * B + pos;
                    const unsigned rnv = yp[0] >> 1;
                    const int i = seg_start + s;
                    const int c1 = (int) ((G >> (pos + 1)) & 1ull);                       /* 6 + r2 % 20 > kseg, crt_core.c:350 */
                    const unsigned r3 = yp[2] >> 1;                                       /* (read whether or not it is a call: in bounds) */
                    /* (i * line) / HRES: i * line < 2^23, HRES < 2^11: exact as (x * ceil(2^40 / HRES)) >> 40 */
                    const unsigned x = (unsigned) (i * vhs_line);
                    const int ln = (int) (((unsigned long long) x * VHS_DIV_MAGIC(H)) >> 40);
                    const bool band = c1 && i < N - H * (5 + ((int) (r3 & 7u) - 4));      /* crt_core.c:351 */
                    const int nn = band ? s_cos[ln - ln_first] : noise;
                    const int sv = (int) s_a[s] + (mul_lo_mad64((int) ((rnv >> 16) & 0xffu) - 0x7f, nn) >> 8);
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
    /* the generator's state after the field (word 31 of a slot is padding: carried over, so that hist_out can be a
     * side buffer that is copied back whole) */
    if (lane < 32) hist_out[(size_t) f * 32 + lane] = lane < 31 ? s_h[lane] : h[31];
    if (lane == 0) {
        state[f].rn = (int) s_misc[1];                             /* crt_core.c:367 */
        signed char *tail = dst + N;
        store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
    }
}

/*
 * Sequence mode of the VHS build (crthip_sequence): the fields of ONE video share ONE rand() stream, field k+1
 * starts where field k stopped, and where it stops depends on the values drawn in its last 25 lines.  Those
 * decisions do not depend on the picture, only on the stream, so the chain can be run ahead of everything else:
 * one wave walks the fields in order with the same speculative block walk as k_vhs_tail minus the samples, and
 * leaves in hist[k] the generator history at the start of field k's crt_demodulate (and, if asked to, draws the
 * aberration height of crt_modulate, crt_ntscvhs.c:205-207, from the stream: one call per field, before it).
 * Afterwards every field is independent again.  Serial by nature: ~0.15 ms per field.
 */
template <class S>
__global__ void __launch_bounds__(64)
k_vhs_chain(int n_fields, crthip_state *__restrict__ state, unsigned *__restrict__ hist,
            const unsigned *__restrict__ tail_row, const unsigned *__restrict__ blk_rows, int draw_aberration)
{
    constexpr int N = S::INPUT_SIZE, H = S::HRES, B = VHS_BLK;
    constexpr int T0 = vhs_tail_start(N, H);
    __shared__ unsigned s_y[64 * B + 8];
    __shared__ unsigned s_h[64];
    __shared__ unsigned s_cur[32];                                 /* the generator's history between fields */
    __shared__ unsigned s_misc[2];
    const int lane = threadIdx.x;
    unsigned cb[31];
#pragma unroll
    for (int m = 0; m < 31; m++) cb[m] = blk_rows[m * 64 + lane];
    if (lane < 31) s_cur[lane] = hist[lane];
    __syncthreads();

    for (int f = 0; f < n_fields; f++) {
        if (draw_aberration) {
            /* one call: y = y[n-31] + y[n-3]; the history slides by one */
            const unsigned y = s_cur[0] + s_cur[28];
            const unsigned nxt = lane < 30 ? s_cur[lane + 1] : y;
            __syncthreads();
            if (lane < 31) s_cur[lane] = nxt;
            if (lane == 0) state[f].aux = (int) ((y >> 1) % 12u) - 8 + 14;
            __syncthreads();
        }
        if (lane < 31) hist[(size_t) f * 32 + lane] = s_cur[lane];  /* start of field f's crt_demodulate */
        /* history in front of call 1 + 2*T0 (call #0 and the parallel region make a fixed number of calls) */
        {
            unsigned zf[61];
#pragma unroll
            for (int j = 0; j < 31; j++) zf[j] = s_cur[j];
#pragma unroll
            for (int j = 31; j < 61; j++) zf[j] = zf[j - 31] + zf[j - 3];
            __syncthreads();
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
            __syncthreads();
        }
        int seg_start = T0;
        while (seg_start < N) {
            const int kseg = (N - seg_start) / H;
            int seg_end = N - H * kseg;
            if (seg_end > N - 1) seg_end = N - 1;
            const int n_s = seg_end - seg_start + 1;
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
            unsigned glo = 0, ghi = 0;
#pragma unroll
            for (int t = 0; t < B; t++) {
                const unsigned v = w[t % 31] + w[(t + 28) % 31];
                w[t % 31] = v;
                s_y[lane * B + t] = v;
                const unsigned flag = (6 + (int) ((v >> 1) % 20u) > kseg) ? 1u : 0u;
                if (t < 32) glo |= flag << t; else ghi |= flag << (t - 32);
            }
            {
                const unsigned nb = (unsigned) __shfl_down((int) (glo & 1u), 1);
                if (lane < 63) ghi |= nb << (B - 32);
            }
            const unsigned long long G = ((unsigned long long) ghi << 32) | glo;
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
            /* only the positions: where does the call after the segment's last sample sit? */
            {
                int pos = e_in;
                for (int k = 0; k < (B + 1) / 2; k++) {
                    const int sidx = first + k;
                    const bool valid = pos < B && sidx < n_s;
                    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) break;
                    if (valid) {
                        pos += 2 + (int) ((G >> (pos + 1)) & 1ull);
                        if (sidx == n_s - 1) s_misc[0] = (unsigned) (lane * B + pos);
                    }
                }
            }
            __syncthreads();
            const int pos_end = (int) s_misc[0];
            unsigned hv = 0;
            if (lane < 31) hv = s_y[pos_end - 31 + lane];
            __syncthreads();
            if (lane < 31) s_h[lane] = hv;
            __syncthreads();
            seg_start += n_s;
        }
        if (lane < 31) s_cur[lane] = s_h[lane];
        __syncthreads();
    }
}

/* rn <- rn after INPUT_SIZE steps (crt_core.c:367) */
__global__ void k_advance_rn(int n_fields, crthip_state *state, uint2 whole_field)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_fields) state[f].rn = (int) (whole_field.x * (unsigned) state[f].rn + whole_field.y);
}


int crt_run_advance_rn(crthip_ctx *c, int n, crthip_state *d_state)
{
    hipLaunchKernelGGL(k_advance_rn, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, d_state, c->whole_field);
    return CRTHIP_OK;
}

/* D1 for every field: VHS -> the rand() model (always produces rn); other systems -> LCG noise, and rn advanced
 * by a whole field if advance_rn (the fused path lets k_vsync do that) */
int crt_run_noise(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_analog, signed char *d_inp,
                  crthip_state *d_state, bool advance_rn)
{
    if (c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE)) {
        if (!c->d_vhs_hist) return set_err(c, CRTHIP_E_ARG, "VHS: no generator histories bound (crthip_vhs_bind_history)", hipSuccess);
        return dispatch_system(c->system, c->pattern, [&](auto tag) {
            using S = decltype(tag);
            ProfScope ps(c, CRTHIP_K_NOISE);
            if constexpr (S::IS_VHS) {
                /* The parallel region (throughput bound) and the tail (one wave per field, pure latency) write disjoint
                 * parts of the field and both start from the field's generator history, which the tail also advances:
                 * it leaves the new history in a side buffer, runs on the internal stream BESIDE the parallel region,
                 * and a copy after the join publishes the histories.  (Without the internal stream: one after the
                 * other, in place.) */
                const bool side = c->d_vhs_next && n <= c->cap_fields && crt_ensure_aux(c) == CRTHIP_OK && c->stream != c->aux_stream &&
                                  hipEventRecord(c->ev_fork, c->stream) == hipSuccess &&
                                  hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0) == hipSuccess;
                /* The tail goes first and on the caller's stream, the parallel region on the internal one: behind its event
                 * wait it starts a few microseconds later, by when the tail's lone waves are resident.  The other way
                 * round the tail's workgroups queue behind the parallel region's 60 000 and run when those are done
                 * (measured: 1.06 ms side by side = 0.51 + 0.55 one after the other). */
                hipStream_t ns = side ? c->aux_stream : c->stream;
                const unsigned *tail_row = c->d_vhs_rows + (size_t) c->vhs_chunks * 31, *blk_rows = c->d_vhs_rows + (size_t) (c->vhs_chunks + 1) * 31;
                const signed char *dig_blocks = c->d_vhs_dig + (size_t) c->vhs_chunks * VHS_DIG_ROW;
                const bool mfma = c->vhs_mfma != 0;            /* the 31 x 31 jumps on the matrix cores (CRTHIP_VHS_MFMA=0: on the vector unit) */
#define CRT_LAUNCH_TAIL(HOUT) do { \
                if (mfma) hipLaunchKernelGGL((k_vhs_tail<S, true>), dim3(n), dim3(64), 0, c->stream, *p, n, d_analog, d_inp, c->fstride, d_state, c->d_vhs_hist, HOUT, tail_row, blk_rows, dig_blocks); \
                else hipLaunchKernelGGL((k_vhs_tail<S, false>), dim3(n), dim3(64), 0, c->stream, *p, n, d_analog, d_inp, c->fstride, d_state, c->d_vhs_hist, HOUT, tail_row, blk_rows, dig_blocks); } while (0)
                if (side) CRT_LAUNCH_TAIL(c->d_vhs_next);
                if (mfma)
                    hipLaunchKernelGGL((k_vhs_noise<S, true>), dim3((n * c->vhs_chunks + 63) / 64), dim3(64), 0, ns,
                                       *p, n, d_analog, d_inp, c->fstride, c->d_vhs_hist, c->d_vhs_rows, c->vhs_chunks, c->d_vhs_dig);
                else
                    hipLaunchKernelGGL((k_vhs_noise<S, false>), dim3((n * c->vhs_chunks + 63) / 64), dim3(64), 0, ns,
                                       *p, n, d_analog, d_inp, c->fstride, c->d_vhs_hist, c->d_vhs_rows, c->vhs_chunks, c->d_vhs_dig);
                if (!side) CRT_LAUNCH_TAIL(c->d_vhs_hist);
#undef CRT_LAUNCH_TAIL
                if (side) {
                    if (hipEventRecord(c->ev_join, c->aux_stream) != hipSuccess ||
                        hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess ||
                        hipMemcpyAsync(c->d_vhs_hist, c->d_vhs_next, sizeof(unsigned) * 32 * (size_t) n, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                        return CRTHIP_E_HIP;
                }
            }
            return CRTHIP_OK;
        });
    }
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
        {
            ProfScope ps(c, CRTHIP_K_NOISE);
            hipLaunchKernelGGL((k_noise<S>), dim3((n * CHUNKS + 255) / 256), dim3(256), 0, c->stream,
                               *p, n, d_analog, d_inp, c->fstride, d_state, c->d_jump16);
        }
        if (advance_rn) hipLaunchKernelGGL(k_advance_rn, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, d_state, c->whole_field);
        return CRTHIP_OK;
    });
}

/* VHS sequence mode: hist[0] = the generator before field 0 -> hist[k] for every field (see k_vhs_chain) */
int crt_run_vhs_chain(crthip_ctx *c, int n, crthip_state *d_state, int draw_aberration)
{
    if (!c->d_vhs_hist) return set_err(c, CRTHIP_E_ARG, "VHS: no generator histories bound (crthip_vhs_bind_history)", hipSuccess);
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        if constexpr (S::IS_VHS) {
            hipLaunchKernelGGL((k_vhs_chain<S>), dim3(1), dim3(64), 0, c->stream, n, d_state, c->d_vhs_hist,
                               c->d_vhs_rows + (size_t) c->vhs_chunks * 31, c->d_vhs_rows + (size_t) (c->vhs_chunks + 1) * 31,
                               draw_aberration);
        }
        return CRTHIP_OK;
    });
}
