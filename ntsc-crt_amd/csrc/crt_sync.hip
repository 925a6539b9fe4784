/* crt_sync.hip -- D2-D7: vertical / horizontal sync search, burst lock, per-line carrier table.  See crt_dev.h. */
#include "crt_dev.h"

/* ------------------------------------------------------------------------- */
/* D2-D7: the serial sync chain                                                */
/* ------------------------------------------------------------------------- */
/* The chain is serial in the line index (hsync and the burst integrators carry over, crt_core.c:447,456-467).  One kernel,
 * k_hsync_wave: a wave per field, the vertical search (D2) included; k_vsync is the same search as a kernel of its own for the
 * CRT_DO_VSYNC 0 variant, which looks at the CLEAN signal before the noise stage.  (Rounds 1-4 also carried a 16-lanes-per-field
 * kernel and a speculative variant of the chain that ran beside the encoder: measured slower / neutral, removed in round 5 --
 * profiles/r02_shape_sweep.txt, profiles/r04_spec_sync.txt.) */
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

/* D2 vsync, crt_core.c:379-396: first (line, j) whose running line sum <= VTHR.  Wave-wide: all 2 * VWIN candidate lines
 * are fetched up front, then searched in order with a prefix sum across the wave.  Returns the line found (the last
 * candidate if none) and j (HRES if none), the same in every lane. */
template <class S, int LP = S::HRES>       /* LP: bytes between line starts (HRES: flat; PadGeom::PITCH: the fused path's padded lines) */
__device__ __forceinline__ void vsync_search(const signed char *__restrict__ in, const int vsync, const int lane, int &vline, int &vj)
{
    vline = 0; vj = S::HRES;
    constexpr int PIECES = (S::HRES + 1023) / 1024;      /* 64 lanes x 16 samples per piece; 2 pieces for the PV-1000's 1920 */
    constexpr int GROUP = PIECES > 1 ? S::VWIN : 2 * S::VWIN;   /* candidate lines in flight at once (register budget) */
    bool found = false;
#pragma unroll
    for (int g0 = 0; g0 < 2 * S::VWIN; g0 += GROUP) {
        if (found) break;
        v4i cand[GROUP][PIECES];
#pragma unroll
        for (int i = 0; i < GROUP; i++) {
            const int l = posmod(vsync + g0 + i - S::VWIN, S::VRES);
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) cand[i][pc] = load16u(in + l * LP + pc * 1024 + lane * 16);
        }
        /* all of them in flight together: without this the compiler sinks each load into the conditional block that
         * uses it, one memory round trip per candidate line (13 in a row in the steady state) */
#pragma unroll
        for (int i = 0; i < GROUP; i++) {
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) asm volatile("" : "+v"(cand[i][pc]));
        }
#pragma unroll
        for (int i = 0; i < GROUP; i++) {
            int carry = 0;                                   /* sum of the line's earlier pieces */
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) {
                if (!found) {
                    vline = posmod(vsync + g0 + i - S::VWIN, S::VRES);
                    const int wds[4] = { cand[i][pc].x, cand[i][pc].y, cand[i][pc].z, cand[i][pc].w };
                    const int s0 = pc * 1024 + lane * 16;
                    int pre[16];
                    int run = 0;
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        int s = (wds[k >> 2] << (24 - 8 * (k & 3))) >> 24;
                        if (s0 + k >= S::HRES) s = 0;
                        run += s;
                        pre[k] = run;
                    }
                    const int incl = wave_incl_scan(run);
                    const int excl = carry + incl - run;
                    int first = 16;
#pragma unroll
                    for (int k = 15; k >= 0; k--) {
                        if (s0 + k < S::HRES && excl + pre[k] <= S::VTHR) first = k;
                    }
                    const unsigned long long m = __ballot(first < 16);
                    if (m) {
                        const int L = __ffsll((long long) m) - 1;
                        vj = pc * 1024 + L * 16 + __builtin_amdgcn_readlane(first, L);
                        found = true;
                    }
                    carry += __builtin_amdgcn_readlane(incl, 63);
                }
            }
        }
    }
    if (!found) vj = S::HRES;
}

/* D2 as a kernel of its own: the CRT_DO_VSYNC 0 variant searches the CLEAN field before the noise stage (k_hsync_wave searches for itself) */
template <class S>
__global__ void __launch_bounds__(64)
k_vsync(int n_fields, const signed char *__restrict__ inp, size_t fstride, crthip_state *__restrict__ state,
        uint2 whole_field, int advance_rn, int clean_signal)
{
    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= n_fields) return;
    crthip_state *st = state + f;
    int vline, vj;
    vsync_search<S>(inp + (size_t) f * fstride, st->vsync, lane, vline, vj);
    if (lane == 0) {
        st->vsync = clean_signal ? -3 : vline;      /* CRT_DO_VSYNC 0 (crt_core.c:323-341): searched in analog[], then pinned */
        st->odd_field = vj > S::HRES / 2;
        if (advance_rn) st->rn = (int) (whole_field.x * (unsigned) st->rn + whole_field.y);
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* D2, D4-D7 (crt_core.c:379-396, 428-510): ONE WAVE PER FIELD, scalar control                                     */
/* ------------------------------------------------------------------------------------------------------------ */
/* A serial walk over a field's 240 lines with everything -- sync search, burst integrators, carrier table, row
 * bookkeeping -- inside one iteration costs ~1800 cycles per line (round 1's kernel: 0.18 ms however small the batch).
 * Here the two serial recurrences are separated and stripped to their cores:
 *   pass 1   hsync chain: per line 16 LDS bytes -> DPP row scan -> ballot -> scalar update           (~200 cycles)
 *   pass 2   burst integrators (crt_core.c:462-467): 10 steps per line and carrier phase of
 *            acc' = acc + s - ((acc >> 7) + (acc > 0 && (acc & 127) != 0))   [== acc * 127 / 128 + s in C]
 *            on CC_VPER x CC_SAMPLES lanes, each walking the lines of its own line class          (~25 cycles a step)
 *   pass 3   everything else (carrier table, positions, rows, ranks, flags) in parallel, one lane per line
 * The lines are processed in chunks of 64 (one per lane): the sync windows of chunk c+1 and the burst samples of chunk
 * c are fetched (global -> registers -> LDS) while chunk c / c-1 are being processed, so a field needs ~12 KB of LDS
 * and 16 fields per CU run interleaved.  A line whose sync window is not where the prefetch put it (hsync far from
 * both 0 and HRES: first lines after a caller-supplied hsync, runaway sync) is served by direct loads -- slower, same
 * result. */
template <class S, bool EXACT_MUL>
__device__ __forceinline__ int burst_step(int acc, int s)
{
    if (EXACT_MUL) {
        const int t127 = (int) (((unsigned) acc << 7) - (unsigned) acc);      /* acc * 127 with 32-bit wrap */
        return ((t127 + ((t127 >> 31) & 127)) >> 7) + s;                       /* C's truncating / 128 */
    }
    /* |acc| < 2^24: acc * 127 / 128 == acc - (acc >> 7) - (acc > 0 && (acc & 127) != 0), the last term being
     * "(acc & 0x8000007f) > 0" as a signed compare = that value clamped to [0, 1].  The chain is ONE dependent stream per
     * lane (1200 steps per field, three quarters of the sync kernel's latency), so what counts is the depth of a step:
     * {and -> med3 | add, shift -> sub} -> sub here, three levels; as a compare into VCC and a subtract-with-borrow it was
     * and -> cmp -> (wait state) -> subb -> add */
    int a1 = acc + s;
    asm volatile("" : "+v"(a1));                             /* (keeps the compiler from forming acc - (acc >> 7) first: one level deeper) */
    int t = a1 - (acc >> 7);
    asm volatile("" : "+v"(t));                              /* (... nor (acc >> 7) + c) */
    const int x = (int) ((unsigned) acc & 0x8000007fu);
    int c;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(c) : "v"(x));      /* (spelled out: the compiler makes min + compare + select of it) */
    return t - c;
}

/* FPB = fields (= waves) per workgroup.  1: the latency shape, everything wave-synchronous.  4 (large batches): the
 * burst integrators of the workgroup's 4 fields are stepped by ONE wave, 4 fields x CC_VPER x CC_SAMPLES lanes at a
 * time, while the other three wait at a barrier -- the chain is the only part that keeps the vector unit busy for long,
 * and with 4 of 64 lanes working per field it costs a quarter this way. */
/* preset_ccf (the fused field-pass): the burst integrators do not start from the state's ccf but from what crt_modulate leaves
 * there (crt_ntsc.c:325-329 and siblings: the burst level of the line class, << 7) -- k_encoder_state's work, done by the lanes
 * that are about to read it, which saves the fused path a launch; field / frame are masked like there. */
/* PAD (r6): inp holds the fused path's padded signal lines (crt_dev.h, sig_layout) -- `shift` bytes, then line n at n * PITCH,
 * `padv` valid bytes of the next line's head behind each.  Every window below is then addressed through sig_phys (line and column
 * of its first byte; contiguous from there), the line table's pos becomes an offset in the padded field, and the few lines whose
 * decoder window would run past the valid copy (hsync far from lock) get that window copied into a scratch row behind the field. */
template <class S, int FPB, bool PAD>
__global__ void __launch_bounds__(64 * FPB, 4)
k_hsync_wave(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
             crthip_state *__restrict__ state, crthip_line *__restrict__ lines, uint2 whole_field, int advance_rn,
             int preset_ccf, int shift, int padv, signed char *__restrict__ scratch)
{
    /* scratch (PAD): the same workspace as inp, as the pointer the scratch rows behind every field are WRITTEN through (rows VRES + 1 ...;
     * nothing this kernel reads lies there: its windows end in line VRES) */
    using G = PadGeom<S>;
    constexpr int LP = PAD ? G::PITCH : S::HRES;
    constexpr int CCS = S::CCS, NB = S::CB_LEN / S::CCS, VPER = S::VPER;
    constexpr int WOFF = S::SYNC_BEG - S::HWIN;          /* first byte of the search window relative to ln + hsync */
    constexpr int CH = 64, NCH = (S::LINES + CH - 1) / CH;
    constexpr int WBACK = 24, WPIECES = 5, WLEN = WPIECES * 16;   /* per line: bytes [ln + WOFF - 24, + 80): hsync -24 .. 40 */
    constexpr int WSTR = 21;                             /* dwords per window row (84 bytes: odd stride, no bank conflicts) */
    constexpr int BPIECES = (S::CB_LEN + 15) / 16;       /* 16-byte pieces covering the CB_LEN burst bytes */
    /* burst row in LDS: the line's NB samples of every carrier phase side by side -- [phase][NB bytes in 3 dwords] -- so that
     * a chain lane gets its line in three dword reads and takes the samples out of them with byte selects that are the same
     * for every lane (SDWA).  Odd row stride: the line lanes write their rows without bank conflicts. */
    constexpr int BDW = (NB + 3) / 4;                    /* dwords per phase string */
    constexpr int BSTR = (CCS * BDW) | 1;                /* dwords per burst row */
    __shared__ int s_win_[FPB][(CH + 1) * WSTR];               /* sync windows of the chunk's lines (+ the line after it) */
    __shared__ int s_bur_[FPB][CH * BSTR];                     /* burst samples, one row per non-skipped line, grouped by line class */
    __shared__ int s_acc_[FPB][CH + 1][CCS == 4 ? 4 : 8];      /* the class's integrators after each of those lines (+ a row nobody reads) */
    __shared__ int s_cnt_[FPB][VPER], s_off_[FPB][VPER];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int vblock = blockIdx.x;
    const int f_raw = vblock * FPB + wv;
    const bool live = f_raw < n_fields;                  /* a wave without a field shadows the last one and stores nothing */
    const int f = live ? f_raw : n_fields - 1;
    int *const s_win = s_win_[wv], *const s_bur = s_bur_[wv];
    int (*const s_acc)[CCS == 4 ? 4 : 8] = s_acc_[wv];
    int *const s_cnt = s_cnt_[wv], *const s_off = s_off_[wv];
#define HSW_SYNC() do { if (FPB > 1) __syncthreads(); else wave_lds_fence(); } while (0)
    const signed char *in = inp + (size_t) f * fstride + (PAD ? shift : 0);
    /* offset of flat sample index a (>= 0) from `in` */
    auto phys = [&](int a) { return PAD ? sig_phys<S>(a) : a; };
    crthip_state *st = state + f;
    int hsync = __builtin_amdgcn_readfirstlane(st->hsync);
    /* D2: the vertical sync search of this field, by the field's own wave (one launch and one trip through memory less
     * than a kernel of its own; what a single field-pass costs is mostly this chain's latency) */
    int vsync, vj_, odd_field;
    if (P.flags & CRTHIP_F_NO_VSYNC) {
        /* CRT_DO_VSYNC 0: k_vsync looked at the clean signal before the noise stage (crt_core.c:323-341) */
        vsync = __builtin_amdgcn_readfirstlane(st->vsync);
        odd_field = __builtin_amdgcn_readfirstlane(st->odd_field);
    } else {
        vsync_search<S, LP>(in, __builtin_amdgcn_readfirstlane(st->vsync), lane, vsync, vj_);
        vsync = __builtin_amdgcn_readfirstlane(vsync);
        odd_field = __builtin_amdgcn_readfirstlane(vj_) > S::HRES / 2;
    }
    if (live && lane == 0) {
        st->vsync = vsync;
        st->odd_field = odd_field;
        if (advance_rn) st->rn = (int) (whole_field.x * (unsigned) st->rn + whole_field.y);
        if (preset_ccf && (!S::LINE_ROWS || !S::NES_TIMING)) { st->field &= 1; st->frame &= 1; }      /* k_encoder_state */
    }
    const int field_rows = odd_field * (P.ratio / 2);                                          /* crt_core.c:407 */
    const unsigned span = (unsigned) P.outh + P.v_fac;
    crthip_line *out_lines = lines + (size_t) f * S::LINES;

    /* chain lanes (wave 0 of the workgroup): lane = (field slot * VPER + r) * CCS + p integrates carrier phase p of line
     * class r of the workgroup's field `slot` */
    constexpr int LPF = VPER * CCS;                      /* chain lanes per field */
    const int my_slot = lane / LPF, my_rp = lane - my_slot * LPF;
    const int my_r = my_rp / CCS, my_p = my_rp - my_r * CCS;
    const bool chain_lane = wv == 0 && my_slot < FPB && vblock * FPB + my_slot < n_fields;
    crthip_state *st_chain = state + (chain_lane ? vblock * FPB + my_slot : 0);
    int acc = chain_lane ? st_chain->ccf[my_r][my_p] : 0;
    if constexpr (!S::IS_VHS) {                          /* (the rand()-noise VHS build runs k_encoder_state itself: it also resets hsync) */
        if (preset_ccf && chain_lane && (S::LINE_ROWS || my_r == 0)) {
            /* what crt_modulate leaves in ccf (k_encoder_state, crt_encode.hip): entry [(cls + CCF_SHIFT) % VPER][p] = the burst
             * level of line class cls at carrier phase p, << 7; the systems without line classes preset row 0 from class 0 */
            const int cls = S::LINE_ROWS ? (my_r + S::VPER - S::CCF_SHIFT % S::VPER) % S::VPER : 0;
            /* field / frame are read here while lane 0 of the field's own wave may be masking them in place (above): harmless ONLY
             * because carrier_row looks at bit 0 of each (or, with LINE_ROWS, at neither) -- masked locally so that this stays true
             * whatever carrier_row becomes */
            const int row = carrier_row<S>(cls, st_chain->field & 1, st_chain->frame & 1, st_chain->aux);
            acc = ((int) (signed char) ((S::BLANK + P.burst[row][my_p] * S::BURST) >> 5)) << 7;
        }
    }
    const bool big = chain_lane && (acc >= (1 << 23) || acc <= -(1 << 23));
    const bool exact_mul = __ballot(big) != 0ull;        /* caller-supplied garbage in ccf: keep the wrapping multiply (wave 0 only) */
    const int k0 = CCS == 4 ? ((my_p - S::CB_BEG) & 3) : ((my_p - S::CB_BEG) % CCS + CCS) % CCS;

    auto lidx_of = [&](int line) { int l = line + vsync; return l >= S::VRES ? l - S::VRES : l; };   /* 0 <= vsync < VRES */
    /* flat start of the sync window of decoded line `line` (lines TOP .. BOT: one past the end is needed too) */
    auto win_base = [&](int line) { return lidx_of(line) * S::HRES + WOFF - WBACK; };

    v4i wreg[WPIECES], wext[WPIECES], breg[BPIECES];
    auto fetch_windows = [&](int c) {
        const int line = S::TOP + c * CH + lane;
        int b = win_base(line < S::BOT ? line : S::BOT);
        if (b < 0) b = 0;
        b = phys(b);                                     /* (padded: 80 bytes from at most 24 columns before a line's end: inside the copy, padv >= 80) */
#pragma unroll
        for (int q = 0; q < WPIECES; q++) wreg[q] = load16u(in + b + q * 16);
        /* every lane also brings the window of the line after the chunk (wrapped hsync values look there) */
        int be = win_base(S::TOP + (c + 1) * CH < S::BOT ? S::TOP + (c + 1) * CH : S::BOT);
        if (be < 0) be = 0;
        be = phys(be);
#pragma unroll
        for (int q = 0; q < WPIECES; q++) wext[q] = load16u(in + be + q * 16);
    };
    auto park_windows = [&]() {
#pragma unroll
        for (int q = 0; q < WPIECES; q++) {
            int *d = s_win + lane * WSTR + q * 4;
            d[0] = wreg[q].x; d[1] = wreg[q].y; d[2] = wreg[q].z; d[3] = wreg[q].w;
            if (lane == 0) {
                int *e = s_win + CH * WSTR + q * 4;
                e[0] = wext[q].x; e[1] = wext[q].y; e[2] = wext[q].z; e[3] = wext[q].w;
            }
        }
    };
    fetch_windows(0);

    /* per-line records of the chunk whose burst chain / line table is still to come (lane = line of the chunk) */
    int rec_hs = 0, rec_rid = 0;
    bool rec_skip = true;

    for (int c = 0; c <= NCH; c++) {
        int new_hs = 0;                                  /* hsync after line `lane` of chunk c (pass 1 fills it lane by lane) */
        bool new_skip = true;
        /* ================= pass 1 of chunk c: the hsync chain ================= */
        if (c < NCH) {
            wave_lds_fence();
            park_windows();
            wave_lds_fence();
            if (c + 1 < NCH) fetch_windows(c + 1);
            const int nl = S::LINES - c * CH < CH ? S::LINES - c * CH : CH;
            /* The recurrence hsync_l = F_l(hsync_{l-1}) is solved as a FIXED POINT, one lane per line: every pass
             * evaluates all 64 lines of the chunk in parallel, line l starting from the previous pass's result of line
             * l-1 (the chunk's first line from the carried-in value).  After pass k the first k lines are final, so it
             * terminates with the serial loop's result (at worst after 64 passes); but F_l hardly depends on its
             * argument -- the search window slides, the sync edge it finds does not -- so a steady picture needs 2
             * passes and a noisy one a handful (crt_core.c:437-450). */
            const int my_line = S::TOP + c * CH + lane;
            const int my_beg = (int) ((unsigned) (my_line - S::TOP) * span / (unsigned) S::LINES + (unsigned) field_rows);
            new_skip = lane >= nl || my_beg >= P.outh;                                   /* D4, crt_core.c:428-432 */
            const int my_lidx = lidx_of(my_line < S::BOT ? my_line : S::TOP);
            const bool own_ok = my_lidx * S::HRES + WOFF - WBACK >= 0;                   /* my window is not clipped at 0 */
            const bool next_ok = my_line + 1 <= S::BOT && my_lidx + 1 < S::VRES;         /* row lane+1 = the next analog line */
            int h_in = hsync;                                                            /* every line starts from the carried-in value */
            for (int pass = 0; pass <= CH; pass++) {
                int h_out = h_in;
                if (!new_skip) {
                    /* the 2*HWIN bytes from ln + h_in + SYNC_BEG - HWIN: from my parked window, the next line's (wrapped
                     * hsync), or -- rarely -- from memory */
                    int a = -1;
                    if (h_in >= 0 && h_in <= WLEN - 2 * S::HWIN - WBACK && own_ok) a = lane * (WSTR * 4) + h_in + WBACK;
                    else if (h_in >= S::HRES - WBACK && h_in < S::HRES && next_ok) a = (lane + 1) * (WSTR * 4) + h_in - S::HRES + WBACK;
                    unsigned w0, w1, w2, w3;
                    if (a >= 0) {
                        const unsigned *q = (const unsigned *) s_win + (a >> 2);
                        const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                        const unsigned sh = (unsigned) (a & 3);
                        w0 = __builtin_amdgcn_alignbyte(d1, d0, sh); w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                        w2 = __builtin_amdgcn_alignbyte(d3, d2, sh); w3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
                    } else {
                        const long ga = (long) my_lidx * S::HRES + h_in + WOFF;
                        v4i g = { 0, 0, 0, 0 };
                        if (PAD) { if (ga >= 0 && ga < (long) (S::INPUT_SIZE + S::HRES)) g = load16u(in + phys((int) ga)); }
                        else if (ga >= 0 && ga + 16 <= (long) fstride) g = load16u(in + ga);
                        w0 = (unsigned) g.x; w1 = (unsigned) g.y; w2 = (unsigned) g.z; w3 = (unsigned) g.w;
                    }
                    /* running sum biased by -(HTHR + 1): its sign bit says "sum <= HTHR"; the sign bits are shifted into
                     * a mask, first sample in the highest position */
                    const unsigned wd[4] = { w0, w1, w2, w3 };
                    int run = -(S::HTHR + 1);
                    unsigned mask = 0;
#pragma unroll
                    for (int k = 0; k < 2 * S::HWIN; k++) {
                        run += (int) (wd[k >> 2] << (24 - 8 * (k & 3))) >> 24;
                        mask = __builtin_amdgcn_alignbit(mask, (unsigned) run, 31);
                    }
                    /* sample k sits at bit 2*HWIN-1-k: the first crossing is the highest set bit */
                    const int hi = mask ? (31 - (int) __builtin_clz(mask)) : -1;
                    const int idx = mask ? (2 * S::HWIN - 1 - hi) - S::HWIN : S::HWIN;
                    int h = idx + h_in;                                                  /* POSMOD(i + hsync, HRES), :447 */
                    if (h_in >= 0 && h_in < S::HRES) {                                   /* |idx| <= HWIN: one wrap either way */
                        if (h < 0) h += S::HRES;
                        if (h >= S::HRES) h -= S::HRES;
                    } else {
                        h = posmod(h, S::HRES);
                    }
                    if (P.flags & CRTHIP_F_NO_HSYNC) h = 0;                              /* CRT_DO_HSYNC 0: crt_core.c:448-450 */
                    h_out = h;
                }
                new_hs = h_out;
                /* next pass: line l starts from this pass's line l-1 */
                int h_next = __shfl_up(h_out, 1);
                if (lane == 0) h_next = hsync;
                const bool changed = h_next != h_in;
                h_in = h_next;
                if (__ballot(changed) == 0ull) break;
            }
            hsync = __builtin_amdgcn_readlane(new_hs, nl - 1);
        }
        /* ================= pass 2 of chunk c - 1: the burst integrators ================= */
        if (c > 0) {
            const int cc = c - 1;
            const int nl = S::LINES - cc * CH < CH ? S::LINES - cc * CH : CH;
            /* the burst samples fetched one iteration ago -> LDS, one row per non-skipped line in class order */
            if (!rec_skip) {
                /* de-interleave: sample q of phase string k is byte k + CCS * q of the fetched window */
                int raw[BPIECES * 4];
#pragma unroll
                for (int q = 0; q < BPIECES; q++) { raw[4 * q] = breg[q].x; raw[4 * q + 1] = breg[q].y; raw[4 * q + 2] = breg[q].z; raw[4 * q + 3] = breg[q].w; }
#pragma unroll
                for (int k = 0; k < CCS; k++) {
#pragma unroll
                    for (int wd = 0; wd < BDW; wd++) {
                        unsigned v = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int q = 4 * wd + b, i = k + CCS * q;       /* compile-time */
                            if (q < NB) v |= (((unsigned) raw[i >> 2] >> (8 * (i & 3))) & 0xffu) << (8 * b);
                        }
                        s_bur[rec_rid * BSTR + k * BDW + wd] = (int) v;
                    }
                }
            }
            HSW_SYNC();
            if (wv == 0) {
            const int n_mine = chain_lane ? s_cnt_[my_slot][my_r] : 0;
            const int row0 = chain_lane ? s_off_[my_slot][my_r] : 0;
            int n_max = n_mine;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(n_max, o); n_max = v > n_max ? v : n_max; }
            /* my phase string of my class's first line: BDW dwords; sample q = byte q & 3 of dword q >> 2 */
            const int *bp = s_bur_[chain_lane ? my_slot : 0] + row0 * BSTR + k0 * BDW;
            int cur[BDW], nxt[BDW];
#pragma unroll
            for (int wd = 0; wd < BDW; wd++) { cur[wd] = n_mine > 0 ? bp[wd] : 0; nxt[wd] = 0; }
            int j = 0;
            if (!exact_mul) {
                /* The lines every chain lane has, in a loop without a single divergent branch: all 64 lanes run it (the lanes
                 * without a chain integrate row 0 into a register nobody reads and store into the spare row of s_acc); a lone
                 * wave pays ~9 cycles per dependent instruction, so the exec-mask bookkeeping of a predicated body (two
                 * branches per line) cost as much as the ten steps of the line. */
                int n_min = chain_lane ? n_mine : 0x7fffffff;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(n_min, o); n_min = v < n_min ? v : n_min; }
                if (n_min == 0x7fffffff) n_min = 0;                      /* (no chain lane at all: cannot happen, slot 0 always has a field) */
                int *ap = chain_lane ? &s_acc_[my_slot][row0][my_p] : &s_acc_[0][CH][0];
                const int astep = chain_lane ? (int) (sizeof(s_acc_[0][0]) / sizeof(int)) : 0;
                for (; j < n_min; j++) {
                    const int jn = j + 1 < n_mine ? j + 1 : j;           /* (the last prefetch stays inside my rows) */
#pragma unroll
                    for (int wd = 0; wd < BDW; wd++) nxt[wd] = bp[jn * BSTR + wd];
#pragma unroll
                    for (int q = 0; q < NB; q++) acc = burst_step<S, false>(acc, (cur[q >> 2] << (24 - 8 * (q & 3))) >> 24);
                    *ap = acc;
                    ap += astep;
#pragma unroll
                    for (int wd = 0; wd < BDW; wd++) cur[wd] = nxt[wd];
                }
            }
            for (; j < n_max; j++) {                                     /* the rest (and caller-supplied garbage in ccf), predicated */
                if (j + 1 < n_mine) {                                    /* next line's samples while this one's chain runs */
#pragma unroll
                    for (int wd = 0; wd < BDW; wd++) nxt[wd] = bp[(j + 1) * BSTR + wd];
                }
                if (j < n_mine) {
                    if (exact_mul) {
#pragma unroll
                        for (int q = 0; q < NB; q++) acc = burst_step<S, true>(acc, (cur[q >> 2] << (24 - 8 * (q & 3))) >> 24);
                    } else {
#pragma unroll
                        for (int q = 0; q < NB; q++) acc = burst_step<S, false>(acc, (cur[q >> 2] << (24 - 8 * (q & 3))) >> 24);
                    }
                    s_acc_[my_slot][row0 + j][my_p] = acc;
                }
#pragma unroll
                for (int wd = 0; wd < BDW; wd++) cur[wd] = nxt[wd];
            }
            }
            HSW_SYNC();
            /* ================= pass 3 of chunk c - 1: the line table, one lane per line ================= */
            if (lane < nl && live) {
                const int i = lane, line = S::TOP + cc * CH + i;
                const int hs = rec_hs;
                const bool skip = rec_skip;
                int beg = (int) ((unsigned) (line - S::TOP + 0) * span / (unsigned) S::LINES + (unsigned) field_rows);
                int end = (int) ((unsigned) (line - S::TOP + 1) * span / (unsigned) S::LINES + (unsigned) field_rows);
                if (end > P.outh) end = P.outh;
                crthip_line lp;
                lp.dx = P.dx; lp.scanl = 0;                                /* :528-529; k_bloom rewrites them per line */
                lp.hsync = hs;
                if (skip) {
                    lp.pos = 0; lp.wave0 = 0; lp.wave1 = 0; lp.beg = 0; lp.nrows = 0;
                } else {
                    const int *ac = s_acc[rec_rid];
                    const int lidx = lidx_of(line);
                    const int xpos = posmod(S::AV_BEG + hs - 3, S::HRES);  /* :452-454 */
                    int ypos = lidx + 3;
                    if (ypos >= S::VRES) ypos -= S::VRES;
                    lp.pos = xpos + ypos * S::HRES;
                    const bool reads_tail = lp.pos + S::AV_LEN + 8 > S::INPUT_SIZE;
                    if constexpr (PAD) {
                        /* where the decoders find the window: in place, or -- it would run past the valid copy behind its line -- in
                         * the line's scratch row, filled here piece by piece (a piece starts inside a line: contiguous) */
                        const int flat = lp.pos;
                        lp.pos = shift + ypos * G::PITCH + xpos;
                        if (xpos + G::DECWIN > S::HRES + padv) {
                            const int srow = shift + (G::SCR_LINE0 + (line - S::TOP)) * G::PITCH;
                            signed char *scr = scratch + (size_t) f * fstride + srow;
                            for (int o = 0; o < G::DECWIN; o += 16) store16u(scr + o, load16u(in + sig_phys<S>(flat + o)));
                            lp.pos = srow;
                        }
                    }
                    int dci, dcq;
                    const int pa = posmod(hs, CCS);
                    if constexpr (CCS == 4) {                              /* :471-472 */
                        dci = ac[(pa + 1) & 3] - ac[(pa + 3) & 3];
                        dcq = ac[(pa + 2) & 3] - ac[(pa + 0) & 3];
                        lp.wave0 = ((dci * P.huecs - dcq * P.huesn) >> 4) * P.saturation;
                        lp.wave1 = ((dcq * P.huecs + dci * P.huesn) >> 4) * P.saturation;
                    } else {                                               /* :480-494 */
                        const int peak_a = pa + CCS / 4, peak_b = pa;
                        const int dci_a = ac[peak_a % CCS];
                        const int dci_b = (ac[(peak_a + CCS / 2) % CCS] + ac[(peak_a + CCS / 2 + 1) % CCS]) / 2;
                        const int dcq_a = ac[(peak_b + CCS / 2) % CCS];
                        const int dcq_b = ac[peak_b % CCS];
                        dci = dci_a - dci_b;
                        dcq = dcq_a - dcq_b;
                        lp.wave0 = dci;
                        lp.wave1 = dcq;
                    }
                    lp.beg = beg;
                    int nrows = end - P.scanlines - beg;                    /* rows beg .. end-scanlines-1, :662 */
                    nrows = nrows < 1 ? 1 : nrows;
                    if (nrows > CRTHIP_LINE_NROWS_MASK) nrows = CRTHIP_LINE_NROWS_MASK;
                    /* rank among the lines that start on the same output row (outh + v_fac < LINES): beg is monotone in
                     * the line and no line before a non-skipped one is skipped, so it is the distance to the first
                     * line with this beg */
                    int rank = 0;
                    if (span < (unsigned) S::LINES) {
                        while (line - rank - 1 >= S::TOP &&
                               (int) ((unsigned) (line - rank - 1 - S::TOP) * span / (unsigned) S::LINES + (unsigned) field_rows) == beg)
                            rank++;
                    }
                    nrows |= (rank & CRTHIP_LINE_RANK_MASK) << CRTHIP_LINE_RANK_SHIFT;
                    nrows |= line_tier_flags<CCS>(lp.wave0, lp.wave1, P.saturation, P.loskip_wave_max, reads_tail);
                    lp.nrows = nrows;
                }
                v4i a, b;
                a.x = lp.pos; a.y = lp.wave0; a.z = lp.wave1; a.w = lp.beg;
                b.x = lp.nrows; b.y = lp.hsync; b.z = lp.dx; b.w = lp.scanl;
                store16u(out_lines + cc * CH + i, a);
                store16u((int *) (out_lines + cc * CH + i) + 4, b);
            }
            wave_lds_fence();
        }
        /* ================= hand-over: chunk c's per-line records for passes 2 / 3, burst fetch ================= */
        if (c < NCH) {
            rec_hs = new_hs;
            rec_skip = new_skip;
            const int line = S::TOP + c * CH + lane;
            const int lidx = lidx_of(line < S::BOT ? line : S::TOP);
            int ypos = lidx + 3;
            if (ypos >= S::VRES) ypos -= S::VRES;
            const int cls = VPER == 1 ? 0 : ypos % VPER;                   /* :456 */
            /* row of the line in s_bur / s_acc: lines of class 0 first, then class 1, ...; line order inside a class */
            int off = 0;
            rec_rid = 0;
#pragma unroll
            for (int r = 0; r < VPER; r++) {
                const unsigned long long m = __ballot(!rec_skip && cls == r);
                if (!rec_skip && cls == r) rec_rid = off + __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) { s_cnt[r] = __popcll(m); s_off[r] = off; }
                off += __popcll(m);
            }
            /* burst samples: CB_LEN bytes from ln + halign + CB_BEG (:459-461) */
            const int halign = CCS == 4 ? (rec_hs & ~3) : rec_hs - rec_hs % CCS;
            const int baddr = phys(lidx * S::HRES + halign + S::CB_BEG);      /* (padded: BPIECES * 16 <= padv bytes from any column) */
#pragma unroll
            for (int q = 0; q < BPIECES; q++) {
                breg[q] = !rec_skip ? load16u(in + baddr + q * 16) : v4i{0, 0, 0, 0};
            }
            wave_lds_fence();
        }
    }
    if (chain_lane) st_chain->ccf[my_r][my_p] = acc;
    if (lane == 0 && live) st->hsync = hsync;
#undef HSW_SYNC
}

/* CRT_DO_BLOOM (crt_core.c:399-402, 512-526): the beam energy of every decoded line (sum of its AV_LEN samples)
 * drives a leaky integrator prev_e that runs over the lines of a field in order; each line gets its own width,
 * i.e. its own resampler step dx and start scanL.  One workgroup per field:
 *   1. the four waves sum the lines, a wave per line at a time: coalesced dword loads, v_dot4 against 0x01010101, a wave
 *      reduction (the first version gave every lane its own line: 64 cache lines per load instruction, 0.65 ms per 4096
 *      fields, 2.5x the bytes);
 *   2. the input term of :518 does not depend on the chain: one lane per line computes it (the division by max_e);
 *   3. the 240-step chain is now multiply, divide by 128, add: walked by wave 0 on the scalar unit;
 *   4. one lane per line turns prev_e into line_w, dx (the division by outw) and scanL, and patches the line table. */
template <class S>
__global__ void __launch_bounds__(256)
k_bloom(const crthip_params P, int n_fields, const signed char *__restrict__ inp, size_t fstride,
        crthip_line *__restrict__ lines)
{
    __shared__ int s_sum[S::LINES];            /* the line sum, then its chain term, then prev_e after the line */
    __shared__ int s_nrows[S::LINES];
    __shared__ int s_pos[S::LINES];
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= n_fields) return;
    crthip_line *lt = lines + (size_t) f * S::LINES;
    if (t < S::LINES) {
        const crthip_line lp = lt[t];
        s_nrows[t] = lp.nrows & CRTHIP_LINE_NROWS_MASK;
        s_pos[t] = lp.pos;
    }
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    /* 16 bytes per lane: one load instruction covers 1024 samples of a line; eight lines per wave and round, straight-line
     * (lines nobody sees are summed too and dropped) */
    constexpr int Q = S::AV_LEN / 16, REM = S::AV_LEN - Q * 16;
    constexpr int UNR = 8;
    for (int l0 = wave; l0 < S::LINES; l0 += 4 * UNR) {
        int sum[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int l = l0 + 4 * u < S::LINES ? l0 + 4 * u : l0;
            const signed char *sig = inp + (size_t) f * fstride + s_pos[l];
            sum[u] = 0;
#pragma unroll
            for (int q0 = 0; q0 < Q; q0 += 64) {
                if (q0 + lane < Q) {
                    const v4i v = gload16u((unsigned long long) (sig + 16 * (q0 + lane)));
                    sum[u] = __builtin_amdgcn_sdot4(v.x, 0x01010101, sum[u], false);
                    sum[u] = __builtin_amdgcn_sdot4(v.y, 0x01010101, sum[u], false);
                    sum[u] = __builtin_amdgcn_sdot4(v.z, 0x01010101, sum[u], false);
                    sum[u] = __builtin_amdgcn_sdot4(v.w, 0x01010101, sum[u], false);
                }
            }
            if (lane == 63) {
#pragma unroll
                for (int b = 0; b < REM; b++) sum[u] += sig[16 * Q + b];
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
#pragma unroll
            for (int u = 0; u < UNR; u++) sum[u] += __shfl_xor(sum[u], off);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
            if (lane == 0 && l0 + 4 * u < S::LINES) s_sum[l0 + 4 * u] = sum[u];
    }
    __syncthreads();
    const int max_e = P.bloom_max_e;                                      /* :400 */
    if (t < S::LINES && s_nrows[t] != 0) s_sum[t] = (((max_e >> 1) - s_sum[t]) << 10) / max_e;
    __syncthreads();
    if (t < 64) {
        /* the chain on the scalar unit: the terms sit in registers (lane = line & 63), v_readlane at a scalar
         * index instead of two dependent LDS round trips per line */
        constexpr int NJ = (S::LINES + 63) / 64;
        int term[NJ], pe[NJ];
        unsigned long long valid[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int l = j * 64 + t;
            const bool on = l < S::LINES && s_nrows[l < S::LINES ? l : 0] != 0;
            term[j] = on ? s_sum[l] : 0;
            pe[j] = 0;
            valid[j] = __ballot(on);                                      /* :431: skipped lines do not reach :512 */
        }
        int prev_e = 16384 / 8;                                           /* :401 */
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            for (int i = 0; i < 64; i++) {
                if (!((valid[j] >> i) & 1ull)) continue;
                prev_e = (prev_e * 123 / 128) + __builtin_amdgcn_readlane(term[j], i);      /* :518 */
                pe[j] = t == i ? prev_e : pe[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++)
            if (j * 64 + t < S::LINES) s_sum[j * 64 + t] = pe[j];
    }
    __syncthreads();
    if (t < S::LINES && s_nrows[t] != 0) {
        const int line_w = (S::AV_LEN * 112 / 128) + (s_sum[t] >> 9);     /* :519 */
        lt[t].dx = (line_w << 12) / P.outw;                               /* :521 */
        lt[t].scanl = ((S::AV_LEN / 2) - (line_w >> 1) + 8) << 12;        /* :522 */
    }
}


/* CRT_DO_VSYNC 0 (crt_core.c:323-341): the field parity is looked for in the CLEAN signal, before the noise stage, and
 * vsync is pinned to -3; the sync kernels then skip their own search (CRTHIP_F_NO_VSYNC) */
int crt_run_clean_vsync(crthip_ctx *c, int n, const signed char *d_analog, crthip_state *d_state)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        ProfScope ps(c, CRTHIP_K_SYNC);
        hipLaunchKernelGGL((k_vsync<S>), dim3(n), dim3(64), 0, c->stream, n, d_analog, c->fstride, d_state, c->whole_field, 0, 1);
        return CRTHIP_OK;
    });
}

/* the sync chain of n fields on the context's stream: vertical search, hsync fixed point, burst integrators, line table (+ the
 * bloom pass).  preset_ccf: see k_hsync_wave. */
int crt_run_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
                 crthip_line *d_lines, int advance_rn, int preset_ccf, const sig_layout *lay)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        ProfScope ps(c, CRTHIP_K_SYNC);
        /* 4 fields per workgroup share one wave for their burst chains (large batches); CRTHIP_SYNC_KERNEL=2 / 3 force 1 / 4
         * fields per workgroup (A/B).  (r5) threshold re-measured, session r5s23: 512 fields 0.078 (one) / 0.085 ms (four) and the
         * field-pass 0.566 / 0.606; 1024 fields 0.097 / 0.090; 4096 fields 0.188 / 0.165; 1080p x 2048 0.140 / 0.119 */
        constexpr bool FPB4_OK = 64 / (S::VPER * S::CCS) >= 4;
        const bool pad = lay && lay->pitch != S::HRES;
        const size_t fstride = lay ? lay->fstride : c->fstride;
        const int shift = pad ? lay->shift : 0, padv = pad ? lay->padv : 0;
        const bool fpb4 = FPB4_OK && c->sync_kernel != 2 && (n >= SYNC_FPB4_MIN_FIELDS || c->sync_kernel == 3);
#define CRTHIP_LAUNCH_SYNC(FPB, PADV) \
    hipLaunchKernelGGL((k_hsync_wave<S, FPB, PADV>), dim3((n + FPB - 1) / FPB), dim3(64 * FPB), 0, c->stream, *p, n, d_inp, fstride, d_state, d_lines, \
                       c->whole_field, advance_rn, preset_ccf, shift, padv, pad ? const_cast<signed char *>(d_inp) : nullptr)   /* (padded: the library's own workspace) */
        if constexpr (FPB4_OK) {
            if (fpb4) { if (pad) CRTHIP_LAUNCH_SYNC(4, true); else CRTHIP_LAUNCH_SYNC(4, false); }
        }
        if (!fpb4) { if (pad) CRTHIP_LAUNCH_SYNC(1, true); else CRTHIP_LAUNCH_SYNC(1, false); }
#undef CRTHIP_LAUNCH_SYNC
        if (p->bloom)
            hipLaunchKernelGGL((k_bloom<S>), dim3(n), dim3(256), 0, c->stream, *p, n, d_inp, fstride, d_lines);
        return CRTHIP_OK;
    });
}
