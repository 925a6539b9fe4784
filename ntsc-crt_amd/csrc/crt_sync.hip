/* crt_sync.hip -- D2-D7: vertical / horizontal sync search, burst lock, per-line carrier table.  See crt_dev.h. */
#include "crt_dev.h"

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


int crt_run_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
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
