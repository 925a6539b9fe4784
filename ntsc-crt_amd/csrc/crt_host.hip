/* crt_host.hip -- context, the crthip_* C ABI (include/crt_hip.h), sequence mode.  See crt_dev.h. */
#include "crt_dev.h"

/* ------------------------------------------------------------------------- */
/* Sequence mode: n consecutive fields of ONE television set  */
/* ------------------------------------------------------------------------- */
/* Sequential semantics of   for k: crt_modulate(field k); crt_demodulate(); save(out)   (the loop of
 * extra/video_convert.c:246-277) reproduced with parallel kernels:
 *   rn      : closed form, rn_k = J^k(rn_0), J = INPUT_SIZE steps of the LCG;
 *   encoder : independent per field (fused, writes the noisy field);
 *   sync    : field k starts from field k-1's final (hsync, vsync).  Solved as a fixed point: every
 *             pass runs k_hsync_wave for ALL fields in parallel with init_k = final_{k-1} of the
 *             previous pass; after pass j fields 0..j-1 are final for good, and because a field's final
 *             state hardly ever depends on its initial one the iteration normally stops after 2-3 passes;
 *   decoder : independent per field given its line table (blend must be 0: with blend the output is a
 *             recurrence over fields);
 *   weave   : output image k = the single output buffer after field k: rows field k does not write come
 *             from the latest earlier field that wrote them (or the initial buffer). */
/* state[k].rn = J^(first_index + k)(rn0): the set's generator before field first_index + k of the video, rn0 = before
 * field 0 (a shard of a longer video starts at first_index > 0) */
__global__ void k_seq_rn(int n_fields, crthip_state *state, uint2 whole_field, unsigned first_index, unsigned rn0)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_fields) return;
    /* (m, a) = J^e by square and multiply */
    unsigned pm = whole_field.x, pa = whole_field.y, m = 1u, a = 0u;
    for (unsigned e = first_index + (unsigned) k; e; e >>= 1) {
        if (e & 1u) { m = pm * m; a = pm * a + pa; }
        pa = pm * pa + pa;
        pm = pm * pm;
    }
    state[k].rn = (int) (m * rn0 + a);
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
            const int *latest, int patch_only)
{
    const int row = blockIdx.x % outh, k = blockIdx.x / outh;
    if (k >= n_fields) return;
    const int src_k = latest[(size_t) k * outh + row];
    if (src_k == k) return;
    if (patch_only && src_k >= 0) return;       /* second visit (cross-shard chain): only the rows nobody here wrote */
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

/* blend != 0 in sequence mode (crt_main.c:235 sets blend = 1): the picture is a recurrence over the fields,
 *     v_k = ((new_k & 0xfefeff) >> 1) + ((v_{k-1} & 0xfefeff) >> 1)        (crt_core.c:584-609)
 * on the first row of every line; the duplicated rows below it are copies of that BLENDED row (:661-664).  The fields
 * are decoded in parallel without blend (image k = new_k, duplicates included), then one pass per field, parallel
 * over the picture, folds the previous image in:  src[k][row] = the first row of the line that wrote `row` (-1: nobody
 * did, the row is carried over). */
__global__ void k_seq_rowsrc(int n_fields, int lines_per_field, int outh, const crthip_line *lines, int *rowsrc)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_fields * lines_per_field) return;
    const int k = gid / lines_per_field;
    const crthip_line lp = lines[gid];
    const int nrows = lp.nrows & CRTHIP_LINE_NROWS_MASK;
    for (int r = 0; r < nrows; r++) {
        if (lp.beg + r < outh) rowsrc[(size_t) k * outh + lp.beg + r] = lp.beg;
    }
}

__global__ void __launch_bounds__(256)
k_seq_blend_step(int k, int outh, size_t pitch, unsigned char *out, size_t ostride, const unsigned char *init,
                 const int *rowsrc, unsigned alpha_mask)
{
    const int row = blockIdx.x;
    const int src = rowsrc[(size_t) k * outh + row];
    unsigned char *cur = out + (size_t) k * ostride + (size_t) row * pitch;
    const unsigned char *prev_img = k ? out + (size_t) (k - 1) * ostride : init;      /* nullptr: zeros (calloc) */
    const unsigned char *old = prev_img ? prev_img + (size_t) (src < 0 ? row : src) * pitch : nullptr;
    for (size_t b = (size_t) threadIdx.x * 4; b < pitch; b += 256 * 4) {
        const size_t nb = pitch - b < 4 ? pitch - b : 4;
        unsigned o = 0, n = 0;
        for (size_t c = 0; c < nb; c++) {
            o |= (unsigned) (old ? old[b + c] : 0) << (8 * c);
            n |= (unsigned) cur[b + c] << (8 * c);
        }
        /* every colour byte: (new >> 1) + (old >> 1) -- what the 0xfefeff mask computes per channel; alpha stays 0xff */
        const unsigned v = src < 0 ? o : ((((n >> 1) & 0x7f7f7f7fu) + ((o >> 1) & 0x7f7f7f7fu)) | alpha_mask);
        for (size_t c = 0; c < nb; c++) cur[b + c] = (unsigned char) (v >> (8 * c));
    }
}


/* (r6) the fused path's padded signal (crt_dev.h, sig_layout) back in the reference's flat layout, 16 samples per lane: what
 * crthip_fieldpass_signal hands out (tests compare it with the oracle's inp[] byte for byte; nothing in a field-pass needs it) */
template <class S>
__global__ void __launch_bounds__(256)
k_unpad(int n_fields, const signed char *__restrict__ src, size_t sstride, int shift, signed char *__restrict__ dst, size_t dstride)
{
    constexpr int CHUNKS = (S::INPUT_SIZE + CRTHIP_TAIL + 15) / 16;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= CHUNKS) return;
    for (int f = blockIdx.y; f < n_fields; f += (int) gridDim.y)
        store16u(dst + (size_t) f * dstride + q * 16, load16u(src + (size_t) f * sstride + shift + sig_phys<S>(q * 16)));
}

int crt_run_unpad(crthip_ctx *c, int n, const sig_layout *lay, const signed char *d_src, signed char *d_dst)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        if (lay->pitch == S::HRES) {
            if (hipMemcpy2DAsync(d_dst, c->fstride, d_src, lay->fstride, (size_t) S::INPUT_SIZE + CRTHIP_TAIL, (size_t) n,
                                 hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return CRTHIP_E_HIP;
            return CRTHIP_OK;
        }
        constexpr int CHUNKS = (S::INPUT_SIZE + CRTHIP_TAIL + 15) / 16;
        hipLaunchKernelGGL((k_unpad<S>), dim3((CHUNKS + 255) / 256, n < 65535 ? n : 65535), dim3(256), 0, c->stream, n, d_src, lay->fstride,
                           lay->shift, d_dst, c->fstride);
        return CRTHIP_OK;
    });
}

/* Which layout the signal of a fused field-pass takes (crt_dev.h, sig_layout): padded lines whenever the geometry allows --
 *   the row's overhang over its line (wrap = xo + destw - HRES) is at most 16 samples and ends inside the field,
 *   the row starts at or behind column PADW (what is copied behind a line is then margin, never picture), and
 *   the copies the margin kernel makes reach at least 80 columns (sync windows: 69, burst windows: 48 / 64)
 * -- and the flat layout otherwise (odd x offsets, the rand()-noise VHS build, CRT_DO_VSYNC 0, the NES, small batches,
 * crthip_set_signal_layout(ctx, 0)). */
/* (the rule without a context: crthip_signal_layout_query lets a host -- and the CPU tests -- ask it) */
static bool layout_rule(int system, int pattern, const struct crt_sysdef &sd, size_t flat_stride, const crthip_params *p, int n, int shape,
                        int sig_pad, sig_layout *lay)
{
    lay->pitch = sd.hres; lay->shift = 0; lay->padv = 0; lay->wrap = 0; lay->fstride = flat_stride;
    if (!sig_pad || system == CRTHIP_SYSTEM_NTSCVHS || (p->flags & CRTHIP_F_NO_VSYNC)) return false;
    /* ... and only where it pays (profiles/r06_ab_padded_by_system.txt, one box, padded against flat): the lane-per-row RGB encoder,
     * whose row stores it aligns -- NTSC 640x480 x 1024 +5 %, SNES +5.6 %, PV-1000 +4.5 %, bloom +5.4 % -- but not the NES's table
     * encoder (never store-bound: the margin kernel's copies cost 1.1 % and buy nothing) and not the batches the library gives to the
     * scanline-parallel encoder by itself (dword stores per lane: 640x480 x 64 -0.8 %, 1080p x 64 -3.5 %).  A FORCED scanline-parallel
     * shape keeps the padded lines (tests run k_active_row's padded stores that way). */
    if (sd.ppu_input) return false;
    if (shape == 0 && n <= ROWS_SHAPE_MAX_FIELDS_ENC) return false;
    if (p->in_bpp == 0) return false;                 /* crt_modulate refuses the format: the noise kernel writes the (flat) field */
    return dispatch_system(system, pattern, [&](auto tag) {
        using S = decltype(tag);
        using G = PadGeom<S>;
        const int over = p->xo + p->destw - S::HRES, wrap = over > 0 ? over : 0;
        if (wrap > 16 || p->xo < G::PADW || p->destw < 16 || p->yo < 0 || p->yo + p->desth + (wrap ? 1 : 0) > S::VRES) return 0;
        /* k_margin_pad copies whole 16-byte chunks: [wrap, wrap + 16 m) behind a line that carries a row, [0, 16 m') elsewhere */
        const int a = wrap + (G::PADC - wrap) / 16 * 16, b = G::PADC / 16 * 16;
        const int padv = a < b ? a : b;
        if (padv < 80) return 0;
        lay->pitch = G::PITCH;
        lay->shift = (128 - p->xo % 128) % 128;
        lay->padv = padv;
        lay->wrap = wrap;
        lay->fstride = G::FSTRIDE;
        return 1;
    }) == 1;
}

bool crt_fused_layout(const crthip_ctx *c, const crthip_params *p, int n, sig_layout *lay)
{
    return layout_rule(c->system, c->pattern, c->sd, c->fstride, p, n, c->shape, c->sig_pad, lay);
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
    c->overlap_chunks = 0;      /* automatic */
    { const char *e = getenv("CRTHIP_SYNC_KERNEL"); c->sync_kernel = e ? atoi(e) : 0; }
    { const char *e = getenv("CRTHIP_ROW_TILE"); c->row_tile = e ? atoi(e) : 0; }
    { const char *e = getenv("CRTHIP_WIDE_DECODE"); c->wide_decode = e ? atoi(e) != 0 : 1; }     /* A/B switch, crt_decode4.hip */
    { const char *e = getenv("CRTHIP_AC_TILE"); c->ac_tile_env = e && (atoi(e) == 16 || atoi(e) == 32) ? atoi(e) : 0; }   /* A/B switch, k_active */
    { const char *e = getenv("CRTHIP_WIDE_LPW"); c->wide_lpw_env = e && (atoi(e) == 8 || atoi(e) == 16) ? atoi(e) : 0; }   /* A/B switch, k_decode_wide */
    { const char *e = getenv("CRTHIP_MARGIN_SIDE"); c->margin_side = e ? atoi(e) != 0 : 1; }   /* A/B switch: k_margin beside k_active (crt_encode.hip) */
    { const char *e = getenv("CRTHIP_SIG_PAD"); c->sig_pad = e ? atoi(e) != 0 : 1; }       /* A/B switch: the fused path's signal layout (crt_dev.h, sig_layout) */
    c->fstride_pad = 0;
    dispatch_system(system, chroma_pattern, [&](auto tag) { c->fstride_pad = PadGeom<decltype(tag)>::FSTRIDE; return CRTHIP_OK; });
    { const char *e = getenv("CRTHIP_WIDE_ORDER"); c->wide_order_env = e ? atoi(e) : 0; }   /* A/B switches: workgroup order (crt_dev.h, block_item) */
    { const char *e = getenv("CRTHIP_DEC_ORDER"); c->dec_order_env = e ? atoi(e) : 0; }
    { const char *e = getenv("CRTHIP_ACT_ORDER"); c->act_order_env = e ? atoi(e) : 0; }
    { const char *e = getenv("CRTHIP_SIG_TILE"); c->sig_tile_env = e && (atoi(e) == 16 || atoi(e) == 32 || atoi(e) == 64) ? atoi(e) : 0; }   /* A/B switch, k_active */
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
    if (system == CRTHIP_SYSTEM_NES && (hipMalloc((void **) &c->d_nes_tab, NES_TAB_SIZE) != hipSuccess ||
                                        hipMalloc((void **) &c->d_nes_tab_alt, NES_TAB_SIZE) != hipSuccess)) {
        crthip_destroy(c);
        return CRTHIP_E_NOMEM;
    }
    {
        uint2 j1[16];
        j1[0] = make_uint2(1u, 0u);
        for (int k = 1; k < 16; k++) j1[k] = make_uint2(LCG_MUL * j1[k - 1].x, LCG_MUL * j1[k - 1].y + LCG_ADD);
        if (hipMalloc((void **) &c->d_skel, (size_t) SKEL_VARIANTS * c->fstride) != hipSuccess ||
            hipMalloc((void **) &c->d_skel_alt, (size_t) SKEL_VARIANTS * c->fstride) != hipSuccess ||
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
        /* the same coefficients as SIGNED BYTE DIGITS for the matrix-core jump (crt_noise.hip, vhs_jump_mfma): a 32-bit word
         * x == d0 + d1 * 2^8 + d2 * 2^16 + d3 * 2^24 (mod 2^32), every d in [-128, 127]; per row 4 digit planes of 32 bytes
         * (coefficient m at byte m, byte 31 = 0): rows 0 .. chunks-1 the parallel chunks, then the tail's 64 blocks */
        const size_t dig_rows = (size_t) chunks + 64;
        signed char *dig = (signed char *) calloc(dig_rows, VHS_DIG_ROW);
        if (!dig) { free(rows); crthip_destroy(c); return CRTHIP_E_NOMEM; }
        for (size_t r = 0; r < dig_rows; r++) {
            for (int m = 0; m < 31; m++) {
                unsigned x = r < (size_t) chunks ? rows[r * 31 + m] : rows[31 * (size_t) (chunks + 1) + (size_t) m * 64 + (r - chunks)];
                for (int pl = 0; pl < 4; pl++) {
                    const int d = (int) (signed char) (x & 255u);
                    dig[r * VHS_DIG_ROW + pl * 32 + m] = (signed char) d;
                    x = (x - (unsigned) d) >> 8;
                }
            }
        }
        { const char *e = getenv("CRTHIP_VHS_MFMA"); c->vhs_mfma = e ? atoi(e) != 0 : 1; }       /* A/B switch: 0 = the jump on the vector unit */
        if (hipMalloc((void **) &c->d_vhs_rows, sizeof(unsigned) * words) != hipSuccess ||
            hipMemcpy(c->d_vhs_rows, rows, sizeof(unsigned) * words, hipMemcpyHostToDevice) != hipSuccess ||
            hipMalloc((void **) &c->d_vhs_dig, dig_rows * VHS_DIG_ROW) != hipSuccess ||
            hipMemcpy(c->d_vhs_dig, dig, dig_rows * VHS_DIG_ROW, hipMemcpyHostToDevice) != hipSuccess) {
            free(rows); free(dig);
            crthip_destroy(c);
            return CRTHIP_E_HIP;
        }
        free(rows); free(dig);
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
    if (c->aux_stream) {
        hipStreamSynchronize(c->aux_stream); hipStreamDestroy(c->aux_stream); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join);
        if (c->ev_mfork) hipEventDestroy(c->ev_mfork);
        if (c->ev_mjoin) hipEventDestroy(c->ev_mjoin);
        for (int k = 0; k < CRTHIP_MAX_CHUNKS; k++) hipEventDestroy(c->ev_chunk[k]);
    }
    if (c->d_jump16) hipFree(c->d_jump16);
    if (c->d_vhs_rows) hipFree(c->d_vhs_rows);
    if (c->d_vhs_dig) hipFree(c->d_vhs_dig);
    if (c->d_vhs_next) hipFree(c->d_vhs_next);
    if (c->d_seq) hipFree(c->d_seq);
    if (c->d_bloom) hipFree(c->d_bloom);
    if (c->table_stream) { hipStreamSynchronize(c->table_stream); hipStreamDestroy(c->table_stream); }
    if (c->d_nes_tab) hipFree(c->d_nes_tab);
    if (c->d_nes_tab_alt) hipFree(c->d_nes_tab_alt);
    if (c->d_skel) hipFree(c->d_skel);
    if (c->d_skel_alt) hipFree(c->d_skel_alt);
    for (int i = 0; i < c->n_retired; i++) if (c->retired[i]) hipFree(c->retired[i]);
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
    if (c->d_vhs_next) hipFree(c->d_vhs_next);
    c->d_analog = 0; c->d_inp = 0; c->d_lines = 0; c->d_vhs_next = 0; c->cap_fields = 0; c->last_n = 0;
    const size_t bytes = c->fstride * (size_t) n + 4096;
    /* inp[] of the fused path: padded lines + a scratch row per decoded line (crt_dev.h, sig_layout) -- about twice the flat field */
    const size_t inp_bytes = (c->fstride_pad > c->fstride ? c->fstride_pad : c->fstride) * (size_t) n + 4096;
    if (hipMalloc((void **) &c->d_inp, inp_bytes) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc inp", hipSuccess);
    if (hipMalloc((void **) &c->d_analog, bytes) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc analog", hipSuccess);
    if (hipMalloc((void **) &c->d_lines, sizeof(crthip_line) * (size_t) n * c->sd.lines) != hipSuccess)
        return set_err(c, CRTHIP_E_NOMEM, "hipMalloc lines", hipSuccess);
    if (c->system == CRTHIP_SYSTEM_NTSCVHS &&
        hipMalloc((void **) &c->d_vhs_next, sizeof(unsigned) * 32 * (size_t) n) != hipSuccess)
        return set_err(c, CRTHIP_E_NOMEM, "hipMalloc VHS histories", hipSuccess);
    /* the internal second stream (VHS noise pair, overlap chunks) exists from here on: a field-pass creates nothing (graph capture) */
    if (crt_ensure_aux(c) != CRTHIP_OK) return set_err(c, CRTHIP_E_HIP, "internal stream", hipGetLastError());
    HIPCHK(c, hipMemsetAsync(c->d_inp, 0, inp_bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_analog, 0, bytes, c->stream));
    if (!c->sd.nes_timing) {                    /* bloom builds (none with the NES timing): the decoder's sort scratch, 4 bytes per scanline */
        const int rc = crt_reserve_bloom(c, n);
        if (rc) return rc;
    }
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

/* The encoder contract.  The reference writes analog[(x + xo) + (y + yo) * HRES] (crt_ntsc.c:322) -- a FLAT index:
 * a rectangle that runs over the end of a line (xoffset = 4 in standard NTSC: 160 + 753 > 910) simply continues in
 * the next line's front porch, and the kernels do exactly the same (they work on flat indices too).  What is refused
 * is only what is undefined in the reference: samples outside analog[], and negative origins (negative carrier
 * table index, crt_ntsc.c:314). */
static int check_encoder(crthip_ctx *c, const crthip_params *p)
{
    if (c->system != CRTHIP_SYSTEM_NES && p->in_bpp == 0) return 1;   /* silent no-op, crt_ntsc.c:190-193 */
    if (p->col_step_hi == 0 && p->col_step_lo == 0)                    /* ceil(2^32 * w / destw) is never 0 */
        return set_err(c, CRTHIP_E_ARG, "params were finalized by an older library (no col_step): call crthip_params_finalize again", hipSuccess);
    const long long end = (long long) p->yo * c->sd.hres + p->xo + (long long) (p->desth - 1) * c->sd.hres + p->destw;
    if (p->xo < 0 || p->yo < 0 || p->destw <= 0 || p->desth <= 0 || p->destw > c->sd.hres || end > c->sd.input_size)
        return set_err(c, CRTHIP_E_ARG, "active rectangle leaves analog[] (xoffset/yoffset out of contract)", hipSuccess);
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
    rc = crt_run_encoder_prepare(c, p, false);
    if (rc) return rc;
    rc = crt_run_encoder(c, p, n, d_images, istride, d_analog, d_state, false, (p->flags & CRTHIP_F_NES_SETUP) != 0, true);
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
    if (p->flags & CRTHIP_F_NO_VSYNC) {                         /* CRT_DO_VSYNC 0: crt_core.c:323-341, before the noise */
        rc = crt_run_clean_vsync(c, n, d_analog, d_state);
        if (rc) return rc;
    }
    rc = crt_run_noise(c, p, n, d_analog, d_inp, d_state, true);
    HIPCHK(c, hipGetLastError());
    return rc;
}

int crthip_sync(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, crthip_state *d_state,
                crthip_line *d_lines)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (p->out_bpp == 0) return CRTHIP_OK;
    if (!d_inp || !d_state || !d_lines) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rc = crt_run_sync(c, p, n, d_inp, d_state, d_lines, 0);
    HIPCHK(c, hipGetLastError());
    return rc;
}

int crthip_decode(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp, const crthip_line *d_lines,
                  void *d_out, size_t ostride)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (p->out_bpp == 0) return CRTHIP_OK;
    if (!d_inp || !d_lines || !d_out) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    rc = crt_run_decode(c, p, n, d_inp, d_lines, d_out, ostride);
    HIPCHK(c, hipGetLastError());
    return rc;
}

/* The blob the sync chain of a FUSED launch sees: inp[] was produced here, from a clean analog[] and the encoder, so its
 * sample range is known and the decoder's no-low-cascade envelope can be as wide as that range allows
 * (crthip_params.loskip_wave_max; crt_setup_signal_range).  The stage-level crthip_sync keeps the any-signal bound. */
static crthip_params with_signal_envelope(const crthip_params *p)
{
    crthip_params q = *p;
    int lo, hi;
    crt_setup_signal_range(p, &lo, &hi);
    q.loskip_wave_max = crt_setup_loskip_bound(lo, hi);
    return q;
}

/* one chunk of a batch, fields [first, first+n), on the context's current stream.  `part` (bits): 1 = encoder + channel
 * noise, 4 = the sync chain up to the line table, 2 = the decoder (the rand()-noise VHS build and CRT_DO_VSYNC 0 run 1 and 4
 * as one unit under bit 1) */
static int fieldpass_chunk(crthip_ctx *c, const crthip_params *p, int enc, int first, int n, int part, const sig_layout &lay,
                           const void *d_images, size_t istride, void *d_out, size_t ostride, crthip_state *d_state)
{
    /* lay: where this field-pass keeps its signal (crt_fused_layout; the same for every chunk and part of a call) */
    const unsigned char *img = (const unsigned char *) d_images + (size_t) first * istride;
    unsigned char *out = (unsigned char *) d_out + (size_t) first * ostride;
    crthip_state *st = d_state + first;
    signed char *inp = c->d_inp + (size_t) first * lay.fstride;
    signed char *analog = c->d_analog + (size_t) first * c->fstride;
    crthip_line *ln = c->d_lines + (size_t) first * c->sd.lines;
    int rc = CRTHIP_OK;
    bool preset = false;
    if (part & 1) {
        const bool vhs_rand = c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE);
        if (vhs_rand || (p->flags & CRTHIP_F_NO_VSYNC)) {
            /* (CRT_DO_VSYNC 0 needs the clean field as well: its vertical sync search reads analog[], crt_core.c:323-341) */
            /* VHS noise follows the C library's rand() stream, not the LCG: the fused encoder (margins + active
             * rectangle = every sample of the field) runs with noise 0 into analog[], then the dedicated noise
             * kernels (which also produce rn) */
            if (enc == 0) {
                crthip_params clean = *p;
                clean.noise = 0;
                rc = crt_run_encoder(c, &clean, n, img, istride, analog, st, true, 1, true);
            } else {
                hipMemsetAsync(analog, 0, c->fstride * (size_t) n, c->stream);   /* crt_modulate refused the format */
            }
            if (rc) return rc;
            if (p->out_bpp == 0) return CRTHIP_OK;
            unsigned *saved = c->d_vhs_hist;
            if (vhs_rand) c->d_vhs_hist = saved + (size_t) first * 32;
            rc = crthip_noise(c, p, n, analog, inp, st);
            c->d_vhs_hist = saved;
            if (rc) return rc;
            rc = crt_run_sync(c, p, n, inp, st, ln, 0);
            part &= ~4;
        } else {
            if (enc == 0) {
                /* the encoder writes the noisy field straight into inp[]; analog[] is never materialised.  What crt_modulate
                 * leaves in the state (ccf preset, M6) is applied by the sync chain's own waves when it follows at once
                 * (k_hsync_wave, preset_ccf): one launch less per field-pass */
                preset = (part & 4) && p->out_bpp != 0 && c->system != CRTHIP_SYSTEM_NTSCVHS;   /* (VHS also resets hsync there) */
                rc = crt_run_encoder(c, p, n, img, istride, inp, st, true, 1, !preset, &lay);
            } else {
                /* invalid input format: crt_modulate is a no-op, the decoder sees a clean field + noise
                 * (rn is advanced by k_vsync below) */
                hipMemsetAsync(analog, 0, c->fstride * (size_t) n, c->stream);
                rc = crt_run_noise(c, p, n, analog, inp, st, false);
            }
        }
        if (rc) return rc;
    }
    if ((part & 4) && p->out_bpp != 0) {
        const crthip_params q = enc == 0 ? with_signal_envelope(p) : *p;
        rc = crt_run_sync(c, &q, n, inp, st, ln, 1, preset ? 1 : 0, &lay);
        if (rc) return rc;
    }
    if ((part & 2) && p->out_bpp != 0) rc = crt_run_decode(c, p, n, inp, ln, out, ostride, lay.fstride);
    return rc;
}

int crthip_fieldpass(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                     void *d_out, size_t ostride, crthip_state *d_state)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (!d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    if (c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE) && !c->d_vhs_hist)
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
    if (enc == 0) {
        /* tables shared by all chunks are (re)built here, on the caller's stream, before any fork */
        rc = crt_run_encoder_prepare(c, p, true);
        if (rc) return rc;
    }
    /* the signal between this call's encoder and decoder: padded lines where the fused LCG-noise encoder writes it (crt_dev.h) */
    sig_layout lay;
    const bool vhs_rand_path = c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE);
    if (enc != 0 || vhs_rand_path || !crt_fused_layout(c, p, n, &lay)) {
        lay.pitch = c->sd.hres; lay.shift = 0; lay.padv = 0; lay.wrap = 0; lay.fstride = c->fstride;
    }
    c->last_lay = lay;
    c->last_n = n;
    /* 0 = automatic.  The idea: run the vector / latency bound encoder + sync chain of one chunk under the HBM-write
     * bound decoder of the previous one.  Measured on MI355X (profiles/r02_overlap_sweep.txt): at 1080p the encoder's
     * image reads and the decoder's picture writes already saturate what HBM delivers for this access mix, running
     * them side by side is slower than one after the other (2 chunks -3 %, 8 chunks -19 %); at 640x480 every kernel is
     * vector bound and chunks only add latency.  So automatic means: one chunk. */
    int want_chunks = c->overlap_chunks;
    if (want_chunks == 0) want_chunks = 1;       /* measured: no configuration gains (profiles/r02_overlap_sweep.txt); round 3 also
                                                    tried only the sync chain of the second half of the batch under the decoder
                                                    of the first: 3-4 % slower at 640x480 and 1080p alike (profiles/r03_1080p_experiments.txt) */
    const int nchunks = (want_chunks > 1 && n >= 256 * want_chunks && !c->prof) ? want_chunks : 1;
    if (nchunks == 1) {
        rc = fieldpass_chunk(c, p, enc, 0, n, 7, lay, d_images, istride, d_out, ostride, d_state);
    } else {
        /* Two-stage software pipeline over the chunks: the encoder + sync chain of ALL chunks run back to back on an
         * internal stream, the decoders on the caller's stream, decoder k waiting for the event behind sync chain k.
         * While decoder k streams its picture out (HBM-write bound at 1080p), the vector-bound encoder of chunk k+1
         * and the latency-bound sync chain run beside it.  (Alternating whole chunks between two streams, the first
         * version of this, ends up running decoder next to decoder: profiles/r02_overlap_timeline_1080p.txt.) */
        if (crt_ensure_aux(c) != CRTHIP_OK) return set_err(c, CRTHIP_E_HIP, "internal stream", hipGetLastError());
        hipStream_t main_stream = c->stream;
        HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        const int per = ((n + nchunks - 1) / nchunks + 3) & ~3;
        c->stream = c->aux_stream;
        int used = 0;
        for (int k = 0, first = 0; first < n && rc == CRTHIP_OK; k++, first += per) {
            const int cnt = n - first < per ? n - first : per;
            rc = fieldpass_chunk(c, p, enc, first, cnt, 1 | 4, lay, d_images, istride, d_out, ostride, d_state);
            if (rc == CRTHIP_OK && hipEventRecord(c->ev_chunk[k], c->aux_stream) != hipSuccess) rc = CRTHIP_E_HIP;
            used = k + 1;
        }
        c->stream = main_stream;
        for (int k = 0, first = 0; k < used && rc == CRTHIP_OK; k++, first += per) {
            const int cnt = n - first < per ? n - first : per;
            HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_chunk[k], 0));
            rc = fieldpass_chunk(c, p, enc, first, cnt, 2, lay, d_images, istride, d_out, ostride, d_state);
        }
        /* the caller's stream has waited for every event of the internal stream: nothing is left running there */
    }
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

unsigned crthip_table_generation(const crthip_ctx *c) { return c ? c->table_gen : 0u; }

int crthip_set_signal_tile(crthip_ctx *c, int dwords)
{
    if (!c || (dwords != 0 && dwords != 16 && dwords != 32 && dwords != 64)) return CRTHIP_E_ARG;
    c->sig_tile_env = dwords;
    return CRTHIP_OK;
}

int crthip_signal_layout_query(const crthip_params *p, int n_fields, int shape, int layout[4], size_t *field_stride)
{
    if (!p || p->finalized != CRTHIP_PARAMS_MAGIC || n_fields <= 0 || shape < 0 || shape > 2) return CRTHIP_E_ARG;
    struct crt_sysdef sd;
    if (crt_sysdef_get(&sd, p->system, p->chroma_pattern) != CRTHIP_OK) return CRTHIP_E_ARG;
    sig_layout lay;
    const bool padded = layout_rule(p->system, p->chroma_pattern, sd, crthip_field_stride(p->system, p->chroma_pattern), p, n_fields, shape, 1, &lay);
    if (layout) { layout[0] = lay.pitch; layout[1] = lay.shift; layout[2] = lay.padv; layout[3] = lay.wrap; }
    if (field_stride) *field_stride = lay.fstride;
    return padded ? 1 : 0;
}

int crthip_set_signal_layout(crthip_ctx *c, int padded)
{
    if (!c || (padded != 0 && padded != 1)) return CRTHIP_E_ARG;
    c->sig_pad = padded;
    return CRTHIP_OK;
}

int crthip_fieldpass_signal(crthip_ctx *c, int n, signed char *d_inp_flat, int *padded)
{
    if (!c || n <= 0 || !d_inp_flat) return CRTHIP_E_ARG;
    if (n > c->last_n || !c->d_inp) return set_err(c, CRTHIP_E_ARG, "crthip_fieldpass_signal: no field-pass of that many fields went through this context", hipSuccess);
    HIPCHK(c, hipSetDevice(c->device));
    if (padded) *padded = c->last_lay.pitch != c->sd.hres;
    const int rc = crt_run_unpad(c, n, &c->last_lay, c->d_inp, d_inp_flat);
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

int crthip_set_wide_lpw(crthip_ctx *c, int lpw)
{
    if (!c || (lpw != 0 && lpw != 8 && lpw != 16)) return CRTHIP_E_ARG;
    c->wide_lpw_env = lpw;
    return CRTHIP_OK;
}

int crthip_set_pixel_tile(crthip_ctx *c, int px)
{
    if (!c || (px != 0 && px != 16 && px != 32)) return CRTHIP_E_ARG;
    c->px_tile = px;
    c->ac_tile = px;
    return CRTHIP_OK;
}

/* scratch of the sequence phases: guess[n] (int2), changed flag, owner[n][outh] (u8), latest[n][outh] (int) */
struct SeqScratch { int2 *guess; int *changed; unsigned char *owner; int *latest; };
static int seq_scratch(crthip_ctx *c, int n, int outh, SeqScratch *sc)
{
    const size_t need = sizeof(int2) * (size_t) n + 256 + (size_t) n * outh + 256 + sizeof(int) * (size_t) n * outh + 256;
    if (need > c->seq_cap) {
        if (c->d_seq) hipFree(c->d_seq);
        c->d_seq = 0; c->seq_cap = 0; c->seq_guess_n = 0;
        if (hipMalloc((void **) &c->d_seq, need) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc sequence scratch", hipSuccess);
        c->seq_cap = need;
    }
    sc->guess = (int2 *) c->d_seq;
    sc->changed = (int *) (c->d_seq + sizeof(int2) * (size_t) n);
    sc->owner = c->d_seq + sizeof(int2) * (size_t) n + 256;
    sc->latest = (int *) (sc->owner + (((size_t) n * outh + 255) & ~(size_t) 255));
    return CRTHIP_OK;
}

static int seq_check(crthip_ctx *c, const crthip_params *p, int n)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE) && !c->d_vhs_hist)
        return set_err(c, CRTHIP_E_ARG, "VHS: no generator histories bound (crthip_vhs_bind_history)", hipSuccess);
    if (p->flags & CRTHIP_F_NO_VSYNC)
        return set_err(c, CRTHIP_E_ARG, "sequence mode is not available for the CRT_DO_VSYNC 0 variant (its sync search reads the clean field)", hipSuccess);
    if (p->blend && (unsigned) p->outh + p->v_fac < (unsigned) c->sd.lines)
        return set_err(c, CRTHIP_E_ARG, "sequence mode with blend needs outh + v_fac >= CRT_LINES (one line per output row)", hipSuccess);
    if (p->out_bpp == 0) return set_err(c, CRTHIP_E_ARG, "sequence mode: unknown output pixel format", hipSuccess);
    int enc = check_encoder(c, p);
    if (enc != 0) return enc < 0 ? enc : set_err(c, CRTHIP_E_ARG, "sequence mode: unknown input pixel format", hipSuccess);
    return CRTHIP_OK;
}

/* phase 1: the noise generator of every field in closed form, all fields encoded in parallel (noise fused) */
int crthip_seq_encode(crthip_ctx *c, const crthip_params *p, int n, int first_index, int rn0,
                      const void *d_images, size_t istride, crthip_state *d_state)
{
    int rc = seq_check(c, p, n);
    if (rc) return rc;
    if (!d_images || !d_state || first_index < 0) return CRTHIP_E_ARG;
    const bool vhs = c->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE);
    if (vhs && first_index != 0 && !c->vhs_prechained)
        return set_err(c, CRTHIP_E_ARG, "VHS: a video shares ONE rand() stream; its fields cannot start in the middle (first_index != 0)", hipSuccess);
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->cap_fields) {
        rc = crthip_reserve(c, n);
        if (rc) return rc;
    }
    c->seq_guess_n = 0;                                    /* a new video: no warm start for the sync chain */
    /* sequence mode keeps the reference's flat signal layout (its phases are separate calls over the same workspace) */
    c->last_lay.pitch = c->sd.hres; c->last_lay.shift = 0; c->last_lay.padv = 0; c->last_lay.wrap = 0; c->last_lay.fstride = c->fstride;
    c->last_n = n;
    rc = crt_run_encoder_prepare(c, p, true);
    if (rc) return rc;
    if (vhs) {
        /* the fields share one rand() stream: run the chain ahead (k_vhs_chain: hist[k] = generator at the start
         * of field k, aberration heights drawn in-stream if asked), after which every field is independent;
         * then clean encode -> rand() noise (which also leaves rn), as in crthip_fieldpass */
        if (!c->vhs_prechained) {
            rc = crt_run_vhs_chain(c, n, d_state, (p->flags & CRTHIP_F_VHS_DRAW_ABERRATION) != 0);
            if (rc) return rc;
        }
        crthip_params clean = *p;
        clean.noise = 0;
        rc = crt_run_encoder(c, &clean, n, d_images, istride, c->d_analog, d_state, true, 1, false);
        if (rc) return rc;
        rc = crt_run_noise(c, p, n, c->d_analog, c->d_inp, d_state, false);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_seq_rn, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, d_state, c->whole_field,
                           (unsigned) first_index, (unsigned) rn0);
        /* encode every field (noise fused), ccf presets */
        rc = crt_run_encoder(c, p, n, d_images, istride, c->d_inp, d_state, true, 1, false);
        if (rc) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

/* phase 2: the sync chain.  Field k starts from field k-1's final (hsync, vsync), field 0 from (hsync_in, vsync_in);
 * solved as a fixed point over ALL fields in parallel (see the comment at the top).  Called again with another
 * incoming pair (a video cut over several shards: the predecessor's final state became known) it restarts from the
 * finals of the previous call, so only the fields whose state really depends on the incoming pair are recomputed
 * in more than one pass. */
int crthip_seq_sync(crthip_ctx *c, const crthip_params *p, int n, crthip_state *d_state, int hsync_in, int vsync_in,
                    int *hsync_out, int *vsync_out, int *passes_out)
{
    int rc = seq_check(c, p, n);
    if (rc) return rc;
    if (!d_state || n > c->cap_fields) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    SeqScratch sc;
    rc = seq_scratch(c, n, p->outh, &sc);
    if (rc) return rc;
    const dim3 gn((n + 63) / 64), b64(64);
    if (c->seq_guess_n != n) {
        /* first guess: nobody's sync state moves */
        int2 *h = (int2 *) malloc(sizeof(int2) * (size_t) n);
        if (!h) return CRTHIP_E_NOMEM;
        for (int k = 0; k < n; k++) h[k] = make_int2(hsync_in, vsync_in);
        hipError_t e = hipMemcpyAsync(sc.guess, h, sizeof(int2) * (size_t) n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        free(h);
        HIPCHK(c, e);
        c->seq_guess_n = n;
    }
    int passes = 0;
    const crthip_params q = with_signal_envelope(p);                  /* inp[] comes from crthip_seq_encode */
    for (;;) {
        passes++;
        HIPCHK(c, hipMemsetAsync(sc.changed, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_seq_load, gn, b64, 0, c->stream, n, d_state, sc.guess, make_int2(hsync_in, vsync_in));
        rc = crt_run_encoder_state(c, p, n, d_state);                 /* ccf preset, crt_ntsc.c:325-329 */
        if (rc) return rc;
        rc = crt_run_sync(c, &q, n, c->d_inp, d_state, c->d_lines, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_seq_compare, gn, b64, 0, c->stream, n, d_state, sc.guess, sc.changed);
        int flag = 0;
        HIPCHK(c, hipMemcpyAsync(&flag, sc.changed, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!flag) break;
        if (passes > n + 1)               /* after pass k fields 0 .. k-1 are final: cannot happen (say so instead of returning a half-converged chain) */
            return set_err(c, CRTHIP_E_HIP, "crthip_seq_sync: the sync chain over the fields did not converge", hipSuccess);
    }
    if (passes_out) *passes_out = passes;
    if (hsync_out || vsync_out) {
        int hv[2];
        HIPCHK(c, hipMemcpyAsync(hv, &d_state[n - 1].hsync, sizeof(hv), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (hsync_out) *hsync_out = hv[0];
        if (vsync_out) *vsync_out = hv[1];
    }
    return CRTHIP_OK;
}

/* phase 3: rn after each field, all fields decoded in parallel (without blend: phase 4 folds the fields) */
int crthip_seq_decode(crthip_ctx *c, const crthip_params *p, int n, void *d_out, size_t ostride, crthip_state *d_state)
{
    int rc = seq_check(c, p, n);
    if (rc) return rc;
    if (!d_out || !d_state || n > c->cap_fields) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->system != CRTHIP_SYSTEM_NTSCVHS || (p->flags & CRTHIP_F_VHS_LCG_NOISE)) crt_run_advance_rn(c, n, d_state);
    crthip_params pb = *p;
    pb.blend = 0;
    rc = crt_run_decode(c, &pb, n, c->d_inp, c->d_lines, d_out, ostride);
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

/* phase 4: image k = the single output buffer as it stands after field k.  d_out_init = the buffer before the shard's
 * first field (NULL = zeros).  Without blend: rows a field does not write come from the latest earlier field that
 * wrote them, or from d_out_init; patch_only != 0 revisits only the latter (the images were woven before with a
 * placeholder init: a later shard of a video whose predecessor's last picture arrives late).  With blend: the
 * recurrence over the fields, one pass per field in order (patch_only is not available). */
int crthip_seq_weave(crthip_ctx *c, const crthip_params *p, int n, void *d_out, size_t ostride, const void *d_out_init,
                     int patch_only)
{
    int rc = seq_check(c, p, n);
    if (rc) return rc;
    if (!d_out || n > c->cap_fields || (p->blend && patch_only)) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    SeqScratch sc;
    rc = seq_scratch(c, n, p->outh, &sc);
    if (rc) return rc;
    const int outh = p->outh;
    const size_t pitch = (size_t) p->outw * p->out_bpp;
    if (p->blend) {
        HIPCHK(c, hipMemsetAsync(sc.latest, 0xff, sizeof(int) * (size_t) n * outh, c->stream));       /* -1 everywhere */
        hipLaunchKernelGGL(k_seq_rowsrc, dim3((n * c->sd.lines + 255) / 256), dim3(256), 0, c->stream, n, c->sd.lines, outh, c->d_lines, sc.latest);
        const int fmt = p->out_format;
        const unsigned alpha = p->out_bpp == 3 ? 0u : ((fmt == CRTHIP_FMT_ARGB || fmt == CRTHIP_FMT_ABGR) ? 0x000000ffu : 0xff000000u);
        for (int k = 0; k < n; k++)
            hipLaunchKernelGGL(k_seq_blend_step, dim3((unsigned) outh), dim3(256), 0, c->stream, k, outh, pitch,
                               (unsigned char *) d_out, ostride, (const unsigned char *) d_out_init, sc.latest, alpha);
        HIPCHK(c, hipGetLastError());
        return CRTHIP_OK;
    }
    if (!patch_only) {
        HIPCHK(c, hipMemsetAsync(sc.owner, 0, (size_t) n * outh, c->stream));
        hipLaunchKernelGGL(k_seq_rows, dim3((n * c->sd.lines + 255) / 256), dim3(256), 0, c->stream, n, c->sd.lines, outh, c->d_lines, sc.owner);
        hipLaunchKernelGGL(k_seq_latest, dim3((outh + 63) / 64), dim3(64), 0, c->stream, n, outh, sc.owner, sc.latest);
    }
    hipLaunchKernelGGL(k_seq_weave, dim3((unsigned) n * (unsigned) outh), dim3(256), 0, c->stream, n, outh, pitch,
                       (unsigned char *) d_out, ostride, (const unsigned char *) d_out_init, sc.latest, patch_only);
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

int crthip_vhs_chain(crthip_ctx *c, int n, crthip_state *d_state, int draw_aberration)
{
    if (!c || n <= 0 || !d_state || c->system != CRTHIP_SYSTEM_NTSCVHS) return CRTHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = crt_run_vhs_chain(c, n, d_state, draw_aberration);
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return CRTHIP_OK;
}

int crthip_seq_vhs_prechained(crthip_ctx *c, int on)
{
    if (!c || c->system != CRTHIP_SYSTEM_NTSCVHS) return CRTHIP_E_ARG;
    c->vhs_prechained = on != 0;
    return CRTHIP_OK;
}

int crthip_sequence(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                    void *d_out, size_t ostride, const void *d_out_init, crthip_state *d_state, int *passes_out)
{
    int rc = check_params(c, p, n);
    if (rc) return rc;
    if (!d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    if (p->out_bpp == 0) return CRTHIP_OK;                /* unknown output format: like crt_demodulate, nothing happens (crt_core.c:312-315) */
    HIPCHK(c, hipSetDevice(c->device));
    crthip_state first;
    HIPCHK(c, hipMemcpyAsync(&first, d_state, sizeof(first), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = crthip_seq_encode(c, p, n, 0, first.rn, d_images, istride, d_state);
    if (rc) return rc;
    rc = crthip_seq_sync(c, p, n, d_state, first.hsync, first.vsync, nullptr, nullptr, passes_out);
    if (rc) return rc;
    rc = crthip_seq_decode(c, p, n, d_out, ostride, d_state);
    if (rc) return rc;
    return crthip_seq_weave(c, p, n, d_out, ostride, d_out_init, 0);
}

int crthip_set_shape(crthip_ctx *c, int shape)
{
    if (!c || shape < 0 || shape > 2) return CRTHIP_E_ARG;
    c->shape = shape;
    return CRTHIP_OK;
}

int crthip_set_overlap(crthip_ctx *c, int chunks)
{
    if (!c || chunks < 0 || chunks > CRTHIP_MAX_CHUNKS) return CRTHIP_E_ARG;
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

int crthip_host_register(crthip_ctx *c, void *h, size_t bytes)
{
    if (!c || !h || !bytes) return CRTHIP_E_ARG;
    if (hipHostRegister(h, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void) hipGetLastError();
        return CRTHIP_E_HIP;
    }
    return CRTHIP_OK;
}

int crthip_host_unregister(crthip_ctx *c, void *h)
{
    if (!c || !h) return CRTHIP_E_ARG;
    if (hipHostUnregister(h) != hipSuccess) {
        (void) hipGetLastError();
        return CRTHIP_E_HIP;
    }
    return CRTHIP_OK;
}

int crthip_memset(crthip_ctx *c, void *d, int value, size_t bytes)
{
    if (!c) return CRTHIP_E_ARG;
    HIPCHK(c, hipMemsetAsync(d, value, bytes, c->stream));
    return CRTHIP_OK;
}

}  /* extern "C" */
