/* crt_encode.hip -- crt_modulate on the GPU: skeleton (M4), active video (M5), ccf preset (M6).  See crt_dev.h. */
#include "crt_dev.h"

/* ------------------------------------------------------------------------- */
/* M4: blanking / sync / burst skeleton                                        */
/* ------------------------------------------------------------------------- */
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

/* value of skeleton sample (line n, column t) and whether crt_modulate writes it.
 * RGB systems: crt_ntsc.c:205-252 (+ crt_ntscvhs.c:234-238), crt_snes.c:203-243, crt_template.c, crt_pv1k.c:190-231;
 * NES timing: crt_nes.c:81-104,173-178, crt_nesrgb.c:24-46,104-109 */
template <class S>
__device__ __forceinline__ bool
skeleton(const crthip_params &P, int n, int t, int field, int frame, int aux, bool nes_setup, int &val)
{
    const int row = carrier_row<S>(n, field, frame, aux);
    const int tc = S::CCS == 4 ? (t & 3) : t % S::CCS;
    if constexpr (S::NES_TIMING) {
        bool written = nes_setup;
        val = S::BLANK;
        if (t >= S::SYNC_BEG && t < (n >= 259 ? S::VS_SEP_END : S::BW_BEG)) val = S::SYNC;
        if (n >= P.yo && n < P.yo + S::LINES && t >= S::CB_BEG && t < S::CB_BEG + S::CB_LEN) {
            int cb = P.burst[row][tc];
            val = (int) (signed char) ((S::BLANK + cb * S::BURST) >> 5);
            written = true;
        }
        if constexpr (S::IS_NES) {
            /* NES_BORDER 1 (crt_nes.c:138-160): the border colour from LAV_BEG to the end of lines TOP .. BOT + 2, written by
             * every crt_modulate before the picture (which then covers its own rectangle) */
            if ((P.flags & CRTHIP_F_NES_BORDER) && n >= S::TOP && n <= S::BOT + 2 && t >= S::LAV_BEG) {
                const int phase = 4 * ((n + (row - n % S::VPER)) % 3) + 6 + 3 * (t - S::LAV_BEG);   /* phasetab[(n + dco) % 3] + 6 */
                const int p = t == S::LAV_BEG ? 0xf0 : P.nes_border_color;
                int ire = S::BLACK + P.black_point;
                ire += ppu_level(p, phase + 0);
                ire += ppu_level(p, phase + 1);
                ire += ppu_level(p, phase + 2);
                ire += ppu_level(p, phase + 3);
                val = (int) (signed char) ((ire * P.white_point / 100) >> 12);
                written = true;
            }
        }
        return written;
    } else {
        if ((n >= S::EQU_A_LO && n <= S::EQU_A_HI) || (n >= S::EQU_B_LO && n <= S::EQU_B_HI)) {   /* equalising pulses */
            val = (t < 4 * S::HRES / 100 || (t >= 50 * S::HRES / 100 && t < 54 * S::HRES / 100)) ? S::SYNC : S::BLANK;
            return true;
        }
        if (n >= S::VS_LO && n <= S::VS_HI) {          /* vertical sync */
            int a = ((S::VS_BY_FIELD && field == 1) ? 4 : 46) * S::HRES / 100;
            val = (t < a || (t >= 50 * S::HRES / 100 && t < 96 * S::HRES / 100)) ? S::SYNC : S::BLANK;
            return true;
        }
        if (t >= S::AV_BEG) {                          /* active part: only cleared above CRT_TOP */
            val = S::BLANK;
            return n < S::TOP;
        }
        val = S::BLANK;
        if (t >= S::SYNC_BEG && t < S::BW_BEG && (!S::IS_VHS || n < S::VRES - aux)) val = S::SYNC;
        if (t >= S::CB_BEG && t < S::CB_BEG + S::CB_LEN) {
            int cb = P.burst[row][tc];
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
        const bool wr = skeleton<S>(P, line, t, field, st.frame & 1, st.aux, nes_setup != 0, v);
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

/* 16 bytes of the input image.  Wide tiles (32 pixels = whole 128-byte lines per row) stream past the caches:
 * the image is read exactly once.  Narrow tiles fetch a line in two halves at different times, and the second
 * half must still find it in L2 (measured: streaming loads cost +35 % there). */
template <int ACT> __device__ __forceinline__ v4i image_piece(unsigned long long a)
{
    return ACT == 32 ? gload16u_nt(a) : gload16u(a);
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

/* the same without the permute: colour bytes moved to bits 0-23 (one shift for the two alpha-first orders) and the
 * RGB->YIQ coefficients of crt_ntsc.c:307-309 ordered by byte position instead */
__device__ __forceinline__ bool format_alpha_first(int format) { return format == CRTHIP_FMT_ARGB || format == CRTHIP_FMT_ABGR; }
__device__ __forceinline__ bool format_blue_low(int format)
{
    return format == CRTHIP_FMT_BGR || format == CRTHIP_FMT_BGRA || format == CRTHIP_FMT_ABGR;
}

/* Cooperative tile I/O of the lane-per-row encoders (see the comment above k_decode): the wave moves 64 rows x ACT
 * dwords between global memory and LDS with row-contiguous 16-byte pieces; each lane then reads / writes only its
 * own tile row.  Input side: `have` = tile in LDS, stage[] = tile have+1 in flight / in registers.  A piece that
 * would run past the row end is moved back to the row's last 16 bytes, so nothing beyond the row is touched
 * (row_bytes >= 16); moved-back pieces of several lanes overlap and carry identical bytes. */
/* OT: dwords per row of the OUTPUT (sample) tile.  The wide-input encoder (ACT = 32: whole 128-byte image lines per
 * piece group) keeps the 16-dword sample tile of the narrow one: 64-byte sample pieces either way, 4 KB less LDS per
 * wave (11 instead of 8 waves per CU behind the image fetches). */
/* (r6) OL: dwords per row of the sample tile that are IN LDS.  OL = OT: all of it (rounds 1-5).  OL = OT / 2 ("staged"): the lane
 * pulls its first half back into OL registers when that half is complete, the second half reuses the LDS rows, and the drain goes
 * through the same LDS bytes in two passes of 32 rows x OT dwords -- the signal still leaves in 4 * OT-byte pieces, but the wave
 * holds half the LDS: 8.4 instead of 12.7 KB beside the 16-dword image tile (16 waves per CU instead of 12; the register file has the
 * room: 77 + 16 + 16 transient of 128), 16.9 instead of 25 KB beside the 32-dword one.  k_active is not bound by its vector unit (61 %
 * busy at 11 waves per CU, profiles/r06_headline_sq_counters.json), but waves are not everything either: staging pays for the 64-dword
 * tile only (launch_active). */
template <int ACT, int OT = ACT, int OL = OT>
struct RowTiles {
    static constexpr int TILE = ACT, PIECES = ACT / 4;                        /* 16-byte pieces per tile row */
    static constexpr bool STAGED = OL != OT;
    static_assert(OL == OT || 2 * OL == OT, "the staged sample tile keeps exactly one half in registers");
    using TP = TileRows<ACT>;                                                 /* row addressing of the two tiles (crt_dev.h): conflict-free */
    using TO = TileRows<OL>;                                                  /*   lane-per-row and cooperatively                          */
    static_assert(!STAGED || TO::DWORDS >= 32 * (OT + 1), "the two-pass drain lays 32 rows of OT + 1 dwords over the tile");
    static constexpr int ROWS = 64 / PIECES;                                  /* rows per load instruction */
    static constexpr int OTILE = OT, OPIECES = OL / 4, OROWS = 64 / OPIECES;  /* (pieces of the tile in LDS) */
    unsigned r0[STAGED ? OL : 1];                                             /* staged: the first half of the current OT-dword block */
    unsigned *s_pix, *s_out;
    unsigned long long my_src, my_dst;                  /* my own row's image row / signal row (0: none); other rows' by lane_u64 */
    int lane, prow, piece, oprow, opiece;
    int row_bytes, last_tile, have;
    bool shift8;                 /* alpha-first pixel formats: the colour bytes are moved to bits 0-23 once per tile, here */
    v4i stage[PIECES];

    __device__ __forceinline__ void init(unsigned *pix, unsigned *out, unsigned long long src, unsigned long long dst, int lane_, int row_bytes_)
    {
        s_pix = pix; s_out = out; my_src = src; my_dst = dst;
        lane = lane_; prow = lane_ / PIECES; piece = lane_ % PIECES;
        oprow = lane_ / OPIECES; opiece = lane_ % OPIECES;
        shift8 = false;
        row_bytes = row_bytes_;
        last_tile = (row_bytes_ - 1) / (TILE * 4);
        have = 0;
    }
    /* (r6) the row pointers of the 64 rows live in their lanes' registers and travel by ds_bpermute_b32 (the LDS crossbar, no LDS
     * memory) where rounds 1-5 kept two 64-entry tables in LDS: 1 KB less per wave -- 12 672 instead of 13 696 B for the 16-dword
     * image / 32-dword sample tile pair, i.e. 12 instead of 11 waves per CU (all lanes are active wherever this is called) */
    __device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int src_lane) const
    {
        const int a = src_lane << 2;
        const unsigned lo = (unsigned) __builtin_amdgcn_ds_bpermute(a, (int) (unsigned) v);
        const unsigned hi = (unsigned) __builtin_amdgcn_ds_bpermute(a, (int) (unsigned) (v >> 32));
        return ((unsigned long long) hi << 32) | lo;
    }
    __device__ __forceinline__ int piece_offset(int tile) const
    {
        const int off = tile * (TILE * 4) + piece * 16;
        return off > row_bytes - 16 ? row_bytes - 16 : off;
    }
    __device__ __forceinline__ void fetch(int tile)
    {
        const int off = piece_offset(tile);
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
#if defined(ENC_DBG) && ENC_DBG == 2          /* measurement build: no image loads (tools/sessions; never shipped) */
            stage[i] = v4i{ off + i, lane, tile, 7 };
#else
            stage[i] = image_piece<ACT>(lane_u64(my_src, i * ROWS + prow) + off);
#endif
        }
    }
    __device__ __forceinline__ void stash(int tile)
    {
        /* dword index inside the tile where my (possibly moved-back) piece belongs; may be negative then */
        const int dw0 = (piece_offset(tile) - tile * (TILE * 4)) >> 2;
        __syncthreads();
        if (shift8) {                                       /* wave-uniform */
#pragma unroll
            for (int i = 0; i < PIECES; i++) {
                stage[i].x = (int) ((unsigned) stage[i].x >> 8); stage[i].y = (int) ((unsigned) stage[i].y >> 8);
                stage[i].z = (int) ((unsigned) stage[i].z >> 8); stage[i].w = (int) ((unsigned) stage[i].w >> 8);
            }
        }
        if (tile < last_tile || (row_bytes & (TILE * 4 - 1)) == 0) {
            /* every piece sits where it belongs (only the row's LAST tile can hold moved-back pieces): plain stores, no
             * per-dword predicates (16 execute-mask round trips per tile as compiled before) */
#pragma unroll
            for (int i = 0; i < PIECES; i++) {
                unsigned *d = s_pix + TP::row(i * ROWS + prow) + piece * 4;
                d[0] = (unsigned) stage[i].x; d[1] = (unsigned) stage[i].y; d[2] = (unsigned) stage[i].z; d[3] = (unsigned) stage[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PIECES; i++) {
                unsigned *d = s_pix + TP::row(i * ROWS + prow);
                if (dw0 + 0 >= 0) d[dw0 + 0] = (unsigned) stage[i].x;
                if (dw0 + 1 >= 0) d[dw0 + 1] = (unsigned) stage[i].y;
                if (dw0 + 2 >= 0) d[dw0 + 2] = (unsigned) stage[i].z;
                if (dw0 + 3 >= 0) d[dw0 + 3] = (unsigned) stage[i].w;
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ void start()
    {
        fetch(0);
        stash(0);
        if (last_tile > 0) fetch(1);
    }
    /* make tile `need` (wave-uniform) the resident one */
    __device__ __forceinline__ void want(int need)
    {
        if (need != have) {
            if (need != have + 1) fetch(need);          /* only when the source is more than TILE x wider than the line */
            stash(need);
            have = need;
            if (need < last_tile) fetch(need + 1);
        }
    }
    __device__ __forceinline__ unsigned pixel_dword(int idx) const { return s_pix[TP::row(lane) + (idx & (TILE - 1))]; }
    __device__ __forceinline__ void put(int g, unsigned pack) { s_out[TO::row(lane) + (g & (OL - 1))] = pack; }
    /* sample k of group g as one LDS byte store: no packing arithmetic on the vector unit */
    __device__ __forceinline__ void put_byte(int g, int k, int v)
    {
        ((unsigned char *) s_out)[(TO::row(lane) + (g & (OL - 1))) * 4 + k] = (unsigned char) v;
    }
    /* drain the sample tile: dwords [g0, g0+ng) of every row = samples [4*g0, ...) clipped to destw */
    __device__ __forceinline__ void drain(int g0, int ng, int destw)
    {
        __syncthreads();
        const int first = (g0 + opiece * 4) * 4;              /* first sample of my piece */
        const int nbytes = destw - first < 16 ? destw - first : 16;
#pragma unroll 2
        for (int i = 0; i < OPIECES; i++) {
            const int r = i * OROWS + oprow;
            const unsigned long long d = lane_u64(my_dst, r);
            if (d != 0 && opiece * 4 < ng && nbytes > 0) {
                const unsigned *sp = s_out + TO::row(r) + opiece * 4;
                v4i o; o.x = (int) sp[0]; o.y = (int) sp[1]; o.z = (int) sp[2]; o.w = (int) sp[3];
#if defined(ENC_DBG) && ENC_DBG == 1          /* measurement build: no signal stores (-128 never leaves the clamp) */
                if (o.x != (int) 0x80808080) continue;
#endif
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
    }
    /* (r6) padded signal lines (crt_dev.h, sig_layout): the row was stored contiguously, its last `wrapn` samples behind the line's
     * end -- where they are the COPY of the next line's head.  Their home, the head of the next line, is `delta` = PITCH - HRES bytes
     * further on.  Called after the last drain (which ends with a barrier): the samples still sit in my own tile row -- groups of the
     * row's last 16 samples are never overwritten by the (shorter) last tile. */
    __device__ __forceinline__ void wrap_home(int destw, int wrapn, int delta)
    {
        const unsigned long long d = my_dst;
        if (d == 0) return;
        for (int x = destw - wrapn; x < destw; x++) {
            const unsigned dw = s_out[TO::row(lane) + ((x >> 2) & (OL - 1))];
            gstore8(d + (unsigned) (x + delta), dw >> (8 * (x & 3)));
        }
    }
    /* staged tile: the block's dwords [0, OL) are in r0, [OL, ng) in LDS.  Second half to registers, then twice: 32 lanes lay their
     * row out at OT + 1 dwords per row (conflict-free both ways), the wave stores those 32 rows in 4 * OT-byte pieces */
    __device__ __forceinline__ void drain_staged(int g0, int ng, int destw)
    {
        unsigned r1[OL];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OL; j++) r1[j] = s_out[TO::row(lane) + j];
        __syncthreads();
        constexpr int DSTRIDE = OT + 1, OP = OT / 4, ORW = 64 / OP;
        const int orow = lane / OP, op = lane % OP;
        const int first = (g0 + op * 4) * 4;
        const int nbytes = destw - first < 16 ? destw - first : 16;
        for (int ph = 0; ph < 2; ph++) {
            if ((lane >> 5) == ph) {
                unsigned *d = s_out + (lane & 31) * DSTRIDE;
#pragma unroll
                for (int j = 0; j < OL; j++) { d[j] = r0[j]; d[OL + j] = r1[j]; }
            }
            __syncthreads();
#pragma unroll 2
            for (int i = 0; i < 32 / ORW; i++) {
                const int r = i * ORW + orow;
                const unsigned long long d = lane_u64(my_dst, ph * 32 + r);
                if (d != 0 && op * 4 < ng && nbytes > 0) {
                    const unsigned *sp = s_out + r * DSTRIDE + op * 4;
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
        }
    }
    /* staged tile: sample x of my row straight to its home when it belongs to the `wrapn` samples that run over the line's end
     * (wrap_home reads them back from the tile, which the two-pass drain has re-laid by then); wave-uniform condition at the caller */
    __device__ __forceinline__ void wrap_byte(int x, int v, int delta)
    {
        if (my_dst != 0) gstore8(my_dst + (unsigned) (x + delta), (unsigned) v);
    }
    /* after sample group g (4 samples = 1 dword per row): flush when the tile is full or the line ends */
    __device__ __forceinline__ void group_done(int g, int ngroups, int destw)
    {
        if constexpr (STAGED) {
            const int gi = g & (OTILE - 1);
            const bool last = g == ngroups - 1;
            if (gi == OL - 1 && !last) {                 /* first half complete, more of this block to come: to registers */
                __syncthreads();
#pragma unroll
                for (int j = 0; j < OL; j++) r0[j] = s_out[TO::row(lane) + j];
                __syncthreads();
            } else if (gi == OTILE - 1 || last) {
                if (gi < OL) drain(g & ~(OTILE - 1), gi + 1, destw);        /* the line ends inside a first half: still all in LDS */
                else drain_staged(g & ~(OTILE - 1), gi + 1, destw);
            }
        } else {
            if ((g & (OTILE - 1)) == OTILE - 1 || g == ngroups - 1) drain(g & ~(OTILE - 1), (g & (OTILE - 1)) + 1, destw);
        }
    }
};

/* source image row of destination row y: crt_ntsc.c:256-263 (field_offset where the system has one),
 * crt_nes.c:165-168 / crt_nesrgb.c:94-97 (NES timing).  The reference's `if (sy >= h) sy = h` (sic) addresses the
 * row BEHIND the image; that row is only read when the caller vouches for it (CRTHIP_F_IMAGE_SPARE_ROW). */
template <class S> __device__ __forceinline__ int source_row(const crthip_params &P, int y, int field)
{
    int sy;
    if constexpr (S::NES_TIMING) {
        sy = (y * P.h) / S::LINES;
    } else {
        const int field_offset = S::FIELD_ROWS ? (field * P.h + P.desth) / P.desth / 2 : 0;
        sy = (y * P.h) / P.desth + field_offset;
    }
    if (sy >= P.h) sy = (P.flags & CRTHIP_F_IMAGE_SPARE_ROW) ? P.h : P.h - 1;
    if (sy < 0) sy = 0;
    return sy;
}

/* FAST: 24-bit multiplies (always in range for the IIRs and the carrier products: 8-bit pixels
 * bound every state; `white` and `noise` are range-checked on the host).
 * IN4: 4-byte input pixels moved through a cooperative LDS tile (RowTiles);
 * otherwise (3-byte formats, tiny images) each lane reads its own pixels bytewise.
 * The produced samples always leave through a cooperative LDS tile. */
/* ACT = dwords per row and tile (ACT pixels in, 4*ACT samples out): 32 moves full 128-byte lines per row
 * piece group, 16 halves the LDS footprint (more waves per SIMD) -- chosen by input width at launch */
/* CLAMP: output is inp[] (fused path), i.e. the +-127 clamp of crt_core.c:363-364 applies even when
 * no noise is added (only matters for NES, whose samples can be -128) */
/* OT = dwords per row of the sample tile: the signal leaves in 4 * OT-byte pieces per row.  16 (64-byte pieces) keeps the LDS
 * small; 64 (256-byte pieces, 16.6 KB) is what large batches take: the encoder is bound by its MEMORY PATTERN -- image pieces in,
 * signal pieces out, no arithmetic at all reproduces its time (tools/ubench_enc.hip, profiles/r05_encoder_memory_shapes.txt) -- and
 * 64-byte pieces at the reference's odd line starts are its dearest part */
template <class S, bool NOISE, bool FAST, bool IN4, bool CLAMP, int ACT, int OT = 16, int OL = OT>
__global__ void __launch_bounds__(64)
k_active(const crthip_params P, int n_fields, const unsigned char *__restrict__ images, size_t istride,
         signed char *__restrict__ dst, size_t fstride, const crthip_state *__restrict__ state,
         const uint2 *__restrict__ jump16, int order_k, int order_per, int pitch, int shift, int wrapn)
{
    /* pitch / shift / wrapn: where a row goes -- HRES, 0, 0 = the reference's flat lines; the fused path's padded lines otherwise
     * (crt_dev.h, sig_layout; wrapn = the row's samples that run over the end of its line) */
    using T = RowTiles<ACT, OT, OL>;
    constexpr int AC_SHIFT = ACT == 32 ? 5 : 4;
    __shared__ unsigned s_pix[T::TP::DWORDS];
    __shared__ unsigned s_out[T::TO::DWORDS];

    const int lane = threadIdx.x;
    const int gid = block_item(blockIdx.x, order_k, order_per) * 64 + lane;     /* workgroup order: crt_dev.h */
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
    const int wrap_x0 = wrapn > 0 ? destw - wrapn : 0x7fffffff;   /* staged sample tile: first sample that runs over the line's end */

    /* per-row source / destination, published to the whole wave */
    const int sy = source_row<S>(P, y, st.field & 1);
    const int in_bpp = S::IS_NES ? 2 : P.in_bpp;
    const unsigned char *row = img + (size_t) sy * w * in_bpp;
    T tiles;
    tiles.init(s_pix, s_out, (unsigned long long) row,
               live ? (unsigned long long) (dst + (size_t) f * fstride + (size_t) (shift + (y + P.yo) * pitch + P.xo)) : 0ull, lane, w * 4);

    if constexpr (S::IS_NES) {
        /* crt_nes.c:162-193, images too narrow for the tile path (k_active_nes otherwise) */
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
            tiles.put(g, pack);
            tiles.group_done(g, ngroups, destw);
        }
    } else {
        /* crt_ntsc.c:254-324 and its siblings crt_ntscvhs.c, crt_snes.c:246-319, crt_template.c, crt_pv1k.c:233-312,
         * crt_nesrgb.c:91-163.  Carrier row of this line: (h * ph) * cc == h * (ph * cc) in wrapping arithmetic,
         * the +-1 line phase of crt_ntsc.c:199-200 is folded into row 1 of the tables on the host.  xo is a multiple
         * of CC_SAMPLES (crt_ntsc.c:203, crt_snes.c:201), so the carrier phase (x + xo) % CCS is x % CCS. */
        const int crow = carrier_row<S>(y + P.yo, st.field, st.frame, st.aux);
        int cI[S::CCS], cQ[S::CCS];
#pragma unroll
        for (int k = 0; k < S::CCS; k++) { cI[k] = P.modI[crow][k] * (FAST ? 4096 : 1); cQ[k] = P.modQ[crow][k] * (FAST ? 4096 : 1); }
        /* 5 samples per chroma cycle (PV-1000): the carrier phase x % 5 is no compile-time constant of the 4-sample unrolling;
         * the line's five carrier pairs live in LDS, [lane][phase]{I, Q} at an odd row stride */
        constexpr int CC5_STRIDE = 11;
        __shared__ int s_cc5[S::CCS == 5 ? 64 * CC5_STRIDE : 1];
        if constexpr (S::CCS == 5) {
#pragma unroll
            for (int k = 0; k < 5; k++) { s_cc5[lane * CC5_STRIDE + 2 * k] = cI[k]; s_cc5[lane * CC5_STRIDE + 2 * k + 1] = cQ[k]; }
        }
        const int cy_ = P.iir_c[0], ci_ = P.iir_c[1], cq_ = P.iir_c[2];
        const int white = P.white, ire_base = P.ire_base, noise = P.noise;
        int hy = 0, hi = 0, hq = 0;
        const bool hipass = (P.flags & CRTHIP_F_HIPASS) != 0;
        int cph = 0;                                              /* x % CCS for the 5-sample system (wave-uniform) */
        /* RGB -> YIQ coefficients by byte position (wave-uniform) */
        const bool alpha_first = format_alpha_first(P.format), blue_low = format_blue_low(P.format);
        const int ky0 = blue_low ? 7471 : 19595, ky1 = 38470, ky2 = blue_low ? 19595 : 7471;
        const int ki0 = blue_low ? -21103 : 39059, ki1 = -18022, ki2 = blue_low ? 39059 : -21103;
        const int kq0 = blue_low ? 20382 : 13894, kq1 = -34275, kq2 = blue_low ? 13894 : 20382;
        /* (r6) the same sums as TWO-term dot products: bytes 0 and 2 of the pixel side by side as 16-bit halves (one v_perm), their
         * two coefficients packed into one scalar register, byte 1 (green) times its coefficient through the accumulator --
         * v_mul_u32_u24_sdwa + v_dot2_i32_i16 per row instead of three multiplies and a three-operand add.  A packed coefficient
         * must fit 16 signed bits: 38470 (Y) and -34275 (Q) are green's, i.e. the lone multiply's; the I row's 39059 goes in as
         * 32767 + 6292, the second part by a second dot product whose other half is 0.  Exact integer sums, below 2^31: the same
         * values bit for bit.  15 -> 11 vector instructions per pixel (the encoder runs at its arithmetic, profiles/r06_ab_*). */
        const int kyp = (ky2 << 16) | ky0, kqp = (int) (((unsigned) kq2 << 16) | ((unsigned) kq0 & 0xffffu));
        const int ki0a = ki0 > 32767 ? 32767 : ki0, ki2a = ki2 > 32767 ? 32767 : ki2;
        const int kipa = (int) (((unsigned) ki2a << 16) | ((unsigned) ki0a & 0xffffu));
        const int kipb = (int) (((unsigned) (ki2 - ki2a) << 16) | ((unsigned) (ki0 - ki0a) & 0xffffu));

        /* FAST + cooperative tiles: the three one-pole low-passes (iirf, crt_ntsc.c:117-126) as one v_mad_i64_i32
         * each.  h' = h + ((c*(s-h)) >> 11) is the high half of (c << 21)*(s-h) + {0, h}; for c >= 1024 the
         * multiplier would not fit 32 bits, but h + (s-h) = s gives the equivalent ((c-2048) << 21)*(s-h) + {0, s}.
         * The 64-bit product is exact where the reference's 32-bit one wraps: equal inside the FAST envelope
         * (|c*(s-h)| < 2^31: |s-h| <= 2*1275, c <= 2048).  Which form a channel takes is a property of the system
         * (IIR_Y_NEAR), checked against the actual coefficients at launch. */
        constexpr bool I64 = FAST && IN4 && S::BANDLIMIT;
        const int my_ = (S::IIR_Y_NEAR ? cy_ - 2048 : cy_) * (1 << 21), mi_ = ci_ * (1 << 21), mq_ = cq_ * (1 << 21);   /* (may be negative: no <<) */
        long hyp = 0, hip = 0, hqp = 0, fyp = 0;                   /* state (and the luma input) in the high halves */
        constexpr long HI_HALF = (long) 0xffffffff00000000ul;

        tiles.shift8 = IN4 && alpha_first;
        if (IN4) tiles.start();
        /* loop invariants the compiler would otherwise re-materialise per sample (constant-bus limit of VOP3) */
        int neg_noise127 = -0x7f * noise;
        asm volatile("" : "+v"(neg_noise127));
        int neg_noise127_256 = -0x7f * noise * 256, ire_base_1024 = ire_base * 1024, ire_base_65536 = ire_base * 65536;
        const int white64 = white * 64;
        asm volatile("" : "+v"(neg_noise127_256));
        asm volatile("" : "+v"(ire_base_1024));
        asm volatile("" : "+v"(ire_base_65536));
        const int noise256 = noise * 256;
        v2u lcg_add = { LCG_ADD, 0u };
        asm volatile("" : "+v"(lcg_add));
        int fy = 0, fi = 0, fq = 0, have_col = -1;                 /* YIQ of source column have_col (wave-uniform) */
        /* source column on the scalar unit: one 64-bit add per sample (crthip_params.col_step), instead of the
         * incremental quotient / remainder pair (8 scalar instructions as compiled) */
        const unsigned long long cstep = ((unsigned long long) P.col_step_hi << 32) | P.col_step_lo;
        unsigned long long cpos = 0;
        for (int g = 0; g < ngroups; g++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = 4 * g + k;
                /* With the cooperative tiles the up to 3 samples behind the line end are simply computed: they land
                 * in the tile and are never stored (drain clips to destw), their pixel reads stay inside the row
                 * (piece_offset) -- one compare and branch per sample less */
                if (IN4 || x < destw) {
                    if constexpr (!S::IS_NES) col = (int) (cpos >> 32);
                    /* an upscaled line (w < destw) samples some source pixels twice: all 64 lanes are at the same
                     * column, so "same pixel as before" is a scalar branch that skips the fetch and the conversion */
                    if (col != have_col) {
                        unsigned pixel;
                        if (IN4) {
                            tiles.want(col >> AC_SHIFT);           /* wave-uniform */
                            pixel = tiles.pixel_dword(col);
                        } else {
                            const unsigned char *pp = row + (size_t) col * in_bpp;
                            pixel = (unsigned) pp[0] | (unsigned) pp[1] << 8 | (unsigned) pp[2] << 16;
                            if (in_bpp == 4) pixel |= (unsigned) pp[3] << 24;
                        }
                        if (!IN4 && alpha_first) {                 /* tiles: shifted once per tile (RowTiles::stash) */
                            pixel >>= 8;
                            asm volatile("" : "+v"(pixel));
                        }
                        const int c0 = pixel & 255, c1 = (pixel >> 8) & 255, c2 = (pixel >> 16) & 255;
                        int y_;
                        if (FAST && IN4) {
                            const int p02 = (int) __builtin_amdgcn_perm(pixel, pixel, 0x0c020c00u);     /* { byte 2, byte 0 } as 16-bit halves */
                            y_ = dot2_vs(p02, kyp, ky1 * c1) >> 14;
                            fi = dot2_vs(p02, kipb, dot2_vs(p02, kipa, ki1 * c1)) >> 14;
                            fq = dot2_vs(p02, kqp, kq1 * c1) >> 14;
                        } else {
                            y_ = (ky0 * c0 + ky1 * c1 + ky2 * c2) >> 14;
                            fi = (ki0 * c0 + ki1 * c1 + ki2 * c2) >> 14;
                            fq = (kq0 * c0 + kq1 * c1 + kq2 * c2) >> 14;
                        }
                        if (I64) fyp = (fyp & ~HI_HALF) | (long) ((unsigned long) (unsigned) y_ << 32);
                        else fy = y_;
                        have_col = col;
                    }
                    if (I64) {
                        /* (r6) the three low-passes side by side -- differences, then the multiply-adds, the noise generator's step with
                         * them -- instead of one after the other: no instruction waits on the one issued just before it (cf. eq_step64_yiq,
                         * crt_decode_lane.h; k_active 0.874 -> 0.856 ms at 640x480 x 4096, profiles/r06_ab_encoder_side_by_side.txt) */
                        const int dy = pair_hi(fyp) - pair_hi(hyp), di = fi - pair_hi(hip), dq = fq - pair_hi(hqp);
                        const long ay = S::IIR_Y_NEAR ? fyp : (hyp & HI_HALF), ai = hip & HI_HALF, aq = hqp & HI_HALF;
                        __builtin_amdgcn_sched_barrier(0);
                        hyp = mad64_vs(dy, my_, ay); hip = mad64_vs(di, mi_, ai); hqp = mad64_vs(dq, mq_, aq);
                        if (NOISE) rn = lcg_step_mad64(rn, lcg_add);
                        __builtin_amdgcn_sched_barrier(0);
                        hy = pair_hi(hyp); hi = pair_hi(hip); hq = pair_hi(hqp);
                    } else if (S::BANDLIMIT) {
                        hy += mulq<FAST>(fy - hy, cy_) >> 11;       /* iirf, crt_ntsc.c:117-126 */
                        hi += mulq<FAST>(fi - hi, ci_) >> 11;
                        hq += mulq<FAST>(fq - hq, cq_) >> 11;
                    } else {                                        /* CRT_DO_BANDLIMITING 0, crt_snes.c:113-122 */
                        hy = fy; hi = fi; hq = fq;
                    }
                    /* what iirf returns: the state -- or, in a HIPASS 1 build (crt_ntsc.c:121-122; exact kernels only),
                     * input minus state */
                    int oy = hy, oi = hi, oq = hq;
                    if (!FAST && S::BANDLIMIT && hipass) { oy = (I64 ? pair_hi(fyp) : fy) - hy; oi = fi - hi; oq = fq - hq; }
                    int ccI, ccQ;
                    if constexpr (S::CCS == 4) {
                        ccI = k == 0 ? cI[0] : k == 1 ? cI[1] : k == 2 ? cI[2] : cI[3];
                        ccQ = k == 0 ? cQ[0] : k == 1 ? cQ[1] : k == 2 ? cQ[2] : cQ[3];
                    } else {
                        ccI = s_cc5[lane * CC5_STRIDE + 2 * cph];           /* one LDS read at a scalar offset instead of a */
                        ccQ = s_cc5[lane * CC5_STRIDE + 2 * cph + 1];       /* five-way register select per carrier          */
                        cph = cph == S::CCS - 1 ? 0 : cph + 1;
                    }
                    int ire;
                    if constexpr (FAST && NOISE && I64) {
                        /* (r6, with the low-passes above) the noise term -- its own chain: generator state (stepped above) -> byte -> scaled --
                         * between the steps of the level's chain; the arithmetic is the FAST && NOISE branch below, line for line */
                        const int pI = __mul24(oi, ccI);
                        const int nb = (int) ((rn >> 16) & 0xffu);
                        __builtin_amdgcn_sched_barrier(0);
                        const int pQ = __mul24(oq, ccQ);
                        const int nz = mad24_vv(nb, noise256, neg_noise127_256);
                        __builtin_amdgcn_sched_barrier(0);
                        const int miq = add_hiwords(pI, pQ);
                        ire = mad24_vv(oy + miq, white64, ire_base_65536);
                        ire = clampi(ire, 0, (110 << 16) | 0xffff);
                        ire = add_hiwords(ire, nz);
                        ire = clampi(ire, -127, 127);
                        tiles.put_byte(g, k, ire);
                        if constexpr (T::STAGED) { if (x >= wrap_x0 && x < destw) tiles.wrap_byte(x, ire, pitch - S::HRES); }
                        cpos += cstep;
                        continue;
                    }
                    if (FAST) {
                        /* (h * cc) >> 4 twice: the carriers are pre-scaled by 2^12 (cI / cQ above), so that each
                         * shift is "take the high word" and both ride on the add;
                         * base + (v * white >> 10) == (v * white + (base << 10)) >> 10: one multiply-add */
                        const int miq = add_hiwords(__mul24(oi, ccI), __mul24(oq, ccQ));
                        /* (r6) with noise to add, the >> 10 is not taken at all: white and the base scaled by another 2^6 leave the
                         * level in the HIGH word, the clamp to [0, 110] works on the scaled value (monotone: same result), and the
                         * noise add below takes the high words of both its operands (|white| < 2^17, host-checked) */
                        if (NOISE) ire = mad24_vv(oy + miq, white64, ire_base_65536);
                        else ire = mad24_vv(oy + miq, white, ire_base_1024) >> 10;
                    } else {
                        const int mi = (oi * ccI) >> 4;
                        const int mq = (oq * ccQ) >> 4;
                        ire = ire_base + (((oy + mi + mq) * white) >> 10);
                    }
                    if (FAST && NOISE) ire = clampi(ire, 0, (110 << 16) | 0xffff);
                    else ire = clampi(ire, 0, 110);
                    if (NOISE) {
                        rn = lcg_step_mad64(rn, lcg_add);
                        const int nb = (int) ((rn >> 16) & 0xffu);
                        if (FAST) {
                            /* ((byte - 0x7f) * noise) >> 8 added to ire: the product scaled by 256 so that the
                             * shift becomes "take the high word" and rides on the add (|noise| < 2^15 on this path) */
                            ire = add_hiwords(ire, mad24_vv(nb, noise256, neg_noise127_256));
                        } else {
                            ire += (nb * noise + neg_noise127) >> 8;
                        }
                        ire = clampi(ire, -127, 127);
                    }
                    tiles.put_byte(g, k, ire);
                    if constexpr (T::STAGED) { if (x >= wrap_x0 && x < destw) tiles.wrap_byte(x, ire, pitch - S::HRES); }
                    cpos += cstep;
                } else {
                    tiles.put_byte(g, k, 0);
                }
            }
            tiles.group_done(g, ngroups, destw);
        }
    }
    if constexpr (!T::STAGED) { if (wrapn > 0) tiles.wrap_home(destw, wrapn, pitch - S::HRES); }
}

/* ------------------------------------------------------------------------- */
/* M5 in the scanline-parallel shape (small batches)                           */
/* ------------------------------------------------------------------------- */
/* k_active gives a destination row to ONE lane, which walks its 753 samples with ~45 vector instructions each: a single
 * field is 4 waves and 0.15 ms however empty the chip is.  Here a wave takes R = 8 rows and cuts the work by kind, in
 * tiles of 64 samples:
 *   A  pixel fetch + RGB -> YIQ for (row, sample): fully parallel, 64 lanes = the 64 samples of one row at a time
 *   B  the three one-pole low-passes (crt_ntsc.c:117-126): the only serial part; lane = (row, channel), 24 lanes,
 *      64 steps of 4 instructions per tile, operands from / to LDS
 *   C  modulate, scale, clamp, + noise, pack: fully parallel, a lane takes 4 consecutive samples (one dword store)
 * so a field is 30 waves and the serial chain per wave is 753 x 4 instructions.  Same arithmetic, exact 32-bit
 * multiplies throughout.  RGB-input systems only (the NES's table encoder is cheap as it is). */
template <class S, bool NOISE, bool CLAMP>
__global__ void __launch_bounds__(64)
k_active_row(const crthip_params P, int n_fields, const unsigned char *__restrict__ images, size_t istride,
             signed char *__restrict__ dst, size_t fstride, const crthip_state *__restrict__ state,
             const uint2 *__restrict__ jump16, const uint2 *__restrict__ jump1, int pitch, int shift, int wrapn)
{
    constexpr int R = 8, TS = 64, CCS = S::CCS;
    __shared__ int s_f[R][3][TS + 1];                    /* YIQ of the tile, then (in place) the low-passed values */

    const int lane = threadIdx.x;
    const int rows = P.desth, total = n_fields * rows;
    const int w = P.w, destw = P.destw, in_bpp = P.in_bpp;
    const unsigned isel = input_selector(P.format);

    /* the wave's rows: global row g0 + r.  Per-lane copies for the phases' lane -> row mappings */
    const int g0 = blockIdx.x * R;
    auto row_info = [&](int r, int &f, int &y, bool &live) {
        const int gid = g0 + r;
        live = gid < total;
        f = live ? gid / rows : 0;
        y = live ? gid - f * rows : 0;
    };

    /* phase B role: lane = 3 * row + channel */
    const int b_row = lane / 3, b_ch = lane - b_row * 3;
    const bool b_lane = lane < 3 * R;
    const int b_coef = b_ch == 0 ? P.iir_c[0] : b_ch == 1 ? P.iir_c[1] : P.iir_c[2];
    int hstate = 0;

    /* phase C role: 16 lanes per row, 4 rows per pass, R / 4 passes; per pass the row's constants */
    constexpr int CP = R / 4;
    const int c_rp = lane >> 4, c_j = lane & 15;
    int c_live[CP], c_start[CP], c_cI[CP][CCS], c_cQ[CP][CCS];
    unsigned c_rn0[CP];
    unsigned long long c_dst[CP];
#pragma unroll
    for (int ps = 0; ps < CP; ps++) {
        int f, y; bool live;
        row_info(ps * 4 + c_rp, f, y, live);
        const crthip_state st = state[f];
        c_live[ps] = live;
        c_start[ps] = (y + P.yo) * S::HRES + P.xo;                  /* flat sample index: the noise generator's position */
        c_dst[ps] = (unsigned long long) (dst + (size_t) f * fstride + (size_t) (shift + (y + P.yo) * pitch + P.xo));   /* k_active: pitch / shift / wrapn */
        c_rn0[ps] = (unsigned) st.rn;
        const int crow = carrier_row<S>(y + P.yo, st.field, st.frame, st.aux);
#pragma unroll
        for (int k = 0; k < CCS; k++) { c_cI[ps][k] = P.modI[crow][k]; c_cQ[ps][k] = P.modQ[crow][k]; }
    }
    /* phase A: source row pointers of all R rows (wave-uniform values, computed by every lane) */
    unsigned long long a_src[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int f, y; bool live;
        row_info(r, f, y, live);
        const int field = live ? state[f].field & 1 : 0;
        const int sy = source_row<S>(P, y, field);
        a_src[r] = (unsigned long long) (images + (size_t) f * istride + (size_t) sy * w * in_bpp);
    }
    const int white = P.white, ire_base = P.ire_base, noise = P.noise;

    for (int t0 = 0; t0 < destw; t0 += TS) {
        /* ---- A: fetch + convert, lane = sample t0 + lane of every row ---- */
        {
            const int x = t0 + lane;
            const int col = x < destw ? (int) ((unsigned) x * (unsigned) w / (unsigned) destw) : 0;   /* crt_ntsc.c:276 */
            unsigned px[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned long long a = a_src[r] + (size_t) col * in_bpp;
                if (in_bpp == 4) px[r] = gload32(a);
                else px[r] = gload8(a) | gload8(a + 1) << 8 | gload8(a + 2) << 16;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned rgb = __builtin_amdgcn_perm(px[r], px[r], isel);
                const int rr = (rgb >> 16) & 255, gg = (rgb >> 8) & 255, bb = rgb & 255;
                s_f[r][0][lane] = (19595 * rr + 38470 * gg + 7471 * bb) >> 14;
                s_f[r][1][lane] = (39059 * rr - 18022 * gg - 21103 * bb) >> 14;
                s_f[r][2][lane] = (13894 * rr - 34275 * gg + 20382 * bb) >> 14;
            }
        }
        wave_lds_fence();
        /* ---- B: the low-passes, lane = (row, channel) ---- */
        if (S::BANDLIMIT) {
            if (b_lane) {
                int *fp = &s_f[b_row][b_ch][0];
#pragma unroll 1
                for (int h0 = 0; h0 < TS; h0 += 16) {
                    int v[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) v[k] = fp[h0 + k];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        hstate += ((v[k] - hstate) * b_coef) >> 11;                /* iirf, crt_ntsc.c:117-126 */
                        v[k] = (P.flags & CRTHIP_F_HIPASS) ? v[k] - hstate : hstate;   /* HIPASS 1: crt_ntsc.c:121-122 */
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) fp[h0 + k] = v[k];
                }
            }
            wave_lds_fence();
        }
        /* ---- C: modulate + noise + store, lane = 4 consecutive samples of one of 4 rows ---- */
#pragma unroll
        for (int ps = 0; ps < CP; ps++) {
            const int r = ps * 4 + c_rp;
            const int x0 = t0 + 4 * c_j;
            unsigned rn = 0;
            if (NOISE) {
                const int idx = c_start[ps] + x0;
                const uint2 j = jump16[idx >> 4], q = jump1[idx & 15];
                rn = q.x * (j.x * c_rn0[ps] + j.y) + q.y;
            }
            int smp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xl = 4 * c_j + k;
                const int hy = s_f[r][0][xl], hi = s_f[r][1][xl], hq = s_f[r][2][xl];
                const int ph = CCS == 4 ? k : (x0 + k) % CCS;      /* xo is a multiple of CCS: phase (x + xo) % CCS == x % CCS */
                int ccI = c_cI[ps][0], ccQ = c_cQ[ps][0];
#pragma unroll
                for (int m = 1; m < CCS; m++) if (ph == m) { ccI = c_cI[ps][m]; ccQ = c_cQ[ps][m]; }
                const int mi = (hi * ccI) >> 4, mq = (hq * ccQ) >> 4;
                int ire = ire_base + (((hy + mi + mq) * white) >> 10);
                ire = clampi(ire, 0, 110);
                if (NOISE) { rn = lcg_step(rn); ire = noisy(ire, rn, noise); }
                else if (CLAMP) ire = clampi(ire, -127, 127);
                smp[k] = ire;
            }
            if (c_live[ps] && x0 < destw) {
                const unsigned long long d = c_dst[ps] + x0;
                if (x0 + 4 <= destw) {
                    gstore32(d, pack4(smp[0], smp[1], smp[2], smp[3]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (x0 + k < destw) gstore8(d + k, (unsigned) smp[k]);
                }
                if (x0 + 4 > destw - wrapn) {               /* padded lines: the home of the samples behind the line's end (RowTiles::wrap_home) */
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (x0 + k >= destw - wrapn && x0 + k < destw) gstore8(d + (unsigned) (k + pitch - S::HRES), (unsigned) smp[k]);
                }
            }
        }
        wave_lds_fence();
    }
}

/* ------------------------------------------------------------------------- */
/* M5, NES flavour (crt_nes.c:162-193) with a lookup table                     */
/* ------------------------------------------------------------------------- */
/* A composite sample of the NES is  ((BLACK + black_point + sum_{k<4} square(p, phase+k)) * white_point / 100) >> 12
 * truncated to a signed char.  square() depends on the phase only through (hue+phase)%12 and (phase>>1)%6,
 * both 12-periodic, so the sample is a function of (9-bit pixel, phase mod 12): 512 x 12 bytes, rebuilt whenever
 * the black / white point knobs change (cached in the context). */
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
             const uint2 *__restrict__ jump16, const signed char *__restrict__ tab, int pitch, int shift, int wrapn)
{
    using T = RowTiles<ACT>;
    constexpr int AC_SHIFT = ACT == 32 ? 5 : 4;
    __shared__ unsigned s_pix[T::TP::DWORDS];
    __shared__ unsigned s_out[T::TO::DWORDS];
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
    const int sy = source_row<S>(P, y, 0);                      /* crt_nes.c:165-168 */
    __syncthreads();                                            /* s_tab */
    /* pixel tiles: ACT dwords = 2*ACT PPU pixels per row */
    T tiles;
    tiles.init(s_pix, s_out, (unsigned long long) (img + (size_t) sy * w * 2),
               live ? (unsigned long long) (dst + (size_t) f * fstride + (size_t) (shift + (y + P.yo) * pitch + P.xo)) : 0ull,   /* k_active: pitch / shift / wrapn */
               lane, w * 2);
    tiles.start();

    int ph = 4 * ((y + P.yo + st.aux) % 3);                    /* phasetab {0,4,8}; advances by 3 per sample, mod 12 */
    const signed char *tb = (const signed char *) s_tab;
    for (int g = 0; g < ngroups; g++) {
        int smp[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = 4 * g + k;
            if (x < destw) {
                tiles.want(col >> (AC_SHIFT + 1));              /* wave-uniform */
                const unsigned dw = tiles.pixel_dword(col >> 1);
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
        tiles.put(g, pack4(smp[0], smp[1], smp[2], smp[3]));
        tiles.group_done(g, ngroups, destw);
    }
    if (wrapn > 0) tiles.wrap_home(destw, wrapn, pitch - S::HRES);
}

/* The clean skeleton (blanking / sync / burst, 0 where crt_modulate writes nothing) of a whole field depends
 * only on a handful of per-field inputs: NTSC / VHS (field, frame parity) -> 4 variants; the systems with
 * per-line-class carriers: dot crawl offset 0..5 (x field parity where the vertical sync depends on it) -> up to
 * 12.  k_skeleton writes the variants (cached in the context, rebuilt when the burst table changes), k_margin then
 * only copies 16 bytes and adds the channel noise. */
template <class S> __device__ __forceinline__ int skeleton_variant(int field, int frame, int aux)
{
    if constexpr (S::LINE_ROWS) {
        int dco = aux;
        if (dco < 0 || dco > CRTHIP_DCO_MAX) dco = ((dco % S::VPER) + S::VPER) % S::VPER;
        return (S::VS_BY_FIELD && !S::NES_TIMING ? (field & 1) * (CRTHIP_DCO_MAX + 1) : 0) + dco;
    } else {
        return ((field & 1) << 1) | (frame & 1);
    }
}

template <class S>
__global__ void __launch_bounds__(256)
k_skeleton(const crthip_params P, signed char *__restrict__ skel, size_t fstride)
{
    constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= SKEL_VARIANTS * CHUNKS) return;
    const int var = gid / CHUNKS;
    const int idx0 = (gid - var * CHUNKS) * 16;
    /* inverse of skeleton_variant(); VHS: the aberration band is patched in by k_margin, aux = 0 here */
    const int field = S::LINE_ROWS ? var / (CRTHIP_DCO_MAX + 1) : var >> 1;
    const int frame = S::LINE_ROWS ? 0 : var & 1;
    const int aux = S::LINE_ROWS ? var % (CRTHIP_DCO_MAX + 1) : 0;
    int line = idx0 / S::HRES;
    int t = idx0 - line * S::HRES;
    int vals[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int v = 0;
        if (!skeleton<S>(P, line, t, field, frame, aux, true, v)) v = 0;
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
 * (flat indices, exactly like the reference's analog[(x + xo) + (y + yo) * HRES]: a rectangle that runs over the
 * end of a line simply continues in the next one) and each lane takes one run of up to 16 samples of it: 16 bytes
 * of the cached skeleton variant (k_skeleton), + noise (LCG state by the 16-step jump table and a 16-entry table
 * for the remainder). */
template <class S, bool NOISE>
__global__ void __launch_bounds__(256)
k_margin(const crthip_params P, int n_fields, signed char *__restrict__ dst, size_t fstride,
         const crthip_state *__restrict__ state, const uint2 *__restrict__ jump16, const uint2 *__restrict__ jump1,
         const signed char *__restrict__ skel, int head_chunks, int gap_chunks, int tail_chunks, unsigned gap_magic)
{
    /* (r5) the field is the grid's y index and the row of a gap chunk comes from a multiply-high by ceil(2^32 / gap_chunks)
     * (exact for the < 2^16 chunks of a field): two runtime integer divisions per lane were a third of a noise-free lane's work */
    const int per_field = head_chunks + (P.desth - 1) * gap_chunks + tail_chunks;
    const int q0 = blockIdx.x * 256 + threadIdx.x;
    if (q0 >= per_field) return;
  for (int f = blockIdx.y; f < n_fields; f += (int) gridDim.y) {
    int q = q0;
    const int s0 = P.yo * S::HRES + P.xo;
    const int gap_len = S::HRES - P.destw;
    int idx0, len, region0;
    if (q < head_chunks) {
        idx0 = q * 16;
        len = s0 - idx0;
        region0 = 0;
    } else if (q < head_chunks + (P.desth - 1) * gap_chunks) {
        q -= head_chunks;
        const int yy = gap_chunks == 1 ? q : (int) __umulhi((unsigned) q, gap_magic), c = q - yy * gap_chunks;     /* (ceil(2^32 / 1) does not fit) */
        region0 = s0 + yy * S::HRES + P.destw;
        idx0 = region0 + c * 16;
        len = gap_len - c * 16;
    } else {
        q -= head_chunks + (P.desth - 1) * gap_chunks;
        region0 = s0 + (P.desth - 1) * S::HRES + P.destw;
        idx0 = region0 + q * 16;
        len = S::INPUT_SIZE - idx0;
    }
    if (len > 16) len = 16;
    /* (r5) a region's last, partial chunk is moved BACK so that it ends with the region and is a whole chunk too (it overlaps its
     * neighbour, which writes the same bytes: skeleton and noise are functions of the sample index).  As 1-15 byte stores it made
     * EVERY wave of the kernel issue up to 15 store instructions beside its one 16-byte store -- a gap is 157 samples, every tenth
     * lane was partial: k_margin 0.116 -> see profiles/r05_experiments.txt section 14 */
    if (len < 16 && idx0 - (16 - len) >= region0) { idx0 -= 16 - len; len = 16; }
    const crthip_state *st = state + f;
    const int aux = st->aux;
    const int var = skeleton_variant<S>(st->field, st->frame, aux);
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
        /* the LCG step and the noise product through the full-rate 64-bit multiply-add (v_mul_lo_u32, which `a * b` compiles to,
         * runs at a quarter of it: two of them per sample made this kernel cost 0.12 ms per 4096 fields); same wrapped 32-bit
         * arithmetic as noisy() / lcg_step() */
        v2u lcg_add = { LCG_ADD, 0u };
        asm volatile("" : "+v"(lcg_add));
        const int noise = P.noise;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            rn = lcg_step_mad64(rn, lcg_add);
            const int sk = (wds[k >> 2] << (24 - 8 * (k & 3))) >> 24;
            vals[k] = clampi(sk + (mul_lo_mad64((int) ((rn >> 16) & 0xffu) - 0x7f, noise) >> 8), -127, 127);
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
    if (q0 == 0) {
        /* mirror of the struct members behind inp[] (see CRTHIP_TAIL) */
        signed char *tail = out + S::INPUT_SIZE;
        store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
    }
  }
}


/* (r6) The same for the fused path's PADDED signal lines (crt_dev.h, sig_layout).  Work is cut by line instead of by flat index, so
 * that no 16-byte chunk straddles a line end (in the padded layout its two halves would be 114 bytes apart):
 *     lines [0, yo)                          the whole line, CF = ceil(HRES / 16) chunks, the last one moved back to end with the line
 *     lines [yo, yo + desth)                 left of the row  [a0, xo)  (CL chunks)  and right of it  [xo + destw, HRES)  (CR chunks)
 *     lines [yo + desth, VRES)               [a0, HRES)
 * a0 = wrap for a line whose predecessor carries an active row (the row's last `wrap` samples run into this line: k_active's), else 0.
 * A chunk that lies inside the first PADC columns of line n >= 1 is stored a second time behind line n - 1 (the copy that makes
 * windows over a line end contiguous); chunks that only partly do are not -- `padv` (crt_fused_layout) is what that guarantees. */
template <class S, bool NOISE>
__global__ void __launch_bounds__(256)
k_margin_pad(const crthip_params P, int n_fields, signed char *__restrict__ dst, size_t fstride, int shift,
             const crthip_state *__restrict__ state, const uint2 *__restrict__ jump16, const uint2 *__restrict__ jump1,
             const signed char *__restrict__ skel, size_t skel_stride, int cl, int cr, unsigned act_magic, int wrap)
{
    using G = PadGeom<S>;
    constexpr int CF = (S::HRES + 15) / 16;
    const int head_chunks = P.yo * CF, act_per = cl + cr, act_chunks = P.desth * act_per;
    const int per_field = head_chunks + act_chunks + (S::VRES - P.yo - P.desth) * CF;
    const int q0 = blockIdx.x * 256 + threadIdx.x;
    if (q0 >= per_field) return;
    int line, k, lo, hi;
    if (q0 < head_chunks) {
        line = q0 / CF; k = q0 - line * CF; lo = 0; hi = S::HRES;
    } else if (q0 < head_chunks + act_chunks) {
        const int q = q0 - head_chunks;
        const int y = act_per == 1 ? q : (int) __umulhi((unsigned) q, act_magic);       /* q / act_per (exact below 2^16) */
        k = q - y * act_per;
        line = P.yo + y;
        if (k < cl) { lo = y > 0 ? wrap : 0; hi = P.xo; }
        else { k -= cl; lo = P.xo + P.destw < S::HRES ? P.xo + P.destw : S::HRES; hi = S::HRES; }
    } else {
        const int q = q0 - head_chunks - act_chunks;
        const int r = q / CF;
        k = q - r * CF;
        line = P.yo + P.desth + r;
        lo = r == 0 ? wrap : 0; hi = S::HRES;
    }
    int col = lo + 16 * k, len = 16;
    if (hi - lo >= 16) { if (col > hi - 16) col = hi - 16; }         /* the interval's last chunk ends with it (it overlaps its neighbour: same bytes) */
    else { if (k > 0 || hi <= lo) return; len = hi - lo; }              /* an interval shorter than a chunk: bytes */
    const int idx0 = line * S::HRES + col;                             /* flat sample index: skeleton and noise are functions of it */
    const bool copy = line >= 1 && col + len <= G::PADC;
  for (int f = blockIdx.y; f < n_fields; f += (int) gridDim.y) {
    const crthip_state *st = state + f;
    const int var = skeleton_variant<S>(st->field, st->frame, st->aux);
    signed char *out = dst + (size_t) f * fstride + shift;
    const v4i sk = load16u(skel + (size_t) var * skel_stride + idx0);
    const int wds[4] = { sk.x, sk.y, sk.z, sk.w };
    v4i pk;
    if (NOISE) {
        unsigned rn;
        {
            const uint2 j = jump16[idx0 >> 4], r = jump1[idx0 & 15];
            rn = r.x * (j.x * (unsigned) st->rn + j.y) + r.y;
        }
        int vals[16];
        v2u lcg_add = { LCG_ADD, 0u };
        asm volatile("" : "+v"(lcg_add));
        const int noise = P.noise;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            rn = lcg_step_mad64(rn, lcg_add);
            const int v = (wds[i >> 2] << (24 - 8 * (i & 3))) >> 24;
            vals[i] = clampi(v + (mul_lo_mad64((int) ((rn >> 16) & 0xffu) - 0x7f, noise) >> 8), -127, 127);
        }
        pk.x = (int) pack4(vals[0], vals[1], vals[2], vals[3]);
        pk.y = (int) pack4(vals[4], vals[5], vals[6], vals[7]);
        pk.z = (int) pack4(vals[8], vals[9], vals[10], vals[11]);
        pk.w = (int) pack4(vals[12], vals[13], vals[14], vals[15]);
    } else {
        pk = sk;
    }
    signed char *home = out + line * G::PITCH + col;
    if (len == 16) {
        store16u(home, pk);
        if (copy) store16u(home - G::PITCH + S::HRES, pk);
    } else {
        const int o4[4] = { pk.x, pk.y, pk.z, pk.w };
        for (int i = 0; i < len; i++) {
            const signed char b = (signed char) (o4[i >> 2] >> (8 * (i & 3)));
            home[i] = b;
            if (copy) home[i - G::PITCH + S::HRES] = b;
        }
    }
    if (q0 == 0) {
        /* mirror of the struct members behind inp[] (CRTHIP_TAIL): flat samples INPUT_SIZE .. + 15 = the head of line VRES, and its copy
         * behind the last line */
        for (int t = 0; t < 2; t++) {
            signed char *tail = out + (t ? S::VRES * G::PITCH : (S::VRES - 1) * G::PITCH + S::HRES);
            store4u(tail + 0, P.outw); store4u(tail + 4, P.outh); store4u(tail + 8, P.out_format); store4u(tail + 12, 0);
        }
    }
  }
}

template <class S>
static bool encoder_fast_ok(const crthip_params *p)
{
    const int wh = p->white < 0 ? -p->white : p->white;
    const int nz = p->noise < 0 ? -p->noise : p->noise;
    const int ib = p->ire_base < 0 ? -p->ire_base : p->ire_base;
    /* noise * 256 is a 24-bit multiplier in k_active; (y + i + q) * white * 64 + (ire_base << 16) must not wrap: |y + i + q| <= 1020 +
     * 608 + 533 (8-bit pixels through crt_ntsc.c:307-309, carriers <= 16 / 16), so 2161 * 2^13 * 64 + 2^13 * 2^16 < 2^31.  (Rounds 1-5
     * let white up to 2^23 into the fast kernels, where the product could wrap differently from the reference's; nothing sane is near.) */
    bool ok = wh < (1 << 13) && ib < (1 << 13) && nz < (1 << 15);
    if (p->flags & CRTHIP_F_HIPASS) ok = false;                   /* the debug build's high-pass lives in the exact kernels */
    for (int r = 0; r < CRTHIP_CARRIER_ROWS; r++)                  /* ... and so are the carriers * 4096 */
        for (int k = 0; k < CRTHIP_MAX_CCS; k++)
            ok = ok && p->modI[r][k] > -2048 && p->modI[r][k] < 2048 && p->modQ[r][k] > -2048 && p->modQ[r][k] < 2048;
    if (S::BANDLIMIT)                                              /* the forms of the 64-bit low-passes (k_active) */
        ok = ok && p->iir_c[0] > 0 && p->iir_c[0] <= 2048 && (p->iir_c[0] >= 1024) == S::IIR_Y_NEAR &&
             p->iir_c[1] > 0 && p->iir_c[1] < 1024 && p->iir_c[2] > 0 && p->iir_c[2] < 1024;
    return ok;
}

template <class S, bool FULL, bool FAST>
static void launch_active(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                          signed char *dst, const crthip_state *d_state, const sig_layout &lay)
{
    const int wrapn = lay.pitch != S::HRES ? lay.wrap : 0;      /* flat lines: a row that runs over its line's end simply continues */
    ProfScope ps(c, CRTHIP_K_ACTIVE);
    const int total = n * p->desth;
    const dim3 grid((total + 63) / 64), block(64);
    const unsigned char *img = (const unsigned char *) d_images;
    if constexpr (S::IS_NES) {
        if (p->w >= 8) {
            /* the sample table depends on the black / white point only: rebuilt when those change, always on the
             * context's main stream (crt_run_encoder_prepare), never concurrently with a reader */
            if (FULL && p->noise != 0)
                hipLaunchKernelGGL((k_active_nes<S, true, FULL, 16>), grid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, c->d_nes_tab, lay.pitch, lay.shift, wrapn);
            else
                hipLaunchKernelGGL((k_active_nes<S, false, FULL, 16>), grid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, c->d_nes_tab, lay.pitch, lay.shift, wrapn);
            return;
        }
    }
    const bool noise = FULL && p->noise != 0;
    if constexpr (!S::IS_NES) {
        /* kernel shape (crthip_set_shape): small batches take the scanline-parallel encoder */
        if (c->shape == 2 || (c->shape == 0 && n <= ROWS_SHAPE_MAX_FIELDS_ENC)) {
            const dim3 rgrid((total + 7) / 8);
            if (noise) hipLaunchKernelGGL((k_active_row<S, true, FULL>), rgrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, c->d_jump1, lay.pitch, lay.shift, wrapn);
            else hipLaunchKernelGGL((k_active_row<S, false, FULL>), rgrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, c->d_jump1, lay.pitch, lay.shift, wrapn);
            return;
        }
    }
    const bool in4 = S::IS_NES || (p->in_bpp == 4 && p->w >= 4);
    const block_order bo = make_block_order((int) grid.x, c->act_order_env == -1 ? n : c->act_order_env);
    const dim3 ogrid(bo.grid);
    const bool wide_in = c->ac_tile_env ? c->ac_tile_env == 32 : c->ac_tile ? c->ac_tile == 32 : p->w >= 1280;
    /* Larger signal pieces where the batch keeps the chip full at the lower occupancy their tile leaves (fused throughput path,
     * 4-byte pixels, fast envelope): 256-byte pieces beside the wide image tile (16.9 KB of LDS with the tile staged, 9 waves per CU), 128-byte pieces
     * beside the narrow one (12.7 KB, 12 waves per CU).  Measured per batch size and system in profiles/r05_experiments.txt
     * (sections 2-4): 1080p x 2048 k_active 0.86 -> 0.79 ms, 640x480 x 4096 1.05 -> 0.98, VHS / bloom / 720p -4..9 %; below the
     * thresholds (crt_dev.h: by wave count, re-measured in round 6) and for the 5-sample system, whose carrier table shares the LDS, the
     * 64-byte pieces stay the faster ones.
     * CRTHIP_SIG_TILE=16 pins the small tile, =32 / =64 the large one whatever the batch (A/B). */
    if constexpr (FULL && FAST && !S::IS_NES && S::CCS == 4) {
        const unsigned nw = grid.x;
        const bool big = c->sig_tile_env ? c->sig_tile_env != 16
                       : wide_in ? nw >= (unsigned) SIG_TILE64_MIN_WAVES_WIDE
                       : (nw > (unsigned) SIG_TILE32_BAND_LO && nw <= (unsigned) SIG_TILE32_ONE_ROUND) || nw > (unsigned) SIG_TILE16_ONE_ROUND;
        if (in4 && big) {
#ifndef CRTHIP_ENC_OL32
/* dwords per row of the large sample tiles that are in LDS (RowTiles: staged when half of OT).  Measured (profiles/r06_ab_encoder_waves.txt):
 * the 64-dword tile beside the wide image tile gains from staging (9 instead of 6 waves per CU: 1280x720 x 2048 k_active 0.779 -> 0.744 ms,
 * 1080p equal), the 32-dword tile beside the narrow one loses (16 instead of 12 waves, but 0.818 -> 0.862 ms at 640x480 x 4096: the two-pass
 * drain costs more than the waves give) and stays all in LDS */
#define CRTHIP_ENC_OL32 32
#endif
#ifndef CRTHIP_ENC_OL64
#define CRTHIP_ENC_OL64 32
#endif
#define CRTHIP_LAUNCH_ACTIVE_BIG(NZ) \
    do { if (wide_in) hipLaunchKernelGGL((k_active<S, NZ, true, true, true, 32, 64, CRTHIP_ENC_OL64>), ogrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, bo.K, bo.per, lay.pitch, lay.shift, wrapn); \
         else hipLaunchKernelGGL((k_active<S, NZ, true, true, true, 16, 32, CRTHIP_ENC_OL32>), ogrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, bo.K, bo.per, lay.pitch, lay.shift, wrapn); } while (0)
            if (noise) CRTHIP_LAUNCH_ACTIVE_BIG(true); else CRTHIP_LAUNCH_ACTIVE_BIG(false);
#undef CRTHIP_LAUNCH_ACTIVE_BIG
            return;
        }
    }
#define CRTHIP_LAUNCH_ACTIVE(NZ, I4) \
    do { if (wide_in) hipLaunchKernelGGL((k_active<S, NZ, FAST, I4, FULL, 32>), ogrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, bo.K, bo.per, lay.pitch, lay.shift, wrapn); \
         else hipLaunchKernelGGL((k_active<S, NZ, FAST, I4, FULL, 16>), ogrid, block, 0, c->stream, *p, n, img, istride, dst, lay.fstride, d_state, c->d_jump16, bo.K, bo.per, lay.pitch, lay.shift, wrapn); } while (0)
    if (noise) { if (in4) CRTHIP_LAUNCH_ACTIVE(true, true); else CRTHIP_LAUNCH_ACTIVE(true, false); }
    else       { if (in4) CRTHIP_LAUNCH_ACTIVE(false, true); else CRTHIP_LAUNCH_ACTIVE(false, false); }
#undef CRTHIP_LAUNCH_ACTIVE
}

/* fused path: skeleton + noise for everything outside the active rectangle */
template <class S>
static void launch_margins(crthip_ctx *c, const crthip_params *p, int n, signed char *dst, const crthip_state *d_state)
{
    ProfScope ps(c, CRTHIP_K_TEMPLATE);
    const int s0 = p->yo * S::HRES + p->xo;
    const int head = (s0 + 15) / 16;
    const int gap = (S::HRES - p->destw + 15) / 16;
    const int tail_len = S::INPUT_SIZE - (s0 + (p->desth - 1) * S::HRES + p->destw);
    const int tail = (tail_len + 15) / 16;
    const int per_field = head + (p->desth - 1) * gap + tail;
    const unsigned gap_magic = gap > 0 ? (unsigned) ((0x100000000ull + (unsigned) gap - 1) / (unsigned) gap) : 0u;
    const dim3 grid((per_field + 255) / 256, n < 65535 ? n : 65535);          /* (fields beyond the grid's y limit: the kernel loops) */
    if (p->noise != 0)
        hipLaunchKernelGGL((k_margin<S, true>), grid, dim3(256), 0, c->stream,
                           *p, n, dst, c->fstride, d_state, c->d_jump16, c->d_jump1, c->d_skel, head, gap, tail, gap_magic);
    else
        hipLaunchKernelGGL((k_margin<S, false>), grid, dim3(256), 0, c->stream,
                           *p, n, dst, c->fstride, d_state, c->d_jump16, c->d_jump1, c->d_skel, head, gap, tail, gap_magic);
}


/* ... into padded signal lines (k_margin_pad) */
template <class S>
static void launch_margins_padded(crthip_ctx *c, const crthip_params *p, int n, signed char *dst, const crthip_state *d_state, const sig_layout &lay)
{
    ProfScope ps(c, CRTHIP_K_TEMPLATE);
    constexpr int CF = (S::HRES + 15) / 16;
    const int cl = (p->xo + 15) / 16;
    const int right = S::HRES - (p->xo + p->destw);
    const int cr = right > 0 ? (right + 15) / 16 : 0;
    const int act_per = cl + cr;
    const unsigned act_magic = act_per > 1 ? (unsigned) ((0x100000000ull + (unsigned) act_per - 1) / (unsigned) act_per) : 0u;
    const int per_field = p->yo * CF + p->desth * act_per + (S::VRES - p->yo - p->desth) * CF;
    const dim3 grid((per_field + 255) / 256, n < 65535 ? n : 65535);
    if (p->noise != 0)
        hipLaunchKernelGGL((k_margin_pad<S, true>), grid, dim3(256), 0, c->stream, *p, n, dst, lay.fstride, lay.shift, d_state,
                           c->d_jump16, c->d_jump1, c->d_skel, c->fstride, cl, cr, act_magic, lay.wrap);
    else
        hipLaunchKernelGGL((k_margin_pad<S, false>), grid, dim3(256), 0, c->stream, *p, n, dst, lay.fstride, lay.shift, d_state,
                           c->d_jump16, c->d_jump1, c->d_skel, c->fstride, cl, cr, act_magic, lay.wrap);
}

template <class S, bool FULL>
static void launch_active_any(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                              signed char *dst, const crthip_state *d_state, const sig_layout &lay)
{
    if (encoder_fast_ok<S>(p) && !c->force_exact) launch_active<S, FULL, true>(c, p, n, d_images, istride, dst, d_state, lay);
    else launch_active<S, FULL, false>(c, p, n, d_images, istride, dst, d_state, lay);
}

template <class S, bool FULL>
static int launch_encoder(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                          signed char *dst, const crthip_state *d_state, int nes_setup, const sig_layout &lay)
{
    /* (r6) The margins and the active rows are disjoint bytes and independent work: the margin kernel is bound by its stores (16 bytes
     * per lane, the copies behind padded lines on top), the active-video kernel by its arithmetic since its rows are aligned -- so for
     * batches that fill the chip the first runs on the context's internal stream BESIDE the second, fenced by events on both sides
     * (to the caller everything is still ordered on its stream; captured into a graph it is a fork and a join).  Not while a profile
     * is being taken (per-kernel durations are of kernels running alone), not on the internal stream itself (overlap chunks).
     * CRTHIP_MARGIN_SIDE=0: in sequence (A/B: profiles/r06_ab_margin_side.txt). */
    bool side = false;
    if (FULL) {
        side = c->margin_side && c->aux_stream && c->ev_mfork && !c->prof && n >= MARGIN_SIDE_MIN_FIELDS && c->stream != c->aux_stream;
        hipStream_t main_stream = c->stream;
        if (side) {
            if (hipEventRecord(c->ev_mfork, main_stream) != hipSuccess || hipStreamWaitEvent(c->aux_stream, c->ev_mfork, 0) != hipSuccess) side = false;
            else c->stream = c->aux_stream;
        }
        if (lay.pitch != S::HRES) launch_margins_padded<S>(c, p, n, dst, d_state, lay);
        else launch_margins<S>(c, p, n, dst, d_state);
        if (side) {
            c->stream = main_stream;
            if (hipEventRecord(c->ev_mjoin, c->aux_stream) != hipSuccess) return CRTHIP_E_HIP;
        }
    } else {
        constexpr int CHUNKS = (S::INPUT_SIZE + 15) / 16;
        ProfScope ps(c, CRTHIP_K_TEMPLATE);
        const int total = n * CHUNKS;
        hipLaunchKernelGGL((k_template<S>), dim3((total + 255) / 256), dim3(256), 0, c->stream,
                           *p, n, dst, c->fstride, d_state, nes_setup);
    }
    launch_active_any<S, FULL>(c, p, n, d_images, istride, dst, d_state, lay);
    if (side && hipStreamWaitEvent(c->stream, c->ev_mjoin, 0) != hipSuccess) return CRTHIP_E_HIP;
    return CRTHIP_OK;
}

/* after crt_modulate: ccf preset (crt_ntsc.c:325-329, crt_nes.c:196-200, crt_snes.c:238-240,321-325),
 * VHS resets (crt_ntscvhs.c:259,332-336) */
template <class S>
__global__ void k_encoder_state(const crthip_params P, int n_fields, crthip_state *state)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fields) return;
    crthip_state *st = state + f;
    if constexpr (S::LINE_ROWS) {
        /* iccf[(n + CCF_SHIFT) % VPER][t % CCS] is last written by the bottom-most line of its class; all lines of
         * a class write the same burst, so any line n of the class will do */
        for (int cls = 0; cls < S::VPER; cls++) {
            const int row = carrier_row<S>(cls, st->field, st->frame, st->aux);
            for (int k = 0; k < S::CCS; k++) {
                const int cb = P.burst[row][k];
                st->ccf[(cls + S::CCF_SHIFT) % S::VPER][k] = ((int) (signed char) ((S::BLANK + cb * S::BURST) >> 5)) << 7;
            }
        }
        if (!S::NES_TIMING) {
            st->field &= 1;
            st->frame &= 1;
        }
    } else {
        const int row = carrier_row<S>(0, st->field, st->frame, st->aux);
        for (int k = 0; k < 4; k++) {
            const int cb = P.burst[row][k];
            const int v = ((int) (signed char) ((S::BLANK + cb * S::BURST) >> 5)) << 7;
            st->ccf[0][k] = S::IS_VHS ? 0 : v;
        }
        if (S::IS_VHS) st->hsync = 0;
        st->field &= 1;
        st->frame &= 1;
    }
}

/* Encoder tables that depend on the parameter blob only (clean skeleton variants, NES sample table): (re)built on
 * the context's CURRENT stream when the inputs they derive from changed.  crthip_fieldpass calls this before it
 * forks onto its second stream, so the chunks only ever read the tables.
 *
 * Under stream capture (a caller recording the FIRST field-pass of a context, or the first one with new settings, into a
 * HIP graph) a launch on the capturing stream would only be recorded: the tables would not exist when the capture ends,
 * the cache would call them valid, and every replay would rebuild them.  So under capture the tables are built FOR REAL,
 * now, on a private stream that is not being captured (the thread's capture mode relaxed for the duration, as allocators
 * do).  The graph then holds only the per-call kernels, reading tables that exist.
 *
 * Which buffer: a set of tables that a captured graph may be reading is NEVER written again.  Every capture
 * marks the current set as referenced; a rebuild that finds the current set referenced (or happens under capture, where
 * kernels enqueued before may still be reading it) goes into a set of its own -- the spare allocated with the context first,
 * fresh allocations after that -- and the old set is retired, alive until crthip_destroy.  A graph captured with generation
 * g (crthip_table_generation) therefore replays with generation g's tables whatever the context was asked to do since; it
 * encodes with ITS settings, as a graph does.  Plain eager use (no capture ever) rebuilds in place as before. */
static signed char *table_target(crthip_ctx *c, signed char **cur, signed char **spare, size_t bytes, bool fresh_needed)
{
    if (!fresh_needed) return *cur;
    signed char *t = *spare;
    *spare = nullptr;
    if (!t && hipMalloc((void **) &t, bytes) != hipSuccess) return nullptr;
    if (c->n_retired < CRTHIP_MAX_RETIRED) c->retired[c->n_retired++] = *cur;      /* (beyond: leaked until the process ends; 64 settings changes under graphs) */
    *cur = t;
    return t;
}

int crt_run_encoder_prepare(crthip_ctx *c, const crthip_params *p, bool fused)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        /* every input of skeleton(): the burst table and, where the burst sits on the active lines only (NES timing,
         * crt_nes.c:173-178), the first active line; field / frame / dot crawl select the variant */
        const int border_key[4] = { p->flags & CRTHIP_F_NES_BORDER, p->nes_border_color, p->black_point, p->white_point };
        const bool border_same = !(p->flags & CRTHIP_F_NES_BORDER) ? c->skel_border[0] == 0
                                                                    : memcmp(c->skel_border, border_key, sizeof(border_key)) == 0;
        const bool need_skel = fused && (!c->skel_valid || memcmp(c->skel_burst, p->burst, sizeof(p->burst)) != 0 || c->skel_yo != p->yo || !border_same);
        bool need_nes = false;
        if constexpr (S::IS_NES) need_nes = !c->nes_tab_valid || c->nes_tab_black != p->black_point || c->nes_tab_white != p->white_point;

        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) { (void) hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        const bool capturing = cap == hipStreamCaptureStatusActive;
        if (capturing) {
            /* the graph being recorded reads the current buffer of every table that is NOT rebuilt below (a rebuilt one gets a fresh
             * buffer and its own flag there) */
            if (!need_skel) c->skel_captured = true;
            if (!need_nes) c->nes_captured = true;
        }
        if (!need_skel && !need_nes) return CRTHIP_OK;
        hipStream_t st = c->stream;
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        if (capturing) {
            if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return set_err(c, CRTHIP_E_HIP, "hipThreadExchangeStreamCaptureMode", hipGetLastError());
            if (!c->table_stream && hipStreamCreateWithFlags(&c->table_stream, hipStreamNonBlocking) != hipSuccess) {
                c->table_stream = nullptr;
                (void) hipThreadExchangeStreamCaptureMode(&mode);
                return set_err(c, CRTHIP_E_HIP, "stream for table builds under capture", hipGetLastError());
            }
            st = c->table_stream;
        }
        /* per table: a buffer a graph may read (or, under capture, that kernels enqueued before may still be reading) is not written */
        int rc = CRTHIP_OK;
        if (need_skel) {
            constexpr int SK_LANES = SKEL_VARIANTS * ((S::INPUT_SIZE + 15) / 16);
            signed char *dst = table_target(c, &c->d_skel, &c->d_skel_alt, (size_t) SKEL_VARIANTS * c->fstride, capturing || c->skel_captured);
            if (!dst) rc = set_err(c, CRTHIP_E_NOMEM, "hipMalloc skeleton tables", hipSuccess);
            else {
                if (!capturing) {
                    ProfScope ps(c, CRTHIP_K_TEMPLATE);
                    hipLaunchKernelGGL((k_skeleton<S>), dim3((SK_LANES + 255) / 256), dim3(256), 0, st, *p, dst, c->fstride);
                } else {
                    hipLaunchKernelGGL((k_skeleton<S>), dim3((SK_LANES + 255) / 256), dim3(256), 0, st, *p, dst, c->fstride);
                }
                memcpy(c->skel_burst, p->burst, sizeof(p->burst));
                c->skel_yo = p->yo;
                memcpy(c->skel_border, border_key, sizeof(border_key));
                c->skel_valid = true;
                c->skel_captured = capturing;                  /* the new buffer: referenced iff this very call is being recorded */
            }
        }
        if constexpr (S::IS_NES) {
            if (need_nes && rc == CRTHIP_OK) {
                signed char *dst = table_target(c, &c->d_nes_tab, &c->d_nes_tab_alt, NES_TAB_SIZE, capturing || c->nes_captured);
                if (!dst) rc = set_err(c, CRTHIP_E_NOMEM, "hipMalloc NES sample table", hipSuccess);
                else {
                    if (!capturing) {
                        ProfScope ps(c, CRTHIP_K_ACTIVE);
                        hipLaunchKernelGGL((k_nes_table<S>), dim3((NES_TAB_SIZE + 255) / 256), dim3(256), 0, st, *p, dst);
                    } else {
                        hipLaunchKernelGGL((k_nes_table<S>), dim3((NES_TAB_SIZE + 255) / 256), dim3(256), 0, st, *p, dst);
                    }
                    c->nes_tab_black = p->black_point;
                    c->nes_tab_white = p->white_point;
                    c->nes_tab_valid = true;
                    c->nes_captured = capturing;
                }
            }
        }
        c->table_gen++;
        if (capturing) {
            const hipError_t e = hipStreamSynchronize(st);
            (void) hipThreadExchangeStreamCaptureMode(&mode);
            if (e != hipSuccess) { c->skel_valid = false; c->nes_tab_valid = false; return set_err(c, CRTHIP_E_HIP, "table build under capture", e); }
        }
        if (rc != CRTHIP_OK) { c->skel_valid = false; c->nes_tab_valid = false; }
        return rc;
    });
}

int crt_run_encoder(crthip_ctx *c, const crthip_params *p, int n, const void *d_images, size_t istride,
                    signed char *dst, crthip_state *d_state, bool fused, int nes_setup, bool with_state, const sig_layout *lay)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        sig_layout flat;                                      /* the reference's layout: what every caller but the fused field-pass gets */
        flat.pitch = S::HRES; flat.shift = 0; flat.padv = 0; flat.wrap = 0; flat.fstride = c->fstride;
        const sig_layout &ly = lay && fused ? *lay : flat;
        int r = fused ? launch_encoder<S, true>(c, p, n, d_images, istride, dst, d_state, nes_setup, ly)
                      : launch_encoder<S, false>(c, p, n, d_images, istride, dst, d_state, nes_setup, ly);
        if (with_state) hipLaunchKernelGGL((k_encoder_state<S>), dim3((n + 63) / 64), dim3(64), 0, c->stream, *p, n, d_state);
        return r;
    });
}

/* the ccf preset as a launch of its own: sequence mode re-runs the sync chain from it (crt_host.hip) */
int crt_run_encoder_state(crthip_ctx *c, const crthip_params *p, int n, crthip_state *d_state)
{
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        hipLaunchKernelGGL((k_encoder_state<S>), dim3((n + 63) / 64), dim3(64), 0, c->stream, *p, n, d_state);
        return CRTHIP_OK;
    });
}
