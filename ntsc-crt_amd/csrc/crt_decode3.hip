/* crt_decode3.hip -- the lane-per-scanline decoder of a CRT_DO_BLOOM build (kernel: crt_decode_lane.h).  See crt_dev.h.
 *
 * With bloom every scanline has its own width line_w (crt_core.c:518-519, the leaky integrator over the beam energy that
 * k_bloom walks), hence its own resampler step dx and start scanL (:521-522).  The lane-per-scanline kernel keeps its pixel
 * schedule on the scalar unit, which needs ONE geometry per wave.  Both depend on line_w alone, and line_w takes few values:
 * prev_e stays inside [-12903, 39322] (term of :518 in [-504, 1536] for max_e >= 128 AV_LEN, i.e. noise >= 0 -- enforced by
 * crt_setup.c -- and |sum| <= 128 AV_LEN; fixed point term * 128 / 5), so line_w - AV_LEN * 112 / 128 lies in [-26, 76].
 * So the lines of the batch are counting-sorted by line_w and every wave decodes 64 lines of EQUAL geometry, wherever in the
 * batch they come from: source and destination of a lane are per-lane addresses already (the cooperative tiles of
 * crt_decode_lane.h move row pieces, not rows of one picture).  Lines nobody sees (nrows == 0) drop out of the sort.
 * The sort is an optimisation, not a correctness condition: a wave that holds several geometries after all (a line table
 * that did not come from k_bloom, through crthip_decode) is decoded in rounds, one geometry at a time.
 *
 *   k_bloom_count    line_w histogram of the batch (LDS histogram per 256 lines, then one global atomic per bucket)
 *   k_bloom_scatter  bucket starts = prefix sum of the counts rounded up to whole waves; line index -> its slot
 *   k_decode<BLOOM>  one wave per 64 slots (-1 = padding)
 */
#include "crt_decode_lane.h"

#define BLOOM_BUCKETS 256
#define BLOOM_KEY_BIAS 64                  /* bucket of line_w = AV_LEN * 112 / 128 */

/* the bucket of a line: line_w back from what k_bloom left in the line table.  scanl gives line_w >> 1 (:522); the low bit
 * decides dx (:521) unless outw is so large that both give the same step -- then both share a bucket rightly. */
template <class S>
__device__ __forceinline__ int bloom_key(const crthip_line &lp, int outw)
{
    if ((lp.nrows & CRTHIP_LINE_NROWS_MASK) == 0) return -1;
    const int half = S::AV_LEN / 2 + 8 - (lp.scanl >> 12);
    const int line_w = 2 * half + (lp.dx != ((2 * half) << 12) / outw ? 1 : 0);
    const int key = line_w - S::AV_LEN * 112 / 128 + BLOOM_KEY_BIAS;
    return key < 0 ? 0 : key >= BLOOM_BUCKETS ? BLOOM_BUCKETS - 1 : key;       /* never clamps for k_bloom's tables, see above */
}

/* lines per workgroup of the two sort kernels: the global atomics are one per bucket and workgroup, on few hot addresses */
#define BLOOM_SORT_ROUNDS 16

template <class S>
__global__ void __launch_bounds__(256)
k_bloom_count(int total, int outw, const crthip_line *__restrict__ lines, int *__restrict__ hist)
{
    __shared__ int s_hist[BLOOM_BUCKETS];
    const int t = threadIdx.x;
    s_hist[t] = 0;
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < BLOOM_SORT_ROUNDS; r++) {
        const int gid = (blockIdx.x * BLOOM_SORT_ROUNDS + r) * 256 + t;
        const int key = gid < total ? bloom_key<S>(lines[gid], outw) : -1;
        if (key >= 0) atomicAdd(&s_hist[key], 1);
    }
    __syncthreads();
    if (s_hist[t]) atomicAdd(&hist[t], s_hist[t]);
}

template <class S>
__global__ void __launch_bounds__(256)
k_bloom_scatter(int total, int outw, const crthip_line *__restrict__ lines, const int *__restrict__ hist,
                int *__restrict__ cursor, int *__restrict__ perm)
{
    __shared__ int s_scan[BLOOM_BUCKETS], s_cnt[BLOOM_BUCKETS], s_base[BLOOM_BUCKETS];
    const int t = threadIdx.x;
    const int mine = (hist[t] + 63) & ~63;                 /* whole waves per bucket */
    s_scan[t] = mine;
    s_cnt[t] = 0;
    __syncthreads();
    for (int d = 1; d < BLOOM_BUCKETS; d <<= 1) {          /* inclusive scan */
        const int v = t >= d ? s_scan[t - d] : 0;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    /* rank of every line among the lines of its bucket in this workgroup; then one reservation per bucket */
    int key[BLOOM_SORT_ROUNDS], rank[BLOOM_SORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < BLOOM_SORT_ROUNDS; r++) {
        const int gid = (blockIdx.x * BLOOM_SORT_ROUNDS + r) * 256 + t;
        key[r] = gid < total ? bloom_key<S>(lines[gid], outw) : -1;
        rank[r] = key[r] >= 0 ? atomicAdd(&s_cnt[key[r]], 1) : 0;
    }
    __syncthreads();
    if (s_cnt[t]) s_base[t] = s_scan[t] - mine + atomicAdd(&cursor[t], s_cnt[t]);      /* exclusive start + reservation */
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BLOOM_SORT_ROUNDS; r++) {
        const int gid = (blockIdx.x * BLOOM_SORT_ROUNDS + r) * 256 + t;
        if (key[r] >= 0) perm[s_base[key[r]] + rank[r]] = gid;
    }
}

/* slots of the sorted order for n fields: every bucket may end in a partial wave */
/* histogram and cursors to 0, every slot to "no line" (-1).  A kernel of our own instead of two hipMemsetAsync calls: with the two
 * memsets -- adjacent ranges, different fill values -- as nodes of a captured graph, the SECOND replay of a bloom field-pass died
 * with a write fault (ROCm 7.2; every eager launch and the first replay were fine, and so was the same graph with this kernel:
 * tools/debug/graph_bloom.py, gpurun_out/r4s18-r4s20); it depended on the box, the test had passed for a round.  One launch is
 * cheaper than two as well. */
__global__ void __launch_bounds__(256)
k_bloom_clear(int *__restrict__ hist2, int n_hist2, int *__restrict__ perm, size_t slots)
{
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, stride = (size_t) gridDim.x * 256;
    if (gid < (size_t) n_hist2) hist2[gid] = 0;
    for (size_t i = gid; i < slots; i += stride) perm[i] = -1;
}

static size_t bloom_slots(const crthip_ctx *c, int n) { return (size_t) n * c->sd.lines + 64 * BLOOM_BUCKETS; }

/* The sort's scratch (histogram, cursors, slot -> line) for n fields.  crthip_reserve sizes it with the rest of the workspace,
 * so that a field-pass allocates nothing (graph capture); a stage-level crthip_decode of more fields than reserved grows it. */
int crt_reserve_bloom(crthip_ctx *c, int n)
{
    const size_t need = sizeof(int) * (bloom_slots(c, n) + 2 * BLOOM_BUCKETS);
    if (need <= c->bloom_cap) return CRTHIP_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_bloom) hipFree(c->d_bloom);
    c->d_bloom = 0; c->bloom_cap = 0;
    if (hipMalloc((void **) &c->d_bloom, need) != hipSuccess) return set_err(c, CRTHIP_E_NOMEM, "hipMalloc bloom sort", hipSuccess);
    c->bloom_cap = need;
    return CRTHIP_OK;
}

int crt_run_decode_bloom_lanes(crthip_ctx *c, const crthip_params *p, int n, const signed char *d_inp,
                               const crthip_line *d_lines, void *d_out, size_t ostride, int min_tier, size_t fstride)
{
    const int total = n * c->sd.lines;
    const size_t slots = bloom_slots(c, n);
    const int rc_ = crt_reserve_bloom(c, n);
    if (rc_) return rc_;
    int *hist = c->d_bloom, *cursor = hist + BLOOM_BUCKETS, *perm = cursor + BLOOM_BUCKETS;
    const bool wide = c->px_tile ? c->px_tile >= 32 : p->outw >= 1280;
    const unsigned span = (unsigned) p->outh + p->v_fac;
    const int passes = span >= (unsigned) c->sd.lines ? 1 : (int) (((unsigned) c->sd.lines + span - 1) / (span ? span : 1));
    /* tier 1 (carriers << 7 beyond 24 bits) exists for the NES, which has no bloom build, and for the PV-1000, whose default
     * saturation sits there; the 4-sample systems send such lines to tier 2 */
    if (min_tier == 1 && c->sd.cc_samples != 5) min_tier = 2;
    return dispatch_system(c->system, c->pattern, [&](auto tag) {
        using S = decltype(tag);
        if constexpr (S::NES_TIMING) {
            return set_err(c, CRTHIP_E_ARG, "no bloom build of the NES systems", hipSuccess);
        } else {
            ProfScope ps(c, CRTHIP_K_DECODE);
            {
                const size_t want = (slots + 255) / 256;
                static_assert(2 * BLOOM_BUCKETS <= 256 * 4, "the first blocks clear the histogram and the cursors");
                hipLaunchKernelGGL(k_bloom_clear, dim3((unsigned) (want < 4 ? 4 : want > 2048 ? 2048 : want)), dim3(256), 0, c->stream,
                                   hist, 2 * BLOOM_BUCKETS, perm, slots);
            }
            const dim3 sgrid((total + 256 * BLOOM_SORT_ROUNDS - 1) / (256 * BLOOM_SORT_ROUNDS)), sblock(256);
            hipLaunchKernelGGL((k_bloom_count<S>), sgrid, sblock, 0, c->stream, total, p->outw, d_lines, hist);
            hipLaunchKernelGGL((k_bloom_scatter<S>), sgrid, sblock, 0, c->stream, total, p->outw, d_lines, hist, cursor, perm);
            const dim3 grid((unsigned) (slots / 64)), block(64);
            unsigned char *o = (unsigned char *) d_out;
            for (int rank = 0; rank < passes; rank++) {
#define CRTHIP_LAUNCH_BLOOM(TG, B3) \
    do { if (wide) hipLaunchKernelGGL((k_decode<S, TG, B3, 32, true>), grid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, min_tier, rank, (const int *) perm, 0, 0); \
         else hipLaunchKernelGGL((k_decode<S, TG, B3, 16, true>), grid, block, 0, c->stream, *p, n, d_inp, fstride, d_lines, o, ostride, min_tier, rank, (const int *) perm, 0, 0); } while (0)
                /* tier groups as in crt_run_decode: 0 = tiers 0 / 1 (tier 1 only ever holds lines of the 5-sample system here), 1 = 2 / 3 */
                if (p->out_bpp == 3) {
                    if (min_tier <= 1) CRTHIP_LAUNCH_BLOOM(0, true);
                    CRTHIP_LAUNCH_BLOOM(1, true);
                } else {
                    if (min_tier <= 1) CRTHIP_LAUNCH_BLOOM(0, false);
                    CRTHIP_LAUNCH_BLOOM(1, false);
                }
#undef CRTHIP_LAUNCH_BLOOM
            }
            return CRTHIP_OK;
        }
    });
}
