/* crt_node.hip -- libcrthip_node.so: one process, all the GPUs of a node (include/crt_hip_node.h).
 *
 * Host logic only: the per-device work is libcrthip's (crthip_fieldpass, crthip_seq_*).  What is here: contiguous
 * sharding, the RCCL broadcast of the settings blob, and -- sequence mode -- the fixed point of the sync state over
 * the shards and the hand-over of the last picture from shard to shard. */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <thread>
#include <vector>

#include "crt_hip_node.h"

struct crthip_node {
    int n_shards;
    int system, pattern;
    std::vector<int> device;            /* per shard */
    std::vector<crthip_ctx *> ctx;      /* per shard */
    std::vector<hipStream_t> stream;    /* per shard (owned) */
    /* RCCL: one rank per DISTINCT device, in order of first appearance */
    int n_ranks;
    std::vector<int> rank_device;       /* per rank */
    std::vector<int> shard_rank;        /* per shard: rank of its device */
    std::vector<ncclComm_t> comm;       /* per rank */
    std::vector<hipStream_t> rstream;   /* per rank: the stream RCCL calls are enqueued on (= first shard's of that device) */
    std::vector<void *> d_blob;         /* per rank: device buffer of sizeof(crthip_params) */
    std::vector<unsigned *> vhs_hist;   /* per shard: the bound generator histories (VHS) */
    /* the last blob that went round and what every shard read back: a caller looping over batches with unchanged settings
     * pays for the broadcast (and its wait on shard 0's stream) once, and the calls stay asynchronous */
    bool have_last;
    crthip_params last_root;
    std::vector<crthip_params> last_blob;
    char err[320];
};

static int node_err(crthip_node *nd, int code, const char *what, const char *detail)
{
    if (nd) snprintf(nd->err, sizeof(nd->err), "%s: %s", what, detail ? detail : "");
    return code;
}
#define NODE_HIP(nd, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return node_err(nd, CRTHIP_E_HIP, #call, hipGetErrorString(e_)); } while (0)
#define NODE_NCCL(nd, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return node_err(nd, CRTHIP_E_HIP, #call, ncclGetErrorString(r_)); } while (0)
#define NODE_CRT(nd, s, call) do { int rc_ = (call); if (rc_ != CRTHIP_OK) return node_err(nd, rc_, #call, crthip_error_string((nd)->ctx[s])); } while (0)

extern "C" {

int crthip_node_create(crthip_node **out, int n_shards, const int *devices, int system, int chroma_pattern)
{
    if (!out || n_shards <= 0 || n_shards > 1024) return CRTHIP_E_ARG;
    *out = nullptr;
    const int ndev = crthip_device_count();
    if (ndev <= 0) return CRTHIP_E_NODEVICE;
    crthip_node *nd = new (std::nothrow) crthip_node();
    if (!nd) return CRTHIP_E_NOMEM;
    nd->n_shards = n_shards; nd->system = system; nd->pattern = chroma_pattern; nd->n_ranks = 0; nd->err[0] = 0;
    nd->vhs_hist.assign(n_shards, nullptr);
    nd->device.resize(n_shards); nd->ctx.assign(n_shards, nullptr); nd->stream.assign(n_shards, nullptr); nd->shard_rank.resize(n_shards);
    for (int s = 0; s < n_shards; s++) {
        const int d = devices ? devices[s] : s % ndev;
        if (d < 0 || d >= ndev) { crthip_node_destroy(nd); return CRTHIP_E_ARG; }
        nd->device[s] = d;
        int r = -1;
        for (int k = 0; k < nd->n_ranks; k++) if (nd->rank_device[k] == d) r = k;
        if (r < 0) { r = nd->n_ranks++; nd->rank_device.push_back(d); }
        nd->shard_rank[s] = r;
    }
    for (int s = 0; s < n_shards; s++) {
        int rc = crthip_create(&nd->ctx[s], nd->device[s], system, chroma_pattern);
        if (rc != CRTHIP_OK) { crthip_node_destroy(nd); return rc; }
        if (hipSetDevice(nd->device[s]) != hipSuccess ||
            hipStreamCreateWithFlags(&nd->stream[s], hipStreamNonBlocking) != hipSuccess) { crthip_node_destroy(nd); return CRTHIP_E_HIP; }
        crthip_set_stream(nd->ctx[s], nd->stream[s]);
    }
    nd->comm.assign(nd->n_ranks, nullptr); nd->rstream.assign(nd->n_ranks, nullptr); nd->d_blob.assign(nd->n_ranks, nullptr);
    for (int s = n_shards - 1; s >= 0; s--) nd->rstream[nd->shard_rank[s]] = nd->stream[s];
    if (ncclCommInitAll(nd->comm.data(), nd->n_ranks, nd->rank_device.data()) != ncclSuccess) {
        nd->comm.assign(nd->n_ranks, nullptr);
        crthip_node_destroy(nd);
        return CRTHIP_E_HIP;
    }
    for (int r = 0; r < nd->n_ranks; r++) {
        if (hipSetDevice(nd->rank_device[r]) != hipSuccess || hipMalloc(&nd->d_blob[r], sizeof(crthip_params)) != hipSuccess) {
            crthip_node_destroy(nd);
            return CRTHIP_E_NOMEM;
        }
    }
    *out = nd;
    return CRTHIP_OK;
}

void crthip_node_destroy(crthip_node *nd)
{
    if (!nd) return;
    for (size_t r = 0; r < nd->comm.size(); r++) if (nd->comm[r]) ncclCommDestroy(nd->comm[r]);
    for (size_t r = 0; r < nd->d_blob.size(); r++) if (nd->d_blob[r]) { hipSetDevice(nd->rank_device[r]); hipFree(nd->d_blob[r]); }
    for (int s = 0; s < nd->n_shards; s++) {
        if (nd->ctx[s]) { crthip_synchronize(nd->ctx[s]); crthip_set_stream(nd->ctx[s], nullptr); crthip_destroy(nd->ctx[s]); }
        if (nd->stream[s]) { hipSetDevice(nd->device[s]); hipStreamDestroy(nd->stream[s]); }
    }
    delete nd;
}

int crthip_node_shards(const crthip_node *nd) { return nd ? nd->n_shards : 0; }
int crthip_node_device(const crthip_node *nd, int s) { return nd && s >= 0 && s < nd->n_shards ? nd->device[s] : -1; }
int crthip_node_rccl_ranks(const crthip_node *nd) { return nd ? nd->n_ranks : 0; }
crthip_ctx *crthip_node_ctx(crthip_node *nd, int s) { return nd && s >= 0 && s < nd->n_shards ? nd->ctx[s] : nullptr; }
const char *crthip_node_error_string(const crthip_node *nd) { return nd ? nd->err : "null node"; }

int crthip_node_synchronize(crthip_node *nd)
{
    if (!nd) return CRTHIP_E_ARG;
    for (int s = 0; s < nd->n_shards; s++) {
        NODE_HIP(nd, hipSetDevice(nd->device[s]));
        NODE_HIP(nd, hipStreamSynchronize(nd->stream[s]));
    }
    return CRTHIP_OK;
}

int crthip_node_vhs_bind_history(crthip_node *nd, int s, unsigned *d_hist)
{
    if (!nd || s < 0 || s >= nd->n_shards) return CRTHIP_E_ARG;
    nd->vhs_hist[s] = d_hist;
    int rc = crthip_vhs_bind_history(nd->ctx[s], d_hist);
    return rc == CRTHIP_OK ? rc : node_err(nd, rc, "crthip_vhs_bind_history", crthip_error_string(nd->ctx[s]));
}

void crthip_node_shard_range(const crthip_node *nd, int n_total, int s, int *first, int *count)
{
    const int shards = nd ? nd->n_shards : 1;
    const int per = (n_total + shards - 1) / shards;
    int lo = s * per; if (lo > n_total) lo = n_total;
    int hi = lo + per; if (hi > n_total) hi = n_total;
    if (first) *first = lo;
    if (count) *count = hi - lo;
}

int crthip_node_broadcast_params(crthip_node *nd, const crthip_params *root, crthip_params *per_shard)
{
    if (!nd || !root || !per_shard) return CRTHIP_E_ARG;
    /* the root blob goes to the device of shard 0 = rank 0, RCCL carries it to every other rank's device */
    NODE_HIP(nd, hipSetDevice(nd->rank_device[0]));
    NODE_HIP(nd, hipMemcpyAsync(nd->d_blob[0], root, sizeof(*root), hipMemcpyHostToDevice, nd->rstream[0]));
    NODE_NCCL(nd, ncclGroupStart());
    for (int r = 0; r < nd->n_ranks; r++) {
        ncclResult_t e = ncclBroadcast(nd->d_blob[r], nd->d_blob[r], sizeof(crthip_params), ncclChar, 0, nd->comm[r], nd->rstream[r]);
        if (e != ncclSuccess) { ncclGroupEnd(); return node_err(nd, CRTHIP_E_HIP, "ncclBroadcast", ncclGetErrorString(e)); }
    }
    NODE_NCCL(nd, ncclGroupEnd());
    for (int s = 0; s < nd->n_shards; s++) {
        const int r = nd->shard_rank[s];
        NODE_HIP(nd, hipSetDevice(nd->device[s]));
        NODE_HIP(nd, hipMemcpyAsync(&per_shard[s], nd->d_blob[r], sizeof(crthip_params), hipMemcpyDeviceToHost, nd->rstream[r]));
    }
    for (int r = 0; r < nd->n_ranks; r++) {
        NODE_HIP(nd, hipSetDevice(nd->rank_device[r]));
        NODE_HIP(nd, hipStreamSynchronize(nd->rstream[r]));
    }
    return CRTHIP_OK;
}

/* the per-shard blobs for `p`: from the last round trip if `p` is byte for byte the blob that made it */
static int node_blobs(crthip_node *nd, const crthip_params *p, std::vector<crthip_params> &blob)
{
    if (nd->have_last && memcmp(&nd->last_root, p, sizeof(*p)) == 0) { blob = nd->last_blob; return CRTHIP_OK; }
    blob.resize(nd->n_shards);
    int rc = crthip_node_broadcast_params(nd, p, blob.data());
    if (rc) return rc;
    nd->last_root = *p; nd->last_blob = blob; nd->have_last = true;
    return CRTHIP_OK;
}

int crthip_node_fieldpass(crthip_node *nd, const crthip_params *p, int n_total, const void *const *d_images, size_t istride,
                          void *const *d_out, size_t ostride, crthip_state *const *d_state)
{
    if (!nd || !p || n_total <= 0 || !d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    std::vector<crthip_params> blob;
    int rc = node_blobs(nd, p, blob);
    if (rc) return rc;
    for (int s = 0; s < nd->n_shards; s++) {
        int first, cnt;
        crthip_node_shard_range(nd, n_total, s, &first, &cnt);
        if (cnt <= 0) continue;
        NODE_CRT(nd, s, crthip_fieldpass(nd->ctx[s], &blob[s], cnt, d_images[s], istride, d_out[s], ostride, d_state[s]));
    }
    return CRTHIP_OK;
}

/* a picture from shard a's device buffer to shard b's: same device = a copy on b's stream behind an event of a's;
 * different devices = RCCL send / recv over xGMI on the two shards' streams */
static int hand_over_picture(crthip_node *nd, int a, const void *src, int b, void *dst, size_t bytes)
{
    if (nd->device[a] == nd->device[b]) {
        hipEvent_t ev;
        NODE_HIP(nd, hipSetDevice(nd->device[a]));
        NODE_HIP(nd, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        NODE_HIP(nd, hipEventRecord(ev, nd->stream[a]));
        NODE_HIP(nd, hipStreamWaitEvent(nd->stream[b], ev, 0));
        NODE_HIP(nd, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nd->stream[b]));
        NODE_HIP(nd, hipEventDestroy(ev));
        return CRTHIP_OK;
    }
    const int ra = nd->shard_rank[a], rb = nd->shard_rank[b];
    NODE_NCCL(nd, ncclGroupStart());
    (void) hipSetDevice(nd->device[a]);
    ncclResult_t e1 = ncclSend(src, bytes, ncclChar, rb, nd->comm[ra], nd->stream[a]);
    (void) hipSetDevice(nd->device[b]);
    ncclResult_t e2 = ncclRecv(dst, bytes, ncclChar, ra, nd->comm[rb], nd->stream[b]);
    ncclResult_t e3 = ncclGroupEnd();
    if (e1 != ncclSuccess || e2 != ncclSuccess || e3 != ncclSuccess)
        return node_err(nd, CRTHIP_E_HIP, "ncclSend/ncclRecv", ncclGetErrorString(e1 != ncclSuccess ? e1 : e2 != ncclSuccess ? e2 : e3));
    return CRTHIP_OK;
}

/* Everything crthip_node_sequence holds beyond its arguments, released on EVERY way out (the early returns of
 * the NODE_* macros used to skip the reset of the shards' vhs_prechained flag -- a context left prechained silently skips
 * the rand() chain of any later sequence -- and leaked the scratch buffers).  The destructor first waits for all shard
 * streams: nothing that still reads the buffers may be in flight when they are freed. */
struct SeqGuard {
    crthip_node *nd;
    bool prechained = false;            /* some shard was switched to "histories already chained" */
    unsigned *hist_all = nullptr;       /* on device[0] */
    crthip_state *st_all = nullptr;     /* on device[0] */
    std::vector<void *> init;           /* per shard: the predecessor's last picture */
    explicit SeqGuard(crthip_node *n) : nd(n), init((size_t) n->n_shards, nullptr) {}
    void free_chain_scratch()
    {
        if (!hist_all && !st_all) return;
        (void) hipSetDevice(nd->device[0]);
        (void) hipStreamSynchronize(nd->stream[0]);
        if (hist_all) (void) hipFree(hist_all);
        if (st_all) (void) hipFree(st_all);
        hist_all = nullptr; st_all = nullptr;
    }
    ~SeqGuard()
    {
        for (int s = 0; s < nd->n_shards; s++) {
            (void) hipSetDevice(nd->device[s]);
            (void) hipStreamSynchronize(nd->stream[s]);
        }
        free_chain_scratch();
        for (int s = 0; s < nd->n_shards; s++) {
            if (init[s]) { (void) hipSetDevice(nd->device[s]); (void) hipFree(init[s]); }
            if (prechained && nd->system == CRTHIP_SYSTEM_NTSCVHS) (void) crthip_seq_vhs_prechained(nd->ctx[s], 0);
        }
    }
};

int crthip_node_sequence(crthip_node *nd, const crthip_params *p, int n_total, const void *const *d_images, size_t istride,
                         void *const *d_out, size_t ostride, const void *d_out_init, crthip_state *const *d_state, int *rounds_out)
{
    if (!nd || !p || n_total <= 0 || !d_images || !d_out || !d_state) return CRTHIP_E_ARG;
    const int S = nd->n_shards;
    const bool vhs = nd->system == CRTHIP_SYSTEM_NTSCVHS && !(p->flags & CRTHIP_F_VHS_LCG_NOISE);
    std::vector<crthip_params> blob;
    int rc = node_blobs(nd, p, blob);
    if (rc) return rc;
    std::vector<int> first(S), cnt(S);
    int last_shard = -1;
    for (int s = 0; s < S; s++) {
        crthip_node_shard_range(nd, n_total, s, &first[s], &cnt[s]);
        if (cnt[s] > 0) last_shard = s;
    }
    SeqGuard g(nd);                     /* from here on every return releases what the call holds */
    /* the set's state before field 0 */
    crthip_state st0;
    NODE_HIP(nd, hipSetDevice(nd->device[0]));
    NODE_HIP(nd, hipMemcpyAsync(&st0, d_state[0], sizeof(st0), hipMemcpyDeviceToHost, nd->stream[0]));
    NODE_HIP(nd, hipStreamSynchronize(nd->stream[0]));

    if (vhs) {
        /* one rand() stream per video: shard 0's device walks it for all n_total fields, then every shard gets the generator
         * states (and drawn aberration heights) of its own fields */
        for (int s = 0; s < S; s++)
            if (cnt[s] > 0 && !nd->vhs_hist[s]) return node_err(nd, CRTHIP_E_ARG, "crthip_node_sequence", "VHS: bind every shard's histories (crthip_node_vhs_bind_history)");
        const bool draw = (p->flags & CRTHIP_F_VHS_DRAW_ABERRATION) != 0;
        NODE_HIP(nd, hipSetDevice(nd->device[0]));
        NODE_HIP(nd, hipMalloc((void **) &g.hist_all, sizeof(unsigned) * 32 * (size_t) n_total));
        NODE_HIP(nd, hipMalloc((void **) &g.st_all, sizeof(crthip_state) * (size_t) n_total));
        NODE_HIP(nd, hipMemsetAsync(g.st_all, 0, sizeof(crthip_state) * (size_t) n_total, nd->stream[0]));
        NODE_HIP(nd, hipMemcpyAsync(g.hist_all, nd->vhs_hist[0], sizeof(unsigned) * 32, hipMemcpyDeviceToDevice, nd->stream[0]));
        NODE_CRT(nd, 0, crthip_vhs_bind_history(nd->ctx[0], g.hist_all));
        rc = crthip_vhs_chain(nd->ctx[0], n_total, g.st_all, draw);
        crthip_vhs_bind_history(nd->ctx[0], nd->vhs_hist[0]);
        if (rc) return node_err(nd, rc, "crthip_vhs_chain", crthip_error_string(nd->ctx[0]));
        NODE_HIP(nd, hipStreamSynchronize(nd->stream[0]));          /* the chain is complete: the scatter below reads it */
        for (int s = 0; s < S; s++) {
            if (cnt[s] <= 0) continue;
            /* Scatter on the RECEIVING shard's stream (the shard streams are non-blocking: a copy on the null stream would
             * not be ordered against them), so the shard's encoder, enqueued behind it, sees the histories.  Between devices:
             * peer copy or staged through the host, the runtime's choice. */
            NODE_HIP(nd, hipSetDevice(nd->device[s]));
            NODE_HIP(nd, hipMemcpyAsync(nd->vhs_hist[s], g.hist_all + 32 * (size_t) first[s], sizeof(unsigned) * 32 * (size_t) cnt[s],
                                        hipMemcpyDefault, nd->stream[s]));
            if (draw)
                NODE_HIP(nd, hipMemcpy2DAsync(&d_state[s][0].aux, sizeof(crthip_state), &g.st_all[first[s]].aux, sizeof(crthip_state),
                                              sizeof(int), (size_t) cnt[s], hipMemcpyDefault, nd->stream[s]));
            g.prechained = true;
            NODE_CRT(nd, s, crthip_seq_vhs_prechained(nd->ctx[s], 1));
        }
        /* the scratch is read by those copies: wait for every receiving stream before it goes */
        for (int s = 0; s < S; s++)
            if (cnt[s] > 0) { NODE_HIP(nd, hipSetDevice(nd->device[s])); NODE_HIP(nd, hipStreamSynchronize(nd->stream[s])); }
        g.free_chain_scratch();
    }

    /* phase 1 + the first sync round: every shard on its own host thread (crthip_seq_sync reads its flag back, i.e.
     * blocks its caller) */
    std::vector<int> hs_in(S, st0.hsync), vs_in(S, st0.vsync), hs_out(S, 0), vs_out(S, 0), rcs(S, CRTHIP_OK);
    auto run_sync = [&](int s, bool encode) {
        if (cnt[s] <= 0) { hs_out[s] = hs_in[s]; vs_out[s] = vs_in[s]; return; }
        int r = CRTHIP_OK;
        if (encode) r = crthip_seq_encode(nd->ctx[s], &blob[s], cnt[s], first[s], st0.rn, d_images[s], istride, d_state[s]);
        if (r == CRTHIP_OK) r = crthip_seq_sync(nd->ctx[s], &blob[s], cnt[s], d_state[s], hs_in[s], vs_in[s], &hs_out[s], &vs_out[s], nullptr);
        rcs[s] = r;
    };
    int rounds = 0;
    std::vector<char> dirty(S, 1);
    for (;;) {
        rounds++;
        std::vector<std::thread> th;
        for (int s = 0; s < S; s++) if (dirty[s]) th.emplace_back(run_sync, s, rounds == 1);
        for (auto &t : th) t.join();
        for (int s = 0; s < S; s++) if (rcs[s] != CRTHIP_OK) return node_err(nd, rcs[s], "sequence shard", crthip_error_string(nd->ctx[s]));
        /* exchange: shard s + 1 starts from what shard s ended with (8 bytes per shard; the host has them already) */
        bool any = false;
        for (int s = 0; s < S; s++) dirty[s] = 0;
        for (int s = 1; s < S; s++) {
            if (hs_in[s] != hs_out[s - 1] || vs_in[s] != vs_out[s - 1]) {
                hs_in[s] = hs_out[s - 1]; vs_in[s] = vs_out[s - 1];
                dirty[s] = 1; any = true;
            }
        }
        if (!any) break;
        if (rounds > S + 1)             /* after round k shards 0 .. k-1 are final: more than S + 1 rounds cannot happen */
            return node_err(nd, CRTHIP_E_HIP, "crthip_node_sequence", "the sync state over the shards did not converge");
    }
    if (rounds_out) *rounds_out = rounds;

    /* decode everywhere (asynchronous) */
    for (int s = 0; s < S; s++)
        if (cnt[s] > 0) NODE_CRT(nd, s, crthip_seq_decode(nd->ctx[s], &blob[s], cnt[s], d_out[s], ostride, d_state[s]));

    /* the output picture across the seams */
    const size_t pic = (size_t) blob[0].outw * blob[0].outh * blob[0].out_bpp;
    std::vector<void *> &init = g.init;
    auto last_picture = [&](int s) { return (const unsigned char *) d_out[s] + (size_t) (cnt[s] - 1) * ostride; };
    if (!blob[0].blend) {
        /* every shard weaves with a placeholder (zeros) at once; then, down the chain, only the rows nobody in the shard
         * wrote are patched from the predecessor's last picture */
        for (int s = 0; s < S; s++)
            if (cnt[s] > 0) NODE_CRT(nd, s, crthip_seq_weave(nd->ctx[s], &blob[s], cnt[s], d_out[s], ostride, s == 0 ? d_out_init : nullptr, 0));
    }
    int prev = -1;
    for (int s = 0; s < S; s++) {
        if (cnt[s] <= 0) continue;
        if (prev >= 0) {
            if (hipSetDevice(nd->device[s]) != hipSuccess || hipMalloc(&init[s], pic) != hipSuccess)
                return node_err(nd, CRTHIP_E_NOMEM, "hipMalloc", "picture hand-over buffer");
            rc = hand_over_picture(nd, prev, last_picture(prev), s, init[s], pic);
            if (rc) return rc;
            NODE_CRT(nd, s, crthip_seq_weave(nd->ctx[s], &blob[s], cnt[s], d_out[s], ostride, init[s], blob[0].blend ? 0 : 1));
        } else if (blob[0].blend) {
            NODE_CRT(nd, s, crthip_seq_weave(nd->ctx[s], &blob[s], cnt[s], d_out[s], ostride, d_out_init, 0));
        }
        prev = s;
    }
    (void) last_shard;
    return crthip_node_synchronize(nd);          /* (the guard then frees the hand-over buffers and clears the VHS flag) */
}

}  /* extern "C" */
