/*
 * node_bench.c -- throughput of the field-pass path through the C ABI alone (include/crt_hip.h, crt_hip_node.h), plain C.
 *
 * S shards on ONE device (crthip_node_create with devices = {0, 0, ...}): every shard is a context with its own stream and
 * its own batch of n independent fields; the settings blob is broadcast once (RCCL), then K steps are enqueued round robin,
 * step k on shard k % S, no host synchronisation until the end -- the "batches in flight" of DESIGN.md section 6a, without
 * Python or torch in the process.  Synthetic input (LCG bytes, 64 distinct frames tiled to the batch), the headline
 * geometry by default.
 * usage: node_bench [shards=2] [fields per shard=4096] [steps=20] [w=640 h=480]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "crt_hip_node.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CRTHIP_OK) { fprintf(stderr, "node_bench: %s failed (%d)\n", #call, rc_); return 1; } } while (0)

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 2, n = argc > 2 ? atoi(argv[2]) : 4096, steps = argc > 3 ? atoi(argv[3]) : 20;
    const int w = argc > 5 ? atoi(argv[4]) : 640, h = argc > 5 ? atoi(argv[5]) : 480, uniq = n < 64 ? n : 64;
    const size_t istride = (size_t) w * (h + 1) * 4, ostride = (size_t) w * h * 4;
    int devices[16] = { 0 }, s, k, rep;
    crthip_node *node;
    crthip_params p, per_shard[16];
    void *d_img[16], *d_out[16];
    crthip_state *d_st[16], *st;
    unsigned char *img;
    unsigned lcg = 12345u;
    size_t i;

    if (S < 1 || S > 16 || n < 1 || steps < 1) return 2;
    CHECK(crthip_node_create(&node, S, devices, CRTHIP_SYSTEM_NTSC, 1));
    CHECK(crthip_params_default(&p, CRTHIP_SYSTEM_NTSC, 1));
    p.w = w; p.h = h; p.format = CRTHIP_FMT_BGRA; p.as_color = 1;
    p.outw = w; p.outh = h; p.out_format = CRTHIP_FMT_BGRA; p.scanlines = 1; p.noise = w <= 640 ? 24 : 0;
    p.flags |= CRTHIP_F_IMAGE_SPARE_ROW;
    CHECK(crthip_params_finalize(&p));
    CHECK(crthip_node_broadcast_params(node, &p, per_shard));          /* RCCL; once */

    img = (unsigned char *) malloc(istride * uniq);
    st = (crthip_state *) calloc(n, sizeof(crthip_state));
    if (!img || !st) return 2;
    for (i = 0; i < istride * uniq; i++) { lcg = lcg * 1664525u + 1013904223u; img[i] = (unsigned char) (lcg >> 8); }
    for (k = 0; k < n; k++) { st[k].field = k & 1; st[k].frame = ((k + 1) >> 1) & 1; st[k].rn = 194 + k; }
    for (s = 0; s < S; s++) {
        crthip_ctx *c = crthip_node_ctx(node, s);
        d_img[s] = crthip_malloc(c, istride * n); d_out[s] = crthip_malloc(c, ostride * n);
        d_st[s] = (crthip_state *) crthip_malloc(c, sizeof(crthip_state) * n);
        if (!d_img[s] || !d_out[s] || !d_st[s]) { fprintf(stderr, "node_bench: out of device memory\n"); return 2; }
        for (k = 0; k < n; k += uniq)
            CHECK(crthip_upload(c, (unsigned char *) d_img[s] + istride * k, img, istride * (size_t) (n - k < uniq ? n - k : uniq)));
        CHECK(crthip_upload(c, d_st[s], st, sizeof(crthip_state) * n));
        CHECK(crthip_reserve(c, n));
        CHECK(crthip_synchronize(c));
    }
    for (rep = 0; rep < 2; rep++) {                                    /* rep 0 = warm-up */
        const int K = rep ? steps : 3 * S;
        double t0, dt;
        CHECK(crthip_node_synchronize(node));
        t0 = now_s();
        for (k = 0; k < K; k++) {
            s = k % S;
            if (crthip_fieldpass(crthip_node_ctx(node, s), &per_shard[s], n, d_img[s], istride, d_out[s], ostride, d_st[s]) != CRTHIP_OK) {
                fprintf(stderr, "node_bench: crthip_fieldpass: %s\n", crthip_error_string(crthip_node_ctx(node, s))); return 1; }
        }
        CHECK(crthip_node_synchronize(node));
        dt = now_s() - t0;
        if (rep) printf("node_bench: %d shard(s) on device 0 x %d fields %dx%d, %d steps: %.0f frames/sec, %.3f ms per step (%d RCCL rank(s))\n",
                        S, n, w, h, steps, (double) steps * n / dt, 1e3 * dt / steps, crthip_node_rccl_ranks(node));
    }
    for (s = 0; s < S; s++) {
        crthip_ctx *c = crthip_node_ctx(node, s);
        crthip_free(c, d_img[s]); crthip_free(c, d_out[s]); crthip_free(c, d_st[s]);
    }
    crthip_node_destroy(node);
    free(img); free(st);
    return 0;
}
