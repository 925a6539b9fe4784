/*
 * node_probe.c -- TEST PROGRAM for include/crt_hip_node.h (libcrthip_node.so), plain C.
 *
 * Runs the multi-shard entry points against the single-context ones of the same library on the same inputs and
 * compares every output byte and every state word (the single-context path's parity with the reference is what
 * tests/test_gpu_parity.py establishes):
 *   crthip_node_fieldpass  vs  crthip_fieldpass      (independent frames, contiguous blocks)
 *   crthip_node_sequence   vs  crthip_sequence       (one video cut over the shards; blend 0 and 1; heavy noise so that
 *                                                     the sync state really travels across the seams)
 *   the same two for the VHS build (CRT_SYSTEM_NTSCVHS, what extra/video_convert.c is built for): per-field rand()
 *   generators in batch mode, ONE rand() stream per video -- aberration heights drawn from it -- in sequence mode
 * Shard layouts: as many shards as the box has devices (1 on a gpurun box), and 2 / 3 shards sharing device 0 --
 * the whole cross-shard protocol on ONE GPU; RCCL runs with crthip_node_rccl_ranks() ranks either way.
 * usage: node_probe [n_fields]      exit code 0 = everything identical
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crt_hip_node.h"

#define W 320
#define H 240
#define OW 400
#define OH 300

static unsigned lcg_state = 12345u;
static unsigned char lcg_byte(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (unsigned char) (lcg_state >> 8); }

#define CHECK(call) do { int rc_ = (call); if (rc_ != CRTHIP_OK) { fprintf(stderr, "node_probe: %s failed (%d)\n", #call, rc_); return 1; } } while (0)

static void field_parity(int k, int *field, int *frame) { *field = k & 1; *frame = ((k + 1) >> 1) & 1; }   /* video_convert.c:261-267 */

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 11;
    const size_t istride = (size_t) W * (H + 1) * 4, ostride = (size_t) OW * OH * 4;
    unsigned char *images = (unsigned char *) malloc(istride * n), *init = (unsigned char *) malloc(ostride);
    unsigned char *want = (unsigned char *) malloc(ostride * n), *got = (unsigned char *) malloc(ostride * n);
    crthip_state *st_in = (crthip_state *) calloc(n, sizeof(crthip_state)), *st_want = (crthip_state *) calloc(n, sizeof(crthip_state)),
                 *st_got = (crthip_state *) calloc(n, sizeof(crthip_state));
    crthip_ctx *one;
    crthip_params p;
    unsigned *hist = (unsigned *) calloc((size_t) n * 32, sizeof(unsigned)), *hist_want = (unsigned *) calloc((size_t) n * 32, sizeof(unsigned)),
             *hist_got = (unsigned *) calloc((size_t) n * 32, sizeof(unsigned));
    int layouts[3][4] = { { 0, 0, 0, 0 }, { 2, 0, 0, 0 }, { 3, 0, 0, 0 } };   /* [shards, devices...]; 0 shards = one per device */
    int mode, lay, k, failures = 0;
    size_t i;

    if (!images || !init || !want || !got || !st_in || !st_want || !st_got || !hist || !hist_want || !hist_got) return 2;
    for (i = 0; i < istride * n; i++) images[i] = lcg_byte();
    for (i = 0; i < ostride; i++) init[i] = lcg_byte();
    one = 0;

    for (mode = 0; mode < 5; mode++) {          /* 0 fieldpass, 1 sequence, 2 sequence with blend; VHS: 3 fieldpass, 4 sequence */
        void *d_img, *d_out, *d_st, *d_init, *d_hist = 0;
        const int vhs = mode >= 3, seq = mode == 1 || mode == 2 || mode == 4;
        const int sysid = vhs ? CRTHIP_SYSTEM_NTSCVHS : CRTHIP_SYSTEM_NTSC;
        if (one) crthip_destroy(one);
        CHECK(crthip_create(&one, 0, sysid, 1));
        CHECK(crthip_params_default(&p, sysid, 1));
        p.w = W; p.h = H; p.format = CRTHIP_FMT_BGRA; p.as_color = 1;
        p.outw = OW; p.outh = OH; p.out_format = CRTHIP_FMT_BGRA; p.scanlines = mode == 2 ? 0 : 1; p.blend = mode == 2;
        p.noise = vhs ? 12 : (mode == 0 ? 24 : 120);
        p.flags |= CRTHIP_F_IMAGE_SPARE_ROW;
        if (mode == 4) p.flags |= CRTHIP_F_VHS_DRAW_ABERRATION;
        CHECK(crthip_params_finalize(&p));
        for (k = 0; k < n; k++) {
            memset(&st_in[k], 0, sizeof(st_in[k]));
            field_parity(k, &st_in[k].field, &st_in[k].frame);
            st_in[k].rn = mode == 0 ? 194 + k : 0;
        }
        st_in[0].rn = 194; st_in[0].hsync = seq ? 7 : 0; st_in[0].vsync = seq ? 2 : 0;
        for (k = 0; k < n; k++) {               /* VHS: per-field generators (batch) / the video's generator in entry 0 (sequence) */
            if (vhs && (k == 0 || !seq)) CHECK(crthip_vhs_history_from_seed(4242u + 77u * (unsigned) k, hist + 32 * k));
            st_in[k].aux = vhs && !seq ? (k % 3) * 7 : 0;      /* aberration heights given by the caller in batch mode */
        }

        /* single context */
        d_img = crthip_malloc(one, istride * n); d_out = crthip_malloc(one, ostride * n);
        d_st = crthip_malloc(one, sizeof(crthip_state) * n); d_init = crthip_malloc(one, ostride);
        if (!d_img || !d_out || !d_st || !d_init) return 2;
        CHECK(crthip_upload(one, d_img, images, istride * n));
        CHECK(crthip_upload(one, d_st, st_in, sizeof(crthip_state) * n));
        CHECK(crthip_upload(one, d_init, init, ostride));
        CHECK(crthip_memset(one, d_out, 0, ostride * n));
        if (vhs) {
            d_hist = crthip_malloc(one, sizeof(unsigned) * 32 * n);
            if (!d_hist) return 2;
            CHECK(crthip_upload(one, d_hist, hist, sizeof(unsigned) * 32 * n));
            CHECK(crthip_vhs_bind_history(one, (unsigned *) d_hist));
        }
        if (!seq) CHECK(crthip_fieldpass(one, &p, n, d_img, istride, d_out, ostride, (crthip_state *) d_st));
        else CHECK(crthip_sequence(one, &p, n, d_img, istride, d_out, ostride, d_init, (crthip_state *) d_st, 0));
        CHECK(crthip_synchronize(one));
        CHECK(crthip_download(one, want, d_out, ostride * n));
        CHECK(crthip_download(one, st_want, d_st, sizeof(crthip_state) * n));
        if (vhs) { CHECK(crthip_download(one, hist_want, d_hist, sizeof(unsigned) * 32 * n)); crthip_free(one, d_hist); }
        crthip_free(one, d_img); crthip_free(one, d_out); crthip_free(one, d_st); crthip_free(one, d_init);

        for (lay = 0; lay < 3; lay++) {
            crthip_node *node;
            const int shards = layouts[lay][0] ? layouts[lay][0] : crthip_device_count();
            const void *s_img[64]; void *s_out[64]; crthip_state *s_st[64]; unsigned *s_hist[64];
            void *n_init = 0;
            int s, rounds = 0, bad_px = 0, bad_st = 0;
            if (shards > 64) continue;
            CHECK(crthip_node_create(&node, shards, layouts[lay][0] ? &layouts[lay][1] : 0, sysid, 1));
            for (s = 0; s < shards; s++) {
                int first, cnt;
                crthip_ctx *c = crthip_node_ctx(node, s);
                crthip_node_shard_range(node, n, s, &first, &cnt);
                s_img[s] = 0; s_out[s] = 0; s_st[s] = 0; s_hist[s] = 0;
                if (cnt <= 0) continue;
                if (vhs) {
                    s_hist[s] = (unsigned *) crthip_malloc(c, sizeof(unsigned) * 32 * cnt);
                    if (!s_hist[s]) return 2;
                    CHECK(crthip_upload(c, s_hist[s], hist + 32 * first, sizeof(unsigned) * 32 * cnt));
                    CHECK(crthip_node_vhs_bind_history(node, s, s_hist[s]));
                }
                s_img[s] = crthip_malloc(c, istride * cnt); s_out[s] = crthip_malloc(c, ostride * cnt);
                s_st[s] = (crthip_state *) crthip_malloc(c, sizeof(crthip_state) * cnt);
                if (!s_img[s] || !s_out[s] || !s_st[s]) return 2;
                CHECK(crthip_upload(c, (void *) s_img[s], images + istride * first, istride * cnt));
                CHECK(crthip_upload(c, s_st[s], st_in + first, sizeof(crthip_state) * cnt));
                CHECK(crthip_memset(c, s_out[s], 0, ostride * cnt));
                CHECK(crthip_synchronize(c));
            }
            if (seq) {
                n_init = crthip_malloc(crthip_node_ctx(node, 0), ostride);
                CHECK(crthip_upload(crthip_node_ctx(node, 0), n_init, init, ostride));
            }
            if (!seq) {
                if (crthip_node_fieldpass(node, &p, n, s_img, istride, s_out, ostride, s_st) != CRTHIP_OK) {
                    fprintf(stderr, "node_probe: crthip_node_fieldpass: %s\n", crthip_node_error_string(node)); return 1; }
            } else {
                if (crthip_node_sequence(node, &p, n, s_img, istride, s_out, ostride, n_init, s_st, &rounds) != CRTHIP_OK) {
                    fprintf(stderr, "node_probe: crthip_node_sequence: %s\n", crthip_node_error_string(node)); return 1; }
            }
            CHECK(crthip_node_synchronize(node));
            for (s = 0; s < shards; s++) {
                int first, cnt;
                crthip_ctx *c = crthip_node_ctx(node, s);
                crthip_node_shard_range(node, n, s, &first, &cnt);
                if (cnt <= 0) continue;
                CHECK(crthip_download(c, got + ostride * first, s_out[s], ostride * cnt));
                CHECK(crthip_download(c, st_got + first, s_st[s], sizeof(crthip_state) * cnt));
                if (vhs) { CHECK(crthip_download(c, hist_got + 32 * first, s_hist[s], sizeof(unsigned) * 32 * cnt)); crthip_free(c, s_hist[s]); }
                crthip_free(c, (void *) s_img[s]); crthip_free(c, s_out[s]); crthip_free(c, s_st[s]);
            }
            if (n_init) crthip_free(crthip_node_ctx(node, 0), n_init);
            for (k = 0; k < n; k++) {
                if (memcmp(got + ostride * k, want + ostride * k, ostride) != 0) bad_px++;
                if (st_got[k].hsync != st_want[k].hsync || st_got[k].vsync != st_want[k].vsync || st_got[k].rn != st_want[k].rn ||
                    st_got[k].aux != st_want[k].aux || memcmp(st_got[k].ccf, st_want[k].ccf, sizeof(st_want[k].ccf)) != 0 ||
                    (vhs && memcmp(hist_got + 32 * k, hist_want + 32 * k, 31 * sizeof(unsigned)) != 0)) bad_st++;
            }
            printf("node_probe: mode %d (%s) %d fields over %d shard(s), %d RCCL rank(s)%s: %d pictures, %d states differ\n", mode,
                   mode == 0 ? "fieldpass" : mode == 1 ? "sequence" : mode == 2 ? "sequence+blend" : mode == 3 ? "VHS fieldpass" : "VHS sequence",
                   n, shards, crthip_node_rccl_ranks(node),
                   seq ? (rounds == 1 ? ", 1 exchange round" : ", several exchange rounds") : "", bad_px, bad_st);
            if (seq) printf("node_probe:   exchange rounds: %d\n", rounds);
            failures += bad_px + bad_st;
            crthip_node_destroy(node);
        }
    }
    crthip_destroy(one);
    printf(failures ? "node_probe FAILED\n" : "node_probe ok\n");
    return failures ? 1 : 0;
}
