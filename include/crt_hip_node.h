/*
 * crt_hip_node.h -- the multi-GPU entry of the field-pass path for C callers: ONE process, all the GPUs of a node.
 *
 * the path shards by field.  A batch of independent field-passes is cut into contiguous
 * blocks of ceil(n / shards) fields, one block per shard; the only thing every shard must agree on is the settings blob
 * (`crthip_params`), which shard 0's copy wins: it is broadcast over RCCL (xGMI) to every device and read back there.
 * No picture data crosses devices in batch mode.
 *
 * SEQUENCE mode across GPUs (last row -- the loop of extra/video_convert.c:246-277 over ONE long video):
 * shard s takes the contiguous fields [first_s, first_s + n_s) of the video.  What the reference carries from field to
 * field has to cross the shard seams:
 *   rn              closed form: J^k(rn0), every shard computes its own (crt_core.c:359-364 is an affine map);
 *   hsync, vsync    data dependent (crt_core.c:379-396, 437-450): every shard runs its sync chain from a guess, the
 *                   finals are exchanged, and a shard whose incoming pair changed re-runs its chain (a fixed point over
 *                   the shards, exactly like the one over the fields inside a shard; after round j the first j shards
 *                   are final for good, in practice two rounds) -- 8 bytes per shard per round, carried by the host;
 *   the output      rows a field does not write keep what the previous fields left there (crt_core.c:431, 608, 662):
 *   picture         shard s's last picture is handed to shard s + 1 -- RCCL send/recv over xGMI between devices, a
 *                   device-to-device copy between shards that share a device -- and only the rows nobody in s + 1
 *                   wrote are taken from it (blend != 0: the blend recurrence itself runs down this chain);
 *   VHS: rand()     one stream per video: walked ahead on shard 0's device, the per-field states scattered (see
 *                   crthip_node_sequence).
 *
 * A shard = one crthip_ctx with its own stream.  Several shards may share a device (`devices[]` may repeat an index):
 * that is how the whole protocol is exercised on a box with ONE GPU.  The RCCL communicator spans the DISTINCT devices.
 *
 * Library: libcrthip_node.so (links libcrthip.so and librccl.so).  No counterpart in the reference (single-threaded,
 * single device); the per-device work is exactly crthip_fieldpass / the crthip_seq_* phases of crt_hip.h.
 */
#ifndef CRT_HIP_NODE_H
#define CRT_HIP_NODE_H

#include "crt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crthip_node crthip_node;

/* n_shards contexts; devices[s] = HIP device of shard s, NULL = shard s on device s % crthip_device_count(). */
int  crthip_node_create(crthip_node **out, int n_shards, const int *devices, int system, int chroma_pattern);
void crthip_node_destroy(crthip_node *node);
int  crthip_node_shards(const crthip_node *node);
int  crthip_node_device(const crthip_node *node, int shard);
int  crthip_node_rccl_ranks(const crthip_node *node);                 /* distinct devices = ranks of the communicator */
crthip_ctx *crthip_node_ctx(crthip_node *node, int shard);            /* the shard's context (knobs: crthip_set_*) */
const char *crthip_node_error_string(const crthip_node *node);
int  crthip_node_synchronize(crthip_node *node);                      /* every shard's stream */
/* VHS: the shard's per-field generator histories (crthip_vhs_bind_history of the shard's context; count x 32 words on the
 * shard's device) */
int  crthip_node_vhs_bind_history(crthip_node *node, int shard, unsigned *d_hist);

/* contiguous block of shard `shard` in a batch / video of n_total fields: blocks of ceil(n_total / shards) */
void crthip_node_shard_range(const crthip_node *node, int n_total, int shard, int *first, int *count);

/* The settings blob of shard 0 (`root`, finalized) to every shard: uploaded to shard 0's device, ncclBroadcast over
 * the communicator, read back from each shard's device into per_shard[shard] (host array of crthip_node_shards()
 * entries).  crthip_node_fieldpass / _sequence call this themselves -- whenever `p` differs from the blob of their last
 * round trip: a loop over batches with unchanged settings pays for it once, and crthip_node_fieldpass stays asynchronous
 * from then on; exposed for callers that stage their own batches (tools/node_bench.c). */
int  crthip_node_broadcast_params(crthip_node *node, const crthip_params *root, crthip_params *per_shard);

/* One batch of n_total independent field-passes (crthip_fieldpass semantics per field).  Per-shard device pointers:
 * d_images[s] / d_out[s] / d_state[s] = the shard's first image / output image / state entry, on the shard's device.
 * Asynchronous: returns when everything is enqueued. */
int  crthip_node_fieldpass(crthip_node *node, const crthip_params *p, int n_total,
                           const void *const *d_images, size_t image_stride,
                           void *const *d_out, size_t out_stride, crthip_state *const *d_state);

/* n_total consecutive fields of ONE television set (crthip_sequence semantics), cut over the shards.
 * d_state[0][0].hsync / .vsync / .rn = the set's state before field 0; every d_state[s][k].field / .frame / .aux = the
 * encoder inputs of that field.  d_out_init: the output buffer before field 0, on shard 0's device (NULL = zeros).
 * *rounds (optional) = exchange rounds of the sync fixed point over the shards.  Synchronous.
 * VHS build (the system extra/video_convert.c is built for): the fields share ONE libc rand() stream.  Every shard's history
 * array is bound with crthip_node_vhs_bind_history; entry 0 of shard 0's array = the generator before field 0.  Shard 0's
 * device walks the stream for the whole video first (crthip_vhs_chain, ~0.15 ms per field, picture independent), the
 * per-field generator states -- and, with CRTHIP_F_VHS_DRAW_ABERRATION, the aberration heights -- are scattered to the
 * shards, after which they work in parallel like the other systems.  On return every shard's entry k = the generator after
 * its field k. */
int  crthip_node_sequence(crthip_node *node, const crthip_params *p, int n_total,
                          const void *const *d_images, size_t image_stride,
                          void *const *d_out, size_t out_stride, const void *d_out_init,
                          crthip_state *const *d_state, int *rounds);

#ifdef __cplusplus
}
#endif
#endif
