/* A C host driving the batched and the multi-GPU entry points of libtinyorb (include/tinyorb.h) the way the Rust host of
 * INTEGRATION.md would:
 *   gcc -Iinclude examples/node_batch.c -Ltinyslam_amd -ltinyorb -Wl,-rpath,$PWD/tinyslam_amd -o node_batch
 *   node_batch frames.rgba W H n_frames out.bin [n_devices]
 * frames.rgba holds n_frames tightly packed RGBA8 frames.  The job is sharded over the first n_devices GPUs (default:
 * 1), collated on the first one (RCCL) and dumped: n_frames (u32), counts[n], offsets[n+1] (u64), total corners, total
 * descriptors.  The same job then goes through the single-device bulk read-back (orb_extract_batch_host +
 * orb_batch_read_all into pinned memory) and both results must be identical byte for byte.  Last, the job is streamed
 * through the node three times in its pipelined form (orb_node_extract_batch_host / orb_node_collate_begin /
 * orb_node_collate_end, two jobs outstanding): every pass must return the counters of the first. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tinyorb.h"

#define CHECK(call, errsrc)                                         \
    do {                                                            \
        int rc_ = (call);                                           \
        if (rc_ != ORB_OK) {                                        \
            fprintf(stderr, "%s: %d %s\n", #call, rc_, (errsrc));  \
            return 1;                                               \
        }                                                           \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: node_batch frames.rgba W H n_frames out.bin [n_devices]\n");
        return 2;
    }
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), F = (uint32_t)atoi(argv[4]);
    const int n_dev = argc > 6 ? atoi(argv[6]) : 1;
    const size_t frame_bytes = (size_t)W * H * 4;
    uint8_t *frames = malloc(frame_bytes * F);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(frames, frame_bytes, F, f) != F) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    fclose(f);

    OrbConfig cfg = {{W, H, 1}, 2048, 2, 20.0f / 255.0f};
    OrbOptions opt;
    memset(&opt, 0, sizeof opt);
    opt.max_batch = F; /* the largest shard one device may get */

    /* ---- multi-GPU entry: shard, extract, collate on the first device ---- */
    int devices[64];
    for (int i = 0; i < n_dev; i++) devices[i] = i;
    OrbNode *node = NULL;
    CHECK(orb_node_create(devices, n_dev, &cfg, &opt, &node), orb_node_last_error(NULL));
    CHECK(orb_node_extract_batch_host(node, frames, F), orb_node_last_error(node));
    uint32_t *counts = calloc(F, sizeof *counts);
    uint64_t *offsets = calloc((size_t)F + 1, sizeof *offsets);
    CHECK(orb_node_collate(node, counts, offsets, NULL, NULL), orb_node_last_error(node));
    const size_t total = (size_t)offsets[F];
    CornerData *kp = calloc(total ? total : 1, sizeof *kp);
    CornerDescriptor *desc = calloc(total ? total : 1, sizeof *desc);
    CHECK(orb_node_read_collated(node, kp, desc, total), orb_node_last_error(node));

    /* ---- single-device bulk read-back into pinned memory ---- */
    OrbProgram *prog = NULL;
    CHECK(orb_program_create(&cfg, &opt, &prog), orb_last_error(NULL));
    void *p_counts, *p_offsets, *p_kp, *p_desc;
    CHECK(orb_host_alloc(sizeof(uint32_t) * F, &p_counts), orb_last_error(NULL));
    CHECK(orb_host_alloc(sizeof(uint64_t) * ((size_t)F + 1), &p_offsets), orb_last_error(NULL));
    CHECK(orb_host_alloc(sizeof(CornerData) * (total ? total : 1), &p_kp), orb_last_error(NULL));
    CHECK(orb_host_alloc(sizeof(CornerDescriptor) * (total ? total : 1), &p_desc), orb_last_error(NULL));
    CHECK(orb_extract_batch_host(prog, frames, F), orb_last_error(prog));
    CHECK(orb_batch_read_all(prog, F, p_counts, p_offsets, p_kp, p_desc, total, NULL), orb_last_error(prog));
    CHECK(orb_batch_sync(prog), orb_last_error(prog));
    int same = memcmp(p_counts, counts, sizeof(uint32_t) * F) == 0 &&
               memcmp(p_offsets, offsets, sizeof(uint64_t) * ((size_t)F + 1)) == 0;
    /* record order inside a frame is unspecified (atomic append, fast.wgsl:146-157): compare per frame as multisets by
     * summing the words of keypoint + descriptor pairs */
    for (uint32_t fr = 0; same && fr < F; fr++) {
        uint64_t sa = 0, sb = 0;
        for (uint64_t i = offsets[fr]; i < offsets[fr + 1]; i++) {
            const CornerData *a = &kp[i], *b = &((CornerData *)p_kp)[i];
            const uint32_t *da = (const uint32_t *)&desc[i], *db = (const uint32_t *)&((CornerDescriptor *)p_desc)[i];
            uint64_t ha = ((uint64_t)a->x * 1315423911u) ^ ((uint64_t)a->y * 2654435761u) ^ ((uint64_t)a->angle << 32) ^ a->octave;
            uint64_t hb = ((uint64_t)b->x * 1315423911u) ^ ((uint64_t)b->y * 2654435761u) ^ ((uint64_t)b->angle << 32) ^ b->octave;
            for (int w = 0; w < 8; w++) {
                ha = ha * 1099511628211ull + da[w];
                hb = hb * 1099511628211ull + db[w];
            }
            sa += ha;
            sb += hb;
        }
        same = sa == sb;
    }
    printf("%u frames on %d device(s): %zu records, node collate %s single-device read-back\n", F,
           orb_node_device_count(node), total, same ? "==" : "!=");

    /* ---- the pipelined form: job k is extracted while job k-1 is exchanged and collated ---- */
    uint32_t *counts2 = calloc(F, sizeof *counts2);
    uint64_t *offsets2 = calloc((size_t)F + 1, sizeof *offsets2);
    int ended = 0;
    for (int k = 0; k < 3 && same; k++) {
        CHECK(orb_node_extract_batch_host(node, frames, F), orb_node_last_error(node));
        if (orb_node_pending(node) == 2) {
            CHECK(orb_node_collate_begin(node), orb_node_last_error(node));
            CHECK(orb_node_collate_end(node, counts2, offsets2, NULL, NULL), orb_node_last_error(node));
            same = memcmp(counts2, counts, sizeof(uint32_t) * F) == 0 && offsets2[F] == offsets[F];
            ended++;
        }
    }
    while (same && orb_node_pending(node) > 0) {
        CHECK(orb_node_collate_end(node, counts2, offsets2, NULL, NULL), orb_node_last_error(node));
        same = memcmp(counts2, counts, sizeof(uint32_t) * F) == 0 && offsets2[F] == offsets[F];
        ended++;
    }
    printf("pipelined: %d jobs collated, counters %s\n", ended, same ? "equal" : "DIFFER");
    free(counts2);
    free(offsets2);

    f = fopen(argv[5], "wb");
    if (!f || fwrite(&F, 4, 1, f) != 1 || fwrite(counts, 4, F, f) != F || fwrite(offsets, 8, (size_t)F + 1, f) != (size_t)F + 1 ||
        fwrite(kp, sizeof *kp, total, f) != total || fwrite(desc, sizeof *desc, total, f) != total) {
        fprintf(stderr, "cannot write %s\n", argv[5]);
        return 1;
    }
    fclose(f);
    orb_host_free(p_counts);
    orb_host_free(p_offsets);
    orb_host_free(p_kp);
    orb_host_free(p_desc);
    orb_program_destroy(prog);
    orb_node_destroy(node);
    free(kp);
    free(desc);
    free(counts);
    free(offsets);
    free(frames);
    return same ? 0 : 3;
}
