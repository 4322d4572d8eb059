/* Minimal C caller of libtinyorb (include/tinyorb.h):
 *   gcc -Iinclude examples/minimal.c -Ltinyslam_amd -ltinyorb -Wl,-rpath,$PWD/tinyslam_amd -o minimal
 * One 640x480 frame in, keypoints and descriptors out -- the six calls of tinyslam::orb (src/orb.rs).
 *   minimal [dump.bin [frame.rgba]]   dump.bin receives total (u32), n (u32), n CornerData, n CornerDescriptor;
 *                                     frame.rgba replaces the built-in checker frame (640*480*4 bytes).
 * tests/test_gpu_round2.py compiles this file with gcc, runs it and checks the dump against the oracle: it stands in
 * for the Rust host of INTEGRATION.md, which cannot be built in this image. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tinyorb.h"

int main(int argc, char **argv) {
    OrbConfig cfg = {{640, 480, 1}, 4096, 2, 20.0f / 255.0f};
    OrbProgram *prog = NULL;
    if (orb_program_create(&cfg, NULL, &prog) != ORB_OK) {
        fprintf(stderr, "create: %s\n", orb_last_error(NULL));
        return 1;
    }
    uint8_t *frame = malloc(640 * 480 * 4);
    for (size_t i = 0; i < 640u * 480u; i++) { /* a checker of small bright squares on black */
        uint8_t v = ((i % 640) / 3 % 5 == 0 && (i / 640) / 3 % 5 == 0) ? 255 : 0;
        frame[4 * i] = frame[4 * i + 1] = frame[4 * i + 2] = v;
        frame[4 * i + 3] = 255;
    }
    if (argc > 2) {
        FILE *f = fopen(argv[2], "rb");
        if (!f || fread(frame, 1, 640 * 480 * 4, f) != 640u * 480u * 4u) {
            fprintf(stderr, "cannot read %s\n", argv[2]);
            return 1;
        }
        fclose(f);
    }
    uint32_t total = 0;
    int rc = orb_write_input_image(prog, frame, 640 * 480 * 4);
    if (rc == ORB_OK) rc = orb_extract_corners(prog, &total); /* ORB_ECAPACITY: more than max_features detected */
    if (rc != ORB_OK && rc != ORB_ECAPACITY) {
        fprintf(stderr, "extract: %s\n", orb_last_error(prog));
        return 1;
    }
    size_t n = total < cfg.max_features ? total : cfg.max_features;
    CornerData *kp = calloc(n ? n : 1, sizeof *kp);
    CornerDescriptor *desc = calloc(n ? n : 1, sizeof *desc);
    orb_read_corners(prog, kp, n);
    orb_read_descriptors(prog, desc, n);
    printf("%u keypoints (%s pipeline)\n", total, orb_pipeline(prog));
    if (n) printf("first: x=%u y=%u angle=%u mrad octave=%u\n", kp[0].x, kp[0].y, kp[0].angle, kp[0].octave);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        uint32_t hdr[2] = {total, (uint32_t)n};
        if (!f || fwrite(hdr, 4, 2, f) != 2 || fwrite(kp, sizeof *kp, n, f) != n || fwrite(desc, sizeof *desc, n, f) != n) {
            fprintf(stderr, "cannot write %s\n", argv[1]);
            return 1;
        }
        fclose(f);
    }
    free(kp);
    free(desc);
    free(frame);
    orb_program_destroy(prog);
    return 0;
}
