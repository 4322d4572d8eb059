/* Minimal C caller of libtinyorb (include/tinyorb.h):
 *   gcc -Iinclude examples/minimal.c -Ltinyslam_amd -ltinyorb -Wl,-rpath,$PWD/tinyslam_amd -o minimal
 * One 640x480 frame in, keypoints and descriptors out -- the six calls of tinyslam::orb (src/orb.rs). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tinyorb.h"

int main(void) {
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
    free(kp);
    free(desc);
    free(frame);
    orb_program_destroy(prog);
    return 0;
}
