// orb_front_launch.h -- host-side seam between orb_api.hip and the translation units that instantiate k_front's variants.
//
// k_front is a template over the level kind, the input format, the band height, row alignment, column tiles, the out-of-level
// policy, the workgroup size AND the arithmetic form of the adapter's shader compiler (FPF: four forms of the luminance x two of
// the blur taps, OrbOptions::fp_contract): some 160 kernels.  orb_front_inst.hip is compiled once per form (-DTINYORB_FRONT_FP=0..7),
// in parallel (tinyslam_amd/build.py), and each of those objects exports the three entry points below under its own suffix; a kernel is
// launched from the translation unit that instantiated it, so no relocatable device code is involved.
#pragma once
#include "orb_kernels_front.h"

namespace orb {

struct FrontLaunch {
    const uint8_t* frames;
    size_t frame_bytes;
    uint16_t* gray;
    uint16_t* blur;
    uint16_t* blur_rowc;
    Pyramid pyr;
    FrontGeom g;
    float thr;
    uint32_t* seg_counts;
    CornerData* segments;
    hipStream_t stream;
    uint32_t grid;        // workgroups
    uint32_t lds;         // dynamic LDS bytes
    uint32_t band_rows;   // 64 / 32 / 16 / 8
    uint32_t ln_threads;  // levels >= 1: kFrontThreadsLN or kFrontThreadsLNBig
    bool input_y8;
    bool general;         // level 0: rows not quad-aligned, or the level-0 plane must be stored (UA)
    bool oob;             // an out-of-level policy other than "zero" (levels >= 1: the OOBK instances)
    bool from_plane;      // take the level >= 1 kernel (rows from the stored grey plane) whatever g.lvl says: the arc / NMS extension's blur-only
                          // launches read level 0's plane, which k_front_i has written (512 threads, 16-row bands)
};

struct FrontPairLaunch {  // k_front_pair: levels 0 and 1 of one frame in one launch (8-row bands)
    const uint8_t* frames;
    size_t frame_bytes;
    uint16_t* gray;
    uint16_t* blur;
    uint16_t* blur_rowc;
    Pyramid pyr;
    FrontGeom g0, g1;
    float thr;
    uint32_t* seg_counts;
    CornerData* segments;
    hipStream_t stream;
    uint32_t lds;
    bool input_y8;
};

// One set per arithmetic form f = lum_form | (blur taps fused ? 4 : 0).  Forms with a luminance other than 0 hold the kernels that
// compute a luminance (level 0 from RGBA, k_front_pair on RGBA) and nothing else: Y8 frames and the levels >= 1 go to form f & 4.
#define ORB_FRONT_DECL(F)                                        \
    hipError_t front_launch_fp##F(const FrontLaunch& L);         \
    hipError_t front_pair_launch_fp##F(const FrontPairLaunch& L); \
    hipError_t front_set_max_lds_fp##F(int max_lds);
ORB_FRONT_DECL(0) ORB_FRONT_DECL(1) ORB_FRONT_DECL(2) ORB_FRONT_DECL(3) ORB_FRONT_DECL(4) ORB_FRONT_DECL(5) ORB_FRONT_DECL(6) ORB_FRONT_DECL(7)
#undef ORB_FRONT_DECL

// the form a launch takes: the luminance bits only where a luminance is computed
inline uint32_t front_form(uint32_t fp_contract, bool computes_luminance) {
    return (computes_luminance ? (uint32_t)lum_form(fp_contract) : 0u) | ((fp_contract & kFpBlur) ? 4u : 0u);
}

}  // namespace orb
