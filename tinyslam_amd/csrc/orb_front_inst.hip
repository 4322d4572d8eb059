// orb_front_inst.hip -- the k_front / k_front_pair instances of ONE arithmetic form (compile with -DTINYORB_FRONT_FP=f, f = 0..7: bits
// 0-1 the luminance's form, bit 2 the blur taps fused; orb_front_launch.h) and their launch switch.
#include <hip/hip_runtime.h>

#include "orb_front_launch.h"

#ifndef TINYORB_FRONT_FP
#error "compile with -DTINYORB_FRONT_FP=0..7 (tinyslam_amd/build.py does)"
#endif

namespace orb {

namespace {
constexpr int FP = TINYORB_FRONT_FP;
constexpr int FPB = FP & 4;              // what a kernel without a luminance sees of the form
constexpr bool kRest = (FP & 3) == 0;    // this unit also holds the kernels that compute no luminance (Y8 level 0, levels >= 1)
}  // namespace

#define ORB_CAT2(a, b) a##b
#define ORB_CAT(a, b) ORB_CAT2(a, b)
#define FRONT_ARGS L.frames, L.frame_bytes, L.gray, L.blur, L.blur_rowc, L.pyr, L.g, L.thr, L.seg_counts, L.segments
#define FRONT_GO(...) hipLaunchKernelGGL((k_front<__VA_ARGS__>), grid, block, L.lds, L.stream, FRONT_ARGS)
// band heights of the full-width variants / of the column tiles
#define FRONT_BY_ROWS(PRE, POST)                                    \
    switch (L.band_rows) {                                          \
        case 64: FRONT_GO(PRE, 64, POST); break;                    \
        case 32: FRONT_GO(PRE, 32, POST); break;                    \
        case 16: FRONT_GO(PRE, 16, POST); break;                    \
        default: FRONT_GO(PRE, 8, POST); break;                     \
    }
#define FRONT_BY_ROWS_TILED(PRE, POST)                              \
    if (L.band_rows == 16u) FRONT_GO(PRE, 16, POST);                \
    else FRONT_GO(PRE, 8, POST);
#define COMMA ,

hipError_t ORB_CAT(front_launch_fp, TINYORB_FRONT_FP)(const FrontLaunch& L) {
    const dim3 grid(L.grid);
    const bool tiled = L.g.tiled != 0u;
    if (L.g.lvl == 0u && !L.input_y8 && !L.from_plane) {  // level 0 from RGBA: the luminance's form is this unit's
        const dim3 block(kFrontThreadsL0);
        if (tiled && L.general) { FRONT_BY_ROWS_TILED(true COMMA false, true COMMA true COMMA false COMMA false COMMA 0 COMMA FP) }
        else if (tiled) { FRONT_BY_ROWS_TILED(true COMMA false, false COMMA true COMMA false COMMA false COMMA 0 COMMA FP) }
        else if (L.general) { FRONT_BY_ROWS(true COMMA false, true COMMA false COMMA false COMMA false COMMA 0 COMMA FP) }
        else { FRONT_BY_ROWS(true COMMA false, false COMMA false COMMA false COMMA false COMMA 0 COMMA FP) }
        return hipGetLastError();
    }
    if constexpr (kRest) {
        if (L.g.lvl == 0u && !L.from_plane) {  // Y8
            const dim3 block(kFrontThreadsL0);
            if (tiled && L.general) { FRONT_BY_ROWS_TILED(true COMMA true, true COMMA true COMMA false COMMA false COMMA 0 COMMA FPB) }
            else if (tiled) { FRONT_BY_ROWS_TILED(true COMMA true, false COMMA true COMMA false COMMA false COMMA 0 COMMA FPB) }
            else if (L.general) { FRONT_BY_ROWS(true COMMA true, true COMMA false COMMA false COMMA false COMMA 0 COMMA FPB) }
            else { FRONT_BY_ROWS(true COMMA true, false COMMA false COMMA false COMMA false COMMA 0 COMMA FPB) }
        } else if (L.ln_threads == (uint32_t)kFrontThreadsLNBig) {  // a level whose bands are large enough for level 0's shape
            const dim3 block(kFrontThreadsLNBig);
            FRONT_BY_ROWS(false COMMA false, false COMMA false COMMA false COMMA false COMMA kFrontThreadsLNBig COMMA FPB)
        } else {
            const dim3 block(kFrontThreadsLN);
            if (L.oob) {  // texels outside the level follow OrbOptions::oob_policy: the OOBK instances
                if (tiled) { FRONT_BY_ROWS_TILED(false COMMA false, false COMMA true COMMA false COMMA true COMMA 0 COMMA FPB) }
                else { FRONT_BY_ROWS(false COMMA false, false COMMA false COMMA false COMMA true COMMA 0 COMMA FPB) }
            } else if (tiled) { FRONT_BY_ROWS_TILED(false COMMA false, false COMMA true COMMA false COMMA false COMMA 0 COMMA FPB) }
            else { FRONT_BY_ROWS(false COMMA false, false COMMA false COMMA false COMMA false COMMA 0 COMMA FPB) }
        }
        return hipGetLastError();
    }
    return hipErrorInvalidValue;  // a kernel without a luminance was asked of a unit that holds none (orb_api.hip: front_form())
}

hipError_t ORB_CAT(front_pair_launch_fp, TINYORB_FRONT_FP)(const FrontPairLaunch& L) {
    const dim3 grid(L.g0.n_bands + L.g1.n_bands), block(kFrontThreadsL0);
    if (!L.input_y8) {
        hipLaunchKernelGGL((k_front_pair<8, 8, false, FP>), grid, block, L.lds, L.stream, L.frames, L.frame_bytes, L.gray, L.blur, L.blur_rowc, L.pyr, L.g0,
                           L.g1, L.thr, L.seg_counts, L.segments);
        return hipGetLastError();
    }
    if constexpr (kRest) {
        hipLaunchKernelGGL((k_front_pair<8, 8, true, FPB>), grid, block, L.lds, L.stream, L.frames, L.frame_bytes, L.gray, L.blur, L.blur_rowc, L.pyr, L.g0,
                           L.g1, L.thr, L.seg_counts, L.segments);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

// The attribute belongs to the function on this device, not to a program: always raised to the device's limit, so that a later,
// smaller program never lowers it under a live, larger one.
hipError_t ORB_CAT(front_set_max_lds_fp, TINYORB_FRONT_FP)(int max_lds) {
#define FN(...) reinterpret_cast<const void*>(&k_front<__VA_ARGS__>)
#define FN4(PRE, POST) FN(PRE, 64, POST), FN(PRE, 32, POST), FN(PRE, 16, POST), FN(PRE, 8, POST)
#define FN2(PRE, POST) FN(PRE, 16, POST), FN(PRE, 8, POST)
    const void* const rgba[] = {
        FN4(true COMMA false, false COMMA false COMMA false COMMA false COMMA 0 COMMA FP), FN4(true COMMA false, true COMMA false COMMA false COMMA false COMMA 0 COMMA FP),
        FN2(true COMMA false, false COMMA true COMMA false COMMA false COMMA 0 COMMA FP), FN2(true COMMA false, true COMMA true COMMA false COMMA false COMMA 0 COMMA FP),
        reinterpret_cast<const void*>(&k_front_pair<8, 8, false, FP>)};
    for (const void* f : rgba)
        if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) return e;
    if constexpr (kRest) {
        const void* const rest[] = {
            FN4(true COMMA true, false COMMA false COMMA false COMMA false COMMA 0 COMMA FPB), FN4(true COMMA true, true COMMA false COMMA false COMMA false COMMA 0 COMMA FPB),
            FN2(true COMMA true, false COMMA true COMMA false COMMA false COMMA 0 COMMA FPB), FN2(true COMMA true, true COMMA true COMMA false COMMA false COMMA 0 COMMA FPB),
            FN4(false COMMA false, false COMMA false COMMA false COMMA false COMMA kFrontThreadsLNBig COMMA FPB),
            FN4(false COMMA false, false COMMA false COMMA false COMMA true COMMA 0 COMMA FPB), FN2(false COMMA false, false COMMA true COMMA false COMMA true COMMA 0 COMMA FPB),
            FN2(false COMMA false, false COMMA true COMMA false COMMA false COMMA 0 COMMA FPB), FN4(false COMMA false, false COMMA false COMMA false COMMA false COMMA 0 COMMA FPB),
            reinterpret_cast<const void*>(&k_front_pair<8, 8, true, FPB>)};
        for (const void* f : rest)
            if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) return e;
    }
    return hipSuccess;
#undef FN
#undef FN4
#undef FN2
}

}  // namespace orb
