/*
 * orb_oracle.h -- CPU restatement of the tinyslam ORB front-end (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference (ccaven/tinyslam @ v2) ships no tests, golden vectors or
 * fixtures for this path and cannot be built or run in this environment (Rust + wgpu/Vulkan,
 * no cargo/rustc/Vulkan ICD).  This file restates the reference's WGSL
 * shaders and the stage order of src/orb.rs in plain C, with the implementation-defined
 * points fixed by SURVEY.md's canonical restatement decisions CRD-1..CRD-13.  It is pinned
 * only by known-answer tests derived from the reference text, by a second, independently
 * written NumPy restatement (oracle/orb_numpy.py) and -- since round 5 -- by the shader text
 * ITSELF, executed by a small WGSL interpreter (tests/wgsl_interp.py, tests/test_reference_text.py:
 * every plane and record bit for bit; tests/golden/reftext/ holds its outputs as data).  That
 * rules out transcription errors; it does not pin the implementation-defined points to any
 * adapter -- the parity stays UNPINNED until a dump of the reference is held
 * (tools/pin_oracle.py, rust/dump_config0).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
 * under oracle/.  The product (tinyslam_amd/, include/) never includes, links or calls it.
 *
 * All images are planes of IEEE binary16 bit patterns (uint16_t), row-major, tightly packed:
 * the reference keeps every image after the input as R16Float (src/orb.rs:151,228,296,311).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 10 /* src/orb.rs:67 MAX_HIERARCHY_DEPTH */

/* src/orb.rs:10-17 CornerData == fast.wgsl:1-6 Feature */
typedef struct {
    uint32_t x, y, angle, octave;
} orc_corner_t;

/* src/orb.rs:19-23 CornerDescriptor (8 little-endian u32 words, brief.wgsl:15) */
typedef struct {
    uint32_t word[8];
} orc_descriptor_t;

/* Level geometry: mip m is (max(1,W>>m), max(1,H>>m)) texels (wgpu mip chain, orb.rs:224-233).
 * offset[] are texel offsets of each level inside one packed pyramid buffer. */
typedef struct {
    uint32_t depth;
    uint32_t w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    size_t offset[ORC_MAX_LEVELS];
    size_t total; /* texels in the packed pyramid */
} orc_pyramid_t;

void orc_pyramid_layout(uint32_t W, uint32_t H, uint32_t depth, orc_pyramid_t *p);

/* scalar helpers (exposed for known-answer tests) */
uint16_t orc_f32_to_f16(float v);  /* CRD-3: round-to-nearest-even, subnormals kept */
float orc_f16_to_f32(uint16_t h);  /* exact */
float orc_atan2f(float y, float x); /* CRD-9 */
uint32_t orc_angle_code(float cy, float cx); /* fast.wgsl:115,153 + CRD-9 */
uint32_t orc_detect_streak_16(uint32_t mask); /* fast.wgsl:51-60 */
float orc_unorm8(uint8_t b); /* CRD-1 */

/* stages */
void orc_grayscale(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray); /* grayscale.wgsl:12-38 */
/* Y8 input variant (NOT in the reference's code; its roadmap item README.md:42): gray(x,y) = f16(Y(x, H-1-y)/255) */
void orc_grayscale_y8(const uint8_t *y8, uint32_t W, uint32_t H, uint16_t *gray);
int orc_extract_y8(const uint8_t *y8, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                   orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr,
                   uint16_t *blur_pyr);
void orc_mip(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd); /* blit.wgsl:17-36 */
void orc_blur_pass(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst); /* gaussian_blur_x.wgsl:32-60 */
/* fast.wgsl:62-159 over every octave in dispatch order (orb.rs:504-520).  Appends in raster
 * order, octave-major.  Stores at most cap records, returns the raw counter in *total. */
void orc_fast(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, orc_corner_t *out, uint32_t cap,
              uint32_t *total);
/* brief.wgsl:20-68 for n corners */
void orc_brief(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
               orc_descriptor_t *out);

/* Whole frame: orb.rs:469-557 extract_corners.  gray_pyr / blur_pyr (may be NULL) receive the
 * packed pyramids (lay.total texels each).  Returns 0, or -1 on invalid arguments. */
int orc_extract(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr,
                uint16_t *blur_pyr);

/* ---- implementation-defined behaviour as switches.  The reference's WGSL leaves two things to the adapter that can
 * change keypoints and descriptor bits; until a dump from the reference itself pins them
 * (tools/pin_oracle.py, rust/dump_config0), the restatement can be run either way:
 *   oob                   what textureLoad returns outside the addressed level (fast.wgsl:78,86,103; brief.wgsl:59-60):
 *                         ORC_OOB_ZERO 0 (robust image access; CRD-6, the default), ORC_OOB_CLAMP each coordinate clamped
 *                         into the level, ORC_OOB_UMIN naga's `Restrict` policy as min(unsigned(coordinate), size - 1):
 *                         negative coordinates land on the LAST column / row.
 *   sampler_weight_bits   precision of a bilinear sampler's weights (gaussian_blur_x.wgsl:53-58; blit.wgsl:35 for odd
 *                         sizes): 0 = the exact binary32 fraction (CRD-5, the default), n = 1..23: the fraction rounded
 *                         to n fractional bits, halves up (8 is what GPUs commonly implement).
 *   contract              CRD-13, a bit per stage (ORC_CONTRACT_LUM | _BLUR | _ROT): 0 = every binary32 product and sum of that
 *                         stage rounded on its own (CRD-2, -5, -10; the default); set = a shader compiler that contracts a
 *                         product and the sum behind it into one fused multiply-add -- WGSL permits it, naga emits no
 *                         NoContraction, and GPU compilers commonly lower `dot()` (grayscale.wgsl:36), `result += sample *
 *                         weight` (gaussian_blur_x.wgsl:58) and matrix * vector (brief.wgsl:53-54) to fma chains.  Per stage
 *                         because a compiler decides expression by expression.
 *   dot_order             the order in which `dot()` and matrix * vector are reduced, which WGSL does not fix: 0 = first
 *                         component / column first (as written; LLVM-based compilers), 1 = last first (Mesa: NIR's inexact
 *                         fdot lowering and spirv_to_nir's matrix * vector start from the last component / column).  Changes the
 *                         luminance with or without contraction ((r + g) + b against (b + g) + r) and the rotation only when
 *                         contracted (which product is the fused one); the blur's accumulation is a sequential loop.
 *   f16_round             the store to the R16Float targets (orb.rs:151, 228, 296, 311): 0 = round to nearest even (CRD-3, the
 *                         default), 1 = toward zero -- Vulkan leaves the rounding of a format conversion to the implementation.
 *                         Restatement and tools/pin_oracle.py only; the kernels round to nearest even.
 *   neg_angle             what the conversion of a negative angle to u32 yields (fast.wgsl:153; half of all keypoints): 0 (default), the
 *                         wrapped value, or all ones -- see orc_angle_code_neg.  Restatement and tool only.
 * The defaults are what every other entry of this header computes. */
#define ORC_OOB_ZERO 0u
#define ORC_OOB_CLAMP 1u
#define ORC_OOB_UMIN 2u
#define ORC_CONTRACT_LUM 1u
#define ORC_CONTRACT_BLUR 2u
#define ORC_CONTRACT_ROT 4u
#define ORC_CONTRACT_ALL 7u
typedef struct {
    uint32_t oob;
    uint32_t sampler_weight_bits;
    uint32_t contract;
    uint32_t dot_order;
    uint32_t f16_round;
    uint32_t neg_angle; /* ORC_NEG_*: what `u32(angle * 1000.0)` (fast.wgsl:153) makes of a NEGATIVE angle -- OpConvertFToU is undefined there */
} orc_impl_t;
#define ORC_NEG_ZERO 0u /* saturates to 0 (GPUs; SURVEY.md Q7, the default) */
#define ORC_NEG_WRAP 1u /* the low 32 bits of the truncated value: 2^32 - m (LLVM's fptoui on x86-64: lavapipe / llvmpipe) */
#define ORC_NEG_ONES 2u /* 0xffffffff (AVX-512's unsigned conversion of an out-of-range operand) */
uint32_t orc_angle_code_neg(float cy, float cx, uint32_t neg);
void orc_fast_impl2(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t oob, uint32_t neg, orc_corner_t *out,
                    uint32_t cap, uint32_t *total);
uint16_t orc_f32_to_f16_mode(float v, uint32_t rtz);
void orc_grayscale_fp(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray, const orc_impl_t *impl);
void orc_grayscale_y8_fp(const uint8_t *y8, uint32_t W, uint32_t H, uint16_t *gray, const orc_impl_t *impl);
void orc_mip_fp(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd, const orc_impl_t *impl);
void orc_blur_pass_fp(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, const orc_impl_t *impl);
void orc_brief_fp(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                  const orc_impl_t *impl, orc_descriptor_t *out);
/* One pattern point (px, py) under the rotation of brief.wgsl:35-54 at an angle code, BEFORE vec2i() truncates it (tools/pin_oracle.py
 * asks how close a rotated coordinate lies to an integer, where an adapter's own cos / sin would flip the truncation). */
void orc_brief_rotate(uint32_t angle_code, int px, int py, uint32_t contract, uint32_t last_first, float *rx, float *ry);
/* round 4's forms: contract != 0 means every stage contracted, first term first */
void orc_grayscale_impl(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray, uint32_t contract);
void orc_blur_pass_impl2(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, uint32_t wbits, uint32_t contract);
void orc_brief_impl2(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                     uint32_t oob, uint32_t contract, orc_descriptor_t *out);
void orc_mip_impl(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd, uint32_t wbits);
void orc_blur_pass_impl(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, uint32_t wbits);
void orc_fast_impl(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t oob, orc_corner_t *out,
                   uint32_t cap, uint32_t *total);
void orc_brief_impl(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                    uint32_t oob, orc_descriptor_t *out);
/* orc_extract / orc_extract_y8 (y8 != 0) with the switches; impl == NULL: the defaults. */
int orc_extract_impl(const uint8_t *frame, int y8, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                     uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                     uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr);

/* ---- opt-in extensions (NOT in the reference, no parity target: the definitions
 * below are the build's own and are pinned only by the NumPy restatement and the GPU tests) ----
 * arc: a corner needs a circular run of >= arc ring pixels (9..16) all brighter or all darker than the centre by
 *      more than the threshold.  arc = 12 is the reference's detector (fast.wgsl:56-60); the reference's 4-point
 *      pre-test is a necessary condition only for arc >= 12, so other arcs test every guarded pixel.
 * nms: 3x3 non-maximum suppression per octave on the score S = sum over the 16 ring pixels of
 *      max(|v - c| - threshold, 0) restricted to the run's polarity (binary32, ring order).  A corner survives
 *      iff every 8-neighbour that is also a corner has a smaller score, or an equal score and a later raster
 *      position (y, then x).  The returned counter is the number of survivors. */
typedef struct {
    uint32_t arc; /* 0 -> 12 */
    uint32_t nms; /* 0 / 1 */
    uint32_t angle_bins; /* intended mode only, IM-6b: 0 = descriptors rotated by the keypoint's milliradian code; 8..6284 = by the centre of
                          * its bin of the full circle */
} orc_options_t;
void orc_fast_ex(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t arc, orc_corner_t *out,
                 float *scores, uint32_t cap, uint32_t *total);
uint32_t orc_nms(const orc_pyramid_t *lay, const orc_corner_t *in, const float *scores, uint32_t n, orc_corner_t *out);
int orc_extract_ex(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                   const orc_options_t *opt, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total);

/* ---- "intended" mode (NOT in the reference, no parity target).  The reference's README
 * describes ORB; its shaders deviate from that in ways that look accidental (SURVEY.md Q1, Q2, Q7, Q8, Q11, Q12,
 * Q14).  This mode is the repaired algorithm, defined here and pinned by the NumPy restatement + the GPU tests:
 * IM-1 grey(x,y) = f16((0.299f*r + 0.587f*g) + 0.114f*b) of input(x,y): BT.601 weight, no vertical mirror.
 * IM-2 mip chain as CRD-4.
 * IM-3 blur = separable 7-tap Gaussian, X pass then Y pass, each stored as f16 (CRD-3).  The taps are the
 *      reference's four bilinear taps read in texel units: weights g0..g3 = 0.282523781f, 0.221251875f,
 *      0.106235079f, 0.0312511548f (centre outwards); one pass = ((g0*t0 + g1*(t-1 + t+1)) + g2*(t-2 + t+2)) +
 *      g3*(t-3 + t+3) in binary32, unfused, indices clamped to the level.
 * IM-4 detector = orc_fast_ex's (arc 9..16, 0 -> 9; no pre-test) with the guard taken from the octave's own
 *      size: 16 < x < w - 16 and 16 < y < h - 16 (octaves with w <= 33 or h <= 33 hold no keypoint).
 * IM-5 angle code = trunc(1000 * a), a = atan2(cy, cx) of the ring centroid (CRD-8/-9), plus 6.28318531f when
 *      negative: 0..6283 milliradians, the full circle.
 * IM-6 BRIEF samples the IM-3 blur at p + trunc(R(+theta) q): (ct*x - st*y, st*x + ct*y), products and sums
 *      rounded on their own, ct/st the correctly rounded cos/sin of fl32(code/1000.0f).
 * IM-6b optional (orc_options_t::angle_bins = N, 8..6284): the rotation of IM-6 uses the centre of the keypoint's angle bin instead of
 *      its own code: bin = code * N / 6284, centre code = (bin * 6284 + 3142) / N (integer arithmetic); the reported angle stays the
 *      milliradian code.  (OpenCV's ORB quantises to 30 bins; a table of 1024 rotated patterns is 1 MB and stays in every XCD's L2.)
 * IM-7 optional 3x3 NMS exactly as orc_nms.
 * IM-8 when more than max_features keypoints remain, the max_features best are kept: larger score first, ties by
 *      smaller (octave, y, x).  The returned counter is still the number before this cut. */
void orc_grayscale_intended(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray);
void orc_gauss_pass(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, int vertical);
uint32_t orc_angle_code_signed(float cy, float cx);
void orc_fast_intended(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t arc, orc_corner_t *out,
                       float *scores, uint32_t cap, uint32_t *total);
uint32_t orc_binned_angle_code(uint32_t code, uint32_t bins);
void orc_brief_intended_bins(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                             uint32_t angle_bins, orc_descriptor_t *out);
void orc_brief_intended(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                        orc_descriptor_t *out);
uint32_t orc_topk(const orc_corner_t *in, const float *scores, uint32_t n, uint32_t k, orc_corner_t *out);
int orc_extract_intended(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                         uint32_t max_features, const orc_options_t *opt, orc_corner_t *corners,
                         orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr);

/* Frame-parallel batch of orc_extract_intended (CPU baseline of `bench.py --mode intended`). */
int orc_extract_intended_batch(const uint8_t *rgba, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth,
                               float threshold, uint32_t max_features, const orc_options_t *opt, orc_corner_t *corners,
                               orc_descriptor_t *descriptors, uint32_t *totals, int n_threads);

/* Frame-parallel batch for the CPU baseline leg of bench.py: n_frames contiguous RGBA frames,
 * outputs strided by max_features.  n_threads <= 1 runs serially. */
int orc_extract_batch_y8(const uint8_t *y8, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                         uint32_t max_features, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *totals,
                         int n_threads);
int orc_extract_batch(const uint8_t *rgba, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                      uint32_t max_features, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *totals,
                      int n_threads);

int orc_extract_batch_impl(const uint8_t *frames, int y8, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                           uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                           uint32_t *totals, int n_threads);

/* Synthetic frames: counter-based, integer only. */
#define ORC_SYN_GRADIENT 1u
#define ORC_SYN_BLOBS 2u
#define ORC_SYN_WEDGES 4u
#define ORC_SYN_NOISE 8u
void orc_synth_frame(uint8_t *rgba, uint32_t W, uint32_t H, uint32_t seed, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
