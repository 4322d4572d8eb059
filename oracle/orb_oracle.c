/*
 * orb_oracle.c -- CPU restatement of the tinyslam ORB front-end.  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED (see orb_oracle.h for what that means and who may use this file).
 *
 * Every function cites the reference text it follows (paths relative to ccaven/tinyslam).
 * Build with -O2 -ffp-contract=off: every product and sum below is meant to be rounded to
 * binary32 on its own, and the HIP kernels are built the same way.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_pattern.h"

static const orc_impl_t ORC_IMPL_DEFAULT = {ORC_OOB_ZERO, 0, 0, 0, 0, 0};

/* ------------------------------------------------------------------------------------------
 * scalar helpers
 * ---------------------------------------------------------------------------------------- */

static uint32_t f32_bits(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return u;
}

static float bits_f32(uint32_t u) {
    float v;
    memcpy(&v, &u, 4);
    return v;
}

/* CRD-3: binary32 -> binary16, round to nearest even, subnormal results kept.  Render-target
 * stores of an R16Float attachment (orb.rs:151, 228, 296, 311).
 * Implementation-defined point (orc_impl_t::f16_round): Vulkan leaves the rounding of a format conversion on a store to the
 * implementation -- either neighbour is a legal result.  rtz != 0 is the other deterministic reading: round toward zero
 * (truncation; a finite value beyond the largest binary16 stays at 65504). */
uint16_t orc_f32_to_f16_mode(float v, uint32_t rtz) {
    uint32_t u = f32_bits(v);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t mag = u & 0x7fffffffu;
    if (mag >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((mag > 0x7f800000u) ? 0x200u : 0u));
    }
    if (mag >= 0x47800000u) { /* >= 65536 -> rounds to inf (65520 and up round to inf below); toward zero: the largest finite */
        return (uint16_t)(sign | (rtz ? 0x7bffu : 0x7c00u));
    }
    if (mag < (rtz ? 0x33800000u : 0x33000000u)) { /* < 2^-25: rounds to zero (2^-25 itself ties to even = 0); toward zero: < 2^-24 */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(mag >> 23) - 127; /* unbiased */
    uint32_t m = (mag & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;
    uint32_t half_exp;
    if (e < -14) { /* subnormal half: value = m * 2^(e-23), unit = 2^-24 */
        shift = (uint32_t)(-1 - e); /* 13 + (-14 - e) */
        half_exp = 0;
    } else {
        shift = 13;
        half_exp = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (!rtz && (rem > halfway || (rem == halfway && (q & 1u)))) q++;
    /* normal: q carries the hidden bit (0x400); adding lets a mantissa overflow bump the exponent */
    uint32_t h = (half_exp == 0) ? q : (((half_exp - 1) << 10) + q);
    return (uint16_t)(sign | h);
}
uint16_t orc_f32_to_f16(float v) { return orc_f32_to_f16_mode(v, 0u); }

float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        /* zero or subnormal: m * 2^-24, exact in binary32 */
        float v = (float)m * 5.9604644775390625e-08f;
        return sign ? -v : v;
    }
    if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
    return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

/* CRD-1: Rgba8Unorm texel -> float (input texture orb.rs:116-121). */
float orc_unorm8(uint8_t b) { return (float)b / 255.0f; }

/* fast.wgsl:51-54.  WGSL precedence makes this (num >> count) | ((num << (16-count)) & 0xffff). */
static uint32_t rotate_bits_16(uint32_t num, uint32_t count) {
    return (num >> count) | ((num << (16u - count)) & 0x0000ffffu);
}

/* fast.wgsl:56-60 */
uint32_t orc_detect_streak_16(uint32_t x) {
    uint32_t o_6 = x & rotate_bits_16(x, 6u);
    uint32_t o_3 = o_6 & rotate_bits_16(o_6, 3u);
    return o_3 & rotate_bits_16(o_3, 2u) & rotate_bits_16(o_3, 1u);
}

/* CRD-9: the canonical atan2 used for fast.wgsl:115.  Only + - * / on binary32, each rounded on
 * its own; octant reduction, then a degree-4 odd polynomial in the reduced argument (the
 * classic Cephes atanf coefficients).  atan2(0,0) = 0. */
float orc_atan2f(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float a = mn / mx;
    float t = a, base = 0.0f;
    if (a > 0.41421356f) {
        t = (a - 1.0f) / (a + 1.0f);
        base = 0.78539816f;
    }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = (p * z) * t + t;
    r = base + r;
    if (ay > ax) r = 1.57079632679f - r;
    if (x < 0.0f) r = 3.14159265f - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* fast.wgsl:153 `u32(angle * 1000.0)`: negative angles saturate to 0. */
uint32_t orc_angle_code(float cy, float cx) { return orc_angle_code_neg(cy, cx, ORC_NEG_ZERO); }

/* Implementation-defined point (orc_impl_t::neg_angle): naga 0.20 emits `u32(f32)` as a bare OpConvertFToU, which Vulkan leaves UNDEFINED
 * for a negative operand -- and the angle is negative for more than half of the keypoints of the test frames (Q7).  GPUs saturate to 0 (the default, ORC_NEG_ZERO).  A CPU
 * adapter need not: LLVM's fptoui compiled for x86-64 converts through a 64-bit signed integer and keeps the low word, so -1234.5 becomes
 * 2^32 - 1234 (ORC_NEG_WRAP -- lavapipe / llvmpipe, the kind of adapter BASELINE.json configs[0] names), and with AVX-512's unsigned
 * conversion every out-of-range operand becomes 0xffffffff (ORC_NEG_ONES).  Such a code then reaches brief.wgsl:35 as an angle of
 * millions of radians, whose cos / sin are the adapter's own business: tools/pin_oracle.py recognises the pattern and compares only the
 * keypoints with non-negative angles. */
uint32_t orc_angle_code_neg(float cy, float cx, uint32_t neg) {
    float r = orc_atan2f(cy, cx);
    if (cy < 0.0f || r < 0.0f) {
        if (neg == ORC_NEG_WRAP) return (uint32_t)(int64_t)truncf(r * 1000.0f); /* two's complement of the truncated value; -0.x -> 0 */
        if (neg == ORC_NEG_ONES) return truncf(r * 1000.0f) < 0.0f ? 0xffffffffu : 0u;
        return 0u;
    }
    return (uint32_t)truncf(r * 1000.0f);
}

/* ------------------------------------------------------------------------------------------
 * pyramid geometry
 * ---------------------------------------------------------------------------------------- */

void orc_pyramid_layout(uint32_t W, uint32_t H, uint32_t depth, orc_pyramid_t *p) {
    memset(p, 0, sizeof(*p));
    p->depth = depth;
    size_t off = 0;
    for (uint32_t m = 0; m < depth && m < ORC_MAX_LEVELS; m++) {
        uint32_t w = W >> m, h = H >> m;
        p->w[m] = w ? w : 1;
        p->h[m] = h ? h : 1;
        p->offset[m] = off;
        off += (size_t)p->w[m] * p->h[m];
    }
    p->total = off;
}

/* ------------------------------------------------------------------------------------------
 * K1 grayscale.wgsl:12-38 (pass at orb.rs:478-496)
 *   texcoord = position*0.5+0.5 maps framebuffer row 0 to v = 1: output row y samples input row
 *   H-1-y at its texel centre (bilinear weight exactly 1, CRD-1); luminance per CRD-2.
 * ---------------------------------------------------------------------------------------- */
void orc_grayscale(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray) { orc_grayscale_impl(rgba, W, H, gray, 0); }

/* Implementation-defined point (CRD-13): WGSL lets a shader compiler contract a product and a sum into one fused
 * multiply-add, and `dot()` (grayscale.wgsl:36) has no evaluation order of its own.  Four forms a real compiler can produce,
 * chosen by orc_impl_t::contract & ORC_CONTRACT_LUM and orc_impl_t::dot_order:
 *   contract 0, order 0   ((r*wr + g*wg) + b*wb) [+ a*0]   every product and sum rounded on its own, first component first
 *                         (CRD-2, the default);
 *   contract 1, order 0   t = r*wr; t = fma(g, wg, t); t = fma(b, wb, t) [; fma(a, 0, t) = t]   one multiply and a chain of
 *                         fmas, first component first (LLVM-based compilers);
 *   contract 0, order 1   ((a*0 + b*wb) + g*wg) + r*wr = (b*wb + g*wg) + r*wr   the reduction from the LAST component down --
 *                         Mesa's NIR lowers an inexact fdot that way (nir_lower_alu_to_scalar, reverse_order), also on
 *                         hardware without an fma such as the reference's stated target, a Raspberry Pi 5 (README.md:20);
 *   contract 1, order 1   t = a*0 = +0; t = fma(b, wb, t) = fl(b*wb); t = fma(g, wg, t); t = fma(r, wr, t)   the same
 *                         lowering on hardware with an fma.
 * The alpha term is +0 in every form (alpha is finite) and changes nothing. */
static float luminance_form(float r, float g, float b, uint32_t contract, uint32_t last_first) {
    const float wr = 0.229f, wg = 0.587f, wb = 0.114f; /* grayscale.wgsl:36: 0.229, not 0.299 */
    if (!last_first) {
        if (contract) {
            float t = wr * r;
            t = fmaf(g, wg, t);
            return fmaf(b, wb, t);
        }
        float pr = wr * r;
        float pg = wg * g;
        float pb = wb * b;
        return (pr + pg) + pb;
    }
    if (contract) {
        float t = wb * b;
        t = fmaf(g, wg, t);
        return fmaf(r, wr, t);
    }
    float pb = wb * b;
    float pg = wg * g;
    float pr = wr * r;
    return (pb + pg) + pr;
}

void orc_grayscale_fp(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray, const orc_impl_t *impl) {
    const uint32_t ct = impl->contract & ORC_CONTRACT_LUM, lf = impl->dot_order, rtz = impl->f16_round;
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t *src = rgba + (size_t)(H - 1 - y) * W * 4;
        for (uint32_t x = 0; x < W; x++) {
            float r = orc_unorm8(src[4 * x + 0]);
            float g = orc_unorm8(src[4 * x + 1]);
            float b = orc_unorm8(src[4 * x + 2]);
            gray[(size_t)y * W + x] = orc_f32_to_f16_mode(luminance_form(r, g, b, ct, lf), rtz);
        }
    }
}

/* contract != 0: the contracting compiler in source order (round 4's one form) */
void orc_grayscale_impl(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray, uint32_t contract) {
    orc_impl_t impl = {ORC_OOB_ZERO, 0, contract ? ORC_CONTRACT_ALL : 0u, 0, 0, 0};
    orc_grayscale_fp(rgba, W, H, gray, &impl);
}

/* ------------------------------------------------------------------------------------------
 * K2 blit.wgsl:17-36 (passes at orb.rs:413-429): bilinear sample of mip m-1 at the centre of
 * each target texel, no flip.  CRD-4.
 * ---------------------------------------------------------------------------------------- */
static uint32_t clamp_idx(int64_t i, uint32_t n) {
    if (i < 0) return 0;
    if (i > (int64_t)n - 1) return n - 1;
    return (uint32_t)i;
}

/* Implementation-defined point: the weight a sampler gives the second texel of a bilinear pair.  WGSL
 * leaves the filter's precision to the adapter; GPUs commonly hold it in 8 fractional bits.  bits == 0: the exact
 * binary32 fraction (CRD-5); bits == n: the fraction rounded to the nearest multiple of 2^-n (halves up). */
static float sampler_weight(float f, uint32_t bits) {
    if (bits == 0 || bits > 23) return f;
    float s = (float)(1u << bits);
    return floorf(f * s + 0.5f) / s; /* s is a power of two: product and quotient are exact */
}

void orc_mip(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd) {
    orc_mip_impl(src, ws, hs, dst, wd, hd, 0);
}

void orc_mip_impl(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd, uint32_t wbits) {
    orc_impl_t impl = {ORC_OOB_ZERO, wbits, 0, 0, 0, 0};
    orc_mip_fp(src, ws, hs, dst, wd, hd, &impl);
}

/* The blit is one textureSample per target texel (blit.wgsl:35): sampler arithmetic, nothing a shader compiler contracts; the
 * store to the R16Float level follows orc_impl_t::f16_round. */
void orc_mip_fp(const uint16_t *src, uint32_t ws, uint32_t hs, uint16_t *dst, uint32_t wd, uint32_t hd, const orc_impl_t *impl) {
    const uint32_t wbits = impl->sampler_weight_bits, rtz = impl->f16_round;
    if (ws == 2 * wd && hs == 2 * hd) { /* weights are exactly 1/2 at any precision */
        for (uint32_t y = 0; y < hd; y++)
            for (uint32_t x = 0; x < wd; x++) {
                float a = orc_f16_to_f32(src[(size_t)(2 * y) * ws + 2 * x]);
                float b = orc_f16_to_f32(src[(size_t)(2 * y) * ws + 2 * x + 1]);
                float c = orc_f16_to_f32(src[(size_t)(2 * y + 1) * ws + 2 * x]);
                float d = orc_f16_to_f32(src[(size_t)(2 * y + 1) * ws + 2 * x + 1]);
                float top = a + b;
                float bot = c + d;
                float v = (top + bot) * 0.25f;
                dst[(size_t)y * wd + x] = orc_f32_to_f16_mode(v, rtz);
            }
        return;
    }
    float rx = (float)ws / (float)wd;
    float ry = (float)hs / (float)hd;
    for (uint32_t y = 0; y < hd; y++) {
        float sy = ((float)y + 0.5f) * ry - 0.5f;
        float fy0 = floorf(sy);
        float fy = sampler_weight(sy - fy0, wbits);
        uint32_t y0 = clamp_idx((int64_t)fy0, hs), y1 = clamp_idx((int64_t)fy0 + 1, hs);
        for (uint32_t x = 0; x < wd; x++) {
            float sx = ((float)x + 0.5f) * rx - 0.5f;
            float fx0 = floorf(sx);
            float fx = sampler_weight(sx - fx0, wbits);
            uint32_t x0 = clamp_idx((int64_t)fx0, ws), x1 = clamp_idx((int64_t)fx0 + 1, ws);
            float a = orc_f16_to_f32(src[(size_t)y0 * ws + x0]);
            float b = orc_f16_to_f32(src[(size_t)y0 * ws + x1]);
            float c = orc_f16_to_f32(src[(size_t)y1 * ws + x0]);
            float d = orc_f16_to_f32(src[(size_t)y1 * ws + x1]);
            float dab = b - a;
            float top = a + fx * dab;
            float dcd = d - c;
            float bot = c + fx * dcd;
            float dtb = bot - top;
            float v = top + fy * dtb;
            dst[(size_t)y * wd + x] = orc_f32_to_f16_mode(v, rtz);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * K3/K4 gaussian_blur_x.wgsl:14-26, 32-41, 53-60.  The SAME shader is used for both blur passes
 * (orb.rs:399-402 builds the "gaussian_blur_y" pipeline from module "gaussian_blur_x").
 * Offsets are added to the normalised u coordinate (not scaled by 1/w), sampler is bilinear
 * clamp-to-edge (orb.rs:123-139), and the vertex shader flips v like grayscale does.  CRD-5.
 * ---------------------------------------------------------------------------------------- */
static const float BLUR_OFFSETS[4] = {-2.2273038885157046f, -0.4391873198428642f, 1.3243948342247673f, 3.0f};
static const float BLUR_WEIGHTS[4] = {0.13748623236806098f, 0.5037756553768409f, 0.32748695702046415f,
                                      0.031251155234634016f};

void orc_blur_pass(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst) { orc_blur_pass_impl(src, w, h, dst, 0); }
void orc_blur_pass_impl(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, uint32_t wbits) {
    orc_blur_pass_impl2(src, w, h, dst, wbits, 0);
}

/* contract (CRD-13): `result += textureSample(..) * weight` (gaussian_blur_x.wgsl:58) as one fma per tap.  The bilinear
 * filter itself is the sampler's arithmetic, not the shader's: it keeps CRD-5 (and the weight precision switch). */
void orc_blur_pass_impl2(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, uint32_t wbits, uint32_t contract) {
    orc_impl_t impl = {ORC_OOB_ZERO, wbits, contract ? ORC_CONTRACT_ALL : 0u, 0, 0, 0};
    orc_blur_pass_fp(src, w, h, dst, &impl);
}

/* orc_impl_t::contract & ORC_CONTRACT_BLUR; the loop is sequential as written (no order for a compiler to choose), so
 * dot_order does not apply; the store follows f16_round. */
void orc_blur_pass_fp(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, const orc_impl_t *impl) {
    const uint32_t wbits = impl->sampler_weight_bits, contract = impl->contract & ORC_CONTRACT_BLUR, rtz = impl->f16_round;
    float fw = (float)w;
    for (uint32_t y = 0; y < h; y++) {
        const uint16_t *row = src + (size_t)(h - 1 - y) * w; /* flipped v, sampled at the row centre */
        for (uint32_t x = 0; x < w; x++) {
            float u = ((float)x + 0.5f) / fw;
            float acc = 0.0f;
            for (int i = 0; i < 4; i++) {
                float uo = u + BLUR_OFFSETS[i];
                float coord = uo * fw - 0.5f;
                float c0 = floorf(coord);
                float f = sampler_weight(coord - c0, wbits);
                uint32_t i0 = clamp_idx((int64_t)c0, w), i1 = clamp_idx((int64_t)c0 + 1, w);
                float t0 = orc_f16_to_f32(row[i0]);
                float t1 = orc_f16_to_f32(row[i1]);
                float d = t1 - t0;
                float s = t0 + f * d;
                if (contract) {
                    acc = fmaf(s, BLUR_WEIGHTS[i], acc);
                } else {
                    float ws = s * BLUR_WEIGHTS[i];
                    acc = acc + ws;
                }
            }
            dst[(size_t)y * w + x] = orc_f32_to_f16_mode(acc, rtz);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * K5 fast.wgsl:62-159, dispatched once per octave (orb.rs:504-520).
 * ---------------------------------------------------------------------------------------- */
static const int RING4[4][2] = {{3, 0}, {-3, 0}, {0, 3}, {0, -3}}; /* fast.wgsl:25-30 */
static const int RING16[16][2] = {{-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}, {0, -3}, {1, -3}, {2, -2}, {3, -1},
                                  {3, 0},  {3, 1},   {2, 2},   {1, 3},   {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}};

/* textureLoad of one mip level.  Implementation-defined point: what a load OUTSIDE the level returns
 * (fast.wgsl:78,86,103 at octaves >= 1, whose guard uses the level-0 size; brief.wgsl:59-60 for samples that leave the
 * level).  WGSL allows either of two behaviours and naga 0.20 picks by adapter (wgpu-hal's Vulkan backend: image_load =
 * Unchecked when the device has robustImageAccess, else Restrict):
 *   ORC_OOB_ZERO   the load returns 0 (Vulkan robust image access) -- CRD-6, the default;
 *   ORC_OOB_CLAMP  every coordinate is clamped into [0, size - 1] on its own;
 *   ORC_OOB_UMIN   naga's `Restrict` as its SPIR-V writer emits it: min(coordinate AS UNSIGNED, size - 1), so a
 *                  negative coordinate lands on the level's LAST column / row, not its first. */
static float level_load_p(const uint16_t *pyr, const orc_pyramid_t *lay, uint32_t lvl, int64_t x, int64_t y, uint32_t oob) {
    const int64_t w = (int64_t)lay->w[lvl], h = (int64_t)lay->h[lvl];
    if (x < 0 || y < 0 || x >= w || y >= h) {
        if (oob == ORC_OOB_ZERO) return 0.0f;
        if (oob == ORC_OOB_CLAMP) {
            x = x < 0 ? 0 : (x >= w ? w - 1 : x);
            y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        } else {
            x = (x < 0 || x >= w) ? w - 1 : x;
            y = (y < 0 || y >= h) ? h - 1 : y;
        }
    }
    return orc_f16_to_f32(pyr[lay->offset[lvl] + (size_t)y * lay->w[lvl] + (size_t)x]);
}
static float level_load(const uint16_t *pyr, const orc_pyramid_t *lay, uint32_t lvl, int64_t x, int64_t y) {
    return level_load_p(pyr, lay, lvl, x, y, ORC_OOB_ZERO);
}

void orc_fast(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, orc_corner_t *out, uint32_t cap,
              uint32_t *total) {
    orc_fast_impl(pyr, lay, threshold, ORC_OOB_ZERO, out, cap, total);
}

void orc_fast_impl(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t oob, orc_corner_t *out,
                   uint32_t cap, uint32_t *total) {
    orc_fast_impl2(pyr, lay, threshold, oob, ORC_NEG_ZERO, out, cap, total);
}

void orc_fast_impl2(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t oob, uint32_t neg, orc_corner_t *out,
                    uint32_t cap, uint32_t *total) {
    uint32_t count = 0;
    uint32_t W0 = lay->w[0], H0 = lay->h[0];
    uint32_t lim_x = W0 - 16u, lim_y = H0 - 16u; /* textureDimensions(texture) is the level-0 size; u32 wrap kept */
    uint32_t width = W0, height = H0;             /* orb.rs:501-502 */
    for (uint32_t oct = 0; oct < lay->depth; oct++) {
        uint32_t gw = ((width + 7) / 8) * 8, gh = ((height + 7) / 8) * 8; /* orb.rs:511-515, 8x8 groups */
        for (uint32_t gy = 0; gy < gh; gy++)
            for (uint32_t gx = 0; gx < gw; gx++) {
                if (!(gx > 16u && gy > 16u && gx < lim_x && gy < lim_y)) continue; /* fast.wgsl:77 */
                float c = level_load_p(pyr, lay, oct, gx, gy, oob);
                uint32_t num_over = 0, num_under = 0;
                for (int i = 0; i < 4; i++) { /* fast.wgsl:85-93 */
                    float v = level_load_p(pyr, lay, oct, (int64_t)gx + RING4[i][0], (int64_t)gy + RING4[i][1], oob);
                    float diff = v - c;
                    if (diff > threshold)
                        num_over++;
                    else if (diff < -threshold)
                        num_under++;
                }
                if (!(num_over >= 3 || num_under >= 3)) continue; /* fast.wgsl:95 */
                uint32_t is_over = 0, is_under = 0;
                float cx = 0.0f, cy = 0.0f;
                for (int i = 0; i < 16; i++) { /* fast.wgsl:102-113 */
                    float v = level_load_p(pyr, lay, oct, (int64_t)gx + RING16[i][0], (int64_t)gy + RING16[i][1], oob);
                    float diff = v - c;
                    float px = v * (float)RING16[i][0];
                    float py = v * (float)RING16[i][1];
                    cx = cx + px;
                    cy = cy + py;
                    if (diff > threshold)
                        is_over |= 1u << i;
                    else if (diff < -threshold)
                        is_under |= 1u << i;
                }
                uint32_t streak = orc_detect_streak_16(is_over) | orc_detect_streak_16(is_under);
                if (streak > 0u) { /* fast.wgsl:121-157 */
                    if (count < cap) {
                        out[count].x = gx;
                        out[count].y = gy;
                        out[count].angle = orc_angle_code_neg(cy, cx, neg);
                        out[count].octave = oct;
                    }
                    count++;
                }
            }
        width /= 2; /* orb.rs:517-518 */
        height /= 2;
    }
    *total = count;
}

/* ------------------------------------------------------------------------------------------
 * K6 brief.wgsl:20-68.  One u32 word per (feature, global_id.x); bit i of word k <-> pattern
 * row 32k+i (brief.wgsl:47).  cos/sin per CRD-10 (double libm rounded to binary32; a test
 * checks all 3142 codes against the product's committed table).
 * ---------------------------------------------------------------------------------------- */
void orc_brief(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
               orc_descriptor_t *out) {
    orc_brief_impl(blur_pyr, lay, corners, n, ORC_OOB_ZERO, out);
}

void orc_brief_impl(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                    uint32_t oob, orc_descriptor_t *out) {
    orc_brief_impl2(blur_pyr, lay, corners, n, oob, 0, out);
}

/* contract (CRD-13): `rotation_matrix * p` (brief.wgsl:53-54) is p.x * column 0 + p.y * column 1: (ct*x + st*y, -st*x + ct*y).
 * Unfused the two-term sum has one value whatever the order.  Contracted (orc_impl_t::contract & ORC_CONTRACT_ROT), one of the
 * two products is fused into the sum, and which one is the compiler's choice (orc_impl_t::dot_order):
 *   order 0   the first term is the product, the second is fused onto it: (fma(st, y, ct*x), fma(ct, y, -st*x));
 *   order 1   the last column first -- Mesa's spirv_to_nir builds matrix * vector from the last column down:
 *             (fma(ct, x, st*y), fma(-st, x, ct*y)). */
void orc_brief_rotate(uint32_t angle_code, int px, int py, uint32_t contract, uint32_t last_first, float *rx, float *ry) {
    float theta = (float)angle_code / 1000.0f; /* brief.wgsl:35 */
    float ct = (float)cos((double)theta);
    float st = (float)sin((double)theta);
    float nst = -st;
    float x = (float)px, y = (float)py;
    /* mat2x2f(ct,-st, st,ct) * p, column-major  brief.wgsl:38-54 */
    float x0 = ct * x, x1 = st * y, y0 = nst * x, y1 = ct * y;
    if (!contract) {
        *rx = x0 + x1;
        *ry = y0 + y1;
    } else if (!last_first) {
        *rx = fmaf(st, y, x0);
        *ry = fmaf(ct, y, y0);
    } else {
        *rx = fmaf(ct, x, x1);
        *ry = fmaf(nst, x, y1);
    }
}

void orc_brief_impl2(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                     uint32_t oob, uint32_t contract, orc_descriptor_t *out) {
    orc_impl_t impl = {oob, 0, contract ? ORC_CONTRACT_ALL : 0u, 0, 0, 0};
    orc_brief_fp(blur_pyr, lay, corners, n, &impl, out);
}

void orc_brief_fp(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                  const orc_impl_t *impl, orc_descriptor_t *out) {
    const uint32_t oob = impl->oob, contract = impl->contract & ORC_CONTRACT_ROT, lf = impl->dot_order;
    for (uint32_t fidx = 0; fidx < n; fidx++) {
        const orc_corner_t *k = &corners[fidx];
        uint32_t oct = k->octave;
        for (uint32_t word = 0; word < 8; word++) {
            uint32_t bits = 0;
            for (uint32_t i = 0; i < 32; i++) {
                const int8_t *row = &ORC_BRIEF_PATTERN[4 * ((word << 5) | i)];
                float rax, ray, rbx, rby;
                orc_brief_rotate(k->angle, row[0], row[1], contract, lf, &rax, &ray);
                orc_brief_rotate(k->angle, row[2], row[3], contract, lf, &rbx, &rby);
                int64_t tax = (int64_t)(int32_t)rax + (int64_t)(int32_t)k->x; /* vec2i() truncates */
                int64_t tay = (int64_t)(int32_t)ray + (int64_t)(int32_t)k->y;
                int64_t tbx = (int64_t)(int32_t)rbx + (int64_t)(int32_t)k->x;
                int64_t tby = (int64_t)(int32_t)rby + (int64_t)(int32_t)k->y;
                float va = 0.0f, vb = 0.0f;
                if (oct < lay->depth) {
                    va = level_load_p(blur_pyr, lay, oct, tax, tay, oob);
                    vb = level_load_p(blur_pyr, lay, oct, tbx, tby, oob);
                }
                if (va > vb) bits |= 1u << i; /* brief.wgsl:62-64 */
            }
            out[fidx].word[word] = bits;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * whole frame, orb.rs:469-557: grayscale -> mips -> blur A (all levels) -> blur B (all levels)
 * -> FAST per octave -> BRIEF.
 * ---------------------------------------------------------------------------------------- */
/* Y8 input variant -- NOT in the reference's code; its README lists "Use Y channel of YUV stream directly" as a roadmap
 * item (README.md:42).  Definition of the build: the grey image is the Y plane itself, gray(x,y) = f16(Y(x, H-1-y)/255):
 * the luminance dot product of grayscale.wgsl:31-38 is replaced by the sample (CRD-1 for the byte, CRD-3 for the store)
 * and the vertical mirror of the full-screen pass (Q2, grayscale.wgsl:16-25) is kept, so that everything downstream --
 * mips, blur, detector, descriptors and every keypoint coordinate -- is the literal path's, unchanged. */
void orc_grayscale_y8(const uint8_t *y8, uint32_t W, uint32_t H, uint16_t *gray) {
    orc_grayscale_y8_fp(y8, W, H, gray, &ORC_IMPL_DEFAULT);
}
void orc_grayscale_y8_fp(const uint8_t *y8, uint32_t W, uint32_t H, uint16_t *gray, const orc_impl_t *impl) {
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t *src = y8 + (size_t)(H - 1 - y) * W;
        for (uint32_t x = 0; x < W; x++) gray[(size_t)y * W + x] = orc_f32_to_f16_mode(orc_unorm8(src[x]), impl->f16_round);
    }
}


static int extract_impl(const uint8_t *frame, int y8, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                        uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                        uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr);

int orc_extract_impl(const uint8_t *frame, int y8, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                     uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                     uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr) {
    if (impl && (impl->oob > ORC_OOB_UMIN || impl->sampler_weight_bits > 23 || impl->contract > ORC_CONTRACT_ALL || impl->dot_order > 1 ||
                 impl->f16_round > 1 || impl->neg_angle > ORC_NEG_ONES))
        return -1;
    return extract_impl(frame, y8, W, H, depth, threshold, max_features, impl ? impl : &ORC_IMPL_DEFAULT, corners, descriptors,
                        total, gray_pyr, blur_pyr);
}

int orc_extract(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr,
                uint16_t *blur_pyr) {
    return extract_impl(rgba, 0, W, H, depth, threshold, max_features, &ORC_IMPL_DEFAULT, corners, descriptors, total, gray_pyr,
                        blur_pyr);
}

int orc_extract_y8(const uint8_t *y8, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                   orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr,
                   uint16_t *blur_pyr) {
    return extract_impl(y8, 1, W, H, depth, threshold, max_features, &ORC_IMPL_DEFAULT, corners, descriptors, total, gray_pyr,
                        blur_pyr);
}

static int extract_impl(const uint8_t *rgba, int y8, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                        uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                        uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr) {
    const uint32_t oob = impl->oob;
    if (!rgba || !W || !H || depth < 1 || depth > ORC_MAX_LEVELS || !total) return -1;
    orc_pyramid_t lay;
    orc_pyramid_layout(W, H, depth, &lay);
    uint16_t *gray = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *tmp = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *blur = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    if (!gray || !tmp || !blur) {
        free(gray);
        free(tmp);
        free(blur);
        return -1;
    }
    if (y8)
        orc_grayscale_y8_fp(rgba, W, H, gray, impl);
    else
        orc_grayscale_fp(rgba, W, H, gray, impl);
    for (uint32_t m = 1; m < depth; m++)
        orc_mip_fp(gray + lay.offset[m - 1], lay.w[m - 1], lay.h[m - 1], gray + lay.offset[m], lay.w[m], lay.h[m], impl);
    for (uint32_t m = 0; m < depth; m++) orc_blur_pass_fp(gray + lay.offset[m], lay.w[m], lay.h[m], tmp + lay.offset[m], impl);
    for (uint32_t m = 0; m < depth; m++) orc_blur_pass_fp(tmp + lay.offset[m], lay.w[m], lay.h[m], blur + lay.offset[m], impl);
    uint32_t count = 0;
    orc_fast_impl2(gray, &lay, threshold, oob, impl->neg_angle, corners, max_features, &count);
    uint32_t stored = count < max_features ? count : max_features;
    if (descriptors) orc_brief_fp(blur, &lay, corners, stored, impl, descriptors);
    *total = count;
    if (gray_pyr) memcpy(gray_pyr, gray, lay.total * sizeof(uint16_t));
    if (blur_pyr) memcpy(blur_pyr, blur, lay.total * sizeof(uint16_t));
    free(gray);
    free(tmp);
    free(blur);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Opt-in extensions: arc length other than 12 and 3x3 non-maximum suppression (definitions in orb_oracle.h).
 * ---------------------------------------------------------------------------------------- */
static int has_run(uint32_t mask16, uint32_t arc) {
    uint32_t dbl = mask16 | (mask16 << 16);
    uint32_t want = arc >= 16 ? 0xffffu : ((1u << arc) - 1u);
    for (int s = 0; s < 16; s++)
        if (((dbl >> s) & want) == want) return 1;
    return 0;
}

void orc_fast_ex(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t arc, orc_corner_t *out,
                 float *scores, uint32_t cap, uint32_t *total) {
    if (arc == 0) arc = 12;
    uint32_t count = 0;
    uint32_t W0 = lay->w[0], H0 = lay->h[0];
    uint32_t lim_x = W0 - 16u, lim_y = H0 - 16u;
    uint32_t width = W0, height = H0;
    for (uint32_t oct = 0; oct < lay->depth; oct++) {
        uint32_t gw = ((width + 7) / 8) * 8, gh = ((height + 7) / 8) * 8;
        for (uint32_t gy = 0; gy < gh; gy++)
            for (uint32_t gx = 0; gx < gw; gx++) {
                if (!(gx > 16u && gy > 16u && gx < lim_x && gy < lim_y)) continue;
                float c = level_load(pyr, lay, oct, gx, gy);
                uint32_t is_over = 0, is_under = 0;
                float cx = 0.0f, cy = 0.0f, s_over = 0.0f, s_under = 0.0f;
                for (int i = 0; i < 16; i++) {
                    float v = level_load(pyr, lay, oct, (int64_t)gx + RING16[i][0], (int64_t)gy + RING16[i][1]);
                    float diff = v - c;
                    float px = v * (float)RING16[i][0];
                    float py = v * (float)RING16[i][1];
                    cx = cx + px;
                    cy = cy + py;
                    if (diff > threshold) {
                        is_over |= 1u << i;
                        float e = diff - threshold;
                        s_over = s_over + e;
                    } else if (diff < -threshold) {
                        is_under |= 1u << i;
                        float nd = -diff;
                        float e = nd - threshold;
                        s_under = s_under + e;
                    }
                }
                int ro = has_run(is_over, arc), ru = has_run(is_under, arc);
                if (ro || ru) {
                    if (count < cap) {
                        out[count].x = gx;
                        out[count].y = gy;
                        out[count].angle = orc_angle_code(cy, cx);
                        out[count].octave = oct;
                        if (scores) scores[count] = ro ? s_over : s_under;
                    }
                    count++;
                }
            }
        width /= 2;
        height /= 2;
    }
    *total = count;
}

uint32_t orc_nms(const orc_pyramid_t *lay, const orc_corner_t *in, const float *scores, uint32_t n, orc_corner_t *out) {
    /* score planes per octave: 0 = no corner (a corner's score is > 0: its run has >= 9 terms each > 0) */
    float *plane = (float *)calloc(lay->total ? lay->total * 4 : 1, sizeof(float));
    size_t *off = (size_t *)calloc(ORC_MAX_LEVELS, sizeof(size_t));
    uint32_t gws[ORC_MAX_LEVELS], ghs[ORC_MAX_LEVELS];
    size_t acc = 0;
    uint32_t width = lay->w[0], height = lay->h[0];
    for (uint32_t m = 0; m < lay->depth; m++) { /* corners live on the 8-rounded dispatch grid of their octave */
        gws[m] = ((width + 7) / 8) * 8 + 2;
        ghs[m] = ((height + 7) / 8) * 8 + 2;
        off[m] = acc;
        acc += (size_t)gws[m] * ghs[m];
        width /= 2;
        height /= 2;
    }
    free(plane);
    plane = (float *)calloc(acc ? acc : 1, sizeof(float));
    for (uint32_t i = 0; i < n; i++)
        plane[off[in[i].octave] + (size_t)(in[i].y + 1) * gws[in[i].octave] + in[i].x + 1] = scores[i];
    uint32_t kept = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t o = in[i].octave;
        float s = scores[i];
        int keep = 1;
        for (int dy = -1; dy <= 1 && keep; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy) continue;
                float t = plane[off[o] + (size_t)((int)in[i].y + 1 + dy) * gws[o] + (size_t)((int)in[i].x + 1 + dx)];
                if (t <= 0.0f) continue; /* not a corner */
                int later = dy > 0 || (dy == 0 && dx > 0); /* neighbour comes later in raster order */
                if (t > s || (t == s && !later)) {
                    keep = 0;
                    break;
                }
            }
        if (keep) out[kept++] = in[i];
    }
    free(plane);
    free(off);
    return kept;
}

int orc_extract_ex(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold, uint32_t max_features,
                   const orc_options_t *opt, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *total) {
    if (!rgba || !W || !H || depth < 1 || depth > ORC_MAX_LEVELS || !total) return -1;
    uint32_t arc = opt && opt->arc ? opt->arc : 12;
    if (arc < 9 || arc > 16) return -1;
    orc_pyramid_t lay;
    orc_pyramid_layout(W, H, depth, &lay);
    uint16_t *gray = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *tmp = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *blur = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    orc_grayscale(rgba, W, H, gray);
    for (uint32_t m = 1; m < depth; m++)
        orc_mip(gray + lay.offset[m - 1], lay.w[m - 1], lay.h[m - 1], gray + lay.offset[m], lay.w[m], lay.h[m]);
    for (uint32_t m = 0; m < depth; m++) orc_blur_pass(gray + lay.offset[m], lay.w[m], lay.h[m], tmp + lay.offset[m]);
    for (uint32_t m = 0; m < depth; m++) orc_blur_pass(tmp + lay.offset[m], lay.w[m], lay.h[m], blur + lay.offset[m]);
    /* provisional list: every detection (the NMS must see all of them) */
    uint32_t cap_all = 0;
    for (uint32_t m = 0; m < depth; m++) cap_all += (((W >> m) + 7) / 8 * 8) * (((H >> m) + 7) / 8 * 8);
    orc_corner_t *all = (orc_corner_t *)malloc((size_t)(cap_all ? cap_all : 1) * sizeof(orc_corner_t));
    float *scores = (float *)malloc((size_t)(cap_all ? cap_all : 1) * sizeof(float));
    uint32_t n = 0;
    orc_fast_ex(gray, &lay, threshold, arc, all, scores, cap_all, &n);
    if (opt && opt->nms) {
        orc_corner_t *kept = (orc_corner_t *)malloc((size_t)(n ? n : 1) * sizeof(orc_corner_t));
        n = orc_nms(&lay, all, scores, n, kept);
        memcpy(all, kept, (size_t)n * sizeof(orc_corner_t));
        free(kept);
    }
    uint32_t stored = n < max_features ? n : max_features;
    memcpy(corners, all, (size_t)stored * sizeof(orc_corner_t));
    if (descriptors) orc_brief(blur, &lay, corners, stored, descriptors);
    *total = n;
    free(all);
    free(scores);
    free(gray);
    free(tmp);
    free(blur);
    return 0;
}

int orc_extract_batch(const uint8_t *rgba, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                      uint32_t max_features, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *totals,
                      int n_threads) {
    int rc = 0;
    size_t frame_bytes = (size_t)W * H * 4;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t f = 0; f < (int64_t)n_frames; f++) {
        int r = orc_extract(rgba + (size_t)f * frame_bytes, W, H, depth, threshold, max_features,
                            corners + (size_t)f * max_features,
                            descriptors ? descriptors + (size_t)f * max_features : NULL, &totals[f], NULL, NULL);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* The same under a setting of the implementation-defined switches (bench.py --contract: the CPU leg follows the GPU's arithmetic). */
int orc_extract_batch_impl(const uint8_t *frames, int y8, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                           uint32_t max_features, const orc_impl_t *impl, orc_corner_t *corners, orc_descriptor_t *descriptors,
                           uint32_t *totals, int n_threads) {
    int rc = 0;
    size_t frame_bytes = (size_t)W * H * (y8 ? 1 : 4);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t f = 0; f < (int64_t)n_frames; f++) {
        int r = orc_extract_impl(frames + (size_t)f * frame_bytes, y8, W, H, depth, threshold, max_features, impl,
                                 corners + (size_t)f * max_features, descriptors ? descriptors + (size_t)f * max_features : NULL,
                                 &totals[f], NULL, NULL);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* The same over Y8 frames (one byte per pixel). */
int orc_extract_batch_y8(const uint8_t *y8, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                         uint32_t max_features, orc_corner_t *corners, orc_descriptor_t *descriptors, uint32_t *totals,
                         int n_threads) {
    int rc = 0;
    size_t frame_bytes = (size_t)W * H;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t f = 0; f < (int64_t)n_frames; f++) {
        int r = orc_extract_y8(y8 + (size_t)f * frame_bytes, W, H, depth, threshold, max_features,
                               corners + (size_t)f * max_features,
                               descriptors ? descriptors + (size_t)f * max_features : NULL, &totals[f], NULL, NULL);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * "intended" mode: the algorithm the reference's README describes, with the literal
 * shaders' defects (Q1, Q2, Q7, Q8, Q11, Q12, Q14) repaired.  NOT in the reference: there is no parity target,
 * the definitions IM-1..IM-8 in orb_oracle.h are the build's own.
 * ---------------------------------------------------------------------------------------- */
void orc_grayscale_intended(const uint8_t *rgba, uint32_t W, uint32_t H, uint16_t *gray) { /* IM-1 */
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t *src = rgba + (size_t)y * W * 4;
        for (uint32_t x = 0; x < W; x++) {
            float r = orc_unorm8(src[4 * x + 0]);
            float g = orc_unorm8(src[4 * x + 1]);
            float b = orc_unorm8(src[4 * x + 2]);
            float pr = 0.299f * r;
            float pg = 0.587f * g;
            float pb = 0.114f * b;
            float lum = (pr + pg) + pb;
            gray[(size_t)y * W + x] = orc_f32_to_f16(lum);
        }
    }
}

/* IM-3: the reference's four bilinear taps read in texel units are this symmetric 7-tap kernel
 * (gaussian_blur_x.wgsl:14-26; centre first). */
static const float GAUSS[4] = {0.282523781f, 0.221251875f, 0.106235079f, 0.0312511548f};

void orc_gauss_pass(const uint16_t *src, uint32_t w, uint32_t h, uint16_t *dst, int vertical) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            float t[7];
            for (int k = -3; k <= 3; k++) {
                uint32_t xx = vertical ? x : clamp_idx((int64_t)x + k, w);
                uint32_t yy = vertical ? clamp_idx((int64_t)y + k, h) : y;
                t[k + 3] = orc_f16_to_f32(src[(size_t)yy * w + xx]);
            }
            float acc = GAUSS[0] * t[3];
            for (int k = 1; k <= 3; k++) {
                float pair = t[3 - k] + t[3 + k];
                float term = GAUSS[k] * pair;
                acc = acc + term;
            }
            dst[(size_t)y * w + x] = orc_f32_to_f16(acc);
        }
}

uint32_t orc_angle_code_signed(float cy, float cx) { /* IM-5 */
    float r = orc_atan2f(cy, cx);
    if (r < 0.0f) r = r + 6.28318531f;
    float m = truncf(r * 1000.0f);
    uint32_t code = (uint32_t)m;
    return code > 6283u ? 6283u : code;
}

void orc_fast_intended(const uint16_t *pyr, const orc_pyramid_t *lay, float threshold, uint32_t arc, orc_corner_t *out,
                       float *scores, uint32_t cap, uint32_t *total) { /* IM-4, IM-5, score of orc_fast_ex */
    if (arc == 0) arc = 9;
    uint32_t count = 0;
    for (uint32_t oct = 0; oct < lay->depth; oct++) {
        uint32_t w = lay->w[oct], h = lay->h[oct];
        if (w <= 33u || h <= 33u) continue; /* no pixel satisfies 16 < x < w - 16 */
        for (uint32_t gy = 17; gy < h - 16u; gy++)
            for (uint32_t gx = 17; gx < w - 16u; gx++) {
                float c = level_load(pyr, lay, oct, gx, gy);
                uint32_t is_over = 0, is_under = 0;
                float cx = 0.0f, cy = 0.0f, s_over = 0.0f, s_under = 0.0f;
                for (int i = 0; i < 16; i++) {
                    float v = level_load(pyr, lay, oct, (int64_t)gx + RING16[i][0], (int64_t)gy + RING16[i][1]);
                    float diff = v - c;
                    float px = v * (float)RING16[i][0];
                    float py = v * (float)RING16[i][1];
                    cx = cx + px;
                    cy = cy + py;
                    if (diff > threshold) {
                        is_over |= 1u << i;
                        float e = diff - threshold;
                        s_over = s_over + e;
                    } else if (diff < -threshold) {
                        is_under |= 1u << i;
                        float nd = -diff;
                        float e = nd - threshold;
                        s_under = s_under + e;
                    }
                }
                int ro = has_run(is_over, arc), ru = has_run(is_under, arc);
                if (ro || ru) {
                    if (count < cap) {
                        out[count].x = gx;
                        out[count].y = gy;
                        out[count].angle = orc_angle_code_signed(cy, cx);
                        out[count].octave = oct;
                        if (scores) scores[count] = ro ? s_over : s_under;
                    }
                    count++;
                }
            }
    }
    *total = count;
}

/* IM-6b: the angle code the descriptor is rotated by when angles are quantised into `bins` bins of the full circle (0: the
 * keypoint's own milliradian code): bin = code * bins / 6284 (integers), rotation by the bin's centre code
 * (bin * 6284 + 3142) / bins.  The keypoint's reported angle is not changed. */
uint32_t orc_binned_angle_code(uint32_t code, uint32_t bins) {
    if (!bins) return code;
    if (code > 6283u) code = 6283u;
    uint32_t bin = (uint32_t)(((uint64_t)code * bins) / 6284u);
    return (uint32_t)(((uint64_t)bin * 6284u + 3142u) / bins);
}

void orc_brief_intended(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                        orc_descriptor_t *out) { /* IM-6 */
    orc_brief_intended_bins(blur_pyr, lay, corners, n, 0, out);
}

void orc_brief_intended_bins(const uint16_t *blur_pyr, const orc_pyramid_t *lay, const orc_corner_t *corners, uint32_t n,
                             uint32_t angle_bins, orc_descriptor_t *out) { /* IM-6, IM-6b */
    for (uint32_t fidx = 0; fidx < n; fidx++) {
        const orc_corner_t *k = &corners[fidx];
        uint32_t oct = k->octave;
        float theta = (float)orc_binned_angle_code(k->angle, angle_bins) / 1000.0f;
        float ct = (float)cos((double)theta);
        float st = (float)sin((double)theta);
        float nst = -st;
        for (uint32_t word = 0; word < 8; word++) {
            uint32_t bits = 0;
            for (uint32_t i = 0; i < 32; i++) {
                const int8_t *row = &ORC_BRIEF_PATTERN[4 * ((word << 5) | i)];
                float ax = (float)row[0], ay = (float)row[1], bx = (float)row[2], by = (float)row[3];
                /* R(+theta) p = (ct*x - st*y, st*x + ct*y) */
                float a0 = ct * ax, a1 = nst * ay, a2 = st * ax, a3 = ct * ay;
                float b0 = ct * bx, b1 = nst * by, b2 = st * bx, b3 = ct * by;
                float rax = a0 + a1, ray = a2 + a3, rbx = b0 + b1, rby = b2 + b3;
                int64_t tax = (int64_t)(int32_t)rax + (int64_t)k->x, tay = (int64_t)(int32_t)ray + (int64_t)k->y;
                int64_t tbx = (int64_t)(int32_t)rbx + (int64_t)k->x, tby = (int64_t)(int32_t)rby + (int64_t)k->y;
                float va = 0.0f, vb = 0.0f;
                if (oct < lay->depth) {
                    va = level_load(blur_pyr, lay, oct, tax, tay);
                    vb = level_load(blur_pyr, lay, oct, tbx, tby);
                }
                if (va > vb) bits |= 1u << i;
            }
            out[fidx].word[word] = bits;
        }
    }
}

/* IM-8: order "better first": larger score, then smaller (octave, y, x). */
typedef struct {
    float score;
    orc_corner_t c;
} scored_t;
static int scored_cmp(const void *pa, const void *pb) {
    const scored_t *a = (const scored_t *)pa, *b = (const scored_t *)pb;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    if (a->c.octave != b->c.octave) return a->c.octave < b->c.octave ? -1 : 1;
    if (a->c.y != b->c.y) return a->c.y < b->c.y ? -1 : 1;
    if (a->c.x != b->c.x) return a->c.x < b->c.x ? -1 : 1;
    return 0;
}
uint32_t orc_topk(const orc_corner_t *in, const float *scores, uint32_t n, uint32_t k, orc_corner_t *out) {
    if (n <= k) {
        memcpy(out, in, (size_t)n * sizeof(orc_corner_t));
        return n;
    }
    scored_t *all = (scored_t *)malloc((size_t)n * sizeof(scored_t));
    for (uint32_t i = 0; i < n; i++) {
        all[i].score = scores[i];
        all[i].c = in[i];
    }
    qsort(all, n, sizeof(scored_t), scored_cmp);
    for (uint32_t i = 0; i < k; i++) out[i] = all[i].c;
    free(all);
    return k;
}

int orc_extract_intended(const uint8_t *rgba, uint32_t W, uint32_t H, uint32_t depth, float threshold,
                         uint32_t max_features, const orc_options_t *opt, orc_corner_t *corners,
                         orc_descriptor_t *descriptors, uint32_t *total, uint16_t *gray_pyr, uint16_t *blur_pyr) {
    if (!rgba || !W || !H || depth < 1 || depth > ORC_MAX_LEVELS || !total) return -1;
    uint32_t arc = opt && opt->arc ? opt->arc : 9;
    if (arc < 9 || arc > 16) return -1;
    orc_pyramid_t lay;
    orc_pyramid_layout(W, H, depth, &lay);
    uint16_t *gray = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *tmp = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    uint16_t *blur = (uint16_t *)malloc(lay.total * sizeof(uint16_t));
    orc_grayscale_intended(rgba, W, H, gray);
    for (uint32_t m = 1; m < depth; m++)
        orc_mip(gray + lay.offset[m - 1], lay.w[m - 1], lay.h[m - 1], gray + lay.offset[m], lay.w[m], lay.h[m]);
    for (uint32_t m = 0; m < depth; m++) {
        orc_gauss_pass(gray + lay.offset[m], lay.w[m], lay.h[m], tmp + lay.offset[m], 0);
        orc_gauss_pass(tmp + lay.offset[m], lay.w[m], lay.h[m], blur + lay.offset[m], 1);
    }
    size_t cap_all = lay.total ? lay.total : 1;
    orc_corner_t *all = (orc_corner_t *)malloc(cap_all * sizeof(orc_corner_t));
    orc_corner_t *kept = (orc_corner_t *)malloc(cap_all * sizeof(orc_corner_t));
    float *scores = (float *)malloc(cap_all * sizeof(float));
    uint32_t n = 0;
    orc_fast_intended(gray, &lay, threshold, arc, all, scores, (uint32_t)cap_all, &n);
    if (opt && opt->nms) {
        /* survivors keep their scores: recompute the index map through a marker pass */
        uint32_t m = orc_nms(&lay, all, scores, n, kept);
        /* orc_nms preserves order, so walk both lists */
        uint32_t j = 0;
        for (uint32_t i = 0; i < n && j < m; i++)
            if (all[i].x == kept[j].x && all[i].y == kept[j].y && all[i].octave == kept[j].octave) scores[j++] = scores[i];
        memcpy(all, kept, (size_t)m * sizeof(orc_corner_t));
        n = m;
    }
    uint32_t stored = orc_topk(all, scores, n, max_features, kept);
    memcpy(corners, kept, (size_t)stored * sizeof(orc_corner_t));
    if (opt && opt->angle_bins && (opt->angle_bins < 8 || opt->angle_bins > 6284)) return -1; /* (checked up front in the wrappers too) */
    if (descriptors) orc_brief_intended_bins(blur, &lay, corners, stored, opt ? opt->angle_bins : 0, descriptors);
    *total = n;
    if (gray_pyr) memcpy(gray_pyr, gray, lay.total * sizeof(uint16_t));
    if (blur_pyr) memcpy(blur_pyr, blur, lay.total * sizeof(uint16_t));
    free(all);
    free(kept);
    free(scores);
    free(gray);
    free(tmp);
    free(blur);
    return 0;
}

int orc_extract_intended_batch(const uint8_t *rgba, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t depth,
                               float threshold, uint32_t max_features, const orc_options_t *opt, orc_corner_t *corners,
                               orc_descriptor_t *descriptors, uint32_t *totals, int n_threads) {
    int rc = 0;
    size_t frame_bytes = (size_t)W * H * 4;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t f = 0; f < (int64_t)n_frames; f++) {
        int r = orc_extract_intended(rgba + (size_t)f * frame_bytes, W, H, depth, threshold, max_features, opt,
                                     corners + (size_t)f * max_features,
                                     descriptors ? descriptors + (size_t)f * max_features : NULL, &totals[f], NULL, NULL);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic frames.  Counter-based so that C, NumPy and a GPU generator
 * produce identical bytes without sharing a sequential random stream.
 * ---------------------------------------------------------------------------------------- */
static uint32_t mix32(uint32_t a) {
    a ^= a >> 16;
    a *= 0x7feb352dU;
    a ^= a >> 15;
    a *= 0x846ca68bU;
    a ^= a >> 16;
    return a;
}

static uint32_t syn_rnd(uint32_t seed, uint32_t stream, uint32_t idx) {
    return mix32(idx ^ mix32(stream + 0x9E3779B9U + mix32(seed + 0x85EBCA6BU)));
}

void orc_synth_frame(uint8_t *rgba, uint32_t W, uint32_t H, uint32_t seed, uint32_t flags) {
    uint32_t ncx_w = (W + 127) / 128, ncx_b = (W + 31) / 32;
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            uint32_t c[3] = {0, 0, 0};
            if (flags & ORC_SYN_GRADIENT) {
                c[0] = W > 1 ? 255u * x / (W - 1) : 0;
                c[1] = H > 1 ? 255u * y / (H - 1) : 0;
                c[2] = (W + H > 2) ? 255u * (x + y) / (W + H - 2) : 0;
            }
            if (flags & ORC_SYN_WEDGES) { /* one acute right triangle per 128x64 cell */
                uint32_t cx = x / 128, cy = y / 64;
                uint32_t hsh = syn_rnd(seed, 2, cy * ncx_w + cx);
                if (hsh & 1u) {
                    uint32_t rise = 4 + ((hsh >> 1) & 15u) % 13u, run = 2 * rise;
                    uint32_t ox = ((hsh >> 8) & 255u) % (128 - run), oy = ((hsh >> 16) & 255u) % (64 - rise);
                    uint32_t level = (hsh >> 24) & 255u;
                    int32_t u = (int32_t)x - (int32_t)(cx * 128 + ox), v = (int32_t)y - (int32_t)(cy * 64 + oy);
                    if (u >= 0 && v >= 0 && u < (int32_t)run && v < (int32_t)rise) {
                        uint32_t uu = (hsh & 32u) ? run - 1 - (uint32_t)u : (uint32_t)u;
                        uint32_t vv = (hsh & 64u) ? rise - 1 - (uint32_t)v : (uint32_t)v;
                        if (vv * run <= uu * rise) c[0] = c[1] = c[2] = level;
                    }
                }
            }
            if (flags & ORC_SYN_BLOBS) { /* at most one bright square (side 1..4) per 32x32 cell */
                uint32_t cx = x / 32, cy = y / 32;
                uint32_t hsh = syn_rnd(seed, 1, cy * ncx_b + cx);
                if (hsh & 1u) {
                    uint32_t s = 1 + ((hsh >> 1) & 3u);
                    uint32_t ox = 1 + ((hsh >> 4) & 255u) % (31 - s), oy = 1 + ((hsh >> 12) & 255u) % (31 - s);
                    uint32_t level = 128 + ((hsh >> 20) & 127u);
                    uint32_t bx = cx * 32 + ox, by = cy * 32 + oy;
                    if (x >= bx && x < bx + s && y >= by && y < by + s) c[0] = c[1] = c[2] = level;
                }
            }
            if (flags & ORC_SYN_NOISE) {
                uint32_t n = syn_rnd(seed, 3, y * W + x);
                for (int k = 0; k < 3; k++) c[k] = (3 * c[k] + ((n >> (8 * k)) & 255u)) / 4;
            }
            uint8_t *p = rgba + ((size_t)y * W + x) * 4;
            p[0] = (uint8_t)c[0];
            p[1] = (uint8_t)c[1];
            p[2] = (uint8_t)c[2];
            p[3] = 255;
        }
}
