// orb_device.h -- device-side scalar building blocks of the ORB path (gfx950).
//
// Everything here is compiled with -ffp-contract=off: each binary32 product and sum is rounded
// on its own, which is what SURVEY.md's canonical decisions CRD-2/-5/-8/-9/-10 fix for the
// points the reference's WGSL leaves implementation-defined.  Division is hipcc's default
// correctly rounded f32 divide.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace orb {

constexpr int kMaxLevels = 10;  // orb.rs:67

// Geometry of the packed binary16 pyramid of one frame (levels back to back).
struct Pyramid {
    uint32_t depth;
    uint32_t w[kMaxLevels];
    uint32_t h[kMaxLevels];
    uint32_t off[kMaxLevels];  // texel offset of each level
    uint32_t stride;           // texels per frame (sum of levels, rounded up to 64)
    uint32_t row_off[kMaxLevels];  // first row of each level in a per-row array (fused path: blur row constants)
    uint32_t row_stride;           // rows per frame over all levels, rounded up to 8
};

typedef _Float16 half_t;

// CRD-3: f32 -> f16 round-to-nearest-even, subnormals kept (v_cvt_f16_f32).
__device__ __forceinline__ half_t to_half(float v) { return (half_t)v; }
__device__ __forceinline__ float from_half(half_t h) { return (float)h; }
// The same conversion of a value that a fused multiply-add produced.  hipcc folds (half_t)fma(a, b, c) into ONE v_fma_mixlo_f16, and
// the hardware rounds that instruction's exact a * b + c once, to binary16 -- not to binary32 and then to binary16 as an fma in a
// shader followed by the store to an R16Float target does (and as LLVM's own pattern assumes): one result in some 10^4 differs by a
// place (found by the blur-plane comparison of tests/test_gpu_round5.py).  The empty asm hides the fma from the conversion.
__device__ __forceinline__ half_t to_half_strict(float v) {
    asm("" : "+v"(v));
    return (half_t)v;
}
__device__ __forceinline__ uint16_t half_bits(half_t h) { return __builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ half_t bits_half(uint16_t b) { return __builtin_bit_cast(half_t, b); }

// fl32(v * s + a) with v a binary16 in the low half of a register, s and a binary32: one v_fma_mix_f32 instead of
// a conversion and the arithmetic.  With s = +-1 and a = -+c the result is +-(v - c), which is exact (CRD-7).
__device__ __forceinline__ float fma_mix_h(uint32_t v_half_bits, float s, float a) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(v_half_bits), "v"(s), "v"(a));
    return d;
}

// Byte address of an LDS object as the DS instructions want it (generic -> local address space).
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// Returning add on an LDS counter with a lane-varying amount, straight on the LDS (`ds_add_rtn_u32`).  Lanes of a wave
// that hit one address are serialised by the LDS (<= 64 of its cycles), which costs the vector unit one instruction;
// hipcc's wave-aggregated form of atomicAdd() with lane-varying amounts is a 20-instruction DPP scan in front of its add.
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t* counter, uint32_t amount) {
    uint32_t old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(lds_offset(counter)), "v"(amount) : "memory");
    return old;
}

// fl32(lo + hi) of the two binary16 halves of a register (one rounding, as the binary32 sum of the converted values).
__device__ __forceinline__ float add_halves(uint32_t packed) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(packed));
    return d;
}

// CRD-1 + CRD-2: grayscale.wgsl:31-38 for one RGBA8 texel (little-endian packed word).
__device__ __forceinline__ float luminance(uint32_t rgba) {
    float r = (float)(rgba & 255u) / 255.0f;
    float g = (float)((rgba >> 8) & 255u) / 255.0f;
    float b = (float)((rgba >> 16) & 255u) / 255.0f;
    float pr = 0.229f * r;  // grayscale.wgsl:36 (0.229, sic)
    float pg = 0.587f * g;
    float pb = 0.114f * b;
    return (pr + pg) + pb;
}

// CRD-13, OrbOptions::fp_contract: the arithmetic WGSL leaves to the adapter's shader compiler, as one word.  A bit per stage
// whose product-and-sum pairs the compiler fuses into fmas -- dot() (grayscale.wgsl:36), `result += sample * weight`
// (gaussian_blur_x.wgsl:58), matrix * vector (brief.wgsl:53-54) -- and one for the order in which dot() and matrix * vector are
// reduced: first component / column first (as written; LLVM-based compilers) or last first (Mesa's NIR lowering).
constexpr uint32_t kFpLum = 1u, kFpBlur = 2u, kFpRot = 4u, kFpLastFirst = 8u, kFpMask = 15u;
// the four forms of the luminance: bit 0 = contracted, bit 1 = last component first
__host__ __device__ constexpr int lum_form(uint32_t fp) { return ((fp & kFpLum) ? 1 : 0) | ((fp & kFpLastFirst) ? 2 : 0); }
// the three forms of the rotation: 0 = every product and sum rounded (the two-term sum has no order), 1 = the second term fused
// onto the first product, 2 = the first term fused onto the second product
__host__ __device__ constexpr int rot_form(uint32_t fp) { return !(fp & kFpRot) ? 0 : ((fp & kFpLastFirst) ? 2 : 1); }

// dot(color, vec4(0.229, 0.587, 0.114, 0.0)) in form `form` (lum_form); r, g, b = byte / 255 (CRD-1).  The alpha term is +0
// in every form.
__device__ __forceinline__ float luminance_dot(float r, float g, float b, int form) {
    const float wr = 0.229f, wg = 0.587f, wb = 0.114f;  // grayscale.wgsl:36 (0.229, sic)
    if (form == 1) return __builtin_fmaf(b, wb, __builtin_fmaf(g, wg, wr * r));
    if (form == 3) return __builtin_fmaf(r, wr, __builtin_fmaf(g, wg, wb * b));
    const float pr = wr * r, pg = wg * g, pb = wb * b;
    if (form == 2) return (pb + pg) + pr;
    return (pr + pg) + pb;
}
__device__ __forceinline__ float luminance_fp(uint32_t rgba, int form) {
    const float r = (float)(rgba & 255u) / 255.0f;
    const float g = (float)((rgba >> 8) & 255u) / 255.0f;
    const float b = (float)((rgba >> 16) & 255u) / 255.0f;
    return luminance_dot(r, g, b, form);
}

// mat2x2f(ct, s2, s1, ct) * (x, y), column-major (brief.wgsl:38-54: s1 = st, s2 = -st): (ct*x + s1*y, s2*x + ct*y) in form
// `form` (rot_form).
__device__ __forceinline__ void rotate_fp(float ct, float s1, float s2, float x, float y, int form, float* rx, float* ry) {
    const float x0 = ct * x, x1 = s1 * y, y0 = s2 * x, y1 = ct * y;
    if (form == 1) {
        *rx = __builtin_fmaf(s1, y, x0), *ry = __builtin_fmaf(ct, y, y0);
    } else if (form == 2) {
        *rx = __builtin_fmaf(ct, x, x1), *ry = __builtin_fmaf(s2, x, y1);
    } else {
        *rx = x0 + x1, *ry = y0 + y1;
    }
}

// "intended" mode IM-1 (not in the reference): BT.601 weight for red.
__device__ __forceinline__ float luminance_601(uint32_t rgba) {
    float r = (float)(rgba & 255u) / 255.0f;
    float g = (float)((rgba >> 8) & 255u) / 255.0f;
    float b = (float)((rgba >> 16) & 255u) / 255.0f;
    float pr = 0.299f * r;
    float pg = 0.587f * g;
    float pb = 0.114f * b;
    return (pr + pg) + pb;
}

// fast.wgsl:51-60 detect_streak_16: non-zero iff the 16-bit circular mask holds a run of >= 12.
__device__ __forceinline__ uint32_t rotate_bits_16(uint32_t v, uint32_t c) {
    return (v >> c) | ((v << (16u - c)) & 0xffffu);
}
__device__ __forceinline__ uint32_t detect_streak_16(uint32_t x) {
    uint32_t o6 = x & rotate_bits_16(x, 6u);
    uint32_t o3 = o6 & rotate_bits_16(o6, 3u);
    return o3 & rotate_bits_16(o3, 2u) & rotate_bits_16(o3, 1u);
}

// CRD-9: canonical atan2 (octant reduction + odd polynomial, only + - * / on binary32).
__device__ __forceinline__ float atan2_canonical(float y, float x) {
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
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

// fast.wgsl:115 + 153: milliradian code, negative angles saturate to 0 (Q7).
__device__ __forceinline__ uint32_t angle_code(float cy, float cx) {
    float r = atan2_canonical(cy, cx);
    if (cy < 0.0f || r < 0.0f) return 0u;
    return (uint32_t)__builtin_truncf(r * 1000.0f);
}

// "intended" mode IM-5: the full circle, 0..6283 milliradians.
__device__ __forceinline__ uint32_t angle_code_signed(float cy, float cx) {
    float r = atan2_canonical(cy, cx);
    if (r < 0.0f) r = r + 6.28318531f;
    const uint32_t code = (uint32_t)__builtin_truncf(r * 1000.0f);
    return code > 6283u ? 6283u : code;
}

// "intended" mode IM-6b (OrbOptions::angle_bins): the code a descriptor is rotated by when angles are quantised into `bins` bins of the
// full circle -- the centre of the code's bin, in integers (0: the code itself).  code <= 6283.
__host__ __device__ __forceinline__ uint32_t binned_angle_code(uint32_t code, uint32_t bins) {
    if (!bins) return code;
    const uint32_t bin = (code * bins) / 6284u;  // < 2^26
    return (bin * 6284u + 3142u) / bins;
}

// FAST ring, fast.wgsl:32-49 (index order matters for the centroid sum, CRD-8).
__device__ constexpr int kRingDx[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
__device__ constexpr int kRingDy[16] = {0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1};

// gaussian_blur_x.wgsl:14-26 (binary32 roundings of the literals)
constexpr float kBlurOffHost = -0.4391873198428642f;  // tap 1, for host-side geometry
__device__ constexpr float kBlurOff[4] = {-2.2273038885157046f, -0.4391873198428642f, 1.3243948342247673f, 3.0f};
__device__ constexpr float kBlurWgt[4] = {0.13748623236806098f, 0.5037756553768409f, 0.32748695702046415f,
                                          0.031251155234634016f};

// "intended" mode IM-3: the same four bilinear taps read in texel units = a symmetric 7-tap kernel (centre first)
__device__ constexpr float kGauss[4] = {0.282523781f, 0.221251875f, 0.106235079f, 0.0312511548f};
// one output of a pass: centre tap, then the three symmetric pairs outwards, every product and sum rounded (IM-3)
__device__ __forceinline__ float gauss7(const float (&t)[7]) {
    float acc = kGauss[0] * t[3];
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        const float pair = t[3 - k] + t[3 + k];
        const float term = kGauss[k] * pair;
        acc = acc + term;
    }
    return acc;
}

// fl32(a + b) and fl32(a * k) with a, b binary16 halves of registers (SA, SB: 0 = low half, 1 = high half) and k binary32:
// v_fma_mix_f32 converts its binary16 sources on the way in (exactly), so the sum is the binary32 sum of the converted
// values and the product their binary32 product, each rounded once -- without the two v_cvt_f32_f16 in front.
template <int SA, int SB>
__device__ __forceinline__ float add_hh(uint32_t a, uint32_t b, float one) {
    float d;
    if constexpr (SA == 0 && SB == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(one), "v"(b));
    if constexpr (SA == 1 && SB == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(one), "v"(b));
    if constexpr (SA == 0 && SB == 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(one), "v"(b));
    if constexpr (SA == 1 && SB == 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(one), "v"(b));
    return d;
}
template <int SA>
__device__ __forceinline__ float mul_h(uint32_t a, float k, float zero) {
    float d;
    if constexpr (SA == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(k), "v"(zero));
    if constexpr (SA == 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(k), "v"(zero));
    return d;
}
// gauss7() on seven binary16 taps that sit in register halves (w_i, half S_i): the same roundings, no conversions.
// (The product with +0 added is the rounded product: grey values are never negative.)
template <int S0, int S1, int S2, int S3, int S4, int S5, int S6>
__device__ __forceinline__ float gauss7_h(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, uint32_t w6,
                                          float one, float zero) {
    float acc = mul_h<S3>(w3, kGauss[0], zero);
    const float p1 = add_hh<S2, S4>(w2, w4, one), p2 = add_hh<S1, S5>(w1, w5, one), p3 = add_hh<S0, S6>(w0, w6, one);
    const float t1 = kGauss[1] * p1;
    acc = acc + t1;
    const float t2 = kGauss[2] * p2;
    acc = acc + t2;
    const float t3 = kGauss[3] * p3;
    acc = acc + t3;
    return acc;
}


// One literal blur tap position (CRD-5): indices of the two texels and the lerp fraction.
struct BlurTap {
    int i0, i1;
    float f;
};
// wq: 0 = the exact binary32 fraction (CRD-5); 2^n = a sampler that holds its weights in n fractional bits -- the
// fraction rounded to the nearest multiple of 2^-n, halves up (OrbOptions::sampler_weight_bits; scaling by a power of
// two is exact).
__host__ __device__ __forceinline__ float sampler_weight(float f, float wq) {
    return wq != 0.0f ? __builtin_floorf(f * wq + 0.5f) / wq : f;
}
__host__ __device__ __forceinline__ BlurTap blur_tap(uint32_t x, uint32_t w, float off, float wq = 0.0f) {
    float fw = (float)w;
    float u = ((float)x + 0.5f) / fw;
    float uo = u + off;
    float coord = uo * fw - 0.5f;
    float c0 = __builtin_floorf(coord);
    BlurTap t;
    t.f = sampler_weight(coord - c0, wq);
    int i = (int)c0;
    int hi = (int)w - 1;
    t.i0 = i < 0 ? 0 : (i > hi ? hi : i);
    int j = i + 1;
    t.i1 = j < 0 ? 0 : (j > hi ? hi : j);
    return t;
}

// What a textureLoad outside the addressed level returns is implementation-defined (fast.wgsl:78,86,103 at
// octaves >= 1, brief.wgsl:59-60): OrbOptions::oob_policy.  kOobZero: 0 (Vulkan robust image access; the default);
// kOobClamp: every coordinate clamped into the level; kOobUmin: naga's `Restrict` as its SPIR-V writer emits it,
// min(unsigned(coordinate), size - 1) -- a negative coordinate lands on the LAST column / row.
constexpr uint32_t kOobZero = 0u, kOobClamp = 1u, kOobUmin = 2u;
__host__ __device__ __forceinline__ int oob_index(int i, int n, uint32_t policy) {  // policy != kOobZero
    if (policy == kOobClamp) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    return (i < 0 || i >= n) ? n - 1 : i;
}

// Synthetic-frame hash; byte-identical to the recipe the tests hold (checked on the GPU).
__device__ __forceinline__ uint32_t mix32(uint32_t a) {
    a ^= a >> 16;
    a *= 0x7feb352dU;
    a ^= a >> 15;
    a *= 0x846ca68bU;
    a ^= a >> 16;
    return a;
}
__device__ __forceinline__ uint32_t syn_rnd(uint32_t seed, uint32_t stream, uint32_t idx) {
    return mix32(idx ^ mix32(stream + 0x9E3779B9U + mix32(seed + 0x85EBCA6BU)));
}

}  // namespace orb
