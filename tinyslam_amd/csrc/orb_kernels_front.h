// orb_kernels_front.h -- the MI355X-first front half: one kernel per pyramid level (k_front, k_front_pair).
// Kept apart from the BRIEF side of the fused pipeline (orb_kernels_fused.h) so that the translation units that instantiate
// k_front's variants (orb_front_inst.hip, one per arithmetic form) see nothing but this header.
//
// k_front<L0> replaces, for one level, the reference's grayscale pass (orb.rs:478-496), the blit
// that produces the next mip (orb.rs:413-429), both blur passes (orb.rs:432-466) and the FAST
// dispatch (orb.rs:504-520).  A workgroup owns a band of R full-width rows of one frame:
//
//   A  stage grey rows [y0-3, y0+R+3) in LDS as f16  (level 0: 16-byte RGBA loads -> luminance;
//      level >= 1: the f16 mip written by the previous level's kernel).  The grey image of
//      level 0 never goes to HBM.
//   B1 FAST 4-point pre-test (fast.wgsl:85-95), 8 pixels per thread from wide LDS reads;
//      survivors are pushed on an LDS queue.
//   B2 the queue is drained densely: 16-point masks, 12-streak test, ring centroid, angle
//      (fast.wgsl:98-121); corners are appended with one wave64 ballot + one global atomic per
//      wave (the reference uses an LDS atomic per thread and two barriers, fast.wgsl:123-157).
//   C0 next mip level (2x2 mean, CRD-4) from the LDS rows.
//   C  both literal blur passes.  The reference's blur is row-local (offsets in UV units, clamp to
//      edge, both passes in X, two vertical flips cancelling: SURVEY.md Q11-Q13), so a band needs
//      no halo for it; the f16-rounded intermediate (blur_tmp, orb.rs:291-304) lives in LDS only.
//
// HBM traffic per level-0 pixel: 4 B RGBA read (+ halo re-reads that hit L2), 2 B blur written,
// 0.5 B mip written -- against 12 B for the staged pipeline.
#pragma once
#include "../../include/tinyorb.h"
#include "orb_device.h"

namespace orb {

constexpr int kFrontThreadsL0 = 1024;  // level 0: 16 waves per band, two bands per CU -> 8 waves/SIMD
constexpr int kFrontThreadsLN = 512;   // the levels above, by default; a level whose bands hold about 20 k pixels runs on kFrontThreadsLNBig threads
constexpr int kFrontThreadsLNBig = 1024;  // with sixteen-pixel pre-test items -- level 0's shape (round 4: k_front<false> 0.0735 -> 0.0697 ms at
                                          // 720p, 0.130 -> 0.102 for the 1280-wide level 1 of 2560x1440); narrow levels keep 512 threads (320x240: 0.051
                                          // against 0.064 on 1024): the program decides per level when it is created (OrbProgram::ln_threads)
constexpr int kFrontRows = 16;       // R: band height at 1280 columns (the bench shape); other widths: kFrontBandHeights, chosen per level at create
constexpr int kFrontRowsWide = 8;    // the flattest band: 14 full-width rows of up to 4096 texels fit in LDS
constexpr int kFrontTmpRows = 2;     // rows per blur chunk (double buffered)
constexpr int kFrontQueue = 4096;    // pre-test survivor queue, 16-bit entries
// Band heights k_front is instantiated for, and the x bits their 16-bit queue entries leave (15 - log2(rows)): a level of
// dispatch width <= 2^bits can run on that height.  Narrow levels take tall bands (less halo per row, and enough pixels
// per workgroup), wide ones flat bands (two workgroups per CU).
constexpr int kFrontBandHeights[] = {64, 32, 16, 8};
__host__ __device__ constexpr int front_x_bits(int rows) { return rows == 64 ? 9 : rows == 32 ? 10 : rows == 16 ? 11 : 12; }
// Survivor mask of half a pre-test item (8 pixels, two polarities): bit p = 8 * (k & 1) + 4 * under + (k >> 1) for pixel k;
// an item is two halves (bits 0..15 and 16..31).  A 16-bit queue entry is [15:5] the item (band row, x / 16) and [4:0] that
// bit number: the push loop of B1 is then ffbl / clear / or / store, and the dense stages decode (locate()).
__host__ __device__ constexpr uint32_t front_mask_bit(int k, bool under) { return 1u << (8 * (k & 1) + (under ? 4 : 0) + (k >> 1)); }
constexpr int kFrontMaxWidth = 2048;     // widest level 0 of the 16-row bands (11-bit x in the 16-bit queue entries)
constexpr int kFrontMaxWidthWide = 4096; // ... of the 8-row bands (12-bit x)
constexpr int kFrontMaxWidthTiled = 16384;  // widest level 0 of the fused literal pipeline: levels too wide for two full-width bands
                                            // per CU are cut into column tiles (k_front<..., TILED>)
constexpr int kFrontTileW = 1280;           // preferred tile width there: the shape the kernel is tuned on (16 rows x 1280 columns)
constexpr int kLdsPad = 8;           // halfs of padding left of column 0

struct FrontGeom {
    uint32_t lvl;       // pyramid level handled by this launch
    uint32_t rows;      // band height (one of kFrontBandHeights)
    // Column tiles (k_front<..., TILED = true>; tiled == 0: a workgroup owns full-width rows and the next six fields are unused).
    // A level too wide for two full-width bands per CU is cut into n_ct tiles of tw columns; a workgroup then owns rows x tw
    // texels, stages 4 columns of halo on either side, and tile 0 also does the band's whole blur (phase C), for which it
    // fetches the three grey texels per row that lie beyond its own columns.
    uint32_t tiled;
    uint32_t tw;        // tile width (multiple of 8); the level's last tile may be narrower
    uint32_t n_ct;      // ceil(max(w, gw) / tw)
    uint32_t xb;        // bits of a tile-local x in a queue entry (tw <= 2^xb, rows <= 2^(15 - xb))
    uint32_t far_i0, far_i1;  // the two grey columns pass 1 lerps at the level's last column (blur_tap(w - 1)); the third far one is w - 1
    uint32_t tmp_halfs; // halfs of the storage shared by the blur column table (phase C) and queues B and C (phase B)
    uint32_t gw, gh;    // FAST dispatch domain of this octave (8-rounded, orb.rs:511-515)
    uint32_t n_bands;   // ceil(max(h, gh) / R)
    uint32_t n_frames;
    uint32_t ls;        // LDS row stride of the grey rows, in halfs (multiple of 8)
    uint32_t ts;        // LDS row stride of the blur intermediate, in halfs
    uint32_t write_mip; // 1: level lvl+1 exists and is an exact 2x2 reduction
    uint32_t store_grey; // level 0 only: 1 = level 1 is NOT an exact half, the band also stores its grey rows for k_mip
    uint32_t xcd_swizzle;
    uint32_t phase_mask;  // debug: bit0 B1, bit1 B2, bit2 C0, bit3 C (timing experiments only)
    uint32_t slot_base;   // index of this level's band 0 among the frame's band slots
    uint32_t n_slots;     // band slots per frame (all levels)
    uint32_t seg_cap;     // CornerData records per band segment (and per class, see n_classes)
    uint32_t n_classes;   // 1: a band's corners form one list.  2: two lists per band, angle code 0 and the rest (their
                          // descriptors need no rotation / a rotation: k_brief_t wants its waves to be of one kind);
                          // the band's memory is then 2 * seg_cap records and it has two counters
    uint32_t blur_p;      // columns [0, blur_p) of blur pass 1 are one constant per row (tap 1 clamps to column 0)
    uint32_t blur_q;      // columns [0, blur_q) of the final blur are one constant per row
    uint32_t n_var;       // w - blur_q: columns whose blur varies along the row (one BlurCol table entry each)
    unsigned long long* stamps;  // diagnostic runs only: 16 cycle sums per kernel flavour (else null)
    uint32_t ovf_words;   // words of one bit set over the band's pre-test items (two sets in LDS: items whose survivors did not fit queue A)
    uint32_t oob;         // OrbOptions::oob_policy (kOobZero / kOobClamp / kOobUmin); phase A of the levels >= 1 follows it through OOBK
    float wq;             // OrbOptions::sampler_weight_bits as 2^bits (0: exact lerp weights), for blur_tap()
    uint32_t fp;          // OrbOptions::fp_contract (kFp*), for the record: the luminance's form and the blur's are the template parameter FPF
};

// Tap positions of one column x >= blur_q of the literal blur (phase C): pass 2 at x lerps pass 1 at columns j0, j1
// with fraction f2; pass 1 at j0 (j1) lerps the grey texels a0, a1 (b0, b1) with fraction fa (fb).  a0 (b0) == 0xffff:
// that pass-1 column lies in the stretch that is one constant per row.  A function of x and the level's width only,
// so a band evaluates blur_tap() once per column instead of three times per pixel.
struct __attribute__((aligned(8))) BlurCol {
    uint16_t a0, a1, b0, b1;
    float fa, fb, f2;
    uint32_t pad;
};
static_assert(sizeof(BlurCol) == 24, "BlurCol layout");

__host__ __device__ inline uint32_t front_lds_bytes(const FrontGeom& g) {
    // grey rows + queues B/C (the blur column table lives there first: 24 B x n_var <= 8 B x ts, checked on the host)
    // + queue A + 5 counters + blur row constants (2 x rows float4)
    // + two bit sets over the pre-test items (8 bytes per 32 items)
    if (g.tiled)  // the same, the shared storage sized for whichever of its two uses is larger, + the far grey columns of tile 0
        return ((g.rows + 6) * g.ls + g.tmp_halfs) * 2u + kFrontQueue * 2u + 32u + 32u * g.rows + 8u * g.rows + 8u * g.ovf_words;
    return ((g.rows + 6) * g.ls + 2 * kFrontTmpRows * g.ts) * 2u + kFrontQueue * 2u + 32u + 32u * g.rows + 8u * g.ovf_words;
}

// Typed buffer loads (the texture path converts): with DATA_FORMAT 8_8_8_8 / NUM_FORMAT UNORM a lane receives byte/255 of
// four consecutive bytes as binary32 -- bit-identical to fl32(byte / 255.0f) (CRD-1) for all 256 bytes and all 2^24
// colours (tools/ubench/fmt_rate.hip checks it on the device).  hipcc has no builtin for them; the LLVM intrinsics are
// reached by name, so the compiler still schedules the loads and tracks their completion (an asm load would not be).
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float3_t __attribute__((ext_vector_type(3)));
__device__ float4_t buffer_load_format_xyzw(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
__device__ float3_t buffer_load_format_xyz(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v3f32");
constexpr int kBufferWord3Raw = 0x00020000;        // raw dword buffer
constexpr int kBufferWord3Unorm8x4 = 0x00050FAC;   // DST_SEL xyzw, NUM_FORMAT UNORM, DATA_FORMAT 8_8_8_8

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ushort2_t as_u16x2(uint32_t v) { return __builtin_bit_cast(ushort2_t, v); }

__device__ __forceinline__ float h2f(uint32_t packed, int hi) {
    return from_half(bits_half((uint16_t)(hi ? (packed >> 16) : (packed & 0xffffu))));
}

// Exact byte/255 (CRD-1) without the divide sequence: 1/255 as a double-float (hi + lo), the product with the
// byte accumulated in one true FMA.  b*hi is exact inside the FMA and b*lo carries a relative error of 2^-24 on
// a term 2^-25 times smaller, so the result is the correctly rounded quotient unless b/255 lies within ~2^-48 of
// a rounding boundary, which no multiple of 1/255 does (checked for all 256 bytes on host and device).
__device__ __forceinline__ float unorm8_exact(float b) {
    const float rc_hi = 0x1.010102p-8f, rc_lo = -0x1.fdfdfep-33f;  // rc_hi + rc_lo = 1/255 to 2^-57
    const float t = b * rc_lo;
    return __builtin_fmaf(b, rc_hi, t);
}
__device__ __forceinline__ float luminance_fast(uint32_t rgba) {
    float r = unorm8_exact((float)(rgba & 255u));
    float g = unorm8_exact((float)((rgba >> 8) & 255u));
    float b = unorm8_exact((float)((rgba >> 16) & 255u));
    float pr = 0.229f * r;
    float pg = 0.587f * g;
    float pb = 0.114f * b;
    return (pr + pg) + pb;
}
__device__ __forceinline__ uint32_t pack_half2(float lo, float hi) {
    return (uint32_t)half_bits(to_half(lo)) | ((uint32_t)half_bits(to_half(hi)) << 16);
}
// Luminance (CRD-1, CRD-2) of two neighbouring texels as f16 in one word.  Written on two-element vectors so that the
// packed binary32 instructions (v_pk_mul/fma/add_f32) work on the pair that v_cvt_pk_f16_f32 then rounds into one
// register: left to itself the vectoriser pairs texels 0/2 and 1/3 and spends four more instructions re-interleaving.
// LUM: the form dot() takes under the adapter's shader compiler (lum_form(), CRD-13): 0 = (pr + pg) + pb, every product and sum
// rounded (CRD-2, the default); 1 = r*wr, fma(g, wg, .), fma(b, wb, .) -- two packed operations per pair fewer; 2 = (pb + pg) + pr;
// 3 = b*wb, fma(g, wg, .), fma(r, wr, .).
template <bool BT601 = false, int LUM = 0>  // BT601 false: the reference's 0.229 red weight (Q1); true: 0.299 (the intended mode's IM-1)
__device__ __forceinline__ uint32_t luminance_pair_f16(uint32_t rgba0, uint32_t rgba1) {
    const float2_t rc_hi = {0x1.010102p-8f, 0x1.010102p-8f}, rc_lo = {-0x1.fdfdfep-33f, -0x1.fdfdfep-33f};
    const float2_t R = {(float)(rgba0 & 255u), (float)(rgba1 & 255u)};
    const float2_t G = {(float)((rgba0 >> 8) & 255u), (float)((rgba1 >> 8) & 255u)};
    const float2_t B = {(float)((rgba0 >> 16) & 255u), (float)((rgba1 >> 16) & 255u)};
    const float2_t tr = R * rc_lo, tb = B * rc_lo;
    const float2_t r = __builtin_elementwise_fma(R, rc_hi, tr);  // exact byte/255, see unorm8_exact
    const float2_t b = __builtin_elementwise_fma(B, rc_hi, tb);
    constexpr float wr1 = BT601 ? 0.299f : 0.229f;
    const float2_t wr = {wr1, wr1}, wg = {0.587f, 0.587f}, wb = {0.114f, 0.114f};
    float2_t l;
    if constexpr (LUM == 1 || LUM == 3) {
        const float2_t tg = G * rc_lo;
        const float2_t g = __builtin_elementwise_fma(G, rc_hi, tg);
        if constexpr (LUM == 1) {
            const float2_t t = r * wr;
            const float2_t u = __builtin_elementwise_fma(g, wg, t);
            l = __builtin_elementwise_fma(b, wb, u);
        } else {
            const float2_t t = b * wb;
            const float2_t u = __builtin_elementwise_fma(g, wg, t);
            l = __builtin_elementwise_fma(r, wr, u);
        }
    } else {
        // The green product fl(fl(G / 255) * 0.587f) -- two roundings -- in TWO operations instead of three: fma(G, kGHi, fl(G * kGLo)) is
        // that value for every byte G (the pair was found by search and is checked for all 256 bytes, exactly, by
        // tests/test_oracle.py::test_green_product_in_two_operations; no such pair exists for 0.229f, 0.299f or 0.114f).
        constexpr float kGHi = 0x1.2db8fcp-9f, kGLo = 0x1.a6bf2ep-33f;
        const float2_t g_hi = {kGHi, kGHi}, g_lo = {kGLo, kGLo};
        const float2_t tg = G * g_lo;
        const float2_t pg = __builtin_elementwise_fma(G, g_hi, tg);
        const float2_t pr = r * wr1, pb = b * 0.114f;
        if constexpr (LUM == 2) {
            const float2_t s = pb + pg;
            l = s + pr;
        } else {
            const float2_t s = pr + pg;
            l = s + pb;
        }
    }
    uint32_t d;  // the instruction hipcc itself uses for two (half) casts (RNE, CRD-3); as asm so that the pairing stays
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(l.x), "v"(l.y));
    return d;
}

// ---- FAST on one pixel; `ctr` points at it inside the LDS grey rows (row stride `ls` halfs) ----
// 16-point masks (fast.wgsl:102-113).  thr >= 0, so `diff > thr` and `diff < -thr` exclude each other
// and the reference's else-if needs no special handling.
__device__ __forceinline__ bool ring_is_corner(const half_t* ctr, int ls, float thr) {
    const float c = from_half(ctr[0]);
    uint32_t m_over = 0, m_under = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float diff = from_half(ctr[kRingDy[i] * ls + kRingDx[i]]) - c;  // CRD-7
        m_over |= (diff > thr) ? (1u << i) : 0u;
        m_under |= (diff < -thr) ? (1u << i) : 0u;
    }
    return (detect_streak_16(m_over) | detect_streak_16(m_under)) != 0u;  // fast.wgsl:117-121
}
// The same for a pixel whose compass pre-test passed with the given polarity: a 12-run holds three of the four
// compass points, so a run of the other polarity is impossible and one mask is enough.
// Exact, on packed 16-bit integers: v - c is exact in binary32 for grey values (f16, <= 1), so `fl32(v - c) > thr` is the
// real comparison v > c + thr, i.e. v >= T with T the smallest f16 above c + thr -- and grey values are non-negative f16,
// whose bit patterns order like the values.  T is found once per pixel (round c + thr to f16, test that candidate with
// the reference's own expression, step one pattern if it fails); "darker" is the mirror image (v <= T', T' the largest
// f16 below c - thr, none if that is not positive), folded into the same subtraction by complementing both sides.
// Ring points i and i + 8 share a register: 8 packed subtractions whose sign bits are the mask, gathered in ring order
// by a packed shift and a shift-or per register and one byte permute.
// The smallest (over) / largest (under) f16 bit pattern t of a ring value v that passes the reference's strict test against centre
// c -- fl32(v - c) > thr, resp. < -thr -- as the two constants of the packed comparison  ((t ^ m) - 1) - (v ^ m) < 0  (see
// ring_is_corner_polar): tt = (t ^ m) - 1 and mm = m, each replicated in both halves of a word.  -1 for "nothing passes".
__device__ __forceinline__ void polar_threshold(float c, float thr, bool over, uint32_t* tt, uint32_t* mm) {
    const float sgn = over ? 1.0f : -1.0f;
    const float s = __builtin_fmaf(sgn, thr, c);           // c + thr / c - thr (rounded: only a first guess)
    const uint32_t h0 = half_bits(to_half(s));
    const float d = from_half(bits_half((uint16_t)h0)) - c;  // exact
    const bool pass = d * sgn > thr;                       // the candidate itself, by the reference's expression (CRD-7; +-d is exact)
    const int isgn = over ? 1 : -1;
    int t = (int)h0 + (pass ? 0 : isgn);                   // over: smallest v that passes; under: largest v that passes
    if (!over && !(s > 0.0f)) t = -1;                      // nothing is darker than a non-positive bound
    const uint32_t m = over ? 0u : 0xffffu;
    const uint32_t tm1 = (uint32_t)(((t ^ (int)m) - 1) & 0xffff);
    *tt = tm1 | (tm1 << 16);
    *mm = m | (m << 16);
}
// 1 in each half of the result where that half of v passes (polar_threshold's constants)
__device__ __forceinline__ uint32_t polar_pass2(uint32_t v, uint32_t tt, uint32_t mm) {
    typedef short short2_t __attribute__((ext_vector_type(2)));
    const short2_t df = __builtin_bit_cast(short2_t, tt) - __builtin_bit_cast(short2_t, v ^ mm);
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2_t, df) >> (unsigned short)15);
}
// The 16-bit ring mask of one polarity (bit i <=> ring point i passes), exact, on packed 16-bit integers.
__device__ __forceinline__ uint32_t ring_mask_polar(const half_t* ctr, int ls, float thr, bool over) {
    uint32_t tt, mm;
    polar_threshold(from_half(ctr[0]), thr, over, &tt, &mm);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const ushort2_t v = {half_bits(ctr[kRingDy[j] * ls + kRingDx[j]]), half_bits(ctr[kRingDy[j + 8] * ls + kRingDx[j + 8]])};
        const uint32_t sb = polar_pass2(__builtin_bit_cast(uint32_t, v), tt, mm);
        acc = j == 0 ? sb : ((sb << j) | acc);
    }
    return __builtin_amdgcn_perm(0u, acc, 0x0c0c0200u);  // bits 0..7: ring 0..7, bits 8..15: ring 8..15
}
// Every second ring point (ring indices 0, 2, .., 14) of one polarity: bit i <=> ring point 2 i passes.
__device__ __forceinline__ uint32_t even_ring_mask_polar(const half_t* ctr, int ls, float thr, bool over) {
    uint32_t tt, mm;
    polar_threshold(from_half(ctr[0]), thr, over, &tt, &mm);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {  // even points j and j + 4 = ring points 2 j and 2 j + 8 share a register
        const ushort2_t v = {half_bits(ctr[kRingDy[2 * j] * ls + kRingDx[2 * j]]), half_bits(ctr[kRingDy[2 * j + 8] * ls + kRingDx[2 * j + 8]])};
        const uint32_t sb = polar_pass2(__builtin_bit_cast(uint32_t, v), tt, mm);
        acc = j == 0 ? sb : ((sb << j) | acc);
    }
    return (acc & 15u) | ((acc >> 12) & 0xf0u);
}

__device__ __forceinline__ bool ring_is_corner_polar(const half_t* ctr, int ls, float thr, bool over) {
    const float c = from_half(ctr[0]);
    const float sgn = over ? 1.0f : -1.0f;
    const float s = __builtin_fmaf(sgn, thr, c);           // c + thr / c - thr (rounded: only a first guess)
    const uint32_t h0 = half_bits(to_half(s));
    const float d = from_half(bits_half((uint16_t)h0)) - c;  // exact
    const bool pass = d * sgn > thr;                       // the candidate itself, by the reference's expression (CRD-7; +-d is exact)
    const int isgn = over ? 1 : -1;
    int t = (int)h0 + (pass ? 0 : isgn);                   // over: smallest v that passes; under: largest v that passes
    if (!over && !(s > 0.0f)) t = -1;                      // nothing is darker than a non-positive bound
    // pass  <=>  over: v >= t  |  under: v <= t  <=>  (v ^ m) >= (t ^ m) as signed 16-bit, m = under ? 0xffff : 0
    //       <=>  ((t ^ m) - 1) - (v ^ m) < 0
    const uint32_t m = over ? 0u : 0xffffu;
    const uint32_t tm1 = (uint32_t)(((t ^ (int)m) - 1) & 0xffff), tt = tm1 | (tm1 << 16), mm = m | (m << 16);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const ushort2_t v = {half_bits(ctr[kRingDy[j] * ls + kRingDx[j]]), half_bits(ctr[kRingDy[j + 8] * ls + kRingDx[j + 8]])};
        typedef short short2_t __attribute__((ext_vector_type(2)));
        const short2_t df = __builtin_bit_cast(short2_t, tt) - __builtin_bit_cast(short2_t, __builtin_bit_cast(uint32_t, v) ^ mm);
        const uint32_t sb = __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2_t, df) >> (unsigned short)15);  // 1 = passes
        acc = j == 0 ? sb : ((sb << j) | acc);
    }
    const uint32_t mask = __builtin_amdgcn_perm(0u, acc, 0x0c0c0200u);  // bits 0..7: ring 0..7, bits 8..15: ring 8..15
    return detect_streak_16(mask) != 0u;
}
// ring centroid -> milliradian code (fast.wgsl:106,115,153; CRD-8: ring order, unfused)
__device__ __forceinline__ uint32_t ring_angle(const half_t* ctr, int ls) {
    float cx = 0.0f, cy = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = from_half(ctr[kRingDy[i] * ls + kRingDx[i]]);
        const float px = v * (float)kRingDx[i];
        const float py = v * (float)kRingDy[i];
        cx = cx + px;
        cy = cy + py;
    }
    return angle_code(cy, cx);
}
__device__ __forceinline__ bool fast_full_test(const half_t* ctr, int ls, float thr, uint32_t* angle) {
    if (!ring_is_corner(ctr, ls, thr)) return false;
    *angle = ring_angle(ctr, ls);
    return true;
}
// A 12-run on the 16-ring contains at least 3 of the 4 diagonal ring points (+-2,+-2) (ring indices
// 2, 6, 10, 14 are four apart), with the run's polarity.  Cheap necessary condition used to thin the
// pre-test survivors (9.4 % of the pixels of a noisy frame) before the 16-point test (-> 2.7 %).
__device__ __forceinline__ bool diagonal_filter(const half_t* ctr, int ls, float thr, bool over) {
    // grey values are non-negative f16: their bit patterns order like the values, so the selection runs on 16-bit
    // integers (v_min_u16 / v_max_u16 issue at twice the rate of v_min_f32) and only the selected value is converted
    const uint16_t a = half_bits(ctr[-2 * ls - 2]), b = half_bits(ctr[-2 * ls + 2]);
    const uint16_t d = half_bits(ctr[2 * ls - 2]), e = half_bits(ctr[2 * ls + 2]);
    const uint16_t lo1 = min(a, b), hi1 = max(a, b), lo2 = min(d, e), hi2 = max(d, e);
    const uint16_t m1 = max(lo1, lo2), m2 = min(hi1, hi2);
    // 2nd smallest / 2nd largest of the four; v -> fl(v - c) is monotone, so ">= 3 diffs beyond thr"
    // is decided by that one value
    const uint16_t sel = over ? min(m1, m2) : max(m1, m2);
    const float diff = from_half(bits_half(sel)) - from_half(ctr[0]);
    return over ? diff > thr : diff < -thr;
}

// Block-local stream compaction: a band's corners go to its own segment of the scratch list, the
// slot comes from an LDS counter.  No global atomic is involved: 180 waves per frame bumping one
// per-frame counter serialise at the memory side and cost more than the rest of the kernel.
// lds_counter[0] counts the first list, lds_counter[1] the second (two_lists: angle code != 0 goes to the second, which
// starts seg_cap records into the band's memory).
__device__ __forceinline__ void segment_append(bool is_corner, uint32_t x, uint32_t y, uint32_t angle, uint32_t oct,
                                               uint32_t* lds_counter, CornerData* seg, uint32_t seg_cap, bool two_lists) {
    if (is_corner) {
        const uint32_t cls = (two_lists && angle != 0u) ? 1u : 0u;
        const uint32_t idx = atomicAdd(lds_counter + cls, 1u);  // hipcc turns this into one ds_add per wave and list
        if (idx < seg_cap) *reinterpret_cast<uint4*>(&seg[cls * seg_cap + idx]) = make_uint4(x, y, angle, oct);
    }
}

// Y8: level 0 reads a one-byte-per-pixel Y plane instead of RGBA (ORB_FLAG_INPUT_Y8), grey = f16(byte/255).
// RB: band height, one of kFrontBandHeights (the host picks it per level: orb_api.hip, program create).
// UA: the general level-0 variant (RGBA or Y8) -- a width that is not a multiple of 4 (rows only 4-byte aligned: texel by texel loads, a
//     partial last quad) and/or a level 1 that is not an exact half (FrontGeom::store_grey: the band also stores its grey rows).
// TILED: column tiles (FrontGeom::tiled).  With TILED = false every tile expression below folds to the full-width form.
// SRC (level 1 only): the band builds its grey rows from the frame itself (RGBA, or Y8 bytes through the typed buffer) -- luminance of the 2x2 blocks, then the mip
//     (CRD-1..4, the arithmetic of phases A and C0) -- instead of reading the mip level 0's launch writes: the two launches
//     of a single frame then do not depend on each other (orb_extract_corners runs them side by side).
// OOBK (levels >= 1 only): the program has an out-of-level policy other than "zero" (OrbOptions::oob_policy): texels outside the
//     level are staged as the level's last row / column instead of 0.  A template flag so that the default's code stays what it was
//     (as a run-time branch it cost k_front<false> a register and 6 % of its time).
// NTO: threads of the workgroup when they are not the level's usual number (k_front_pair runs level 1 on 1024).
// The body lives in orb_front_body.inc and is emitted twice: front_body<...> (a device function, for k_front_pair) and
// the kernel k_front<...> itself.
// FPF: the arithmetic of the adapter's shader compiler (OrbOptions::fp_contract, CRD-13) as far as this kernel computes it -- bits 0-1 the
//     luminance's form (lum_form(): level 0 from RGBA, and SRC; luminance_pair_f16), bit 2 the blur taps as fused multiply-adds
//     (`result += sample * weight`, gaussian_blur_x.wgsl:58).  A template parameter: FPF = 0 is, instruction for instruction, the kernel
//     that existed before the switch did (carried as run-time constants the blur's form cost the default 1.5 % of the step, measured
//     library against library).
template <bool L0, bool Y8 = false, int RB = kFrontRows, bool UA = false, bool TILED = false, bool SRC = false, int NTO = 0, bool OOBK = false, int FPF = 0>
__device__ __forceinline__ void front_body(const uint32_t block_id, const uint8_t* __restrict__ frames, size_t frame_bytes,
                                           uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                           uint16_t* __restrict__ blur_rowc, const Pyramid pyr,
                                           const FrontGeom geo, float thr, uint32_t* __restrict__ seg_counts,
                                           CornerData* __restrict__ segments) {
#include "orb_front_body.inc"
}

// NTK: threads of the workgroup when not the level's default (levels >= 1: kFrontThreadsLNBig)
template <bool L0, bool Y8 = false, int RB = kFrontRows, bool UA = false, bool TILED = false, bool SRC = false, bool OOBK = false, int NTK = 0, int FPF = 0>
__global__ __launch_bounds__(NTK ? NTK : (L0 ? kFrontThreadsL0 : kFrontThreadsLN), (L0 || NTK == kFrontThreadsLNBig) ? 8 : 4) void k_front(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                         uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                         uint16_t* __restrict__ blur_rowc, Pyramid pyr,
                                                         FrontGeom geo, float thr, uint32_t* __restrict__ seg_counts,
                                                         CornerData* __restrict__ segments) {
    constexpr int NTO = NTK;
    const uint32_t block_id = blockIdx.x;
#include "orb_front_body.inc"
}

// Levels 0 and 1 of ONE frame in one launch (the reference's call shape, orb.rs:469-557: one blocking call per frame, where a
// dependent launch costs more than the work it starts): blocks [0, n0) are level 0's bands, the rest level 1's, which build
// their grey rows from the frame itself (SRC) and so wait for nothing.  1024 threads for both.
template <int RB0, int RB1, bool Y8 = false, int FPF = 0>
__global__ __launch_bounds__(kFrontThreadsL0) void k_front_pair(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                              uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                              uint16_t* __restrict__ blur_rowc, Pyramid pyr, FrontGeom geo0,
                                                              FrontGeom geo1, float thr, uint32_t* __restrict__ seg_counts,
                                                              CornerData* __restrict__ segments) {
    if (blockIdx.x < geo0.n_bands)
        front_body<true, Y8, RB0, false, false, false, 0, false, FPF>(blockIdx.x, frames, frame_bytes, gray, blur, blur_rowc, pyr, geo0, thr, seg_counts, segments);
    else
        front_body<false, Y8, RB1, false, false, true, kFrontThreadsL0, false, FPF>(blockIdx.x - geo0.n_bands, frames, frame_bytes, gray, blur, blur_rowc,
                                                                                   pyr, geo1, thr, seg_counts, segments);
}

}  // namespace orb
