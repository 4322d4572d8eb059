// orb_kernels_fused.h -- the MI355X-first front half: one kernel per pyramid level.
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
#include "orb_kernels_staged.h"

namespace orb {

constexpr int kFrontThreadsL0 = 1024;  // level 0: 16 waves per band, two bands per CU -> 8 waves/SIMD
constexpr int kFrontThreadsLN = 512;
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
    if (g.tiled)  // the same, the shared storage sized for whichever of its two uses is larger, + the far grey columns of tile 0
        return ((g.rows + 6) * g.ls + g.tmp_halfs) * 2u + kFrontQueue * 2u + 32u + 32u * g.rows + 8u * g.rows;
    return ((g.rows + 6) * g.ls + 2 * kFrontTmpRows * g.ts) * 2u + kFrontQueue * 2u + 32u + 32u * g.rows;
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
template <bool BT601 = false>  // false: the reference's 0.229 red weight (Q1); true: 0.299 (the intended mode's IM-1)
__device__ __forceinline__ uint32_t luminance_pair_f16(uint32_t rgba0, uint32_t rgba1) {
    const float2_t rc_hi = {0x1.010102p-8f, 0x1.010102p-8f}, rc_lo = {-0x1.fdfdfep-33f, -0x1.fdfdfep-33f};
    const float2_t R = {(float)(rgba0 & 255u), (float)(rgba1 & 255u)};
    const float2_t G = {(float)((rgba0 >> 8) & 255u), (float)((rgba1 >> 8) & 255u)};
    const float2_t B = {(float)((rgba0 >> 16) & 255u), (float)((rgba1 >> 16) & 255u)};
    const float2_t tr = R * rc_lo, tg = G * rc_lo, tb = B * rc_lo;
    const float2_t r = __builtin_elementwise_fma(R, rc_hi, tr);  // exact byte/255, see unorm8_exact
    const float2_t g = __builtin_elementwise_fma(G, rc_hi, tg);
    const float2_t b = __builtin_elementwise_fma(B, rc_hi, tb);
    const float2_t pr = r * (BT601 ? 0.299f : 0.229f), pg = g * 0.587f, pb = b * 0.114f;
    const float2_t s = pr + pg;
    const float2_t l = s + pb;
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
template <bool L0, bool Y8 = false, int RB = kFrontRows, bool UA = false, bool TILED = false>
__global__ __launch_bounds__(L0 ? kFrontThreadsL0 : kFrontThreadsLN, L0 ? 8 : 4) void k_front(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                         uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                         uint16_t* __restrict__ blur_rowc, Pyramid pyr,
                                                         FrontGeom geo, float thr, uint32_t* __restrict__ seg_counts,
                                                         CornerData* __restrict__ segments) {
    constexpr int NT = L0 ? kFrontThreadsL0 : kFrontThreadsLN, R = RB, TC = kFrontTmpRows;
    // 16-bit queue entries: [15:XB+1] row of the band, [XB:4] (x - x0) / 8, [3:0] pixel and polarity (front_mask_bit)
    const int XB = TILED ? (int)geo.xb : front_x_bits(RB);
    static_assert((RB - 1) < (1 << (15 - front_x_bits(RB))), "band row does not fit the queue entry");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int LS = (int)geo.ls, TS = (int)geo.ts;
    const int row3 = 3 * LS + kLdsPad;  // band row 0, column 0 inside the staged rows (three halo rows above, the left pad)
    half_t* const grey = reinterpret_cast<half_t*>(lds_raw);             // (R+6) rows x LS
    half_t* const tmp = grey + (R + 6) * LS;                              // 2 x TC rows x TS (TILED: geo.tmp_halfs)
    // Queues of the FAST phase, 16-bit entries (see XB above).
    //   A: pre-test survivors (own storage); B: survivors of the diagonal filter; C: corners.  B and C
    //   live in the blur intermediate's storage (phase C starts after a barrier).  Whenever a queue is
    //   full the item is finished in place, so capacities only affect speed.
    uint16_t* const queue_a = reinterpret_cast<uint16_t*>(tmp + (TILED ? (int)geo.tmp_halfs : 2 * TC * TS));
    uint16_t* const queue_b = reinterpret_cast<uint16_t*>(tmp);
    const uint32_t cap_b = 3u * (uint32_t)TS, cap_c = (uint32_t)TS;
    uint16_t* const queue_c = queue_b + cap_b;
    uint32_t* const qa_count = reinterpret_cast<uint32_t*>(queue_a + kFrontQueue);
    uint32_t* const qb_count = qa_count + 1;
    uint32_t* const qc_count = qa_count + 2;
    uint32_t* const c_count = qa_count + 3;  // corners found by this band: [0] first list, [1] second list
    float4* const blur_k1 = reinterpret_cast<float4*>(qa_count + 8);  // per band row: taps 0, 2, 3 of blur pass 1
    float4* const blur_k2 = blur_k1 + R;                              // per band row: the same for pass 2, and c2
    half_t* const far = reinterpret_cast<half_t*>(blur_k2 + R);       // TILED, tile 0: per band row grey(w - 1), grey(far_i0), grey(far_i1)
    // per column >= blur_q: tap positions.  Used by phase C only, which runs before the detector: it borrows the storage of
    // queues B and C, which are first written in stage S1 -- behind the barrier that ends the pre-test, which every wave
    // reaches after its share of phase C.
    BlurCol* const blur_cols = reinterpret_cast<BlurCol*>(tmp);

    // ---- which band (tile) of which frame: keep all bands of a frame on one XCD so halo rows hit its L2
    uint32_t frame, band, tile = 0;
    {
        const uint32_t L = blockIdx.x, n_wg = TILED ? geo.n_bands * geo.n_ct : geo.n_bands;  // TILED: band-major, a band's tiles side by side
        uint32_t wg;
        if (geo.xcd_swizzle) {
            const uint32_t xcd = L & 7u, slot = L >> 3;
            frame = (slot / n_wg) * 8u + xcd;
            wg = slot % n_wg;
        } else {
            frame = L / n_wg;
            wg = L % n_wg;
        }
        band = wg;
        if (TILED) band = wg / geo.n_ct, tile = wg - band * geo.n_ct;
    }
    const uint32_t lvl = geo.lvl;
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const int y0 = (int)band * R;
    const int x0 = TILED ? (int)(tile * geo.tw) : 0;                                          // first column of the tile
    const int xe = TILED ? min(x0 + (int)geo.tw, max(w, (int)geo.gw)) : max(w, (int)geo.gw);  // one past its last (level or dispatch domain)
    const bool blur_tile = !TILED || tile == 0u;                                              // tile 0 does the band's blur (phase C)
    const bool far_cols = TILED && geo.n_ct > 1u;                                             // ... with three grey columns from beyond its own
    const int tid = (int)threadIdx.x;
    uint16_t* const gray_f = gray + (size_t)frame * pyr.stride;
    uint16_t* const blur_lvl = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    const size_t slot = (size_t)frame * geo.n_slots + geo.slot_base + (TILED ? band * geo.n_ct + tile : band);
    const bool two_lists = geo.n_classes == 2u;
    CornerData* const seg = segments + slot * geo.seg_cap * geo.n_classes;

    // diagnostic stamps (geo.stamps != null): cycles of wave 0 between consecutive marks, summed over workgroups
#ifdef TINYORB_STAMPS
    unsigned long long t_last = geo.stamps ? __builtin_readcyclecounter() : 0ull;
    auto stamp = [&](int slot) {
        if (geo.stamps && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(geo.stamps + (L0 ? 0 : 16) + slot, now - t_last);
            t_last = now;
        }
    };
#else
    auto stamp = [](int) {};  // the shipped build executes no stamp
#endif
    if (tid < 5) qa_count[tid] = 0u;

    // =========================== A: stage grey rows [y0-3, y0+R+3) ===========================
    // Thread -> (column group tx, row phase ty): a thread keeps its column group and walks down the
    // rows, so per item there is one address increment instead of a division; four 16-byte loads are
    // in flight per thread before the first is consumed.
    {
        // 16-byte items of a staged row (RGBA quads / half8 groups).  TILED: the tile's own columns, one item of halo to its
        // left when it has a neighbour there, and what covers 4 columns to its right (as far as the row goes)
        const int it0 = !TILED ? 0 : (L0 ? (x0 >> 2) - (x0 > 0 ? 1 : 0) : (x0 >> 3) - (x0 > 0 ? 1 : 0));  // first item, counted from the row's start
        const int it1 = L0 ? min(xe + 4 + 3, UA ? w + 3 : w) >> 2 : (xe + 4 + 7) >> 3;                     // TILED: one past the last
        const int per_row = TILED ? it1 - it0 : (L0 ? ((UA ? w + 3 : w) >> 2) : ((LS - kLdsPad) >> 3));
        const int rpp = NT / per_row;                                 // rows covered per pass (>= 1: checked on the host)
        const int ty = (int)(((float)tid + 0.5f) * (1.0f / (float)per_row));
        const int tx = tid - __mul24(ty, per_row);
        const bool lane_ok = ty < rpp;
        const uint8_t* src0 = frames + (size_t)frame * frame_bytes;
        const __amdgpu_buffer_rsrc_t frame_rsrc =  // gfx9 raw buffer: stride 0, num_records = bytes of the frame (< 2^32, checked at create)
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src0), 0, (int)frame_bytes, kBufferWord3Raw);
        // Y8: the same frame as a typed buffer -- four texels come back as byte/255 in binary32, converted by the texture
        // path instead of 4 v_cvt_f32_ubyte + 8 multiply/fma on the vector unit, which is what bounds this kernel
        const __amdgpu_buffer_rsrc_t frame_unorm =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src0), 0, (int)frame_bytes, kBufferWord3Unorm8x4);
        const uint16_t* srcn = gray_f + pyr.off[lvl];
        constexpr int U = L0 ? 8 : 4;  // 16-byte loads in flight per thread (VGPR budget: 64 at 8 waves/SIMD): level 0 issues all of a thread's rows at once
        // Level 0 walks with increments: the mirrored source offset, the LDS offset and the row of the thread's first
        // item are computed once; a step of rpp rows is three additions (the multiplies, clamps and selects of a
        // per-item address are the 4-cycle kind of instruction, and this phase is a third of the kernel's count).
        const int bpp = Y8 ? 1 : 4;
        const int txc = lane_ok ? tx : 0;
        int gy_w = y0 - 3 + ty;                                                        // row of the walking item
        uint32_t off_w = (uint32_t)(__mul24(h - 1 - gy_w, w) + (it0 + txc) * 4) * (uint32_t)bpp;   // its (mirrored) byte offset
        int dst_w = __mul24(ty, LS) + kLdsPad + (it0 + tx) * 4 - x0;                   // its LDS offset (halfs): LDS column kLdsPad <-> image column x0
        const uint32_t off_step = (uint32_t)(rpp * w * bpp);
        const int dst_step = rpp * LS;
        for (int lyb = ty, ly_base = 0; lyb < R + 6; lyb += rpp * U, ly_base += rpp * U) {
            uint4 v[U];
            int dst[U];
            bool live[U];
            // TILED: passes of this round that still reach rows of the tile (uniform: a narrow tile stages many rows per
            // pass, and a load for a row past the tile would be real traffic for nothing)
            const int n_u = TILED ? min(U, (R + 6 - ly_base + rpp - 1) / rpp) : U;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int ly = lyb + u * rpp;
                const int gy = y0 - 3 + ly;
                const bool in_band = lane_ok && ly < R + 6;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (TILED) {
                    live[u] = false;
                    dst[u] = 0;
                    if (u >= n_u) continue;
                }
                if (L0) {
                    // rows outside the image are never read by a pixel that passes the guard (fast.wgsl:77);
                    // the input row is the vertically mirrored one (grayscale.wgsl:16-25).  Byte offsets inside a frame
                    // are < 2^28 (checked at create).  The load itself is unconditional so that all of a thread's loads
                    // are issued back to back.
                    // A row above or below the image needs no test of its own in the aligned variants: its buffer load
                    // returns zeros, which are staged like any other row and never read.  The general variant stores
                    // rows to HBM and reads Y8 bytes through a pointer: it keeps the test.
                    const bool ok = in_band && (!UA || (uint32_t)gy_w < (uint32_t)h);
                    live[u] = ok;  // a lane mask in scalar registers: no select here and no compare at the store
                    dst[u] = dst_w;
                    // BUFFER loads with the frame as the buffer (stride 0, num_records = its bytes): a 32-bit offset per lane
                    // instead of a 64-bit address, and no select for the items that do not exist -- a row above or below
                    // the image has an offset past the frame ((h-1-gy)*w*4 >= h*w*4, or negative = huge) and reads zeros,
                    // any other lane without an item reads something valid that is never stored.  Buffer loads also only
                    // ask for dword alignment, which is what the general variant's rows have.
                    const int off = (int)off_w;
                    if (Y8 && UA) {  // rows start on any byte; the last group of a row may be partial
                        const uint8_t* q = src0 + (size_t)(ok ? off_w : 0u);
                        const int left = ok ? w - (it0 + tx) * 4 : 4;
                        v[u].x = (uint32_t)q[0] | ((uint32_t)q[left > 1 ? 1 : 0] << 8) | ((uint32_t)q[left > 2 ? 2 : 0] << 16) |
                                 ((uint32_t)q[left > 3 ? 3 : 0] << 24);
                    } else if (Y8) {  // four texels = four bytes, converted on the way in (rows are 4-byte aligned here)
                        v[u] = __builtin_bit_cast(uint4, buffer_load_format_xyzw(frame_unorm, off, 0, 0));
                    } else {  // RGBA quad; in the general variant (UA) the last quad of a row may run into the next row: those
                              // texels land in columns >= w, which nothing ever uses
                        v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(frame_rsrc, off, 0, 0));
                    }
                    gy_w += rpp;
                    off_w -= off_step;
                    dst_w += dst_step;
                } else {
                    // f16 mip from HBM; texels outside the level are stored as 0 (CRD-6): at octaves >= 1 the
                    // reference's guard and dispatch size let pixels near/over the level edge through (Q8).
                    const int x = (it0 + tx) * 8;
                    live[u] = in_band;
                    dst[u] = __mul24(ly, LS) + kLdsPad + x - x0;
                    if (in_band && gy >= 0 && gy < h && x < w) {
                        const uint16_t* row = srcn + (size_t)(uint32_t)__mul24(gy, w);
                        if ((w & 7) == 0 && x + 8 <= w) {
                            v[u] = *reinterpret_cast<const uint4*>(row + x);
                        } else if ((w & 1) == 0) {  // even width: rows start on a dword, texel pairs never straddle the row end
                            const uint32_t* row2 = reinterpret_cast<const uint32_t*>(row + x);
                            v[u] = make_uint4(row2[0], x + 2 < w ? row2[1] : 0u, x + 4 < w ? row2[2] : 0u, x + 6 < w ? row2[3] : 0u);
                        } else {
                            uint32_t e[8];
#pragma unroll
                            for (int k = 0; k < 8; k++) e[k] = (x + k < w) ? (uint32_t)row[x + k] : 0u;
                            v[u] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16),
                                              e[6] | (e[7] << 16));
                        }
                    }
                }
            }
            // The first item is the first to be consumed: pin its load here, in front of the conversions.  Left alone, hipcc
            // sinks that one load into the conditional block that consumes it -- behind the other seven --, and the loads
            // returning in order, the first conversion then waits for all eight instead of one.
            if (L0) asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w));
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (live[u]) {
                    if (L0) {
                        uint2 out;
                        if (Y8 && UA) {
                            const uint32_t b = v[u].x;
                            out.x = pack_half2(unorm8_exact((float)(b & 255u)), unorm8_exact((float)((b >> 8) & 255u)));
                            out.y = pack_half2(unorm8_exact((float)((b >> 16) & 255u)), unorm8_exact((float)(b >> 24)));
                        } else if (Y8) {  // v holds byte/255 of the four texels: two packed conversions (CRD-3)
                            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(out.x) : "v"(__builtin_bit_cast(float, v[u].x)), "v"(__builtin_bit_cast(float, v[u].y)));
                            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(out.y) : "v"(__builtin_bit_cast(float, v[u].z)), "v"(__builtin_bit_cast(float, v[u].w)));
                        } else {
                            out.x = luminance_pair_f16(v[u].x, v[u].y);
                            out.y = luminance_pair_f16(v[u].z, v[u].w);
                        }
                        *reinterpret_cast<uint2*>(grey + dst[u]) = out;
                        // Level 1 is not an exact half of level 0 (odd width or height): the generic blit (k_mip) builds
                        // it from the level-0 plane, which the band's own rows therefore also store.
                        const int ly = lyb + u * rpp;
                        const int qx = (it0 + tx) * 4;  // first column of this quad
                        if (UA && geo.store_grey && ly >= 3 && ly < R + 3 && (!TILED || (qx >= x0 && qx < xe))) {  // TILED: the tile's own columns
                            uint16_t* g = gray_f + pyr.off[lvl] + (size_t)(uint32_t)(__mul24(y0 - 3 + ly, w) + qx);
                            if (w & 3) {
                                const uint32_t t[4] = {out.x & 0xffffu, out.x >> 16, out.y & 0xffffu, out.y >> 16};
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    if (qx + k < w) g[k] = (uint16_t)t[k];
                            } else {
                                *reinterpret_cast<uint2*>(g) = out;
                            }
                        }
                    } else {
                        *reinterpret_cast<uint4*>(grey + dst[u]) = v[u];
                    }
                }
            }
        }
    }
    // TILED: tile 0 of several does the band's blur, whose row constants read three grey texels per row that lie beyond its
    // own columns: the row's last one and the two that pass 1 lerps at the last column.  Same conversion
    // as the staged rows (CRD-1..3); rows outside the level are never used.
    if (TILED && blur_tile && far_cols && (geo.phase_mask & 8u)) {
        for (int i = tid; i < 3 * R; i += NT) {
            const int r = i / 3, k = i - 3 * r, gy = y0 + r;
            const int col = k == 0 ? w - 1 : (k == 1 ? (int)geo.far_i0 : (int)geo.far_i1);
            uint16_t g = 0;
            if (gy < h) {
                if (L0) {
                    const uint8_t* px = frames + (size_t)frame * frame_bytes + (size_t)(uint32_t)(__mul24(h - 1 - gy, w) + col) * (Y8 ? 1u : 4u);
                    if (Y8) {
                        g = half_bits(to_half(unorm8_exact((float)px[0])));
                    } else {
                        const uint32_t t = *reinterpret_cast<const uint32_t*>(px);
                        g = (uint16_t)(luminance_pair_f16(t, t) & 0xffffu);
                    }
                } else {
                    g = gray_f[pyr.off[lvl] + (size_t)(uint32_t)(__mul24(gy, w) + col)];
                }
            }
            far[4 * r + k] = bits_half(g);
        }
    }
    stamp(0);  // A: staging
    __syncthreads();
    stamp(1);  // barrier

    // ---- blur row constants (one thread per band row; published by the barriers of phase B) ----
    // The reference adds its blur offsets to the normalised u coordinate (Q11), so taps 0, 2 and 3 always clamp
    // to column 0 / w-1 and are three constants per row: acc = (((0 + t[0]*w0) + lerp*w1) + t[w-1]*w2) + t[w-1]*w3.
    // Tap 1 samples 0.4392*w texels to the left: for x < blur_p it clamps to column 0 as well and pass 1 is one
    // value per row there (c1); pass 2 reads pass 1 at x - 0.4392*w again, so the final blur is one value per
    // row (c2) for every x < blur_q -- 88 % of the columns.
    //   blur_k1[r] = pass 1: {0 + t0*w0, tl*w2, tl*w3, c1}          (t0, tl: grey row r at columns 0, w-1)
    //   blur_k2[r] = pass 2: {0 + c1*w0, p1l*w2, p1l*w3, c2}        (p1l: pass 1 at column w-1)
    if (blur_tile && tid < R) {
        const half_t* row = grey + (tid + 3) * LS + kLdsPad;
        const float t0 = from_half(row[0]), tl = from_half(far_cols ? far[4 * tid] : row[w - 1]);
        const float a0 = t0 * kBlurWgt[0], a2 = tl * kBlurWgt[2], a3 = tl * kBlurWgt[3];
        const float base = 0.0f + a0;
        auto finish = [&](float bs, float lerp, float k2, float k3) {
            const float ws = lerp * kBlurWgt[1];
            float acc = bs + ws;
            acc = acc + k2;
            acc = acc + k3;
            return from_half(to_half(acc));  // R16Float store (CRD-3)
        };
        const float c1 = finish(base, t0, a2, a3);
        float p1l = c1;  // pass 1 at the last column
        if ((int)geo.blur_p < w) {
            const BlurTap t = blur_tap((uint32_t)(w - 1), (uint32_t)w, kBlurOff[1]);  // (t.i0, t.i1) = (geo.far_i0, geo.far_i1)
            const float v0 = from_half(far_cols ? far[4 * tid + 1] : row[t.i0]), v1 = from_half(far_cols ? far[4 * tid + 2] : row[t.i1]);
            const float d = v1 - v0;
            p1l = finish(base, v0 + t.f * d, a2, a3);
        }
        const float b0 = c1 * kBlurWgt[0], b2 = p1l * kBlurWgt[2], b3 = p1l * kBlurWgt[3];
        const float base2 = 0.0f + b0;
        blur_k1[tid] = make_float4(base, a2, a3, c1);
        blur_k2[tid] = make_float4(base2, b2, b3, finish(base2, c1, b2, b3));
    }

    // ---- blur column table (the last threads of the workgroup, so that wave 0 is not doing both) ----
    if ((geo.phase_mask & 8u) && blur_tile)
    for (int c = NT - 1 - tid; c < (int)geo.n_var; c += NT) {  // one entry per thread while n_var <= NT (it is about 0.12 w)
        const int x = (int)geo.blur_q + c, P = (int)geo.blur_p;
        const BlurTap t2 = blur_tap((uint32_t)x, (uint32_t)w, kBlurOff[1]);
        BlurCol e;
        e.f2 = t2.f;
        e.pad = 0u;
        if (t2.i0 < P) {
            e.a0 = 0xffffu, e.a1 = 0u, e.fa = 0.0f;
        } else {
            const BlurTap t = blur_tap((uint32_t)t2.i0, (uint32_t)w, kBlurOff[1]);
            e.a0 = (uint16_t)t.i0, e.a1 = (uint16_t)t.i1, e.fa = t.f;
        }
        if (t2.i1 < P) {
            e.b0 = 0xffffu, e.b1 = 0u, e.fb = 0.0f;
        } else {
            const BlurTap t = blur_tap((uint32_t)t2.i1, (uint32_t)w, kBlurOff[1]);
            e.b0 = (uint16_t)t.i0, e.b1 = (uint16_t)t.i1, e.fb = t.f;
        }
        blur_cols[c] = e;
    }

    // Phases B (FAST) and C (mip + blur) only share the read-only grey rows.
    auto phase_B = [&]() {
        // =========================== B1: 4-point pre-test, 16 px per item (8 above level 0) ===========================
        if (geo.phase_mask & 1u) {
            // Level 0: an item is sixteen pixels of a row, tested as two halves of eight (two packed pixels per operation,
            // four pixel pairs per half); what is done once per item -- index arithmetic, the fast.wgsl:77 guard, the slot
            // reservation for its survivors -- then weighs half as much per pixel as with items of eight (k_front<true>
            // 0.354 -> 0.346 ms).  The levels above keep items of eight: their bands hold 1.25 sixteen-pixel items per
            // thread, and the two waves with a second item set the pace (0.072 -> 0.076 ms).
            constexpr int IH = L0 ? 2 : 1, IW = 8 * IH;  // halves and pixels per item
            const int cols_t = TILED ? min(x0 + (int)geo.tw, (int)geo.gw) - x0 : (int)geo.gw;  // dispatch columns (of this tile): a multiple of 8
            const int g16 = (cols_t + IW - 1) / IW;                                           // items of a row
            const float inv_g16 = 1.0f / (float)max(g16, 1);
            const int n_items = R * g16;
            // fast.wgsl:77 -- level-0 dimensions for every octave, u32 arithmetic (Q8)
            const uint32_t lim_x = pyr.w[0] - 16u, lim_y = pyr.h[0] - 16u;
            // the item that holds column lim_x (the first one past the guard) and the mask bits of its pixels below lim_x;
            // the item whose second half lies outside the dispatch domain (a row of 8 (mod 16) columns)
            const int x_cut = x0 + (((int)lim_x - x0) & ~(IW - 1));
            const int x_half = (IH == 2 && (cols_t & 8)) ? x0 + (cols_t & ~15) : -1;
            uint32_t keep_cut = 0;
#pragma unroll
            for (int k = 0; k < IW; k++)
                if (k < (int)lim_x - x_cut) keep_cut |= (front_mask_bit(k & 7, false) | front_mask_bit(k & 7, true)) << (16 * (k >> 3));
            // thr_lo: one f16 ulp below RD16(thr) (see below); -min_subnormal when that would pass zero
            uint32_t tb = half_bits(to_half(thr));
            if (from_half(bits_half((uint16_t)tb)) > thr) tb--;
            tb = tb ? tb - 1u : 0x8001u;
            const half2_t thr_lo2 = __builtin_bit_cast(half2_t, tb | (tb << 16));
            for (int i = tid; i < n_items; i += NT) {
                const int lyc = (int)(((float)i + 0.5f) * inv_g16);
                const int xl = (i - __mul24(lyc, g16)) * IW, x = x0 + xl;  // tile-local and image column of the item's first pixel
                const uint32_t gy = (uint32_t)(y0 + lyc);
                if (!(gy < geo.gh && gy > 16u && gy < lim_y)) continue;
                if ((uint32_t)x + (uint32_t)(IW - 1) <= 16u || (uint32_t)x >= lim_x) continue;
                const half_t* row16 = grey + row3 + (int)__umul24((uint32_t)lyc, (uint32_t)LS) + xl;  // row lyc + 3 of the staged rows
                // Pre-test (fast.wgsl:85-95), as a CONSERVATIVE filter: every pixel the reference's pre-test
                // passes is kept, a few extra may be; the decision itself is made by the 16-point test, whose
                // 12-run already implies the 3-of-4 compass condition, so results do not change.
                //  * ">= 3 of the 4 compass diffs beyond thr" <=> the 2nd smallest (2nd largest) neighbour value
                //    minus the centre is beyond thr; grey values are non-negative f16, so the selection network
                //    runs on their bit patterns as packed u16 (2 pixels per op);
                //  * the two differences and compares run in packed f16 against thr_lo, one f16 ulp below
                //    RD16(thr): x > thr  =>  RN16(x) >= RD16(thr) > thr_lo, so nothing is missed;
                //  * compares are subtractions whose sign bits are the answer (a float subtraction has the
                //    sign of the exact difference).
                uint32_t cand = 0;  // bit 16 h + front_mask_bit(k, under): pixel 8 h + k survives
    #pragma unroll
                for (int hh = 0; hh < IH; hh++) {
                    const half_t* rowc = row16 + 8 * hh;
                    const uint2 qa = *reinterpret_cast<const uint2*>(rowc - 4);
                    const uint4 qb = *reinterpret_cast<const uint4*>(rowc);
                    const uint2 qc = *reinterpret_cast<const uint2*>(rowc + 8);
                    const uint4 qu = *reinterpret_cast<const uint4*>(rowc - 3 * LS);
                    const uint4 qd = *reinterpret_cast<const uint4*>(rowc + 3 * LS);
                    const uint32_t dw[8] = {qa.x, qa.y, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y};  // dw[j] = grey(x-4+2j, x-3+2j) of this half
                    const uint32_t upw[4] = {qu.x, qu.y, qu.z, qu.w}, dnw[4] = {qd.x, qd.y, qd.z, qd.w};
                    uint32_t e_ovr[4], e_und[4];  // sign bit of each half word = the answer for that pixel
    #pragma unroll
                    for (int j = 0; j < 4; j++) {  // pixel pair (x+2j, x+2j+1)
                        const ushort2_t left = as_u16x2(__builtin_amdgcn_alignbit(dw[j + 1], dw[j], 16));       // x+2j-3, x+2j-2
                        const ushort2_t right = as_u16x2(__builtin_amdgcn_alignbit(dw[j + 4], dw[j + 3], 16));  // x+2j+3, x+2j+4
                        const ushort2_t upp = as_u16x2(upw[j]), dwn = as_u16x2(dnw[j]);
                        const ushort2_t lo1 = __builtin_elementwise_min(left, right), hi1 = __builtin_elementwise_max(left, right);
                        const ushort2_t lo2 = __builtin_elementwise_min(upp, dwn), hi2 = __builtin_elementwise_max(upp, dwn);
                        const ushort2_t m1 = __builtin_elementwise_max(lo1, lo2), m2 = __builtin_elementwise_min(hi1, hi2);
                        const half2_t second_lo = __builtin_bit_cast(half2_t, __builtin_elementwise_min(m1, m2));
                        const half2_t second_hi = __builtin_bit_cast(half2_t, __builtin_elementwise_max(m1, m2));
                        const half2_t c2 = __builtin_bit_cast(half2_t, dw[j + 2]);
                        const half2_t e_over = thr_lo2 - (second_lo - c2);   // negative  <=>  second_lo - c > thr_lo
                        const half2_t e_under = (second_hi - c2) + thr_lo2;  // negative  <=>  second_hi - c < -thr_lo
                        e_ovr[j] = __builtin_bit_cast(uint32_t, e_over);
                        e_und[j] = __builtin_bit_cast(uint32_t, e_under);
                    }
                    // The sixteen sign bits (8 pixels x 2 polarities, which exclude each other) as one 16-bit mask, bit
                    // front_mask_bit(k, under): every word's two sign bits become 0/1 in its halves (one packed shift), a
                    // shift-or per word places them -- even pixels in the low half, odd ones in the high half --, one byte
                    // permute folds the halves (into the upper 16 bits for the item's second half).
                    uint32_t acc = 0;
    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t so = __builtin_bit_cast(uint32_t, as_u16x2(e_ovr[j]) >> (unsigned short)15);
                        const uint32_t su = __builtin_bit_cast(uint32_t, as_u16x2(e_und[j]) >> (unsigned short)15);
                        acc = j == 0 ? so : ((so << j) | acc);
                        acc = (su << (4 + j)) | acc;
                    }
                    cand |= __builtin_amdgcn_perm(0u, acc, hh == 0 ? 0x0c0c0200u : 0x02000c0cu);  // byte 0 | byte 2 << 8 (<< 16)
                }
                // fast.wgsl:77 guard on x: keep pixels k with 16 < x+k < lim_x.  Only two items of a row are cut -- the one
                // at x = 16 loses pixel 0, the one that holds column lim_x loses its tail --, and both masks are the same for
                // every row: two compares and selects here instead of sixteen each.  A third item may have no second half.
                cand &= x == 16 ? ~(front_mask_bit(0, false) | front_mask_bit(0, true)) : ~0u;
                cand &= x == x_cut ? keep_cut : ~0u;
                if (IH == 2) cand &= x == x_half ? 0xffffu : ~0u;
                if (cand) {  // one LDS atomic for all survivors of this item
                    // every lane reserves its own slots with the LDS's returning add (lds_add_rtn, orb_device.h)
                    const uint32_t n_cand = (uint32_t)__builtin_popcount(cand);
                    uint32_t qs = lds_add_rtn(qa_count, n_cand);
                    const uint32_t base = ((uint32_t)lyc << (XB + 1)) | ((uint32_t)xl << 1);  // xl is a multiple of 16 (8): the low five (four) bits are free
                    if (qs + n_cand <= (uint32_t)kFrontQueue) {  // all survivors of the item fit: no test per entry
                        while (cand) {
                            const uint32_t p = (uint32_t)__builtin_ctz(cand);
                            cand &= cand - 1u;
                            queue_a[qs++] = (uint16_t)(base | p);
                        }
                    } else {
                        while (cand) {
                            const uint32_t p = (uint32_t)__builtin_ctz(cand);
                            cand &= cand - 1u;
                            if (qs < (uint32_t)kFrontQueue) {
                                queue_a[qs] = (uint16_t)(base | p);
                            } else {  // queue full (pathological frame): finish in place
                                const int k = (int)(((p & 3u) << 1) | ((p >> 3) & 1u) | ((p >> 4) << 3));
                                uint32_t angle;
                                const bool hit = fast_full_test(row16 + k, LS, thr, &angle);
                                segment_append(hit, (uint32_t)(x + k), gy, angle, lvl, c_count, seg, geo.seg_cap, two_lists);
                            }
                            qs++;
                        }
                    }
                }
            }
        }
        stamp(2);  // B1
        __syncthreads();
        stamp(3);

        // =========================== B2: thin, test, orient -- each stage on densely packed lanes ===========
        if (geo.phase_mask & 2u) {
            auto locate = [&](uint32_t e, uint32_t* x, uint32_t* gy) -> const half_t* {
                const uint32_t lyc = e >> (XB + 1);                                       // e is a 16-bit entry
                const uint32_t k = ((e & 3u) << 1) | ((e >> 3) & 1u) | ((e >> 1) & 8u);   // pixel of the item: front_mask_bit in [3:0], the half in [4]
                const uint32_t xl = ((e >> 1) & (((1u << XB) - 1u) & ~15u)) | k;  // tile-local column
                *x = (uint32_t)x0 + xl;
                *gy = (uint32_t)y0 + lyc;
                return grey + row3 + (int)__umul24(lyc, (uint32_t)LS) + (int)xl;
            };
            auto is_over = [](uint32_t e) { return (e & 4u) == 0u; };  // polarity of the pre-test that passed
            // stage 1: diagonal 3-of-4 filter (a necessary condition of a 12-run), A -> B
            const uint32_t n_a = min(*qa_count, (uint32_t)kFrontQueue);
            for (uint32_t i = (uint32_t)tid; i < n_a; i += NT) {
                const uint32_t e = queue_a[i];
                uint32_t x, gy;
                const half_t* ctr = locate(e, &x, &gy);
                if (diagonal_filter(ctr, LS, thr, is_over(e))) {
                    const uint32_t qs = atomicAdd(qb_count, 1u);
                    if (qs < cap_b) {
                        queue_b[qs] = (uint16_t)e;
                    } else {
                        uint32_t angle;
                        const bool hit = fast_full_test(ctr, LS, thr, &angle);
                        segment_append(hit, x, gy, angle, lvl, c_count, seg, geo.seg_cap, two_lists);
                    }
                }
            }
            stamp(4);  // S1
            __syncthreads();
            stamp(5);
            // stage 2: 16-point masks + 12-streak, B -> C
            const uint32_t n_b = min(*qb_count, cap_b);
            for (uint32_t i = (uint32_t)tid; i < n_b; i += NT) {
                const uint32_t e = queue_b[i];
                uint32_t x, gy;
                const half_t* ctr = locate(e, &x, &gy);
                if (ring_is_corner_polar(ctr, LS, thr, is_over(e))) {
                    const uint32_t qs = atomicAdd(qc_count, 1u);
                    if (qs < cap_c)
                        queue_c[qs] = (uint16_t)e;
                    else
                        segment_append(true, x, gy, ring_angle(ctr, LS), lvl, c_count, seg, geo.seg_cap, two_lists);
                }
            }
            stamp(6);  // S2
            __syncthreads();
            stamp(7);
            // stage 3: orientation of the corners, append to the band's segment
            const uint32_t n_c = min(*qc_count, cap_c);
            for (uint32_t i = (uint32_t)tid; i < n_c; i += NT) {
                uint32_t x, gy;
                const half_t* ctr = locate(queue_c[i], &x, &gy);
                segment_append(true, x, gy, ring_angle(ctr, LS), lvl, c_count, seg, geo.seg_cap, two_lists);
            }
        }
        stamp(8);  // S3
        __syncthreads();  // queues B and C share storage with phase C's blur intermediate
        stamp(9);
    };
    auto phase_C = [&]() {
        // =========================== C0: next mip level (blit.wgsl, exact 2x2 case) ===========================
        if (geo.write_mip && (geo.phase_mask & 4u)) {
            const int wd = (int)pyr.w[lvl + 1], hd = (int)pyr.h[lvl + 1];
            uint16_t* dst = gray_f + pyr.off[lvl + 1];
            const int xd0 = x0 >> 1, xd1 = TILED ? min(min(x0 + (int)geo.tw, w) >> 1, wd) : wd;  // the tile's columns of the next level
            const int g4 = max(xd1 - xd0 + 3, 0) >> 2;
            const float inv_g4 = 1.0f / (float)max(g4, 1);
            const int n_items = (R / 2) * g4;
            const bool vec_ok = (wd & 3) == 0;
            for (int i = tid; i < n_items; i += NT) {
                const int r = (int)(((float)i + 0.5f) * inv_g4);
                const int xdl = (i - __mul24(r, g4)) * 4, xd = xd0 + xdl;
                const int yd = (y0 >> 1) + r;
                if (yd >= hd) continue;
                const half_t* top = grey + row3 + (int)__umul24((uint32_t)r, (uint32_t)(2 * LS)) + 2 * xdl;
                const uint4 qt = *reinterpret_cast<const uint4*>(top);
                const uint4 qb = *reinterpret_cast<const uint4*>(top + LS);
                const uint32_t tw[4] = {qt.x, qt.y, qt.z, qt.w}, bw[4] = {qb.x, qb.y, qb.z, qb.w};
                uint16_t o[4];
    #pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float st = add_halves(tw[k]);  // a + b, c + d: the two texels of a row share a register (CRD-4)
                    const float sb = add_halves(bw[k]);
                    o[k] = half_bits(to_half((st + sb) * 0.25f));
                }
                uint16_t* out = dst + (size_t)(uint32_t)(__mul24(yd, wd) + xd);
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(out) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
                } else {
    #pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (xd + k < xd1) out[k] = o[k];
                }
            }
        }

        stamp(10);  // C0
        // =========================== C: literal blur, both passes ===========================
        // Columns < blur_q get the row constant c2 (16-byte stores).  For a column x >= blur_q, pass 2 needs
        // pass 1 at two neighbouring columns j0, j1 (>= blur_p - 1), each of which needs two grey texels: the
        // thread evaluates those two pass-1 values itself (f16-rounded like the R16Float intermediate,
        // orb.rs:291-304) -- no intermediate plane, no barrier.  Pass 1 at any other column is never read.
        if ((geo.phase_mask & 8u) && blur_tile) {
            const int rows = min(R, h - y0);  // band rows that exist in this level (uniform per block)
            const int Q = (int)geo.blur_q;
            if (rows > 0) {
                // constant stretch: one f16 per row in the row-constant array covers columns [0, Qa), Qa = Q rounded
                // down to a multiple of 8 (the plane itself is only written from column Qa on: k_brief_rows
                // samples the constants directly, the first 88 % of the plane never travel through HBM)
                const int Qa = Q & ~7;
                if (tid < rows)
                    blur_rowc[(size_t)frame * pyr.row_stride + pyr.row_off[lvl] + (uint32_t)(y0 + tid)] =
                        half_bits(to_half(blur_k2[tid].w));
                for (int i = tid; i < rows * (Q - Qa); i += NT) {
                    const int r = i / (Q - Qa), x = Qa + i % (Q - Qa);
                    blur_lvl[(size_t)(uint32_t)(__mul24(y0 + r, w) + x)] = half_bits(to_half(blur_k2[r].w));
                }
                // per-pixel stretch [Q, w): a thread takes two neighbouring columns (level 0) and RPI rows of the band; the columns' tap
                // positions come from the band's table, so no blur_tap() (a division and a floor) runs here.  Pass 2 at
                // column x lerps pass 1 at columns j, j + 1 and at x + 1 at j + 1, j + 2: the middle one is evaluated once
                // (when the table says it is the same sample -- it is, save for clamping at the row's end).
                const int nvar = w - Q;
                if (nvar > 0) {
                    constexpr int RPI = 4;
                    constexpr int CPI = L0 ? 2 : 1;  // columns per item: pairs at level 0 (-1.8 %); the narrower levels have too few items (+3 % there)
                    const int npair = (nvar + CPI - 1) / CPI;
                    const float inv_npair = 1.0f / (float)npair;
                    const int n_items = ((rows + RPI - 1) / RPI) * npair;
                    for (int i = tid; i < n_items; i += NT) {
                        const int rg = (int)(((float)i + 0.5f) * inv_npair);
                        const int c = (i - __mul24(rg, npair)) * CPI;
                        const bool second = CPI == 2 && c + 1 < nvar;
                        const BlurCol e = blur_cols[c], e2 = blur_cols[second ? c + 1 : c];
                        const bool shared = e2.a0 == e.b0 && e2.a1 == e.b1 && e2.fa == e.fb;
#pragma unroll
                        for (int k = 0; k < RPI; k++) {
                            const int r = rg * RPI + k;
                            if (r >= rows) break;
                            const half_t* row = grey + (r + 3) * LS + kLdsPad;
                            const float4 k1 = blur_k1[r], k2 = blur_k2[r];
                            auto pass1 = [&](uint32_t i0, uint32_t i1, float f) {  // pass 1 as stored (f16, orb.rs:291-304)
                                if (i0 == 0xffffu) return k1.w;
                                const float v0 = from_half(row[i0]), v1 = from_half(row[i1]);
                                const float d = v1 - v0;
                                const float fd = f * d;
                                const float lerp = v0 + fd;
                                const float ws = lerp * kBlurWgt[1];
                                float acc = k1.x + ws;
                                acc = acc + k1.y;
                                acc = acc + k1.z;
                                return from_half(to_half(acc));
                            };
                            auto pass2 = [&](float u0, float u1, float f2) {
                                const float d = u1 - u0;
                                const float fd = f2 * d;
                                const float lerp = u0 + fd;
                                const float ws = lerp * kBlurWgt[1];
                                float acc = k2.x + ws;
                                acc = acc + k2.y;
                                acc = acc + k2.z;
                                return half_bits(to_half(acc));
                            };
                            const float u0 = pass1(e.a0, e.a1, e.fa);
                            const float u1 = pass1(e.b0, e.b1, e.fb);
                            uint16_t* out = blur_lvl + (size_t)(uint32_t)(__mul24(y0 + r, w) + Q + c);
                            out[0] = pass2(u0, u1, e.f2);
                            if (second) {
                                const float v0 = shared ? u1 : pass1(e2.a0, e2.a1, e2.fa);
                                const float v1 = pass1(e2.b0, e2.b1, e2.fb);
                                out[1] = pass2(v0, v1, e2.f2);
                            }
                        }
                    }
                }
            }
        }
    };
    __syncthreads();  // blur row constants published
    phase_C();
    stamp(11);  // C
    phase_B();
    __syncthreads();
    stamp(12);
    if (tid < (int)geo.n_classes) seg_counts[slot * geo.n_classes + tid] = c_count[tid];  // raw counts (may exceed seg_cap)
}

// Band slots of a frame: one per kFrontRows-row band per level, in level order.
struct BandGeom {
    uint32_t n_slots, seg_cap, n_frames, xcd_swizzle;
    uint32_t slot_base[kMaxLevels + 1];  // first slot of each level; [depth] = n_slots
};

constexpr int kBriefHalo = 18;  // |trunc(R(-theta) p)| <= 18 for every pattern point (max radius 18.38)

// ---------------------------------------------------------------------------------------------
// K6  brief.wgsl:20-68 for the fused pipeline, plus the compaction of the band segments into the final
// lists (orb.rs:159-164 `corners`, 195-199 `descriptors`).
// The literal blur is one value per row for every column below qa (88 % of the width, see k_front phase
// C), so for a keypoint with 18 <= x < qa - 18 the 37x37 patch is 37 values.  One wave64 per keypoint, no
// LDS: lane r holds the row constant of row y - 18 + r (0 outside the level, CRD-6), a sample is a
// cross-lane read (ds_bpermute) at the rotated point's row, four ballots give the eight u32 words
// (brief.wgsl:47,63,67).  The other 12 % of the keypoints (right end of the rows, left border) take
// the general path: per sample 0 / row constant / a 2-byte load from the plane's stored tail.
// Workgroup = one band slot of one frame; output index = seg_before[slot] + index in the segment, so the
// final lists are the band segments back to back; counts[frame] comes from k_slot_prefix.
// ---------------------------------------------------------------------------------------------
struct RowsGeom {
    uint32_t n_slots, seg_cap;
    uint32_t slot_base[kMaxLevels + 1];
    uint32_t flat_end[kMaxLevels];  // qa - 18: a keypoint with 18 <= x < flat_end samples only columns in [0, qa)
    uint32_t qa[kMaxLevels];        // columns [0, qa) of the level's blur are the row constants
    uint32_t split;                 // workgroups per band slot (> 1 for small batches: more waves in flight)
};

__global__ __launch_bounds__(256) void k_brief_rows(const uint16_t* __restrict__ blur,
                                                    const uint16_t* __restrict__ blur_rowc, Pyramid pyr, RowsGeom rg,
                                                    const uint32_t* __restrict__ seg_counts,
                                                    const uint32_t* __restrict__ seg_before,
                                                    const CornerData* __restrict__ segments,
                                                    CornerData* __restrict__ corners, uint32_t cap,
                                                    CornerDescriptor* __restrict__ descriptors, BriefTables tab) {
    const uint32_t slot = blockIdx.x / rg.split, frame = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u;
    // this wave takes keypoints wave, wave + stride, ... of the band segment
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x % rg.split) * 4u + (threadIdx.x >> 6));
    const uint32_t stride = 4u * rg.split;
    uint32_t lvl = 0;
    for (uint32_t m = 1; m < pyr.depth; m++)
        if (slot >= rg.slot_base[m]) lvl = m;
    const uint32_t fe = rg.flat_end[lvl];
    const int qa = (int)rg.qa[lvl];
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const size_t sidx = (size_t)frame * rg.n_slots + slot;
    const uint32_t n = min(seg_counts[sidx], rg.seg_cap);
    const uint32_t before = seg_before[sidx];
    if (wave >= n) return;
    const CornerData* seg = segments + sidx * rg.seg_cap;
    const uint16_t* rowc = blur_rowc + (size_t)frame * pyr.row_stride + pyr.row_off[lvl];
    const uint16_t* plane = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    CornerData* out_kp = corners + (size_t)frame * cap;
    uint32_t* out_desc = reinterpret_cast<uint32_t*>(descriptors + (size_t)frame * cap);
    // this lane's four tests (l, 64+l, 128+l, 192+l) of the pattern (packed int8 x 4)
    uint32_t pat[4];
#pragma unroll
    for (int e = 0; e < 4; e++) pat[e] = tab.pattern[64u * (uint32_t)e + lane];

    uint4 nxt = *reinterpret_cast<const uint4*>(&seg[wave]);
    for (uint32_t j = wave; j < n; j += stride) {
        const uint4 rec = nxt;  // x, y, angle, octave
        if (j + stride < n) nxt = *reinterpret_cast<const uint4*>(&seg[j + stride]);
        const uint32_t k = before + j;
        if (k >= cap) break;  // frame is full (indices only grow)
        const int gy = (int)rec.y - kBriefHalo + (int)lane;  // lanes 0..36 are the patch rows
        const uint32_t rowv = (gy >= 0 && gy < h && lane < 37u) ? (uint32_t)rowc[gy] : 0u;
        uint64_t bal[4];
        if (rec.x >= (uint32_t)kBriefHalo && rec.x < fe) {
            // ---- every sample column lies in [0, qa): only the rows of the rotated points matter
            int ra[4], rb[4];
            if (rec.z == 0u) {  // R = I (more than half of all keypoints, Q7)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    ra[e] = (int)(int8_t)((pat[e] >> 8) & 255u);
                    rb[e] = (int)(int8_t)(pat[e] >> 24);
                }
            } else {
                const uint32_t code = min(rec.z, (uint32_t)(ORB_ANGLE_STEPS - 1));
                const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;  // CRD-10 table
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float pax = (float)(int8_t)(pat[e] & 255u), pay = (float)(int8_t)((pat[e] >> 8) & 255u);
                    const float pbx = (float)(int8_t)((pat[e] >> 16) & 255u), pby = (float)(int8_t)(pat[e] >> 24);
                    // mat2x2f(ct,-st, st,ct) * p (column-major): (ct*x + st*y, -st*x + ct*y); rows only
                    const float a2 = nst * pax, a3 = ct * pay, b2 = nst * pbx, b3 = ct * pby;
                    const float ray = a2 + a3, rby = b2 + b3;
                    ra[e] = (int)ray;  // vec2i() truncates
                    rb[e] = (int)rby;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t va = (uint32_t)__shfl((int)rowv, ra[e] + kBriefHalo);
                const uint32_t vb = (uint32_t)__shfl((int)rowv, rb[e] + kBriefHalo);
                bal[e] = __ballot(va > vb);  // non-negative f16: bit patterns order like the values (brief.wgsl:62)
            }
        } else {
            // ---- general case (12 % of the keypoints: near the right end of the row, or within 18 px of the left
            //      border): a sample is 0 outside the level (CRD-6), the row constant for columns < qa, and a
            //      2-byte load from the plane's stored tail otherwise
            const uint32_t code = min(rec.z, (uint32_t)(ORB_ANGLE_STEPS - 1));
            const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float pax = (float)(int8_t)(pat[e] & 255u), pay = (float)(int8_t)((pat[e] >> 8) & 255u);
                const float pbx = (float)(int8_t)((pat[e] >> 16) & 255u), pby = (float)(int8_t)(pat[e] >> 24);
                const float a0 = ct * pax, a1 = st * pay, a2 = nst * pax, a3 = ct * pay;
                const float b0 = ct * pbx, b1 = st * pby, b2 = nst * pbx, b3 = ct * pby;
                const float rax = a0 + a1, ray = a2 + a3, rbx = b0 + b1, rby = b2 + b3;
                const int dya = (int)ray, dyb = (int)rby;
                const int xa = (int)rec.x + (int)rax, ya = (int)rec.y + dya;
                const int xb = (int)rec.x + (int)rbx, yb = (int)rec.y + dyb;
                uint32_t va = (uint32_t)__shfl((int)rowv, dya + kBriefHalo);  // 0 when the row is outside the level
                uint32_t vb = (uint32_t)__shfl((int)rowv, dyb + kBriefHalo);
                const bool ina = xa >= 0 && xa < w && ya >= 0 && ya < h;
                const bool inb = xb >= 0 && xb < w && yb >= 0 && yb < h;
                if (!ina)
                    va = 0u;
                else if (xa >= qa)
                    va = plane[(size_t)(uint32_t)(__mul24(ya, w) + xa)];
                if (!inb)
                    vb = 0u;
                else if (xb >= qa)
                    vb = plane[(size_t)(uint32_t)(__mul24(yb, w) + xb)];
                bal[e] = __ballot(va > vb);
            }
        }
        if (lane < 8u) {
            const uint64_t src = lane < 2u ? bal[0] : (lane < 4u ? bal[1] : (lane < 6u ? bal[2] : bal[3]));
            out_desc[(size_t)k * 8u + lane] = (uint32_t)(src >> ((lane & 1u) * 32u));
        } else if (lane == 8u) {
            *reinterpret_cast<uint4*>(&out_kp[k]) = rec;
        }
    }
}

// Exclusive prefix of the stored keypoints over a frame's band slots (= where each band's keypoints start in the final
// lists) and the frame's raw counter (orb.rs:550-556).  One wave per frame.  With n_classes == 2 a band has two lists
// (seg_counts[slot][class]); the final list is all first lists in slot order, then all second lists:
// seg_before[class * n_slots + slot].
__global__ __launch_bounds__(64) void k_slot_prefix(const uint32_t* __restrict__ seg_counts,
                                                    uint32_t* __restrict__ seg_before, uint32_t* __restrict__ counts,
                                                    uint32_t n_slots, uint32_t seg_cap, uint32_t n_classes) {
    const uint32_t frame = blockIdx.x, lane = threadIdx.x;
    const uint32_t* sc = seg_counts + (size_t)frame * n_slots * n_classes;
    uint32_t* sb = seg_before + (size_t)frame * n_slots * n_classes;
    uint32_t carry = 0, total = 0;
    for (uint32_t cls = 0; cls < n_classes; cls++) {
        for (uint32_t s0 = 0; s0 < n_slots; s0 += 64u) {
            const uint32_t s = s0 + lane;
            const uint32_t raw = s < n_slots ? sc[s * n_classes + cls] : 0u;
            const uint32_t stored = min(raw, seg_cap);
            uint32_t incl = stored;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d);
                if ((int)lane >= d) incl += t;
            }
            if (s < n_slots) sb[cls * n_slots + s] = carry + incl - stored;
            carry += __shfl(incl, 63);
            uint32_t r = raw;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) r += __shfl_xor(r, sh);
            total += r;
        }
    }
    if (lane == 0u) counts[frame] = total;
}

}  // namespace orb
