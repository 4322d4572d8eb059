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

constexpr int kFrontThreads = 512;
constexpr int kFrontRows = 16;       // R: band height (even)
constexpr int kFrontTmpRows = 2;     // rows per blur chunk (double buffered)
constexpr int kFrontQueue = 2048;    // candidate queue entries
constexpr int kFrontMaxCols = 4;     // blur columns per thread -> level width <= 4 * kFrontThreads
constexpr int kLdsPad = 8;           // halfs of padding left of column 0

struct FrontGeom {
    uint32_t lvl;       // pyramid level handled by this launch
    uint32_t gw, gh;    // FAST dispatch domain of this octave (8-rounded, orb.rs:511-515)
    uint32_t n_bands;   // ceil(max(h, gh) / R)
    uint32_t n_frames;
    uint32_t ls;        // LDS row stride of the grey rows, in halfs (multiple of 8)
    uint32_t ts;        // LDS row stride of the blur intermediate, in halfs
    uint32_t write_mip; // 1: level lvl+1 exists and is an exact 2x2 reduction
    uint32_t xcd_swizzle;
    uint32_t phase_mask;  // debug: bit0 B1, bit1 B2, bit2 C0, bit3 C (timing experiments only)
    uint32_t slot_base;   // index of this level's band 0 among the frame's band slots
    uint32_t n_slots;     // band slots per frame (all levels)
    uint32_t seg_cap;     // CornerData records per band segment
};

__host__ __device__ inline uint32_t front_lds_bytes(const FrontGeom& g) {
    return ((kFrontRows + 6) * g.ls + 2 * kFrontTmpRows * g.ts) * 2u + (kFrontQueue + 4) * 4u;  // queue + q_count + c_count
}

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ushort2_t as_u16x2(uint32_t v) { return __builtin_bit_cast(ushort2_t, v); }

__device__ __forceinline__ float h2f(uint32_t packed, int hi) {
    return from_half(bits_half((uint16_t)(hi ? (packed >> 16) : (packed & 0xffffu))));
}

// Exact byte/255 (CRD-1) without the divide sequence: one Newton correction with true FMAs gives
// the correctly rounded quotient for all 256 inputs (checked exhaustively on host and device).
__device__ __forceinline__ float unorm8_exact(float b) {
    const float rc = 1.0f / 255.0f;
    float q = b * rc;
    float r = __builtin_fmaf(-q, 255.0f, b);
    return __builtin_fmaf(r, rc, q);
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

// Full FAST test of one pre-test survivor; `ctr` points at the pixel inside the LDS grey rows.
__device__ __forceinline__ bool fast_full_test(const half_t* ctr, int ls, float thr, uint32_t* angle) {
    const float c = from_half(ctr[0]);
    uint32_t m_over = 0, m_under = 0;
    float cx = 0.0f, cy = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = from_half(ctr[kRingDy[i] * ls + kRingDx[i]]);
        const float diff = v - c;
        const float px = v * (float)kRingDx[i];
        const float py = v * (float)kRingDy[i];
        cx = cx + px;  // CRD-8: ring order, unfused
        cy = cy + py;
        // thr >= 0, so `diff > thr` and `diff < -thr` exclude each other (fast.wgsl:108-112's else-if)
        m_over |= (diff > thr) ? (1u << i) : 0u;
        m_under |= (diff < -thr) ? (1u << i) : 0u;
    }
    if ((detect_streak_16(m_over) | detect_streak_16(m_under)) == 0u) return false;
    *angle = angle_code(cy, cx);
    return true;
}

// Block-local stream compaction: a band's corners go to its own segment of the scratch list, the
// slot comes from an LDS counter.  No global atomic is involved: 180 waves per frame bumping one
// per-frame counter serialise at the memory side and cost more than the rest of the kernel.
__device__ __forceinline__ void segment_append(bool is_corner, uint32_t x, uint32_t y, uint32_t angle, uint32_t oct,
                                               uint32_t* lds_counter, CornerData* seg, uint32_t seg_cap) {
    if (is_corner) {
        const uint32_t idx = atomicAdd(lds_counter, 1u);  // hipcc turns this into one ds_add per wave
        if (idx < seg_cap) *reinterpret_cast<uint4*>(&seg[idx]) = make_uint4(x, y, angle, oct);
    }
}

template <bool L0>
__global__ __launch_bounds__(kFrontThreads) void k_front(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                         uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                         Pyramid pyr, FrontGeom geo, float thr,
                                                         uint32_t* __restrict__ seg_counts,
                                                         CornerData* __restrict__ segments) {
    constexpr int NT = kFrontThreads, R = kFrontRows, TC = kFrontTmpRows;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int LS = (int)geo.ls, TS = (int)geo.ts;
    half_t* const grey = reinterpret_cast<half_t*>(lds_raw);             // (R+6) rows x LS
    half_t* const tmp = grey + (R + 6) * LS;                              // 2 x TC rows x TS
    uint32_t* const queue = reinterpret_cast<uint32_t*>(tmp + 2 * TC * TS);
    uint32_t* const q_count = queue + kFrontQueue;
    uint32_t* const c_count = q_count + 1;  // corners found by this band

    // ---- which band of which frame: keep all bands of a frame on one XCD so halo rows hit its L2
    uint32_t frame, band;
    {
        const uint32_t L = blockIdx.x;
        if (geo.xcd_swizzle) {
            const uint32_t xcd = L & 7u, slot = L >> 3;
            frame = (slot / geo.n_bands) * 8u + xcd;
            band = slot % geo.n_bands;
        } else {
            frame = L / geo.n_bands;
            band = L % geo.n_bands;
        }
    }
    const uint32_t lvl = geo.lvl;
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const int y0 = (int)band * R;
    const int tid = (int)threadIdx.x;
    uint16_t* const gray_f = gray + (size_t)frame * pyr.stride;
    uint16_t* const blur_lvl = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    const size_t slot = (size_t)frame * geo.n_slots + geo.slot_base + band;
    CornerData* const seg = segments + slot * geo.seg_cap;

    if (tid == 0) {
        *q_count = 0u;
        *c_count = 0u;
    }

    // =========================== A: stage grey rows [y0-3, y0+R+3) ===========================
    if (L0) {
        // RGBA8 -> luminance of the vertically mirrored row (grayscale.wgsl:16-38), 4 px per item.
        // Loads are issued kLoadBatch at a time before any is consumed, so a thread has that many
        // 16-byte HBM requests in flight instead of one.
        constexpr int U = 8;
        const int w4 = w >> 2;
        const float inv_w4 = 1.0f / (float)w4;
        const uint8_t* src = frames + (size_t)frame * frame_bytes;
        const int n_items = (R + 6) * w4;
        for (int ib = tid; ib < n_items; ib += NT * U) {
            uint4 px[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = ib + u * NT;
                const int ly = (int)(((float)i + 0.5f) * inv_w4);
                const int xi = i - ly * w4;
                const int gy = y0 - 3 + ly;
                // rows outside the image are never read by a pixel that passes the guard (fast.wgsl:77)
                const bool ok = i < n_items && gy >= 0 && gy < h;
                dst[u] = ok ? ly * LS + kLdsPad + xi * 4 : -1;
                px[u] = make_uint4(0u, 0u, 0u, 0u);
                if (ok) px[u] = *reinterpret_cast<const uint4*>(src + ((size_t)(h - 1 - gy) * w + (size_t)xi * 4) * 4);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (dst[u] >= 0) {
                    uint2 out;
                    out.x = pack_half2(luminance_fast(px[u].x), luminance_fast(px[u].y));
                    out.y = pack_half2(luminance_fast(px[u].z), luminance_fast(px[u].w));
                    *reinterpret_cast<uint2*>(grey + dst[u]) = out;
                }
            }
        }
    } else {
        // f16 mip from HBM; texels outside the level read as 0 (CRD-6) because at octaves >= 1 the
        // reference's guard and dispatch size let pixels near/over the level edge through (Q8).
        constexpr int U = 4;
        const uint16_t* src = gray_f + pyr.off[lvl];
        const int w8 = (LS - kLdsPad) >> 3;  // 8-texel groups per LDS row
        const float inv_w8 = 1.0f / (float)w8;
        const int n_items = (R + 6) * w8;
        const bool vec_ok = (w & 7) == 0;
        for (int ib = tid; ib < n_items; ib += NT * U) {
            uint4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = ib + u * NT;
                const int ly = (int)(((float)i + 0.5f) * inv_w8);
                const int xg = i - ly * w8;
                const int gy = y0 - 3 + ly;
                const int x = xg * 8;
                dst[u] = i < n_items ? ly * LS + kLdsPad + x : -1;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < n_items && gy >= 0 && gy < h) {
                    const uint16_t* row = src + (size_t)gy * w;
                    if (vec_ok && x + 8 <= w) {
                        v[u] = *reinterpret_cast<const uint4*>(row + x);
                    } else {
                        uint32_t e[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) e[k] = (x + k < w) ? (uint32_t)row[x + k] : 0u;
                        v[u] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(grey + dst[u]) = v[u];
        }
    }
    __syncthreads();

    // =========================== B1: 4-point pre-test, 8 px per item ===========================
    if (geo.phase_mask & 1u) {
        const int g8 = (int)geo.gw >> 3;
        const float inv_g8 = 1.0f / (float)g8;
        const int n_items = R * g8;
        // fast.wgsl:77 -- level-0 dimensions for every octave, u32 arithmetic (Q8)
        const uint32_t lim_x = pyr.w[0] - 16u, lim_y = pyr.h[0] - 16u;
        for (int i = tid; i < n_items; i += NT) {
            const int lyc = (int)(((float)i + 0.5f) * inv_g8);
            const int x = (i - lyc * g8) * 8;
            const uint32_t gy = (uint32_t)(y0 + lyc);
            if (!(gy < geo.gh && gy > 16u && gy < lim_y)) continue;
            if ((uint32_t)x + 7u <= 16u || (uint32_t)x >= lim_x) continue;
            const half_t* rowc = grey + (lyc + 3) * LS + kLdsPad + x;
            const uint2 qa = *reinterpret_cast<const uint2*>(rowc - 4);
            const uint4 qb = *reinterpret_cast<const uint4*>(rowc);
            const uint2 qc = *reinterpret_cast<const uint2*>(rowc + 8);
            const uint4 qu = *reinterpret_cast<const uint4*>(rowc - 3 * LS);
            const uint4 qd = *reinterpret_cast<const uint4*>(rowc + 3 * LS);
            // "At least 3 of the 4 compass diffs exceed thr" (fast.wgsl:85-95) <=> the 2nd smallest of the four
            // neighbour values, minus the centre, exceeds thr (v -> fl(v - c) is monotone); likewise the
            // 2nd largest for the "under" case.  Grey values are non-negative f16, so their bit patterns
            // order like the values and the selection network runs on packed u16 pairs (2 pixels per op).
            const uint32_t dw[8] = {qa.x, qa.y, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y};  // dw[j] = grey(x-4+2j, x-3+2j)
            const uint32_t upw[4] = {qu.x, qu.y, qu.z, qu.w}, dnw[4] = {qd.x, qd.y, qd.z, qd.w};
            uint32_t cand = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {  // pixel pair (x+2j, x+2j+1)
                const ushort2_t left = as_u16x2(__builtin_amdgcn_alignbit(dw[j + 1], dw[j], 16));       // x+2j-3, x+2j-2
                const ushort2_t right = as_u16x2(__builtin_amdgcn_alignbit(dw[j + 4], dw[j + 3], 16));  // x+2j+3, x+2j+4
                const ushort2_t upp = as_u16x2(upw[j]), dwn = as_u16x2(dnw[j]);
                const ushort2_t lo1 = __builtin_elementwise_min(left, right), hi1 = __builtin_elementwise_max(left, right);
                const ushort2_t lo2 = __builtin_elementwise_min(upp, dwn), hi2 = __builtin_elementwise_max(upp, dwn);
                const ushort2_t m1 = __builtin_elementwise_max(lo1, lo2), m2 = __builtin_elementwise_min(hi1, hi2);
                const ushort2_t second_lo = __builtin_elementwise_min(m1, m2), second_hi = __builtin_elementwise_max(m1, m2);
                const uint32_t cw = dw[j + 2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float c = h2f(cw, e);
                    const float d_lo = from_half(bits_half(second_lo[e])) - c;  // CRD-7: one f32 subtraction
                    const float d_hi = from_half(bits_half(second_hi[e])) - c;
                    const uint32_t gx = (uint32_t)(x + 2 * j + e);
                    const uint32_t hit = (uint32_t)(d_lo > thr) | (uint32_t)(d_hi < -thr);  // branch-free on purpose
                    const uint32_t ok = (uint32_t)(gx > 16u) & (uint32_t)(gx < lim_x);
                    cand |= (hit & ok) << (2 * j + e);
                }
            }
            while (cand) {
                const int k = __builtin_ctz(cand);
                cand &= cand - 1u;
                const uint32_t slot = atomicAdd(q_count, 1u);
                if (slot < (uint32_t)kFrontQueue) {
                    queue[slot] = ((uint32_t)lyc << 16) | (uint32_t)(x + k);
                } else {  // queue full (pathological frame): test in place
                    uint32_t angle;
                    const bool hit = fast_full_test(rowc + k, LS, thr, &angle);
                    segment_append(hit, (uint32_t)(x + k), gy, angle, lvl, c_count, seg, geo.seg_cap);
                }
            }
        }
    }
    __syncthreads();

    // =========================== B2: drain the candidate queue densely ===========================
    if (geo.phase_mask & 2u) {
        const uint32_t n_q = min(*q_count, (uint32_t)kFrontQueue);
        for (uint32_t base = (uint32_t)(tid & ~63); base < n_q; base += NT) {  // wave-uniform trip count
            const uint32_t i = base + (uint32_t)(tid & 63);
            bool is_corner = false;
            uint32_t angle = 0, x = 0, gy = 0;
            if (i < n_q) {
                const uint32_t e = queue[i];
                const int lyc = (int)(e >> 16);
                x = e & 0xffffu;
                gy = (uint32_t)(y0 + lyc);
                is_corner = fast_full_test(grey + (lyc + 3) * LS + kLdsPad + (int)x, LS, thr, &angle);
            }
            segment_append(is_corner, x, gy, angle, lvl, c_count, seg, geo.seg_cap);
        }
    }

    __syncthreads();
    if (tid == 0) seg_counts[slot] = *c_count;  // raw count of the band (may exceed seg_cap)

    // =========================== C0: next mip level (blit.wgsl, exact 2x2 case) ===========================
    if (geo.write_mip && (geo.phase_mask & 4u)) {
        const int wd = (int)pyr.w[lvl + 1], hd = (int)pyr.h[lvl + 1];
        uint16_t* dst = gray_f + pyr.off[lvl + 1];
        const int g4 = (wd + 3) >> 2;
        const float inv_g4 = 1.0f / (float)g4;
        const int n_items = (R / 2) * g4;
        const bool vec_ok = (wd & 3) == 0;
        for (int i = tid; i < n_items; i += NT) {
            const int r = (int)(((float)i + 0.5f) * inv_g4);
            const int xd = (i - r * g4) * 4;
            const int yd = (y0 >> 1) + r;
            if (yd >= hd) continue;
            const half_t* top = grey + (2 * r + 3) * LS + kLdsPad + 2 * xd;
            const uint4 qt = *reinterpret_cast<const uint4*>(top);
            const uint4 qb = *reinterpret_cast<const uint4*>(top + LS);
            const uint32_t tw[4] = {qt.x, qt.y, qt.z, qt.w}, bw[4] = {qb.x, qb.y, qb.z, qb.w};
            uint16_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float a = h2f(tw[k], 0), b = h2f(tw[k], 1), c = h2f(bw[k], 0), d = h2f(bw[k], 1);
                const float st = a + b;
                const float sb = c + d;
                o[k] = half_bits(to_half((st + sb) * 0.25f));
            }
            uint16_t* out = dst + (size_t)yd * wd + xd;
            if (vec_ok) {
                *reinterpret_cast<uint2*>(out) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (xd + k < wd) out[k] = o[k];
            }
        }
    }

    // =========================== C: literal blur, both passes ===========================
    if (geo.phase_mask & 8u) {
        const int rows = min(R, h - y0);  // band rows that exist in this level
        if (rows > 0) {                   // uniform per block
            // tap 1 (offset -0.4392 in UV units) per column, kept in registers for every row
            int i0[kFrontMaxCols], i1[kFrontMaxCols];
            float fr[kFrontMaxCols];
#pragma unroll
            for (int c = 0; c < kFrontMaxCols; c++) {
                const int x = tid + c * NT;
                BlurTap t = blur_tap((uint32_t)(x < w ? x : 0), (uint32_t)w, kBlurOff[1]);
                i0[c] = t.i0;
                i1[c] = t.i1;
                fr[c] = t.f;
            }
            const int n_chunks = (rows + TC - 1) / TC;
            auto pass1 = [&](int chunk) {
                half_t* dstbuf = tmp + (chunk & 1) * TC * TS;
                for (int rr = 0; rr < TC; rr++) {
                    const int r = chunk * TC + rr;
                    if (r >= rows) break;
                    const half_t* row = grey + (r + 3) * LS + kLdsPad;
                    const float t0 = from_half(row[0]), tl = from_half(row[w - 1]);
                    const float a0 = t0 * kBlurWgt[0], a2 = tl * kBlurWgt[2], a3 = tl * kBlurWgt[3];
#pragma unroll
                    for (int c = 0; c < kFrontMaxCols; c++) {
                        const int x = tid + c * NT;
                        if (x < w) {
                            const float v0 = from_half(row[i0[c]]), v1 = from_half(row[i1[c]]);
                            const float d = v1 - v0;
                            const float s = v0 + fr[c] * d;
                            const float ws = s * kBlurWgt[1];
                            float acc = 0.0f + a0;
                            acc = acc + ws;
                            acc = acc + a2;
                            acc = acc + a3;
                            dstbuf[rr * TS + x] = to_half(acc);
                        }
                    }
                }
            };
            auto pass2 = [&](int chunk) {
                const half_t* srcbuf = tmp + (chunk & 1) * TC * TS;
                for (int rr = 0; rr < TC; rr++) {
                    const int r = chunk * TC + rr;
                    if (r >= rows) break;
                    const half_t* row = srcbuf + rr * TS;
                    const float t0 = from_half(row[0]), tl = from_half(row[w - 1]);
                    const float a0 = t0 * kBlurWgt[0], a2 = tl * kBlurWgt[2], a3 = tl * kBlurWgt[3];
                    uint16_t* out = blur_lvl + (size_t)(y0 + r) * w;
#pragma unroll
                    for (int c = 0; c < kFrontMaxCols; c++) {
                        const int x = tid + c * NT;
                        if (x < w) {
                            const float v0 = from_half(row[i0[c]]), v1 = from_half(row[i1[c]]);
                            const float d = v1 - v0;
                            const float s = v0 + fr[c] * d;
                            const float ws = s * kBlurWgt[1];
                            float acc = 0.0f + a0;
                            acc = acc + ws;
                            acc = acc + a2;
                            acc = acc + a3;
                            out[x] = half_bits(to_half(acc));
                        }
                    }
                }
            };
            pass1(0);
            for (int k = 0; k < n_chunks; k++) {
                __syncthreads();
                if (k + 1 < n_chunks) pass1(k + 1);
                pass2(k);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6 for the fused pipeline: brief.wgsl:20-68 over the band segments written by k_front.
// One workgroup = one band slot of one frame, so the keypoints a CU works on share a window of
// R + 36 blur rows (L1/L2 hits instead of one HBM line per sample), and the final compact lists
// (orb.rs:159-164 `corners`, 195-199 `descriptors`) are produced here: slot s starts at the sum of
// the stored counts of the slots before it.  One wave64 per keypoint, lane l owns tests l, 64+l,
// 128+l, 192+l; four ballots give the eight u32 words (brief.wgsl:47,63,67).
// ---------------------------------------------------------------------------------------------
struct BandGeom {
    uint32_t n_slots, seg_cap, n_frames, xcd_swizzle;
    uint32_t slot_base[kMaxLevels + 1];  // first slot of each level; [depth] = n_slots
};

__global__ __launch_bounds__(256) void k_brief_bands(const uint16_t* __restrict__ blur, Pyramid pyr, BandGeom bg,
                                                     const uint32_t* __restrict__ seg_counts,
                                                     const CornerData* __restrict__ segments,
                                                     uint32_t* __restrict__ counts, CornerData* __restrict__ corners,
                                                     uint32_t cap, CornerDescriptor* __restrict__ descriptors,
                                                     BriefTables tab) {
    uint32_t frame, slot;
    {
        const uint32_t L = blockIdx.x;
        if (bg.xcd_swizzle) {
            const uint32_t xcd = L & 7u, q = L >> 3;
            frame = (q / bg.n_slots) * 8u + xcd;
            slot = q % bg.n_slots;
        } else {
            frame = L / bg.n_slots;
            slot = L % bg.n_slots;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t* sc = seg_counts + (size_t)frame * bg.n_slots;

    // stored keypoints in the slots before this one, and the frame's raw total (orb.rs:550-556)
    uint32_t before = 0, total = 0;
    for (uint32_t s0 = 0; s0 < bg.n_slots; s0 += 64u) {
        const uint32_t s = s0 + lane;
        const uint32_t raw = s < bg.n_slots ? sc[s] : 0u;
        total += raw;
        before += s < slot ? min(raw, bg.seg_cap) : 0u;
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        before += __shfl_xor(before, sh);
        total += __shfl_xor(total, sh);
    }
    if (slot == 0u && threadIdx.x == 0u) counts[frame] = total;

    uint32_t lvl = 0;
    for (uint32_t m = 1; m < pyr.depth; m++)
        if (slot >= bg.slot_base[m]) lvl = m;
    const uint32_t w = pyr.w[lvl], h = pyr.h[lvl];
    const uint16_t* plane = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    const uint32_t n_here = min(sc[slot], bg.seg_cap);
    const CornerData* seg = segments + ((size_t)frame * bg.n_slots + slot) * bg.seg_cap;
    CornerData* out_kp = corners + (size_t)frame * cap;
    uint32_t* out_desc = reinterpret_cast<uint32_t*>(descriptors + (size_t)frame * cap);

    const uint32_t p0 = tab.pattern[lane], p1 = tab.pattern[64u + lane], p2 = tab.pattern[128u + lane],
                   p3 = tab.pattern[192u + lane];
    uint4 next = make_uint4(0u, 0u, 0u, 0u);
    if (wave < n_here) next = *reinterpret_cast<const uint4*>(&seg[wave]);
    for (uint32_t j = wave; j < n_here; j += 4u) {
        const uint32_t k = before + j;
        if (k >= cap) break;  // frame is full (wave-uniform)
        const uint4 rec = next;  // x, y, angle, octave
        if (j + 4u < n_here) next = *reinterpret_cast<const uint4*>(&seg[j + 4u]);
        const uint32_t code = min(rec.z, (uint32_t)(ORB_ANGLE_STEPS - 1));
        const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;
        const int px = (int)rec.x, py = (int)rec.y;
        const uint64_t b0 = __ballot(brief_test(p0, ct, st, nst, px, py, plane, w, h));
        const uint64_t b1 = __ballot(brief_test(p1, ct, st, nst, px, py, plane, w, h));
        const uint64_t b2 = __ballot(brief_test(p2, ct, st, nst, px, py, plane, w, h));
        const uint64_t b3 = __ballot(brief_test(p3, ct, st, nst, px, py, plane, w, h));
        if (lane < 8u) {
            const uint64_t src = lane < 2u ? b0 : (lane < 4u ? b1 : (lane < 6u ? b2 : b3));
            out_desc[(size_t)k * 8u + lane] = (uint32_t)(src >> ((lane & 1u) * 32u));
        } else if (lane == 8u) {
            *reinterpret_cast<uint4*>(&out_kp[k]) = rec;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6, tiled: brief.wgsl:20-68 with the sampling window staged in LDS.
//
// A 64-lane gather of 2-byte texels through the vector memory path costs about one address per
// cycle, and BRIEF needs 512 of them per keypoint; the same gather from LDS runs at bank speed.
// So a workgroup owns a tile of kBriefTileH x kBriefTileW keypoint positions of one level of one
// frame (two k_front bands high), copies the blurred window tile + 18 px halo into LDS once with
// 16-byte loads (texels outside the level become 0, CRD-6, so the sampling loop needs no bounds
// checks), picks the tile's keypoints out of the two band segments, and then runs one wave64 per
// keypoint.  |trunc(R(-theta) p)| <= 18 for every pattern point (max radius 18.38).
// Output index of a keypoint = stored keypoints in earlier band slots + its index in its band
// segment, i.e. the final lists are the band segments back to back.
// ---------------------------------------------------------------------------------------------
constexpr int kBriefTileH = 2 * kFrontRows;  // 32 rows = two band slots
constexpr int kBriefTileW = 256;
constexpr int kBriefHalo = 18;
constexpr int kBriefPadX = 24;  // halo rounded up to a multiple of 8 texels (16-byte loads)
constexpr int kBriefWinW = kBriefTileW + 2 * kBriefPadX;  // 304
constexpr int kBriefWinH = kBriefTileH + 2 * kBriefHalo;  // 68
constexpr int kBriefList = 256;

struct TileGeom {
    uint32_t n_slots, seg_cap, n_frames, xcd_swizzle;
    uint32_t slot_base[kMaxLevels + 1];
    uint32_t tile_base[kMaxLevels + 1];  // first tile of each level; [depth] = tiles per frame
    uint32_t tile_cols[kMaxLevels];      // column tiles of each level
};

__global__ __launch_bounds__(256) void k_brief_tiles(const uint16_t* __restrict__ blur, Pyramid pyr, TileGeom tg,
                                                     const uint32_t* __restrict__ seg_counts,
                                                     const CornerData* __restrict__ segments,
                                                     uint32_t* __restrict__ counts, CornerData* __restrict__ corners,
                                                     uint32_t cap, CornerDescriptor* __restrict__ descriptors,
                                                     BriefTables tab) {
    __shared__ __attribute__((aligned(16))) uint16_t win[kBriefWinH * kBriefWinW];
    __shared__ uint4 list_rec[kBriefList];   // x, y, angle, octave
    __shared__ uint32_t list_k[kBriefList];  // output index
    __shared__ uint32_t list_n;

    const uint32_t tiles_per_frame = tg.tile_base[pyr.depth];
    uint32_t frame, tile;
    {
        const uint32_t L = blockIdx.x;
        if (tg.xcd_swizzle) {
            const uint32_t xcd = L & 7u, q = L >> 3;
            frame = (q / tiles_per_frame) * 8u + xcd;
            tile = q % tiles_per_frame;
        } else {
            frame = L / tiles_per_frame;
            tile = L % tiles_per_frame;
        }
    }
    uint32_t lvl = 0;
    for (uint32_t m = 1; m < pyr.depth; m++)
        if (tile >= tg.tile_base[m]) lvl = m;
    const uint32_t t_in = tile - tg.tile_base[lvl];
    const uint32_t ty = t_in / tg.tile_cols[lvl], tx = t_in % tg.tile_cols[lvl];
    const uint32_t n_bands = tg.slot_base[lvl + 1] - tg.slot_base[lvl];
    const uint32_t slot_a = tg.slot_base[lvl] + 2u * ty;
    const bool has_b = 2u * ty + 1u < n_bands;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t* sc = seg_counts + (size_t)frame * tg.n_slots;

    // ---- stored keypoints before slot_a, and the frame's raw total (orb.rs:550-556)
    uint32_t before = 0, total = 0;
    for (uint32_t s0 = 0; s0 < tg.n_slots; s0 += 64u) {
        const uint32_t s = s0 + lane;
        const uint32_t raw = s < tg.n_slots ? sc[s] : 0u;
        total += raw;
        before += s < slot_a ? min(raw, tg.seg_cap) : 0u;
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        before += __shfl_xor(before, sh);
        total += __shfl_xor(total, sh);
    }
    if (tile == 0u && tid == 0u) counts[frame] = total;
    const uint32_t n_a = min(sc[slot_a], tg.seg_cap);
    const uint32_t n_b = has_b ? min(sc[slot_a + 1u], tg.seg_cap) : 0u;
    if (tid == 0u) list_n = 0u;
    if (n_a + n_b == 0u) return;  // uniform: nothing detected in these two bands

    // ---- stage the window: rows [y0-18, y0+32+18), columns [x0-24, x0+256+24), zero outside the level
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const int x0 = (int)tx * kBriefTileW, y0 = (int)ty * kBriefTileH;
    const uint16_t* plane = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    {
        constexpr int G = kBriefWinW / 8;  // 16-byte groups per window row
        constexpr int N = kBriefWinH * G;
        constexpr int U = 6;
        const bool vec_ok = (w & 7) == 0;
        for (int ib = (int)tid; ib < N; ib += 256 * U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = ib + u * 256;
                const int r = i / G, g = i - r * G;
                const int gy = y0 - kBriefHalo + r, gx = x0 - kBriefPadX + g * 8;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < N && gy >= 0 && gy < h && gx + 8 > 0 && gx < w) {
                    const uint16_t* row = plane + (size_t)gy * w;
                    if (vec_ok && gx >= 0 && gx + 8 <= w) {
                        v[u] = *reinterpret_cast<const uint4*>(row + gx);
                    } else {
                        uint32_t e[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) e[k] = (gx + k >= 0 && gx + k < w) ? (uint32_t)row[gx + k] : 0u;
                        v[u] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = ib + u * 256;
                if (i < N) *reinterpret_cast<uint4*>(&win[i * 8]) = v[u];
            }
        }
    }

    const CornerData* seg_a = segments + ((size_t)frame * tg.n_slots + slot_a) * tg.seg_cap;
    const CornerData* seg_b = seg_a + tg.seg_cap;
    CornerData* out_kp = corners + (size_t)frame * cap;
    uint32_t* out_desc = reinterpret_cast<uint32_t*>(descriptors + (size_t)frame * cap);
    const uint32_t p0 = tab.pattern[lane], p1 = tab.pattern[64u + lane], p2 = tab.pattern[128u + lane],
                   p3 = tab.pattern[192u + lane];
    const uint32_t pats[4] = {p0, p1, p2, p3};

    // ---- rounds of up to 256 segment records: pick this tile's keypoints, then one wave per keypoint
    const uint32_t n_ab = n_a + n_b;
    for (uint32_t c0 = 0; c0 < n_ab; c0 += 256u) {
        __syncthreads();  // window staged / previous round drained
        const uint32_t j = c0 + tid;
        if (j < n_ab) {
            const uint4 rec = *reinterpret_cast<const uint4*>(j < n_a ? &seg_a[j] : &seg_b[j - n_a]);
            const uint32_t k = before + j;  // segments back to back
            if (k < cap && rec.x >= (uint32_t)x0 && rec.x < (uint32_t)(x0 + kBriefTileW)) {
                const uint32_t idx = atomicAdd(&list_n, 1u);
                list_rec[idx] = rec;
                list_k[idx] = k;
            }
        }
        __syncthreads();
        const uint32_t n_l = list_n;
        for (uint32_t i = wave; i < n_l; i += 4u) {
            const uint4 rec = list_rec[i];
            const uint32_t k = list_k[i];
            const uint32_t code = min(rec.z, (uint32_t)(ORB_ANGLE_STEPS - 1));
            const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;
            // keypoint position inside the window
            const int cx = (int)rec.x - x0 + kBriefPadX, cy = (int)rec.y - y0 + kBriefHalo;
            const uint16_t* ctr = win + cy * kBriefWinW + cx;
            uint64_t bal[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t packed = pats[t];
                const float ax = (float)(int8_t)(packed & 255u), ay = (float)(int8_t)((packed >> 8) & 255u);
                const float bx = (float)(int8_t)((packed >> 16) & 255u), by = (float)(int8_t)(packed >> 24);
                // mat2x2f(ct,-st, st,ct) * p (column-major): (ct*x + st*y, -st*x + ct*y)   brief.wgsl:38-54
                const float a0 = ct * ax, a1 = st * ay, a2 = nst * ax, a3 = ct * ay;
                const float b0 = ct * bx, b1 = st * by, b2 = nst * bx, b3 = ct * by;
                const float rax = a0 + a1, ray = a2 + a3, rbx = b0 + b1, rby = b2 + b3;
                const float va = from_half(bits_half(ctr[(int)ray * kBriefWinW + (int)rax]));  // vec2i() truncates
                const float vb = from_half(bits_half(ctr[(int)rby * kBriefWinW + (int)rbx]));
                bal[t] = __ballot(va > vb);  // brief.wgsl:62
            }
            if (lane < 8u) {
                const uint64_t src = lane < 2u ? bal[0] : (lane < 4u ? bal[1] : (lane < 6u ? bal[2] : bal[3]));
                out_desc[(size_t)k * 8u + lane] = (uint32_t)(src >> ((lane & 1u) * 32u));
            } else if (lane == 8u) {
                *reinterpret_cast<uint4*>(&out_kp[k]) = rec;
            }
        }
        __syncthreads();
        if (tid == 0u) list_n = 0u;
    }
}

}  // namespace orb
