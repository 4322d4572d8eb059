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
};

__host__ __device__ inline uint32_t front_lds_bytes(const FrontGeom& g) {
    return ((kFrontRows + 6) * g.ls + 2 * kFrontTmpRows * g.ts) * 2u + (kFrontQueue + 4) * 4u;
}

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
        if (diff > thr)
            m_over |= 1u << i;
        else if (diff < -thr)
            m_under |= 1u << i;
    }
    if ((detect_streak_16(m_over) | detect_streak_16(m_under)) == 0u) return false;
    *angle = angle_code(cy, cx);
    return true;
}

__device__ __forceinline__ void append_one(uint32_t x, uint32_t y, uint32_t angle, uint32_t oct, uint32_t* counter,
                                           CornerData* out, uint32_t cap) {
    const uint32_t idx = atomicAdd(counter, 1u);
    if (idx < cap) *reinterpret_cast<uint4*>(&out[idx]) = make_uint4(x, y, angle, oct);
}

template <bool L0>
__global__ __launch_bounds__(kFrontThreads) void k_front(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                         uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                         Pyramid pyr, FrontGeom geo, float thr,
                                                         uint32_t* __restrict__ counts,
                                                         CornerData* __restrict__ corners, uint32_t cap) {
    constexpr int NT = kFrontThreads, R = kFrontRows, TC = kFrontTmpRows;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int LS = (int)geo.ls, TS = (int)geo.ts;
    half_t* const grey = reinterpret_cast<half_t*>(lds_raw);             // (R+6) rows x LS
    half_t* const tmp = grey + (R + 6) * LS;                              // 2 x TC rows x TS
    uint32_t* const queue = reinterpret_cast<uint32_t*>(tmp + 2 * TC * TS);
    uint32_t* const q_count = queue + kFrontQueue;

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

    if (tid == 0) *q_count = 0u;

    // =========================== A: stage grey rows [y0-3, y0+R+3) ===========================
    if (L0) {
        // RGBA8 -> luminance of the vertically mirrored row (grayscale.wgsl:16-38), 4 px per item.
        const int w4 = w >> 2;
        const float inv_w4 = 1.0f / (float)w4;
        const uint8_t* src = frames + (size_t)frame * frame_bytes;
        const int n_items = (R + 6) * w4;
        for (int i = tid; i < n_items; i += NT) {
            const int ly = (int)(((float)i + 0.5f) * inv_w4);
            const int xi = i - ly * w4;
            const int gy = y0 - 3 + ly;
            if (gy < 0 || gy >= h) continue;  // never read by a pixel that passes the guard (fast.wgsl:77)
            const uint4 px = *reinterpret_cast<const uint4*>(src + ((size_t)(h - 1 - gy) * w + (size_t)xi * 4) * 4);
            uint2 out;
            out.x = pack_half2(luminance_fast(px.x), luminance_fast(px.y));
            out.y = pack_half2(luminance_fast(px.z), luminance_fast(px.w));
            *reinterpret_cast<uint2*>(grey + ly * LS + kLdsPad + xi * 4) = out;
        }
    } else {
        // f16 mip from HBM; texels outside the level read as 0 (CRD-6) because at octaves >= 1 the
        // reference's guard and dispatch size let pixels near/over the level edge through (Q8).
        const uint16_t* src = gray_f + pyr.off[lvl];
        const int w8 = (LS - kLdsPad) >> 3;  // 8-texel groups per LDS row
        const float inv_w8 = 1.0f / (float)w8;
        const int n_items = (R + 6) * w8;
        const bool vec_ok = (w & 7) == 0;
        for (int i = tid; i < n_items; i += NT) {
            const int ly = (int)(((float)i + 0.5f) * inv_w8);
            const int xg = i - ly * w8;
            const int gy = y0 - 3 + ly;
            const int x = xg * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < h) {
                const uint16_t* row = src + (size_t)gy * w;
                if (vec_ok && x + 8 <= w) {
                    v = *reinterpret_cast<const uint4*>(row + x);
                } else {
                    uint32_t e[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = (x + k < w) ? (uint32_t)row[x + k] : 0u;
                    v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                }
            }
            *reinterpret_cast<uint4*>(grey + ly * LS + kLdsPad + x) = v;
        }
    }
    __syncthreads();

    // =========================== B1: 4-point pre-test, 8 px per item ===========================
    {
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
            // ctr[k] = grey(x - 3 + k), k = 0..13
            const float ctr[14] = {h2f(qa.x, 1), h2f(qa.y, 0), h2f(qa.y, 1), h2f(qb.x, 0), h2f(qb.x, 1),
                                   h2f(qb.y, 0), h2f(qb.y, 1), h2f(qb.z, 0), h2f(qb.z, 1), h2f(qb.w, 0),
                                   h2f(qb.w, 1), h2f(qc.x, 0), h2f(qc.x, 1), h2f(qc.y, 0)};
            const float up[8] = {h2f(qu.x, 0), h2f(qu.x, 1), h2f(qu.y, 0), h2f(qu.y, 1),
                                 h2f(qu.z, 0), h2f(qu.z, 1), h2f(qu.w, 0), h2f(qu.w, 1)};
            const float dn[8] = {h2f(qd.x, 0), h2f(qd.x, 1), h2f(qd.y, 0), h2f(qd.y, 1),
                                 h2f(qd.z, 0), h2f(qd.z, 1), h2f(qd.w, 0), h2f(qd.w, 1)};
            uint32_t cand = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float c = ctr[k + 3];
                const float d0 = ctr[k + 6] - c, d1 = ctr[k] - c, d2 = dn[k] - c, d3 = up[k] - c;  // fast.wgsl:25-30
                const int n_over = (d0 > thr) + (d1 > thr) + (d2 > thr) + (d3 > thr);
                const int n_under = (d0 < -thr) + (d1 < -thr) + (d2 < -thr) + (d3 < -thr);
                const uint32_t gx = (uint32_t)(x + k);
                if ((n_over >= 3 || n_under >= 3) && gx > 16u && gx < lim_x) cand |= 1u << k;
            }
            while (cand) {
                const int k = __builtin_ctz(cand);
                cand &= cand - 1u;
                const uint32_t slot = atomicAdd(q_count, 1u);
                if (slot < (uint32_t)kFrontQueue) {
                    queue[slot] = ((uint32_t)lyc << 16) | (uint32_t)(x + k);
                } else {  // queue full (pathological frame): test in place
                    uint32_t angle;
                    if (fast_full_test(rowc + k, LS, thr, &angle))
                        append_one((uint32_t)(x + k), gy, angle, lvl, counts + frame, corners + (size_t)frame * cap, cap);
                }
            }
        }
    }
    __syncthreads();

    // =========================== B2: drain the candidate queue densely ===========================
    {
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
            append_corners(is_corner, x, gy, angle, lvl, counts + frame, corners + (size_t)frame * cap, cap);
        }
    }

    // =========================== C0: next mip level (blit.wgsl, exact 2x2 case) ===========================
    if (geo.write_mip) {
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
    {
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

}  // namespace orb
