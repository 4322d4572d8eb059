// orb_kernels_fused.h -- the fused literal pipeline behind k_front (orb_kernels_front.h): band slots, the wave-per-keypoint BRIEF
// kernel and the slot prefix.
#pragma once
#include "orb_kernels_staged.h"
#include "orb_kernels_front.h"

namespace orb {

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
    uint32_t flat_end[kMaxLevels];  // Q - 18: a keypoint with 18 <= x < flat_end samples only columns in [0, Q), which all hold the row constant
    uint32_t qa[kMaxLevels];        // columns [0, qa) of the level's blur are the row constants
    uint32_t split;                 // workgroups per band slot (> 1 for small batches: more waves in flight)
    uint32_t oob;                   // OrbOptions::oob_policy for samples that leave the level (brief.wgsl:59-60)
    uint32_t fp;                    // OrbOptions::fp_contract: the rotation's form (rot_form())
};

// One sample of the fused pipeline's blur under an out-of-level policy != kOobZero: the coordinates are mapped into the level,
// columns below qa are the row constant, the rest is the stored tail.
__device__ __forceinline__ uint32_t blur_sample_mapped(const uint16_t* plane, const uint16_t* rowc, int w, int h, int qa, int x, int y,
                                                       uint32_t oob) {
    const int xm = oob_index(x, w, oob), ym = oob_index(y, h, oob);
    return xm < qa ? (uint32_t)rowc[ym] : (uint32_t)plane[(size_t)(uint32_t)(__mul24(ym, w) + xm)];
}

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
        uint32_t rowv = (gy >= 0 && gy < h && lane < 37u) ? (uint32_t)rowc[gy] : 0u;
        if (rg.oob != kOobZero && lane < 37u && !(gy >= 0 && gy < h)) rowv = rowc[oob_index(gy, h, rg.oob)];  // a row of the level instead of 0
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
                    float rax, ray, rbx, rby;
                    rotate_fp(ct, st, nst, pax, pay, rot_form(rg.fp), &rax, &ray);
                    rotate_fp(ct, st, nst, pbx, pby, rot_form(rg.fp), &rbx, &rby);
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
                float rax, ray, rbx, rby;
                rotate_fp(ct, st, nst, pax, pay, rot_form(rg.fp), &rax, &ray);
                rotate_fp(ct, st, nst, pbx, pby, rot_form(rg.fp), &rbx, &rby);
                const int dya = (int)ray, dyb = (int)rby;
                const int xa = (int)rec.x + (int)rax, ya = (int)rec.y + dya;
                const int xb = (int)rec.x + (int)rbx, yb = (int)rec.y + dyb;
                uint32_t va = (uint32_t)__shfl((int)rowv, dya + kBriefHalo);  // 0 when the row is outside the level
                uint32_t vb = (uint32_t)__shfl((int)rowv, dyb + kBriefHalo);
                const bool ina = xa >= 0 && xa < w && ya >= 0 && ya < h;
                const bool inb = xb >= 0 && xb < w && yb >= 0 && yb < h;
                if (!ina)
                    va = rg.oob != kOobZero ? blur_sample_mapped(plane, rowc, w, h, qa, xa, ya, rg.oob) : 0u;
                else if (xa >= qa)
                    va = plane[(size_t)(uint32_t)(__mul24(ya, w) + xa)];
                if (!inb)
                    vb = rg.oob != kOobZero ? blur_sample_mapped(plane, rowc, w, h, qa, xb, yb, rg.oob) : 0u;
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
