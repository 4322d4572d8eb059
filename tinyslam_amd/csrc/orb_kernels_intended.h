// orb_kernels_intended.h -- fused pipeline of the opt-in "intended" mode (ORB_FLAG_INTENDED; definitions IM-1..IM-8
// in DESIGN.md section 8; NOT in the reference, whose literal algorithm is orb_kernels_fused.h).
//
//   k_front_i<L0>   per level: grey tile in LDS (level 0: RGBA -> BT.601 luminance, no mirror; level >= 1: the f16
//                   mip) -> next mip level -> separable 7-tap Gaussian of the tile's own pixels straight from the
//                   tile (IM-3: the grey plane of level 0 never goes to HBM and comes back) -> segment test with
//                   arc 9..16 on the octave's own guard -> score + full-circle angle -> 3x3 NMS inside the tile ->
//                   survivors appended to the tile's segment (record + score).
//   k_gauss         (orb_kernels_staged.h) the same blur as a kernel of its own over a stored grey plane: the
//                   staged cross-check, and TINYORB_I_GAUSS_KERNEL=1.
//   k_select_i      per frame: raw counter, the top-K cut (IM-8) as a 64-bit key threshold, and where each tile's
//                   kept keypoints start in the final lists.
//   k_brief_i       per tile: blur window in LDS, rotated BRIEF-256 (+theta) of the kept keypoints.
//
// A workgroup of k_front_i owns a 16-row x TW-column tile (TW = 320 at 1280) with a 4-px apron: 3 for the
// FAST ring plus 1 so that the 3x3 NMS of the tile's own pixels sees every neighbour's score without another
// pass.  ~26 KB of LDS -> 6 workgroups of 256 threads per CU, each in its own phase (the literal kernel's
// A/B measurement showed column tiles on a par with full-width bands).
#pragma once
#include "orb_kernels_fused.h"

namespace orb {

constexpr int kIThreads = 256;
constexpr int kITileW = 320;   // widest tile (multiple of 8)
constexpr int kIPad = 16;      // LDS columns left of the tile's column 0 (multiple of 8, >= 8 + 4)
constexpr int kIApron = 4;     // grey rows above / below the tile
constexpr int kIRows = kFrontRows + 2 * kIApron;  // 24 LDS rows; LDS row j <-> image row y0 - 4 + j
constexpr int kIQueueA = 3072, kIQueueB = 1024, kIQueueC = 512, kIList = 512;

struct IGeom {
    uint32_t lvl;
    uint32_t tw;         // tile width (multiple of 8, <= kITileW)
    uint32_t n_ct;       // column tiles per band
    uint32_t n_bands;    // ceil(h / 16)
    uint32_t ls;         // LDS row stride in halfs: kIPad + tw + 16
    uint32_t write_mip;  // level lvl+1 exists and is the exact 2x2 reduction
    uint32_t xcd_swizzle;
    uint32_t slot_base;  // first tile slot of this level
    uint32_t n_slots;    // tile slots per frame (all levels)
    uint32_t seg_cap;    // records per tile segment
    uint32_t arc;        // 9..16
    uint32_t nms;        // 0 / 1
    uint32_t phase_mask; // timing experiments only: bit0 B1, bit1 S1, bit2 S2, bit3 S3+S4, bit4 C0, bit5 G
    // Detector domain and guard.  intended (IM-4): the level itself, 16 < x < w - 16, 16 < y < h - 16.
    // literal == 1 (the reference's algorithm with the opt-in arc / NMS): the reference's
    // dispatch grid of the octave (8-rounded, may exceed the level) and its level-0 guard at every octave (fast.wgsl:77,
    // Q8); grey = 0.229 r + ..., mirrored rows (Q1, Q2); angle codes 0..3141 (Q7).
    uint32_t literal;
    uint32_t dw, dh;        // domain: columns / rows covered by tiles
    uint32_t gx1, gy1;      // guard: 16 < x < gx1, 16 < y < gy1
    uint32_t store_grey;    // level 0: the tile's own grey pixels also go to the grey plane (k_mip / k_gauss / the literal blur read it)
    uint32_t blur;          // phase G: the separable Gaussian of the tile's own pixels, written to the blur plane (intended mode)
};

constexpr int kIGaussRows = kFrontRows + 6;  // rows of the X pass a tile's Y pass reads
// the X pass's output of one column half of a tile lives in the queues' storage (phase G runs before the detector)
__host__ __device__ inline bool ifront_gauss_fits(uint32_t tw) { return (uint32_t)kIGaussRows * (tw / 2u) <= (uint32_t)(kIQueueA + kIQueueB + kIQueueC); }

constexpr int kINmsRows = 20;  // region rows 0 .. R + 1 (the tile's rows and its 1-px apron), plus two spare
__host__ __device__ inline uint32_t ifront_lds_bytes(const IGeom& g) {
    return kIRows * g.ls * 2u + (kIQueueA + kIQueueB + kIQueueC) * 2u + 32u + 4u * (2u * kINmsRows + 2u);  // grey tile, queues, 8 counters, the NMS's row counters and row starts
}

__device__ __forceinline__ bool has_run_bits(uint32_t mask, uint32_t n_bits, uint32_t need) {
    // circular run of >= need set bits in an n_bits-bit mask (n_bits <= 16, need <= n_bits)
    uint32_t r = mask | (mask << n_bits);
    for (uint32_t have = 1u; have < need;) {
        const uint32_t s = min(have, need - have);
        r &= r >> s;
        have += s;
    }
    return (r & ((1u << n_bits) - 1u)) != 0u;
}

// Every second ring point (ring indices 0, 2, .., 14): a run of `arc` ring positions contains floor(arc / 2)
// consecutive ones of these eight.  Necessary condition used to thin the pre-test survivors.
__device__ __forceinline__ bool even_ring_filter(const half_t* ctr, int ls, float thr, uint32_t need, bool try_over,
                                                 bool try_under) {
    const float c = from_half(ctr[0]);
    if (try_over != try_under) {
        // one polarity to test (the usual case): d = +-(v - c) is exact (one v_fma_mix_f32), one compare per ring point
        const float sgn = try_over ? 1.0f : -1.0f, cneg = try_over ? -c : c;
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float diff = fma_mix_h(half_bits(ctr[kRingDy[2 * i] * ls + kRingDx[2 * i]]), sgn, cneg);
            m |= (diff > thr) ? (1u << i) : 0u;
        }
        return has_run_bits(m, 8u, need);
    }
    uint32_t m_over = 0, m_under = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float diff = from_half(ctr[kRingDy[2 * i] * ls + kRingDx[2 * i]]) - c;
        m_over |= (diff > thr) ? (1u << i) : 0u;
        m_under |= (diff < -thr) ? (1u << i) : 0u;
    }
    return (try_over && has_run_bits(m_over, 8u, need)) || (try_under && has_run_bits(m_under, 8u, need));
}

// Segment test on the full ring: corner <=> run of >= arc in either polarity.  try_over / try_under: polarities whose
// compass pre-test passed (a run of `arc` implies it, so the other polarity cannot have one).
__device__ __forceinline__ bool ring_has_arc(const half_t* ctr, int ls, float thr, uint32_t arc, bool try_over = true,
                                             bool try_under = true) {
    const float c = from_half(ctr[0]);
    if (try_over != try_under) {
        const float sgn = try_over ? 1.0f : -1.0f, cneg = try_over ? -c : c;
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float diff = fma_mix_h(half_bits(ctr[kRingDy[i] * ls + kRingDx[i]]), sgn, cneg);
            m |= (diff > thr) ? (1u << i) : 0u;
        }
        return has_run_16(m, arc);
    }
    uint32_t m_over = 0, m_under = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float diff = from_half(ctr[kRingDy[i] * ls + kRingDx[i]]) - c;
        m_over |= (diff > thr) ? (1u << i) : 0u;
        m_under |= (diff < -thr) ? (1u << i) : 0u;
    }
    return has_run_16(m_over, arc) || has_run_16(m_under, arc);
}

// Score (sum over the run's polarity of |diff| - thr, ring order, binary32) and full-circle angle of a corner.
__device__ __forceinline__ void ring_score_angle(const half_t* ctr, int ls, float thr, uint32_t arc, bool literal,
                                                 float* score, uint32_t* angle) {
    const float c = from_half(ctr[0]);
    uint32_t m_over = 0;
    float cx = 0.0f, cy = 0.0f, s_over = 0.0f, s_under = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = from_half(ctr[kRingDy[i] * ls + kRingDx[i]]);
        const float diff = v - c;
        const float px = v * (float)kRingDx[i];
        const float py = v * (float)kRingDy[i];
        cx = cx + px;
        cy = cy + py;
        if (diff > thr) {
            m_over |= 1u << i;
            const float e = diff - thr;
            s_over = s_over + e;
        } else if (diff < -thr) {
            const float nd = -diff;
            const float e = nd - thr;
            s_under = s_under + e;
        }
    }
    *score = has_run_16(m_over, arc) ? s_over : s_under;
    *angle = literal ? angle_code(cy, cx) : angle_code_signed(cy, cx);
}

template <bool L0>
__global__ __launch_bounds__(kIThreads, 6) void k_front_i(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                          uint16_t* __restrict__ gray, uint16_t* __restrict__ blur, Pyramid pyr, IGeom geo, float thr,
                                                          uint32_t* __restrict__ seg_counts,
                                                          CornerData* __restrict__ segments,
                                                          float* __restrict__ seg_scores) {
    constexpr int NT = kIThreads, R = kFrontRows;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int LS = (int)geo.ls, TW = (int)geo.tw;
    half_t* const grey = reinterpret_cast<half_t*>(lds_raw);  // kIRows x LS; LDS column kIPad <-> image column cx0
    // 16-bit queue entries: region row, item of the row, bit number of the item's survivor mask (see B1).
    // A: pre-test survivors; B: survivors of the even-ring filter; C: corners.  The corner list of the NMS
    // (position, angle, score) reuses A's storage once A has been drained.
    uint16_t* const queue_a = reinterpret_cast<uint16_t*>(grey + kIRows * LS);
    uint16_t* const queue_b = queue_a + kIQueueA;
    uint16_t* const queue_c = queue_b + kIQueueB;
    uint32_t* const counters = reinterpret_cast<uint32_t*>(queue_c + kIQueueC);
    uint32_t* const qa_count = counters;
    uint32_t* const qb_count = counters + 1;
    uint32_t* const qc_count = counters + 2;
    uint32_t* const list_count = counters + 3;
    uint32_t* const c_count = counters + 4;   // corners appended by this tile
    uint32_t* const overflow = counters + 5;  // some queue was full: the tile is redone by the direct path
    uint32_t* const nms_row_cnt = counters + 8;  // [kINmsRows] corners per region row (S3 counts, S4 sorts by them)
    uint16_t* const list_pos = queue_a;                                          // kIList
    uint16_t* const list_ang = queue_a + kIList;                                 // kIList
    float* const list_score = reinterpret_cast<float*>(queue_a + 2 * kIList);   // kIList (6 KB in all = A's storage)

    uint32_t frame, band, ct;
    {
        const uint32_t per_frame = geo.n_bands * geo.n_ct;
        const uint32_t L = blockIdx.x;
        uint32_t rem;
        if (geo.xcd_swizzle) {
            const uint32_t xcd = L & 7u, s2 = L >> 3;
            frame = (s2 / per_frame) * 8u + xcd;
            rem = s2 % per_frame;
        } else {
            frame = L / per_frame;
            rem = L % per_frame;
        }
        band = rem / geo.n_ct;
        ct = rem % geo.n_ct;
    }
    const uint32_t lvl = geo.lvl;
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const int y0 = (int)band * R, cx0 = (int)ct * TW;
    const bool lit = geo.literal != 0u;
    const int dw = (int)geo.dw, dh = (int)geo.dh, gx1 = (int)geo.gx1, gy1 = (int)geo.gy1;
    const int tw = min(TW, dw - cx0);  // columns of this tile inside the detector's domain
    const int tid = (int)threadIdx.x;
    uint16_t* const gray_f = gray + (size_t)frame * pyr.stride;
    const size_t slot = (size_t)frame * geo.n_slots + geo.slot_base + (size_t)band * geo.n_ct + ct;
    CornerData* const seg = segments + slot * geo.seg_cap;
    float* const seg_sc = seg_scores + slot * geo.seg_cap;
    const uint32_t arc = geo.arc;
    const int apron = geo.nms ? 1 : 0;  // the NMS needs the scores of the pixels around the tile as well
    if (tid < 8 + kINmsRows) counters[tid] = 0u;

    // =========================== A: grey rows [y0-4, y0+R+4) x columns [cx0-8, cx0+tw+8) ===========================
    if (L0) {
        // RGBA quads; BT.601 luminance of input(x, y) (IM-1) -- or the reference's 0.229 weight on the mirrored row
        // (grayscale.wgsl:16-38); the tile's own pixels also go to the grey plane (for the blur kernel)
        // A thread keeps its quad column and walks down the rows (k_front's phase A): the mirrored byte offset, the LDS offset
        // and the row are computed once and stepped; buffer loads with the frame as the buffer -- a row above or below the image
        // has an offset past the frame (or a negative one = huge) and reads zeros, which are staged like any other row and
        // never read by a guarded pixel (the literal setting wants zeros there, Q8; phase G maps such rows to the edge row).
        const int q0 = max(cx0 / 4 - 2, 0), q1 = min((cx0 + tw) / 4 + 2, w / 4);
        const int per_row = q1 - q0;                       // <= kITileW / 4 + 4 quads: at least two rows per pass
        const int rpp = NT / per_row;
        const int ty = (int)(((float)tid + 0.5f) * (1.0f / (float)per_row)), tx = tid - __mul24(ty, per_row);
        const bool lane_ok = ty < rpp;
        const uint8_t* src0 = frames + (size_t)frame * frame_bytes;
        const __amdgpu_buffer_rsrc_t frame_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src0), 0, (int)frame_bytes, kBufferWord3Raw);
        uint16_t* plane0 = gray_f + pyr.off[0];
        const int q = q0 + tx;
        int gy_w = y0 - kIApron + ty;
        uint32_t off_w = (uint32_t)(__mul24(lit ? h - 1 - gy_w : gy_w, w) + q * 4) * 4u;  // input row: mirrored in the literal setting
        const uint32_t off_step = (uint32_t)(__mul24(lit ? -rpp : rpp, w) * 4);
        int dst_w = __mul24(ty, LS) + kIPad + (q * 4 - cx0);
        const int dst_step = rpp * LS;
        const bool own_col = q * 4 >= cx0 && q * 4 < cx0 + tw;
        constexpr int U = 4;
        for (int lyb = ty; lyb < kIRows; lyb += rpp * U) {
            uint4 v[U];
            int dst[U], gys[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(frame_rsrc, (int)off_w, 0, 0));
                dst[u] = dst_w;
                gys[u] = gy_w;
                off_w += off_step;
                dst_w += dst_step;
                gy_w += rpp;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int ly = lyb + u * rpp;
                if (lane_ok && ly < kIRows) {
                    // on texel pairs, as in k_front (packed binary32 instructions, one packed conversion per pair)
                    uint2 out;
                    if (lit) {
                        out.x = luminance_pair_f16<false>(v[u].x, v[u].y);
                        out.y = luminance_pair_f16<false>(v[u].z, v[u].w);
                    } else {
                        out.x = luminance_pair_f16<true>(v[u].x, v[u].y);
                        out.y = luminance_pair_f16<true>(v[u].z, v[u].w);
                    }
                    *reinterpret_cast<uint2*>(grey + dst[u]) = out;
                    if (geo.store_grey && own_col && ly >= kIApron && ly < kIApron + R && gys[u] < h)
                        *reinterpret_cast<uint2*>(plane0 + (size_t)(uint32_t)(__mul24(gys[u], w) + q * 4)) = out;
                }
            }
        }
    } else {
        // f16 mip from HBM in 8-texel groups; texels outside the level are never read by a guarded pixel
        const uint16_t* srcn = gray_f + pyr.off[lvl];
        const int g0 = cx0 / 8 - 1, per_row = TW / 8 + 2;
        const float inv_per_row = 1.0f / (float)per_row;
        const int n_items = kIRows * per_row;
        const bool vec_ok = (w & 7) == 0;
        for (int i = tid; i < n_items; i += NT) {
            const int ly = (int)(((float)i + 0.5f) * inv_per_row);
            const int x = (g0 + (i - __mul24(ly, per_row))) * 8;
            const int gy = y0 - kIApron + ly;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < h && x >= 0 && x < w) {
                const uint16_t* row = srcn + (size_t)(uint32_t)__mul24(gy, w);
                if (vec_ok && x + 8 <= w) {
                    v = *reinterpret_cast<const uint4*>(row + x);
                } else {
                    uint32_t e[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = (x + k < w) ? (uint32_t)row[x + k] : 0u;
                    v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                }
            }
            *reinterpret_cast<uint4*>(grey + __mul24(ly, LS) + kIPad + (x - cx0)) = v;
        }
    }
    __syncthreads();

    // =========================== C0: next mip level (exact 2x2 case) ===========================
    // (before the detector: its stores drain while the detector runs)
    if (geo.write_mip && (geo.phase_mask & 16u)) {
        const int wd = (int)pyr.w[lvl + 1], hd = (int)pyr.h[lvl + 1];
        uint16_t* dst = gray_f + pyr.off[lvl + 1];
        const int g4 = (tw / 2 + 3) >> 2;
        const float inv_g4 = 1.0f / (float)max(g4, 1);
        const int n_items = (R / 2) * g4;
        const bool vec_ok = (wd & 3) == 0;
        for (int i = tid; i < n_items; i += NT) {
            const int r = (int)(((float)i + 0.5f) * inv_g4);
            const int xl = (i - __mul24(r, g4)) * 4;
            const int xd = cx0 / 2 + xl, yd = (y0 >> 1) + r;
            if (yd >= hd || xd >= wd) continue;
            const half_t* top = grey + __mul24(2 * r + kIApron, LS) + kIPad + 2 * xl;
            const uint4 qt = *reinterpret_cast<const uint4*>(top);
            const uint4 qb = *reinterpret_cast<const uint4*>(top + LS);
            const uint32_t tw4[4] = {qt.x, qt.y, qt.z, qt.w}, bw[4] = {qb.x, qb.y, qb.z, qb.w};
            uint16_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float st = add_halves(tw4[k]);  // a + b, c + d: the two texels of a row share a register (CRD-4)
                const float sb = add_halves(bw[k]);
                o[k] = half_bits(to_half((st + sb) * 0.25f));
            }
            uint16_t* out = dst + (size_t)(uint32_t)(__mul24(yd, wd) + xd);
            if (vec_ok && xd + 4 <= wd) {
                *reinterpret_cast<uint2*>(out) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (xd + k < wd) out[k] = o[k];
            }
        }
    }

    // =========================== G: separable 7-tap Gaussian of the tile's own pixels (IM-3) ===========================
    // X pass then Y pass, f16 after each (gauss7, the arithmetic of k_gauss), clamp-to-edge.  The tile holds what both
    // passes need: rows y0-3 .. y0+R+2 and columns cx0-3 .. cx0+tw+2 lie inside its apron.  Rows outside the level are
    // read from the level's first / last row, which this band stages whenever it needs them; columns outside the
    // level are filled in place first (the detector's guard keeps it 16 texels away from them).  The X pass's output
    // (22 rows) of one column HALF of the tile at a time borrows the queues' storage: 7 KB at tw = 320.
    if (geo.blur && (geo.phase_mask & 32u)) {
        uint16_t* const mid = queue_a;
        const bool left_edge = cx0 == 0, right_edge = cx0 + TW + 3 > w;  // uniform; the right apron of a tile may cross the edge without the tile reaching it
        if (left_edge || right_edge) {
            const int we = w - cx0;  // columns of the level in this tile
            for (int i = tid; i < kIRows * 3; i += NT) {
                const int ly = i / 3, k = i - ly * 3;
                half_t* const row = grey + __mul24(ly, LS) + kIPad;
                if (left_edge) row[-1 - k] = row[0];
                if (right_edge) row[we + k] = row[we - 1];
            }
            __syncthreads();
        }
        const int HW = TW >> 1, Q = HW >> 2, P2 = HW >> 1;  // half width (multiple of 4), its quads, its column pairs
        float one = 1.0f, zero = 0.0f;
        asm volatile("" : "+v"(one), "+v"(zero));  // operands of v_fma_mix_f32, kept in registers
        const float inv_q = 1.0f / (float)Q, inv_p2 = 1.0f / (float)P2;
        const int xrpp = NT / Q, xty = (int)(((float)tid + 0.5f) * inv_q), xtx = tid - __mul24(xty, Q);  // X pass: rows per turn, this thread's row phase and quad
        uint16_t* const plane = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
        for (int c0 = 0; c0 < tw; c0 += HW) {  // uniform
            // X pass: item = 4 consecutive columns of one row; inputs x-3 .. x+6 from three 8-byte reads
            for (int r = xty; r < kIGaussRows && xty < xrpp; r += xrpp) {  // a thread keeps its quad column and walks down the rows
                const int x = xtx * 4;
                const int lyc = min(max(y0 - 3 + r, 0), h - 1) - y0 + kIApron;
                const half_t* p = grey + __mul24(lyc, LS) + kIPad + c0 + x - 4;
                const uint2 q0 = *reinterpret_cast<const uint2*>(p), q1 = *reinterpret_cast<const uint2*>(p + 4),
                            q2 = *reinterpret_cast<const uint2*>(p + 8);
                // halfs 0..11 = columns x-4 .. x+7; output column x + c takes halfs c+1 .. c+7 (gauss7_h: no conversions)
                const uint32_t o0 = half_bits(to_half(gauss7_h<1, 0, 1, 0, 1, 0, 1>(q0.x, q0.y, q0.y, q1.x, q1.x, q1.y, q1.y, one, zero)));
                const uint32_t o1 = half_bits(to_half(gauss7_h<0, 1, 0, 1, 0, 1, 0>(q0.y, q0.y, q1.x, q1.x, q1.y, q1.y, q2.x, one, zero)));
                const uint32_t o2 = half_bits(to_half(gauss7_h<1, 0, 1, 0, 1, 0, 1>(q0.y, q1.x, q1.x, q1.y, q1.y, q2.x, q2.x, one, zero)));
                const uint32_t o3 = half_bits(to_half(gauss7_h<0, 1, 0, 1, 0, 1, 0>(q1.x, q1.x, q1.y, q1.y, q2.x, q2.x, q2.y, one, zero)));
                const uint32_t o[4] = {o0, o1, o2, o3};
                *reinterpret_cast<uint2*>(&mid[__mul24(r, HW) + x]) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            }
            __syncthreads();
            // Y pass: item = 2 columns x 6, 5 or 5 rows (three row groups: 240 items for the 256 threads at tw = 320; two groups
            // of eight rows kept 160 busy) from 12 or 11 rows of the X pass
            for (int i = tid; i < 3 * P2; i += NT) {
                const int rg = (int)(((float)i + 0.5f) * inv_p2), x = (i - __mul24(rg, P2)) * 2;
                const int r0 = rg == 0 ? 0 : 5 * rg + 1, n_rows = rg == 0 ? 6 : 5;
                const int gx = cx0 + c0 + x;
                if (gx >= w) continue;
                uint32_t v[12];  // column x in the low halves, x + 1 in the high halves
#pragma unroll
                for (int k = 0; k < 12; k++) v[k] = *reinterpret_cast<const uint32_t*>(&mid[__mul24(min(r0 + k, kIGaussRows - 1), HW) + x]);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int gy = y0 + r0 + k;
                    const uint32_t oa = half_bits(to_half(gauss7_h<0, 0, 0, 0, 0, 0, 0>(v[k], v[k + 1], v[k + 2], v[k + 3], v[k + 4], v[k + 5], v[k + 6], one, zero)));
                    const uint32_t ob = half_bits(to_half(gauss7_h<1, 1, 1, 1, 1, 1, 1>(v[k], v[k + 1], v[k + 2], v[k + 3], v[k + 4], v[k + 5], v[k + 6], one, zero)));
                    if (k < n_rows && gy < h) {
                        uint16_t* out = plane + (size_t)(uint32_t)(__mul24(gy, w) + gx);
                        if (gx + 1 < w && (w & 1) == 0)
                            *reinterpret_cast<uint32_t*>(out) = oa | (ob << 16);
                        else {
                            out[0] = (uint16_t)oa;
                            if (gx + 1 < w) out[1] = (uint16_t)ob;
                        }
                    }
                }
            }
            __syncthreads();  // the next half's X pass, then the detector's queues, overwrite mid
        }
    }

    // Region of this tile: its own pixels plus (with NMS) a 1-px apron.  Region row r <-> image row y0 - 1 + r,
    // LDS row r + 3.  IM-4 guard: 16 < x < w - 16, 16 < y < h - 16.
    const int rx0 = max(cx0 - apron, 17), rx1 = min(cx0 + tw + apron, gx1);  // [rx0, rx1) columns tested
    const int ry0 = max(y0 - apron, 17), ry1 = min(min(y0 + R, dh) + apron, gy1);
    auto in_core = [&](int x, int gy) { return x >= cx0 && x < cx0 + tw && gy >= y0 && gy < y0 + R && gy < dh; };
    auto append = [&](uint32_t x, uint32_t gy, uint32_t angle, float score) {
        const uint32_t idx = atomicAdd(c_count, 1u);
        if (idx < geo.seg_cap) {
            *reinterpret_cast<uint4*>(&seg[idx]) = make_uint4(x, gy, angle, lvl);
            seg_sc[idx] = score;
        }
    };
    // The dense path (a queue overflowed: a tile of texture at a low threshold).  Round 4's form evaluated every pixel of the tile on its
    // own and, with NMS, scored each corner's eight neighbours AGAIN -- 12 ms per 256 frames of noise at threshold 8 / 255, the mode's
    // one cliff.  Now the tile is walked down its region rows with a few rows of (score, angle) resident in the drained queues' storage (the
    // 3x3 NMS of a row needs its two neighbours and nothing else): every pixel is tested once, no corner is scored twice, and no capacity
    // is involved -- results cannot depend on how dense the tile is.
    auto dense_rows = [&]() {
        // Two region rows per step, FOUR resident (the rows of this step and the two before): compass pre-test of the step's pixels ->
        // survivors compacted into a list (densely packed lanes for what follows) -> segment test of the polarities that passed, score and
        // angle of the corners into the rows' (score, angle) buffers -> 3x3 suppression of the two rows whose neighbours are now resident.
        constexpr int NCMAX = kITileW + 2;
        const int nc = tw + 2 * apron;                                    // region columns, x = cx0 - apron + c
        const int nr = min(R, dh - y0) + 2 * apron;                       // region rows, gy = y0 - apron + r
        float* const sc4 = reinterpret_cast<float*>(queue_a);             // [4][NCMAX] scores, 0 = no corner (a corner's score is positive)
        uint16_t* const an4 = reinterpret_cast<uint16_t*>(sc4 + 4 * NCMAX);  // [4][NCMAX] angle codes
        uint16_t* const plist = an4 + 4 * NCMAX;                          // [2 * NCMAX] pre-test survivors of the step: pixel | flags << 14
        static_assert(4 * NCMAX * 6 + 2 * NCMAX * 2 <= (kIQueueA + kIQueueB + kIQueueC) * 2, "the dense path's rows and list do not fit the queues' storage");
        static_assert(2 * NCMAX < (1 << 14), "a step's pixel index and two flags share 16 bits");
        const uint32_t need = arc >= 12u ? 3u : 2u;                       // compass points a run of `arc` holds at least
        auto slot_of_row = [](int r) { return (r + 4) & 3; };
        for (int c = tid; c < NCMAX; c += NT) sc4[slot_of_row(-1) * NCMAX + c] = 0.0f, sc4[slot_of_row(-2) * NCMAX + c] = 0.0f;  // above the region
        const int n_steps = (nr + 2) / 2;
        for (int k = 0; k < n_steps; k++) {
            const int r0 = 2 * k;
            // ---- a: compass pre-test (fast.wgsl:85-95 with the arc's count) of rows r0, r0 + 1; their buffers start as "no corner"
            if (tid == 0) *qa_count = 0u;
            __syncthreads();
            for (int p = tid; p < 2 * nc; p += NT) {
                const int rr = p >= nc ? 1 : 0, c = p - rr * nc, r = r0 + rr;
                const int x = cx0 - apron + c, gy = y0 - apron + r;
                sc4[slot_of_row(r) * NCMAX + c] = 0.0f;
                if (r < nr && gy > 16 && gy < gy1 && x > 16 && x < gx1) {
                    const half_t* ctr = grey + __mul24(gy - y0 + kIApron, LS) + kIPad + (x - cx0);
                    const float cv = from_half(ctr[0]);
                    const float d4[4] = {from_half(ctr[3]) - cv, from_half(ctr[-3]) - cv, from_half(ctr[3 * LS]) - cv, from_half(ctr[-3 * LS]) - cv};
                    uint32_t n_over = 0, n_under = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) n_over += d4[q] > thr ? 1u : 0u, n_under += d4[q] < -thr ? 1u : 0u;
                    const uint32_t fl = (n_over >= need ? 1u : 0u) | (n_under >= need ? 2u : 0u);
                    if (fl) plist[atomicAdd(qa_count, 1u)] = (uint16_t)((uint32_t)p | (fl << 14));
                }
            }
            __syncthreads();
            // ---- b: segment test, score, angle -- on the list
            const uint32_t n_l = *qa_count;
            for (uint32_t i = (uint32_t)tid; i < n_l; i += NT) {
                const uint32_t e = plist[i], p = e & 0x3fffu;
                const int rr = (int)p >= nc ? 1 : 0, c = (int)p - rr * nc, r = r0 + rr;
                const int x = cx0 - apron + c, gy = y0 - apron + r;
                const half_t* ctr = grey + __mul24(gy - y0 + kIApron, LS) + kIPad + (x - cx0);
                if (ring_has_arc(ctr, LS, thr, arc, (e >> 14) & 1u, (e >> 15) & 1u)) {
                    float sv;
                    uint32_t ang;
                    ring_score_angle(ctr, LS, thr, arc, lit, &sv, &ang);
                    if (!geo.nms) {
                        append((uint32_t)x, (uint32_t)gy, ang, sv);  // no apron without suppression: every region pixel is the tile's own
                    } else {
                        sc4[slot_of_row(r) * NCMAX + c] = sv;
                        an4[slot_of_row(r) * NCMAX + c] = (uint16_t)ang;
                    }
                }
            }
            __syncthreads();
            // ---- c: suppression of rows r0 - 1 and r0 (their three rows are resident)
            if (geo.nms)
                for (int p = tid; p < 2 * nc; p += NT) {
                    const int rr = p >= nc ? 1 : 0, c = p - rr * nc, r = r0 - 1 + rr;
                    const int x = cx0 - apron + c, gy = y0 - apron + r;
                    if (c < 1 || c >= nc - 1 || !in_core(x, gy)) continue;  // the region's border belongs to the neighbouring tiles
                    const int sm = slot_of_row(r), su = slot_of_row(r - 1), sd = slot_of_row(r + 1);
                    const float sv = sc4[sm * NCMAX + c];
                    if (!(sv > 0.0f)) continue;
                    bool keep = true;
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            if (dx == 0 && dy == 0) continue;
                            const float t = sc4[(dy < 0 ? su : (dy > 0 ? sd : sm)) * NCMAX + c + dx];
                            const bool later = dy > 0 || (dy == 0 && dx > 0);
                            if (t > sv || (t == sv && !later)) keep = false;  // t == 0: no corner there (sv > 0)
                        }
                    if (keep) append((uint32_t)x, (uint32_t)gy, (uint32_t)an4[sm * NCMAX + c], sv);
                }
            // (the next step's barrier in front of its pre-test orders its writes behind these reads)
        }
    };

    // =========================== B1: compass pre-test, 16 px per item ===========================
    // 16-bit queue entries (k_front's form): [14:10] region row (0 <-> y0-1), [9:5] item of the row (LDS column kIPad - 8 + 16 j),
    // [4:0] bit number in the item's survivor mask: bit 16 h + front_mask_bit(k, under) = pixel 8 h + k of the item passed the
    // pre-test of that polarity.  A pixel that passes both (possible below arc 12: two of the four compass points each way)
    // has two entries; the stages behind test one polarity per entry, and a run of >= 9 of one polarity excludes the other.
    if (geo.phase_mask & 1u)
    // A run of `arc` ring positions holds at least arc / 4 of the four compass points: 3 for arc >= 12 (the
    // reference's shortcut), 2 for 9..11.  Conservative packed test as in k_front (selection network on the f16 bit
    // patterns, packed-f16 compares against a threshold one ulp below RD16(thr)); what is done once per item -- index
    // arithmetic, the region's edges, the slot reservation -- weighs half as much per pixel as with items of eight.
    {
        const bool need3 = arc >= 12u;
        const int r_lo = ry0 - (y0 - 1), r_hi = ry1 - (y0 - 1);          // region rows [r_lo, r_hi)
        const int j_lo = (rx0 - cx0 + 8) >> 4, j_hi = (rx1 - 1 - cx0 + 8) >> 4;  // items: xl = 16 * j - 8
        const int per_row = j_hi - j_lo + 1;
        const int n_items = (r_hi > r_lo && per_row > 0) ? (r_hi - r_lo) * per_row : 0;
        const float inv_per_row = 1.0f / (float)max(per_row, 1);
        uint32_t tb = half_bits(to_half(thr));
        if (from_half(bits_half((uint16_t)tb)) > thr) tb--;
        tb = tb ? tb - 1u : 0x8001u;
        const half2_t thr_lo2 = __builtin_bit_cast(half2_t, tb | (tb << 16));
        for (int i = tid; i < n_items; i += NT) {
            const int rr = (int)(((float)i + 0.5f) * inv_per_row);
            const int r = r_lo + rr;
            const int j = j_lo + (i - __mul24(rr, per_row));
            const int xl = 16 * j - 8;  // column inside the tile of the item's pixel 0
            const int x = cx0 + xl;
            const half_t* row16 = grey + __mul24(r + 3, LS) + kIPad + xl;
            uint32_t cand = 0;  // bit 16 h + front_mask_bit(k, under): pixel 8 h + k survives
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const half_t* rowc = row16 + 8 * hh;
                const uint2 qa = *reinterpret_cast<const uint2*>(rowc - 4);
                const uint4 qb = *reinterpret_cast<const uint4*>(rowc);
                const uint2 qc = *reinterpret_cast<const uint2*>(rowc + 8);
                const uint4 qu = *reinterpret_cast<const uint4*>(rowc - 3 * LS);
                const uint4 qd = *reinterpret_cast<const uint4*>(rowc + 3 * LS);
                const uint32_t dw[8] = {qa.x, qa.y, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y};
                const uint32_t upw[4] = {qu.x, qu.y, qu.z, qu.w}, dnw[4] = {qd.x, qd.y, qd.z, qd.w};
                uint32_t e_ovr[4], e_und[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const ushort2_t left = as_u16x2(__builtin_amdgcn_alignbit(dw[q + 1], dw[q], 16));
                    const ushort2_t right = as_u16x2(__builtin_amdgcn_alignbit(dw[q + 4], dw[q + 3], 16));
                    const ushort2_t upp = as_u16x2(upw[q]), dwn = as_u16x2(dnw[q]);
                    const ushort2_t lo1 = __builtin_elementwise_min(left, right), hi1 = __builtin_elementwise_max(left, right);
                    const ushort2_t lo2 = __builtin_elementwise_min(upp, dwn), hi2 = __builtin_elementwise_max(upp, dwn);
                    const ushort2_t m1 = __builtin_elementwise_max(lo1, lo2), m2 = __builtin_elementwise_min(hi1, hi2);
                    const half2_t second_lo = __builtin_bit_cast(half2_t, __builtin_elementwise_min(m1, m2));
                    const half2_t second_hi = __builtin_bit_cast(half2_t, __builtin_elementwise_max(m1, m2));
                    const half2_t c2 = __builtin_bit_cast(half2_t, dw[q + 2]);
                    // >= 3 brighter <=> 2nd smallest beyond thr; >= 2 brighter <=> 2nd largest beyond thr (and mirrored)
                    const half2_t sel_over = need3 ? second_lo : second_hi, sel_under = need3 ? second_hi : second_lo;
                    const half2_t e_over = thr_lo2 - (sel_over - c2);    // negative  <=>  sel_over - c > thr_lo
                    const half2_t e_under = (sel_under - c2) + thr_lo2;  // negative  <=>  sel_under - c < -thr_lo
                    e_ovr[q] = __builtin_bit_cast(uint32_t, e_over);
                    e_und[q] = __builtin_bit_cast(uint32_t, e_under);
                }
                uint32_t acc = 0;  // the sixteen sign bits as one 16-bit mask (k_front, B1)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t so = __builtin_bit_cast(uint32_t, as_u16x2(e_ovr[q]) >> (unsigned short)15);
                    const uint32_t su = __builtin_bit_cast(uint32_t, as_u16x2(e_und[q]) >> (unsigned short)15);
                    acc = q == 0 ? so : ((so << q) | acc);
                    acc = (su << (4 + q)) | acc;
                }
                cand |= __builtin_amdgcn_perm(0u, acc, hh == 0 ? 0x0c0c0200u : 0x02000c0cu);
            }
            {   // pixels of the item outside [rx0, rx1): only the first and the last item of a row
                const int first = rx0 - x, past = rx1 - x;
                if (first > 0 || past < 16) {
                    uint32_t keep = 0;
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (k >= first && k < past) keep |= (front_mask_bit(k & 7, false) | front_mask_bit(k & 7, true)) << (16 * (k >> 3));
                    cand &= keep;
                }
            }
            if (cand) {
                const uint32_t n_cand = (uint32_t)__builtin_popcount(cand);
                uint32_t qs = lds_add_rtn(qa_count, n_cand);
                const uint32_t base = ((uint32_t)r << 10) | ((uint32_t)j << 5);
                if (qs + n_cand <= (uint32_t)kIQueueA) {
                    while (cand) {
                        const uint32_t pbit = (uint32_t)__builtin_ctz(cand);
                        cand &= cand - 1u;
                        queue_a[qs++] = (uint16_t)(base | pbit);
                    }
                } else {
                    *overflow = 1u;
                }
            }
        }
    }
    __syncthreads();

    auto locate = [&](uint32_t e, int* x, int* gy) -> const half_t* {
        const int r = (int)((e >> 10) & 31u), j = (int)((e >> 5) & 31u);
        const int k16 = (int)(((e & 3u) << 1) | ((e >> 3) & 1u) | ((e >> 1) & 8u));  // pixel of the item: front_mask_bit in [3:0], the half in [4]
        const int xl = 16 * j - 8 + k16;
        *x = cx0 + xl;
        *gy = y0 - 1 + r;
        return grey + __mul24(r + 3, LS) + kIPad + xl;
    };
    auto is_over = [](uint32_t e) { return (e & 4u) == 0u; };  // polarity of the pre-test that passed
    // =========================== S1: even-ring filter, A -> B ===========================
    // every second ring point: a run of `arc` ring positions contains arc / 2 consecutive ones of these eight (one polarity per
    // entry: one exact v_fma_mix_f32 difference and one compare per point.  The packed 16-bit form of the literal kernel's
    // 16-point test -- even_ring_mask_polar -- was measured here and is slower on eight points: finding the exact f16 threshold
    // of the pixel costs as much as half of them: 0.144 against 0.120 ms per batch)
    // (a tile whose pre-test already overflowed queue A takes the dense path: the stages in between would work on a truncated queue for nothing)
    const bool early_overflow = *overflow != 0u;  // uniform: read behind the barrier that ends B1
    if ((geo.phase_mask & 2u) && !early_overflow) {
        const uint32_t need_even = arc >> 1;
        const uint32_t n_a = min(*qa_count, (uint32_t)kIQueueA);
        for (uint32_t i = (uint32_t)tid; i < n_a; i += NT) {
            const uint32_t e = queue_a[i];
            int x, gy;
            const half_t* ctr = locate(e, &x, &gy);
            if (even_ring_filter(ctr, LS, thr, need_even, is_over(e), !is_over(e))) {
                const uint32_t qs = atomicAdd(qb_count, 1u);
                if (qs < (uint32_t)kIQueueB)
                    queue_b[qs] = (uint16_t)e;
                else
                    *overflow = 1u;
            }
        }
    }
    __syncthreads();
    // =========================== S2: full segment test, B -> C ===========================
    if ((geo.phase_mask & 4u) && !early_overflow) {
        const uint32_t n_b = min(*qb_count, (uint32_t)kIQueueB);
        for (uint32_t i = (uint32_t)tid; i < n_b; i += NT) {
            const uint32_t e = queue_b[i];
            int x, gy;
            const half_t* ctr = locate(e, &x, &gy);
            if (has_run_16(ring_mask_polar(ctr, LS, thr, is_over(e)), arc)) {
                const uint32_t qs = atomicAdd(qc_count, 1u);
                if (qs < (uint32_t)kIQueueC)
                    queue_c[qs] = (uint16_t)e;
                else
                    *overflow = 1u;
            }
        }
    }
    __syncthreads();
    // =========================== S3: score and angle of the corners -> list (A's storage) ===========================
    if ((geo.phase_mask & 8u) && !early_overflow) {
        const uint32_t n_c = min(*qc_count, (uint32_t)kIQueueC);
        for (uint32_t i = (uint32_t)tid; i < n_c; i += NT) {
            const uint32_t e = queue_c[i];
            int x, gy;
            const half_t* ctr = locate(e, &x, &gy);
            float s;
            uint32_t ang;
            ring_score_angle(ctr, LS, thr, arc, lit, &s, &ang);
            const uint32_t row = (uint32_t)(gy - (y0 - 1));
            list_pos[i] = (uint16_t)((row << 9) | (uint32_t)(x - cx0 + kIPad));  // region row, LDS column
            list_ang[i] = (uint16_t)ang;
            list_score[i] = s;
            // the corner's place among the corners of its region row (S4 sorts by row): queue B is drained by now
            if (geo.nms) reinterpret_cast<uint16_t*>(queue_b)[i] = (uint16_t)atomicAdd(&nms_row_cnt[min(row, (uint32_t)(kINmsRows - 1))], 1u);
        }
        if (tid == 0) *list_count = n_c;
    }
    __syncthreads();
    // =========================== S4: 3x3 NMS inside the tile, survivors -> segment ===========================
    if (*overflow == 0u) {
        const uint32_t n = *list_count;
        if (geo.nms && n > 1u) {
            // The corners sorted by region row (a counting sort in the storage of queues B and C, both drained by now): a corner
            // then meets only the corners of its own and the two neighbouring rows -- about ten of them where all n * n ordered
            // pairs spread over the workgroup were 3 600 tests per tile (n ~ 60), a tenth of the kernel's instructions.
            constexpr int kRows = kINmsRows;
            // Storage: slot_of (written by S3) at the head of queue B, the sorted indices in queue C, row_start behind the row counters.  Every wave
            // derives the row prefix itself (lanes 0..kRows-1, a wave-wide scan) and writes the same values: no barrier for it.
            const uint16_t* const slot_of = reinterpret_cast<const uint16_t*>(queue_b);         // [kIList]
            uint32_t* const row_start = nms_row_cnt + kRows;                                    // [kRows + 1]
            uint16_t* const sorted = queue_c;                                                   // [kIList] corner indices, row by row
            static_assert(kIList <= kIQueueB && kIList <= kIQueueC, "NMS scratch does not fit queues B and C");
            {
                const uint32_t lane = (uint32_t)tid & 63u;
                const uint32_t cnt = lane < (uint32_t)kRows ? nms_row_cnt[lane] : 0u;
                uint32_t incl = cnt;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
                    if ((int)lane >= d) incl += t;
                }
                if (lane < (uint32_t)kRows) row_start[lane] = incl - cnt;
                if (lane == (uint32_t)kRows - 1u) row_start[kRows] = incl;
            }
            for (uint32_t i = (uint32_t)tid; i < n; i += NT) sorted[row_start[min((uint32_t)list_pos[i] >> 9, (uint32_t)(kRows - 1))] + slot_of[i]] = (uint16_t)i;
            __syncthreads();
            for (uint32_t i = (uint32_t)tid; i < n; i += NT) {
                const uint32_t pi = list_pos[i];
                const int ri = (int)(pi >> 9), ci = (int)(pi & 511u);
                const float sc = list_score[i];
                bool keep = true;
                for (int rr = max(ri - 1, 0); rr <= min(ri + 1, kRows - 1); rr++) {
                    const uint32_t q1 = row_start[rr + 1];
                    for (uint32_t q = row_start[rr]; q < q1; q++) {
                        const uint32_t j = sorted[q];
                        const int dx = (int)(list_pos[j] & 511u) - ci, dy = rr - ri;
                        if (j != i && dx >= -1 && dx <= 1) {
                            const float t = list_score[j];
                            const bool later = dy > 0 || (dy == 0 && dx > 0);
                            if (t > sc || (t == sc && !later)) keep = false;
                        }
                    }
                }
                const int x = cx0 + ci - kIPad, gy = y0 - 1 + ri;
                if (keep && in_core(x, gy)) append((uint32_t)x, (uint32_t)gy, (uint32_t)list_ang[i], sc);
            }
        } else {
            for (uint32_t i = (uint32_t)tid; i < n; i += NT) {
                const uint32_t pi = list_pos[i];
                const int x = cx0 + (int)(pi & 511u) - kIPad, gy = y0 - 1 + (int)(pi >> 9);
                if (in_core(x, gy)) append((uint32_t)x, (uint32_t)gy, (uint32_t)list_ang[i], list_score[i]);
            }
        }
    } else {
        // some queue was full (a dense tile): the queues are drained and useless -- three resident rows at a time
        __syncthreads();  // every wave has read the flag and is done with the queues' storage
        dense_rows();
    }

    __syncthreads();
    if (tid == 0) seg_counts[slot] = *c_count;  // raw count of the tile (may exceed seg_cap)
}

// 64-bit selection key of a candidate (IM-8): larger = better.  Scores are positive, so their bit patterns order
// like the values; ties go to the smaller (octave, y, x).
__device__ __forceinline__ unsigned long long select_key(const uint4 rec, float score) {
    const uint32_t pos = (rec.w << 28) | (rec.y << 14) | rec.x;  // x, y < 2^14 (checked at create)
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xffffffffu - pos);
}

// Per frame: counts[f] = candidates before the cut; thr_key[f] = smallest key that is kept (0: keep everything);
// seg_before[slot] = where the slot's kept keypoints start in the frame's final list.  One workgroup per frame.
__global__ __launch_bounds__(1024) void k_select_i(const uint32_t* __restrict__ seg_counts,
                                                   const CornerData* __restrict__ segments,
                                                   const float* __restrict__ seg_scores, uint32_t n_slots,
                                                   uint32_t seg_cap, uint32_t cap, uint32_t* __restrict__ counts,
                                                   unsigned long long* __restrict__ thr_key,
                                                   uint32_t* __restrict__ seg_before) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t red[16];
    __shared__ unsigned long long sel_prefix;
    __shared__ uint32_t sel_want;
    const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t* sc = seg_counts + (size_t)f * n_slots;
    uint32_t* sb = seg_before + (size_t)f * n_slots;
    auto block_sum = [&](uint32_t v) {  // sum over the 1024 threads, returned to all
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
        __syncthreads();
        if (lane == 0u) red[wv] = v;
        __syncthreads();
        uint32_t t = 0;
        for (int k = 0; k < 16; k++) t += red[k];
        return t;
    };
    uint32_t raw = 0, stored = 0;
    for (uint32_t s = tid; s < n_slots; s += 1024u) {
        raw += sc[s];
        stored += min(sc[s], seg_cap);
    }
    raw = block_sum(raw);
    stored = block_sum(stored);
    if (tid == 0u) counts[f] = raw;
    unsigned long long kth = 0ull;
    if (stored > cap) {  // uniform
        if (tid == 0u) {
            sel_prefix = 0ull;
            sel_want = cap;
        }
        unsigned long long known = 0ull;
        for (int pass = 7; pass >= 0; pass--) {
            if (tid < 256u) hist[tid] = 0u;
            __syncthreads();
            const unsigned long long prefix = sel_prefix;
            for (uint32_t s = wv; s < n_slots; s += 16u) {  // one wave per slot
                const uint32_t n = min(sc[s], seg_cap);
                const size_t base = ((size_t)f * n_slots + s) * seg_cap;
                // The key's high word is the score's bit pattern: the four passes over it read 4 bytes per candidate, not the 20 of record +
                // score, and the four passes over the low word (the position: ties in the score) read a record only where the score IS the
                // selected one -- 32 instead of 160 bytes per candidate over the eight passes (k_select_i 0.78 ms per 256 overflowing frames before).
                // (four loads in flight per lane: the pass is a chain of memory round trips otherwise -- one wave walks a slot, and the
                // LDS atomics between the loads keep hipcc from overlapping the iterations)
                for (uint32_t j0 = 0; j0 < n; j0 += 256u) {
                    uint32_t sb4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u + lane;
                        sb4[u] = j < n ? __float_as_uint(seg_scores[base + j]) : 0u;  // 0: no candidate (a corner's score is positive)
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u + lane, sbits = sb4[u];
                        if (j >= n) continue;
                        if (pass >= 4) {
                            const unsigned long long k = (unsigned long long)sbits << 32;
                            if ((k & known) == prefix) atomicAdd(&hist[(sbits >> (8 * (pass - 4))) & 255u], 1u);
                        } else if (sbits == (uint32_t)(prefix >> 32)) {
                            const unsigned long long k = select_key(*reinterpret_cast<const uint4*>(&segments[base + j]), seg_scores[base + j]);
                            if ((k & known) == prefix) atomicAdd(&hist[(uint32_t)(k >> (8 * pass)) & 255u], 1u);
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0u) {
                uint32_t acc = 0, want = sel_want, digit = 0;
                for (int b = 255; b >= 0; b--) {
                    if (acc + hist[b] >= want) {
                        digit = (uint32_t)b;
                        break;
                    }
                    acc += hist[b];
                }
                sel_want = want - acc;
                sel_prefix = prefix | ((unsigned long long)digit << (8 * pass));
            }
            known |= 0xffull << (8 * pass);
            __syncthreads();
        }
        kth = sel_prefix;  // exactly `cap` keys are >= kth (keys are unique)
    }
    if (tid == 0u) thr_key[f] = kth;
    // kept keypoints per slot -> exclusive prefix over the slots (in slot order)
    __shared__ uint32_t carry;
    if (tid == 0u) carry = 0u;
    __syncthreads();
    if (kth != 0ull) {
        // kept keypoints of every slot, a wave per slot (lanes over the candidates, four loads in flight each; the score decides, the
        // record is read only on a tie with the threshold's score), parked in the slot's place of the prefix array
        const uint32_t kth_score = (uint32_t)(kth >> 32);
        for (uint32_t s = wv; s < n_slots; s += 16u) {
            const uint32_t n = min(sc[s], seg_cap);
            const size_t base = ((size_t)f * n_slots + s) * seg_cap;
            uint32_t kept = 0;
            for (uint32_t j0 = 0; j0 < n; j0 += 256u) {
                uint32_t sb4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + 64u * (uint32_t)u + lane;
                    sb4[u] = j < n ? __float_as_uint(seg_scores[base + j]) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + 64u * (uint32_t)u + lane;
                    if (j >= n) continue;
                    if (sb4[u] != kth_score)
                        kept += sb4[u] > kth_score ? 1u : 0u;
                    else
                        kept += select_key(*reinterpret_cast<const uint4*>(&segments[base + j]), seg_scores[base + j]) >= kth ? 1u : 0u;
                }
            }
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) kept += __shfl_xor(kept, sh);
            if (lane == 0u) sb[s] = kept;
        }
        __syncthreads();  // (workgroup-scope: the counts are read back below by other threads of this workgroup)
    }
    for (uint32_t s0 = 0; s0 < n_slots; s0 += 1024u) {
        const uint32_t s = s0 + tid;
        uint32_t kept = 0;
        if (s < n_slots) kept = kth == 0ull ? min(sc[s], seg_cap) : sb[s];
        // block-wide exclusive scan of `kept`
        uint32_t incl = kept;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        __syncthreads();
        if (lane == 63u) red[wv] = incl;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
        for (uint32_t k = 0; k < 16u; k++) {
            if (k < wv) wave_off += red[k];
            total += red[k];
        }
        if (s < n_slots) sb[s] = carry + wave_off + incl - kept;
        __syncthreads();
        if (tid == 0u) carry += total;
        __syncthreads();
    }
}

// Rotated BRIEF-256 of the kept keypoints (IM-6: R(+theta), full-circle table).  A workgroup takes a column of
// kIBriefStack vertically adjacent tiles: the blur window their keypoints can sample, widened to 16-byte columns
// ([y0-18, y0+16*stack+18) x [cx0-24, cx0+tw+24), texels outside the level = 0, CRD-6), is staged in LDS once with
// 16-byte loads (a single tile's window would be 3.25x its own rows, the stack's is 1.56x); the kept keypoints of
// each tile are dealt to the eight waves round-robin, one wave64 per keypoint, lane l evaluates tests l, 64+l,
// 128+l, 192+l.
constexpr int kIBriefStack = 4;
constexpr bool kIBriefTable = true;  // rotated points from BriefTables::rot (false: computed per keypoint, for A/B measurements)
constexpr int kIBriefThreads = 512;
constexpr int kIBriefRowsMax = kFrontRows * kIBriefStack + 2 * kBriefHalo;  // 100
constexpr int kIBriefApronX = 24;                                            // >= 18, multiple of 8
struct IBriefGeom {
    uint32_t n_slots, seg_cap;
    uint32_t slot_base[kMaxLevels + 1];
    uint32_t tw[kMaxLevels], n_ct[kMaxLevels], n_bands[kMaxLevels];
    uint32_t group_base[kMaxLevels + 1];  // first stack of each level; [depth] = stacks per frame
    uint32_t pitch;        // LDS row pitch in halfs: kITileW + 2 * kIBriefApronX
    uint32_t xcd_swizzle;  // all stacks of a frame on one XCD: its blur planes stay in that L2
    uint32_t phase_mask;   // timing experiments only: bit0 the window is staged, bit1 the keypoints are described
    uint32_t angle_bins;   // OrbOptions::angle_bins (IM-6b): 0 = the table has one entry per milliradian code, N = one per angle bin
};

__global__ __launch_bounds__(kIBriefThreads) void k_brief_i(const uint16_t* __restrict__ blur, Pyramid pyr,
                                                            IBriefGeom bg, const uint32_t* __restrict__ seg_counts,
                                                            const uint32_t* __restrict__ seg_before,
                                                            const unsigned long long* __restrict__ thr_key,
                                                            const CornerData* __restrict__ segments,
                                                            const float* __restrict__ seg_scores,
                                                            CornerData* __restrict__ corners, uint32_t cap,
                                                            CornerDescriptor* __restrict__ descriptors, BriefTables tab) {
    constexpr int NT = kIBriefThreads, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) uint16_t win[];  // up to kIBriefRowsMax x pitch
    __shared__ uint4 kept_rec[256];
    __shared__ uint32_t wave_kept[4];
    __shared__ uint32_t kept_k[256];                   // final index of the kept keypoint
    __shared__ uint32_t tile_first[kIBriefStack];     // kept entries of the chunk in front of the tile's first entry
    __shared__ uint32_t tile_run[kIBriefStack];       // kept entries of the tile in earlier chunks
    const uint32_t n_groups = bg.group_base[pyr.depth];
    uint32_t group, frame;
    if (bg.xcd_swizzle) {
        const uint32_t xcd = blockIdx.x & 7u, s2 = blockIdx.x >> 3;
        frame = (s2 / n_groups) * 8u + xcd;
        group = s2 % n_groups;
    } else {
        frame = blockIdx.x / n_groups;
        group = blockIdx.x % n_groups;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t lvl = 0;
    for (uint32_t m = 1; m < pyr.depth; m++)
        if (group >= bg.group_base[m]) lvl = m;
    const uint32_t rel = group - bg.group_base[lvl];
    const int band0 = (int)(rel / bg.n_ct[lvl]) * kIBriefStack, ct = (int)(rel % bg.n_ct[lvl]);
    const int nb = min(kIBriefStack, (int)bg.n_bands[lvl] - band0);  // tiles in this stack
    const size_t sidx0 = (size_t)frame * bg.n_slots + bg.slot_base[lvl] + (size_t)band0 * bg.n_ct[lvl] + ct;
    // The stack's tiles as ONE list (see below): counts, list starts and the cut's key first -- one memory round trip --, then the
    // first chunk's records, whose round trip runs under the staging of the window.
    uint32_t n_t[kIBriefStack], o_t[kIBriefStack + 1], before_t[kIBriefStack];
    o_t[0] = 0;
#pragma unroll
    for (int t = 0; t < kIBriefStack; t++) {
        const size_t sidx = sidx0 + (size_t)min(t, nb - 1) * bg.n_ct[lvl];
        n_t[t] = t < nb ? min(seg_counts[sidx], bg.seg_cap) : 0u;
        before_t[t] = seg_before[sidx];
        o_t[t + 1] = o_t[t] + n_t[t];
    }
    const unsigned long long kth = thr_key[frame];
    const uint32_t n_all = o_t[kIBriefStack];
    if (n_all == 0u) return;  // uniform
    static_assert(kIBriefStack == 4, "tile_of / entry_of spell out four tiles");
    const uint32_t o1 = o_t[1], o2 = o_t[2], o3 = o_t[3];
    auto tile_of = [o1, o2, o3](uint32_t c) { return (c >= o1 ? 1u : 0u) + (c >= o2 ? 1u : 0u) + (c >= o3 ? 1u : 0u); };
    const size_t seg_stride = (size_t)bg.n_ct[lvl] * bg.seg_cap, seg_base = sidx0 * bg.seg_cap;
    auto entry_of = [o1, o2, o3, seg_stride, seg_base](uint32_t c, uint32_t t) {  // where entry c of the concatenated list lies in the segment arrays
        const uint32_t ot = t == 0u ? 0u : (t == 1u ? o1 : (t == 2u ? o2 : o3));
        return seg_base + (size_t)t * seg_stride + (c - ot);
    };
    uint4 rec_first = make_uint4(0u, 0u, 0u, 0u);
    float score_first = 0.0f;
    if (tid < 256u && tid < n_all) {
        const size_t at = entry_of(tid, tile_of(tid));
        rec_first = *reinterpret_cast<const uint4*>(&segments[at]);
        score_first = seg_scores[at];
    }
    const int w = (int)pyr.w[lvl], h = (int)pyr.h[lvl];
    const int wy0 = band0 * kFrontRows - kBriefHalo, wx0 = ct * (int)bg.tw[lvl] - kIBriefApronX;
    const int tw = min((int)bg.tw[lvl], w - ct * (int)bg.tw[lvl]);
    const int pitch = (int)bg.pitch;
    const uint16_t* plane = blur + (size_t)frame * pyr.stride + pyr.off[lvl];
    if (bg.phase_mask & 1u)
    {   // Stage the window in 8-texel groups.  A thread keeps its column group and walks down the rows (k_front's phase A): what
        // kind of group it is -- inside the level (one 16-byte buffer load with the level's plane as the buffer: a row above or
        // below the level has an offset outside it and reads zeros), outside (zeros, no load) or across its edge (texel by
        // texel) -- is decided once, a step is three additions; five loads per thread are in flight.
        const int groups = (tw + 2 * kIBriefApronX + 7) >> 3;  // <= 46
        const int n_rows = nb * kFrontRows + 2 * kBriefHalo;
        const int rpp = NT / groups;
        const int ty = (int)(((float)tid + 0.5f) * (1.0f / (float)groups)), tx = (int)tid - __mul24(ty, groups);
        const int gx = wx0 + tx * 8;
        const bool lane_ok = ty < rpp;
        const bool inside = (w & 7) == 0 && gx >= 0 && gx + 8 <= w, outside = gx + 8 <= 0 || gx >= w;
        const __amdgpu_buffer_rsrc_t plane_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(plane), 0, (int)((uint32_t)w * (uint32_t)h * 2u), kBufferWord3Raw);
        int gy_w = wy0 + ty;
        uint32_t off_w = (uint32_t)(__mul24(gy_w, w) + (inside ? gx : 0)) * 2u;
        const uint32_t off_step = (uint32_t)__mul24(rpp, w) * 2u;
        int dst_w = __mul24(ty, pitch) + tx * 8;
        const int dst_step = rpp * pitch;
        constexpr int U = 5;
        for (int rb = ty; rb < n_rows; rb += rpp * U) {
            uint4 v[U];
            int dst[U], gys[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (inside) v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)off_w, 0, 0));
                dst[u] = dst_w;
                gys[u] = gy_w;
                off_w += off_step;
                dst_w += dst_step;
                gy_w += rpp;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!(lane_ok && rb + u * rpp < n_rows)) continue;
                uint4 o = v[u];
                if (!inside && !outside && gys[u] >= 0 && gys[u] < h) {  // the level's edge runs through the group (or its rows are not 16-byte aligned)
                    const uint16_t* row = plane + (size_t)(uint32_t)__mul24(gys[u], w);
                    uint32_t e[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = (gx + k >= 0 && gx + k < w) ? (uint32_t)row[gx + k] : 0u;
                    o = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                }
                *reinterpret_cast<uint4*>(&win[dst[u]]) = o;
            }
        }
    }
    CornerData* out_kp = corners + (size_t)frame * cap;
    uint32_t* out_desc = reinterpret_cast<uint32_t*>(descriptors + (size_t)frame * cap);
    // The stack's tiles as ONE list: entry c of the concatenated segments belongs to tile t (c in [o_t, o_t + n_t)), and a kept
    // entry's final index is seg_before[tile] + its rank among the tile's kept entries (segment order).  Chunks of 256
    // entries: the first four waves decide "kept" and compact records and final indices into kept_rec / kept_k, then all
    // eight waves describe them.  (Tile by tile -- a dependent global load, three barriers and one or two turns of the
    // waves per tile -- the kernel spent its time waiting: 0.13 ms of its 0.57 with neither window nor keypoints.)
    uint32_t pat[4] = {0u, 0u, 0u, 0u};
    if constexpr (!kIBriefTable) {
#pragma unroll
        for (int e = 0; e < 4; e++) pat[e] = tab.pattern[64u * (uint32_t)e + lane];
    }
    if (tid < (uint32_t)kIBriefStack) tile_run[tid] = 0u;
    {
        for (uint32_t c0 = 0; c0 < n_all; c0 += 256u) {
            const uint32_t c = c0 + tid;
            uint4 rec = rec_first;
            float score = score_first;
            bool kept = false;
            uint32_t t = 0;
            if (tid < 256u && c < n_all) {
                t = tile_of(c);
                if (c0 != 0u) {  // (the first chunk's records were fetched in front of the window)
                    const size_t at = entry_of(c, t);
                    rec = *reinterpret_cast<const uint4*>(&segments[at]);
                    score = seg_scores[at];
                }
                kept = kth == 0ull || select_key(rec, score) >= kth;
            }
            const uint64_t mask = __ballot(kept);
            __syncthreads();  // previous chunk's kept_rec consumed; (first chunk) the window is complete
            if (lane == 0u && wv < 4u) wave_kept[wv] = (uint32_t)__builtin_popcountll(mask);
            __syncthreads();
            uint32_t off = 0;
            for (uint32_t k = 0; k < wv && k < 4u; k++) off += wave_kept[k];
            const uint32_t chunk_total = wave_kept[0] + wave_kept[1] + wave_kept[2] + wave_kept[3];
            const uint32_t pfx = off + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));  // kept entries of the chunk before this one
            uint32_t ot = o_t[0], nt = n_t[0], bt = before_t[0];
#pragma unroll
            for (int q = 1; q < kIBriefStack; q++) {
                ot = t == (uint32_t)q ? o_t[q] : ot;
                nt = t == (uint32_t)q ? n_t[q] : nt;
                bt = t == (uint32_t)q ? before_t[q] : bt;
            }
            const bool mine = tid < 256u && c < n_all;
            if (mine && c == max(ot, c0)) tile_first[t] = pfx;  // the tile's first entry in this chunk
            __syncthreads();
            if (kept) {
                kept_rec[pfx] = rec;
                kept_k[pfx] = bt + tile_run[t] + (pfx - tile_first[t]);
            }
            __syncthreads();
            if (mine && (c + 1u == ot + nt || tid == 255u)) tile_run[t] += pfx + (kept ? 1u : 0u) - tile_first[t];  // the tile's last entry in this chunk
            auto rot_of = [&](uint32_t r) {
                uint32_t code = min((uint32_t)__builtin_amdgcn_readfirstlane(kept_rec[r].z), (uint32_t)(ORB_ANGLE_STEPS_FULL - 1));
                if constexpr (kIBriefTable) {
                    // IM-6b: the table holds one entry per angle BIN (1024 bins: 1 MB, resident in every XCD's L2 where the 6284 codes'
                    // 6.4 MB are not); the bin of a code is code * bins / 6284 in integers (scalar arithmetic: the code is wave-uniform)
                    const uint32_t entry = bg.angle_bins ? (code * bg.angle_bins) / (uint32_t)ORB_ANGLE_STEPS_FULL : code;
                    return tab.rot[(size_t)entry * 64u + lane];
                } else {
                    code = binned_angle_code(code, bg.angle_bins);  // the rotation on the spot (k_rot_table's arithmetic): no table traffic, 24 more vector instructions per test
                    const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;
                    uint32_t w[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t pk = pat[e];
                        const float pax = (float)(int8_t)(pk & 255u), pay = (float)(int8_t)((pk >> 8) & 255u);
                        const float pbx = (float)(int8_t)((pk >> 16) & 255u), pby = (float)(int8_t)(pk >> 24);
                        const float a0 = ct * pax, a1 = nst * pay, a2 = st * pax, a3 = ct * pay;
                        const float b0 = ct * pbx, b1 = nst * pby, b2 = st * pbx, b3 = ct * pby;
                        const float rax = a0 + a1, ray = a2 + a3, rbx = b0 + b1, rby = b2 + b3;
                        const int oa = 2 * ((int)ray * pitch + (int)rax), ob = 2 * ((int)rby * pitch + (int)rbx);
                        w[e] = ((uint32_t)oa & 0xffffu) | ((uint32_t)ob << 16);
                    }
                    return make_uint4(w[0], w[1], w[2], w[3]);
                }
            };
            // U keypoints per wave and turn: a keypoint is a chain of latencies (record from LDS, table entry from L2, eight
            // samples from LDS, ballots), not arithmetic -- with one at a time the kernel waited 60 % of its wave-cycles at four
            // waves per SIMD (the window leaves room for two workgroups per CU).  The chains of U keypoints are independent.
            constexpr uint32_t U = 4;
            uint4 tt_next[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) tt_next[u] = wv + u * NW < chunk_total ? rot_of(wv + u * NW) : make_uint4(0u, 0u, 0u, 0u);
            for (uint32_t r = wv; r < chunk_total && (bg.phase_mask & 2u); r += NW * U) {
                uint4 kr[U], tt[U];
                uint32_t kk[U];
                bool ok[U];  // wave-uniform
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t ru = r + u * NW;
                    kr[u] = kept_rec[min(ru, 255u)];
                    kk[u] = kept_k[min(ru, 255u)];
                    ok[u] = ru < chunk_total && kk[u] < cap;  // past the capacity: dropped (IM-8's cut keeps the list below it)
                    tt[u] = tt_next[u];
                }
#pragma unroll
                for (uint32_t u = 0; u < U; u++)
                    if (r + (U + u) * NW < chunk_total) tt_next[u] = rot_of(r + (U + u) * NW);
                uint64_t bal[U][4];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t tw[4] = {tt[u].x, tt[u].y, tt[u].z, tt[u].w};
                    // the keypoint's texel inside the window (wave-uniform); a sample adds 2 * (dy * pitch + dx) bytes
                    const int kbase = ok[u] ? __builtin_amdgcn_readfirstlane(((int)kr[u].y - wy0) * pitch + ((int)kr[u].x - wx0)) : 0;
                    const uint8_t* const centre = reinterpret_cast<const uint8_t*>(win + kbase);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int oa = ok[u] ? rot_a(tw[e]) : 0, ob = ok[u] ? rot_b(tw[e]) : 0;
                        const uint32_t va = *reinterpret_cast<const uint16_t*>(centre + oa);
                        const uint32_t vb = *reinterpret_cast<const uint16_t*>(centre + ob);
                        bal[u][e] = __ballot(va > vb);  // non-negative f16: bit patterns order like the values
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    if (!ok[u]) continue;
                    const uint32_t k = kk[u];
                    if (lane < 8u) {
                        const uint64_t srcw = lane < 2u ? bal[u][0] : (lane < 4u ? bal[u][1] : (lane < 6u ? bal[u][2] : bal[u][3]));
                        out_desc[(size_t)k * 8u + lane] = (uint32_t)(srcw >> ((lane & 1u) * 32u));
                    } else if (lane == 8u) {
                        *reinterpret_cast<uint4*>(&out_kp[k]) = kr[u];
                    }
                }
            }
        }
    }
}

}  // namespace orb
