// orb_kernels_staged.h -- one kernel per reference stage, batched over frames (blockIdx.z).
//
// This is the straightforward pipeline: every R16Float plane the reference materialises
// (orb.rs:221-409: image_hierarchy, blur_tmp_hierarchy, blur_hierarchy) exists in HBM here
// too, except blur_tmp which lives in LDS.  It is the cross-check for the fused pipeline
// (orb_kernels_fused.h) and the path taken for shapes the fused kernels do not cover.
#pragma once
#include "orb_device.h"
#include "orb_tables.h"

namespace orb {

// ---------------------------------------------------------------------------------------------
// K1  grayscale.wgsl:12-38 -- RGBA8 -> luminance f16 of the vertically mirrored row.
// One thread = 4 horizontally adjacent pixels: one 16-byte load, one 8-byte store.
// grid: (ceil(W/4/256), H, frames)
// ---------------------------------------------------------------------------------------------
// INTENDED (IM-1, not in the reference): BT.601 red weight and no mirror.
template <bool INTENDED>
__global__ __launch_bounds__(256) void k_grayscale(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                   uint16_t* __restrict__ gray, Pyramid pyr, uint32_t fp = 0u) {
    // fp (OrbOptions::fp_contract, CRD-13; the reference's detector only): which of the four forms the dot product takes
    const int form = lum_form(fp);
    // (to_half_strict: the contracted forms end in an fma, which must not be folded into the conversion)
    auto lum16 = [form](uint32_t v) { return to_half_strict(INTENDED ? luminance_601(v) : (form ? luminance_fp(v, form) : luminance(v))); };
    const uint32_t W = pyr.w[0], H = pyr.h[0];
    const uint32_t y = blockIdx.y, f = blockIdx.z;
    const uint32_t x0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (x0 >= W) return;
    const uint8_t* src_row = frames + (size_t)f * frame_bytes + (size_t)(INTENDED ? y : H - 1u - y) * W * 4u;
    uint16_t* dst_row = gray + (size_t)f * pyr.stride + pyr.off[0] + (size_t)y * W;
    if (x0 + 4u <= W && (W & 3u) == 0u) {
        const uint4 px = *reinterpret_cast<const uint4*>(src_row + (size_t)x0 * 4u);
        ushort4 out;
        out.x = half_bits(lum16(px.x));
        out.y = half_bits(lum16(px.y));
        out.z = half_bits(lum16(px.z));
        out.w = half_bits(lum16(px.w));
        *reinterpret_cast<ushort4*>(dst_row + x0) = out;
    } else {
        for (uint32_t x = x0; x < W && x < x0 + 4u; x++) {
            uint32_t v = *reinterpret_cast<const uint32_t*>(src_row + (size_t)x * 4u);
            dst_row[x] = half_bits(lum16(v));
        }
    }
}

// Y8 input variant (ORB_FLAG_INPUT_Y8; not in the reference's code, its roadmap item README.md:42): the frame is one
// byte per pixel and the grey image is that sample, gray(x,y) = f16(Y(x, H-1-y)/255) -- same mirror as above (Q2).
__global__ __launch_bounds__(256) void k_grayscale_y8(const uint8_t* __restrict__ frames, size_t frame_bytes,
                                                      uint16_t* __restrict__ gray, Pyramid pyr) {
    const uint32_t W = pyr.w[0], H = pyr.h[0];
    const uint32_t y = blockIdx.y, f = blockIdx.z;
    const uint32_t x0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (x0 >= W) return;
    const uint8_t* src_row = frames + (size_t)f * frame_bytes + (size_t)(H - 1u - y) * W;
    uint16_t* dst_row = gray + (size_t)f * pyr.stride + pyr.off[0] + (size_t)y * W;
    for (uint32_t x = x0; x < W && x < x0 + 4u; x++) dst_row[x] = half_bits(to_half((float)src_row[x] / 255.0f));
}

// ---------------------------------------------------------------------------------------------
// K2  blit.wgsl:17-36 -- mip m from mip m-1 (CRD-4).  One thread per target texel.
// grid: (ceil(wd/64), ceil(hd/(4*kMipRows)), frames), block (64,4)
// ---------------------------------------------------------------------------------------------
constexpr int kMipRows = 4;  // target rows per thread
// rx, ry: source/target size ratios of the generic case, (float)ws / (float)wd and (float)hs / (float)hd, divided once on the
// host (IEEE binary32 division on both sides: the same value the kernel used to compute per thread).
__global__ __launch_bounds__(256) void k_mip(uint16_t* __restrict__ gray, Pyramid pyr, uint32_t m, float rx, float ry, float wq = 0.0f) {
    const uint32_t wd = pyr.w[m], hd = pyr.h[m], ws = pyr.w[m - 1], hs = pyr.h[m - 1];
    const uint32_t x = blockIdx.x * 64u + threadIdx.x;
    if (x >= wd) return;
    const uint16_t* src = gray + (size_t)blockIdx.z * pyr.stride + pyr.off[m - 1];
    uint16_t* dst = gray + (size_t)blockIdx.z * pyr.stride + pyr.off[m];
    // a thread takes kMipRows target rows (y, y + 4, ...): their loads are issued together
    const uint32_t yb = blockIdx.y * (4u * (uint32_t)kMipRows) + threadIdx.y;
    if (ws == 2u * wd && hs == 2u * hd) {
#pragma unroll
        for (int r = 0; r < kMipRows; r++) {
            const uint32_t y = yb + 4u * (uint32_t)r;
            if (y >= hd) break;
            const uint32_t two = *reinterpret_cast<const uint32_t*>(src + (size_t)(2u * y) * ws + 2u * x);
            const uint32_t two2 = *reinterpret_cast<const uint32_t*>(src + (size_t)(2u * y + 1u) * ws + 2u * x);
            float a = from_half(bits_half((uint16_t)(two & 0xffffu))), b = from_half(bits_half((uint16_t)(two >> 16)));
            float c = from_half(bits_half((uint16_t)(two2 & 0xffffu))), d = from_half(bits_half((uint16_t)(two2 >> 16)));
            float top = a + b;
            float bot = c + d;
            dst[(size_t)y * wd + x] = half_bits(to_half((top + bot) * 0.25f));
        }
        return;
    }
    const float sx = ((float)x + 0.5f) * rx - 0.5f;
    const float fx0 = __builtin_floorf(sx);
    const float fx = sampler_weight(sx - fx0, wq);  // wq: OrbOptions::sampler_weight_bits (0: exact, CRD-4)
    const int ix = (int)fx0;
    const int x0 = min(max(ix, 0), (int)ws - 1), x1 = min(max(ix + 1, 0), (int)ws - 1);
    uint32_t ta[kMipRows], tb[kMipRows], tc[kMipRows], td[kMipRows];
    float fy[kMipRows];
#pragma unroll
    for (int r = 0; r < kMipRows; r++) {
        const uint32_t y = min(yb + 4u * (uint32_t)r, hd - 1u);  // rows past the level repeat the last one (not stored)
        const float sy = ((float)y + 0.5f) * ry - 0.5f;
        const float fy0 = __builtin_floorf(sy);
        fy[r] = sampler_weight(sy - fy0, wq);
        const int iy = (int)fy0;
        const int y0 = min(max(iy, 0), (int)hs - 1), y1 = min(max(iy + 1, 0), (int)hs - 1);
        ta[r] = src[(size_t)y0 * ws + x0], tb[r] = src[(size_t)y0 * ws + x1];
        tc[r] = src[(size_t)y1 * ws + x0], td[r] = src[(size_t)y1 * ws + x1];
    }
#pragma unroll
    for (int r = 0; r < kMipRows; r++) {
        const uint32_t y = yb + 4u * (uint32_t)r;
        if (y >= hd) break;
        const float a = from_half(bits_half((uint16_t)ta[r])), b = from_half(bits_half((uint16_t)tb[r]));
        const float c = from_half(bits_half((uint16_t)tc[r])), d = from_half(bits_half((uint16_t)td[r]));
        const float dab = b - a;
        const float top = a + fx * dab;
        const float dcd = d - c;
        const float bot = c + fx * dcd;
        const float dtb = bot - top;
        dst[(size_t)y * wd + x] = half_bits(to_half(top + fy[r] * dtb));
    }
}

// ---------------------------------------------------------------------------------------------
// K3+K4  gaussian_blur_x.wgsl:45-61 twice (orb.rs:432-466; both pipelines use the X shader,
// orb.rs:399-402).  Each pass mirrors v (gaussian_blur_x.wgsl:32-41), so after two passes the
// rows line up with the grey image again and blur(.,y) depends on grey row y alone: pass(pass(row)).
// One block = one row of one level of one frame; row and the f16-rounded intermediate in LDS.
// grid: (h_level, 1, frames), dynamic LDS = 2 * w * 2 bytes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float blur_point(const half_t* row, uint32_t x, uint32_t w, float wq, uint32_t fp = 0u) {
    const bool contract = (fp & kFpBlur) != 0u;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        BlurTap t = blur_tap(x, w, kBlurOff[i], wq);
        float t0 = from_half(row[t.i0]), t1 = from_half(row[t.i1]);
        float d = t1 - t0;
        float s = t0 + t.f * d;  // the sampler's arithmetic, not the shader's: CRD-5 either way
        if (contract) {          // CRD-13: `result += sample * weight` as one fma
            acc = __builtin_fmaf(s, kBlurWgt[i], acc);
        } else {
            float ws = s * kBlurWgt[i];
            acc = acc + ws;
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_blur_rows(const uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                                   Pyramid pyr, uint32_t m, float wq, uint32_t fp = 0u) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t w = pyr.w[m];
    half_t* row = reinterpret_cast<half_t*>(lds_raw);
    half_t* tmp = row + w;
    const uint32_t y = blockIdx.x;
    const size_t base = (size_t)blockIdx.z * pyr.stride + pyr.off[m] + (size_t)y * w;
    for (uint32_t x = threadIdx.x; x < w; x += 256u) row[x] = bits_half(gray[base + x]);
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < w; x += 256u) tmp[x] = to_half_strict(blur_point(row, x, w, wq, fp));
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < w; x += 256u) blur[base + x] = half_bits(to_half_strict(blur_point(tmp, x, w, wq, fp)));
}

// ---------------------------------------------------------------------------------------------
// "intended" mode IM-3 (not in the reference, which runs its X shader twice with UV-unit offsets): separable
// 7-tap Gaussian, X pass then Y pass, both f16-rounded.  One block = a 256x16 tile: the grey tile with a 3-px
// apron (clamp-to-edge) goes to LDS as f16 (16-byte loads inside the level), the X pass (4 outputs per thread
// from 10 inputs) fills (16+6) x 256 f16 intermediate values in LDS, the Y pass (2 columns x 8 rows per thread
// from 14 rows) writes 4-byte pairs.
// grid: (ceil(w/256), ceil(h/16), frames)
// ---------------------------------------------------------------------------------------------
constexpr int kGaussTW = 256, kGaussTH = 16, kGaussPad = 8;
constexpr int kGaussPitch = kGaussTW + 2 * kGaussPad;  // grey tile: LDS column kGaussPad <-> image column bx

__global__ __launch_bounds__(256) void k_gauss(const uint16_t* __restrict__ gray, uint16_t* __restrict__ blur,
                                               Pyramid pyr, uint32_t m) {
    constexpr int TW = kGaussTW, TH = kGaussTH, R = 3, ROWS = TH + 2 * R;
    __shared__ __attribute__((aligned(16))) uint16_t src[ROWS * kGaussPitch];
    __shared__ __attribute__((aligned(16))) uint16_t mid[ROWS * TW];
    const int w = (int)pyr.w[m], h = (int)pyr.h[m];
    const size_t base = (size_t)blockIdx.z * pyr.stride + pyr.off[m];
    const uint16_t* plane = gray + base;
    const int bx = (int)blockIdx.x * TW, by = (int)blockIdx.y * TH;
    const int tid = (int)threadIdx.x;
    {   // 8-texel groups: columns [bx - 8, bx + TW + 8), rows [by - 3, by + TH + 3), indices clamped to the level
        constexpr int G = kGaussPitch / 8;
        const bool vec_ok = (w & 7) == 0;
        for (int i = tid; i < ROWS * G; i += 256) {
            const int r = i / G, g = i - r * G;
            const int gy = min(max(by + r - R, 0), h - 1), gx = bx - kGaussPad + g * 8;
            const uint16_t* row = plane + (size_t)gy * w;
            uint4 v;
            if (vec_ok && gx >= 0 && gx + 8 <= w) {
                v = *reinterpret_cast<const uint4*>(row + gx);
            } else {
                uint32_t e[8];
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = row[min(max(gx + k, 0), w - 1)];
                v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
            *reinterpret_cast<uint4*>(&src[r * kGaussPitch + g * 8]) = v;
        }
    }
    __syncthreads();
    // X pass: item = 4 consecutive columns of one row; inputs x-3 .. x+6 = 12 halfs from three 8-byte reads
    for (int i = tid; i < ROWS * (TW / 4); i += 256) {
        const int r = i / (TW / 4), x = (i - r * (TW / 4)) * 4;
        const uint16_t* p = &src[r * kGaussPitch + kGaussPad + x - 4];
        const uint2 q0 = *reinterpret_cast<const uint2*>(p), q1 = *reinterpret_cast<const uint2*>(p + 4),
                    q2 = *reinterpret_cast<const uint2*>(p + 8);
        const uint32_t wds[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
        float f[12];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            f[2 * k] = from_half(bits_half((uint16_t)(wds[k] & 0xffffu)));
            f[2 * k + 1] = from_half(bits_half((uint16_t)(wds[k] >> 16)));
        }
        uint16_t o[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {  // output column x + c: taps f[c + 1 .. c + 7]
            const float t[7] = {f[c + 1], f[c + 2], f[c + 3], f[c + 4], f[c + 5], f[c + 6], f[c + 7]};
            o[c] = half_bits(to_half(gauss7(t)));
        }
        *reinterpret_cast<uint2*>(&mid[r * TW + x]) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
    }
    __syncthreads();
    // Y pass: item = 2 columns x 8 rows; 14 intermediate rows per item
    {
        const int x = (tid & 127) * 2, r0 = (tid >> 7) * 8;  // 128 column pairs x 2 row groups = 256 threads
        float lo[14], hi[14];
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(&mid[(r0 + k) * TW + x]);
            lo[k] = from_half(bits_half((uint16_t)(v & 0xffffu)));
            hi[k] = from_half(bits_half((uint16_t)(v >> 16)));
        }
        const int gx = bx + x;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int gy = by + r0 + k;
            const float ta[7] = {lo[k], lo[k + 1], lo[k + 2], lo[k + 3], lo[k + 4], lo[k + 5], lo[k + 6]};
            const float tb[7] = {hi[k], hi[k + 1], hi[k + 2], hi[k + 3], hi[k + 4], hi[k + 5], hi[k + 6]};
            const uint32_t oa = half_bits(to_half(gauss7(ta))), ob = half_bits(to_half(gauss7(tb)));
            if (gy < h && gx < w) {
                uint16_t* out = blur + base + (size_t)gy * w + gx;
                if (gx + 1 < w && (w & 1) == 0)
                    *reinterpret_cast<uint32_t*>(out) = oa | (ob << 16);
                else {
                    out[0] = (uint16_t)oa;
                    if (gx + 1 < w) out[1] = (uint16_t)ob;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5  fast.wgsl:62-159 for one octave (one dispatch of orb.rs:509-519).
// 16x16 pixels per block, tile + 3-pixel halo staged in LDS as f32; out-of-level texels are 0
// (CRD-6).  The reference's per-thread LDS atomic + one global atomic per workgroup
// (fast.wgsl:123-141) becomes a wave64 ballot/prefix and one global atomic per wave.
// grid: (ceil(gw/16), ceil(gh/16), frames) with gw,gh the reference's 8-rounded dispatch size.
// ---------------------------------------------------------------------------------------------
// Returns the slot the record went to, or ~0u (not a corner / list full).
__device__ __forceinline__ uint32_t append_corners(bool is_corner, uint32_t x, uint32_t y, uint32_t angle, uint32_t oct,
                                                   uint32_t* counter, CornerData* out, uint32_t cap) {
    const uint64_t mask = __ballot(is_corner);
    if (mask == 0ull) return ~0u;
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
    base = __shfl(base, __builtin_ctzll(mask));
    if (is_corner) {
        const uint32_t idx = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (idx < cap) {
            uint4 rec = make_uint4(x, y, angle, oct);
            *reinterpret_cast<uint4*>(&out[idx]) = rec;
            return idx;
        }
    }
    return ~0u;
}

// Opt-in extensions of the detector (not in the reference, default off):
//   arc   a corner needs a circular run of >= arc ring pixels (9..16); 12 is the reference's fast.wgsl:56-60.
//         A run of N contiguous ring positions contains at least floor(N/4) of the 4 compass positions, which
//         is the pre-test count used (3 for arc 12 = the reference's shortcut, 2 for FAST-9).
//   score S = sum over the run's polarity of (|v - c| - threshold), binary32 in ring order; written to a
//         per-octave score plane (one float per dispatch-grid pixel + 1-px border, 0 = no corner) for k_nms.
struct ScoreLayout {
    uint32_t off[kMaxLevels];     // float offset of each level's plane
    uint32_t pitch[kMaxLevels];   // gw + 2
    uint32_t stride;              // floats per frame
};

__device__ __forceinline__ bool has_run_16(uint32_t mask, uint32_t arc) {
    if (arc == 12u) return detect_streak_16(mask) != 0u;  // fast.wgsl:56-60
    // bit i of r <=> positions i .. i+have-1 (circular) are all set; doubling `have` by AND-ing shifted copies
    uint32_t r = mask | (mask << 16);
    for (uint32_t have = 1u; have < arc;) {
        const uint32_t s = min(have, arc - have);
        r &= r >> s;
        have += s;
    }
    return (r & 0xffffu) != 0u;
}

__global__ __launch_bounds__(256) void k_fast(const uint16_t* __restrict__ gray, Pyramid pyr, uint32_t oct,
                                              uint32_t gw, uint32_t gh, float threshold, uint32_t arc,
                                              uint32_t intended, uint32_t* __restrict__ counts, CornerData* __restrict__ corners,
                                              uint32_t cap, float* __restrict__ scores_list,
                                              float* __restrict__ score_planes, ScoreLayout sl, uint32_t oob = kOobZero) {
    constexpr int T = 16, R = 3, S = T + 2 * R;
    __shared__ float tile[S][S + 1];
    const uint32_t w = pyr.w[oct], h = pyr.h[oct];
    const uint32_t f = blockIdx.z;
    const uint16_t* lvl = gray + (size_t)f * pyr.stride + pyr.off[oct];
    const int bx = (int)blockIdx.x * T, by = (int)blockIdx.y * T;
    const int tid = (int)(threadIdx.y * T + threadIdx.x);
    for (int i = tid; i < S * S; i += T * T) {
        int ty = i / S, tx = i - ty * S;
        int gx = bx + tx - R, gy = by + ty - R;
        float v = 0.0f;
        if (gx >= 0 && gy >= 0 && gx < (int)w && gy < (int)h) v = from_half(bits_half(lvl[(size_t)gy * w + gx]));
        else if (oob != kOobZero)  // OrbOptions::oob_policy: a texel of the level instead of 0 (CRD-6)
            v = from_half(bits_half(lvl[(size_t)oob_index(gy, (int)h, oob) * w + (size_t)oob_index(gx, (int)w, oob)]));
        tile[ty][tx] = v;
    }
    __syncthreads();
    const uint32_t gx = (uint32_t)bx + threadIdx.x, gy = (uint32_t)by + threadIdx.y;
    const int lx = (int)threadIdx.x + R, ly = (int)threadIdx.y + R;
    bool is_corner = false;
    uint32_t angle = 0;
    float score = 0.0f;
    const uint32_t need = arc >> 2;  // compass points any run of `arc` must contain (fast.wgsl:95 for arc 12)
    // fast.wgsl:77 -- textureDimensions() is the level-0 size for every octave (Q8); u32 wrap kept.
    // "intended" mode (IM-4): the octave's own size, signed so that tiny octaves hold no keypoint.
    const uint32_t lim_x = pyr.w[0] - 16u, lim_y = pyr.h[0] - 16u;
    const bool guard = intended ? ((int)gx > 16 && (int)gy > 16 && (int)gx < (int)w - 16 && (int)gy < (int)h - 16)
                                : (gx > 16u && gy > 16u && gx < lim_x && gy < lim_y);
    if (gx < gw && gy < gh && guard) {
        const float c = tile[ly][lx];
        uint32_t n_over = 0, n_under = 0;
        const float v4[4] = {tile[ly][lx + 3], tile[ly][lx - 3], tile[ly + 3][lx], tile[ly - 3][lx]};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float diff = v4[i] - c;
            if (diff > threshold)
                n_over++;
            else if (diff < -threshold)
                n_under++;
        }
        if (n_over >= need || n_under >= need) {
            uint32_t m_over = 0, m_under = 0;
            float cx = 0.0f, cy = 0.0f, s_over = 0.0f, s_under = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float v = tile[ly + kRingDy[i]][lx + kRingDx[i]];
                float diff = v - c;
                float px = v * (float)kRingDx[i];
                float py = v * (float)kRingDy[i];
                cx = cx + px;
                cy = cy + py;
                if (diff > threshold) {
                    m_over |= 1u << i;
                    float e = diff - threshold;
                    s_over = s_over + e;
                } else if (diff < -threshold) {
                    m_under |= 1u << i;
                    float nd = -diff;
                    float e = nd - threshold;
                    s_under = s_under + e;
                }
            }
            const bool ro = has_run_16(m_over, arc), ru = has_run_16(m_under, arc);
            if (ro || ru) {
                is_corner = true;
                angle = intended ? angle_code_signed(cy, cx) : angle_code(cy, cx);
                score = ro ? s_over : s_under;
            }
        }
    }
    if (is_corner && score_planes)
        score_planes[(size_t)f * sl.stride + sl.off[oct] + (size_t)(gy + 1u) * sl.pitch[oct] + gx + 1u] = score;
    // append: one global atomic per workgroup (the reference's scheme, fast.wgsl:123-141), wave ballots inside
    __shared__ uint32_t wave_n[4], block_base;
    const uint64_t mask = __ballot(is_corner);
    const uint32_t lane = __lane_id(), wv = (uint32_t)tid >> 6;
    if (lane == 0u) wave_n[wv] = (uint32_t)__builtin_popcountll(mask);
    __syncthreads();
    const uint32_t total = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    if (total == 0u) return;  // uniform
    if (tid == 0) block_base = atomicAdd(counts + f, total);
    __syncthreads();
    if (is_corner) {
        uint32_t idx = block_base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        for (uint32_t k = 0; k < wv; k++) idx += wave_n[k];
        if (idx < cap) {
            *reinterpret_cast<uint4*>(&corners[(size_t)f * cap + idx]) = make_uint4(gx, gy, angle, oct);
            if (scores_list) scores_list[(size_t)f * cap + idx] = score;
        }
    }
}

// 3x3 non-maximum suppression per octave over the provisional corners of k_fast: a corner survives iff every
// 8-neighbour that is also a corner has a smaller score, or an equal score and a later raster position.
// grid: (blocks, 1, frames); the blocks of a frame stride over its provisional list (every wave stays whole until
// its last round, append_corners ballots over the wave).
__global__ __launch_bounds__(256) void k_nms(const uint32_t* __restrict__ prov_counts,
                                             const CornerData* __restrict__ prov, const float* __restrict__ prov_scores,
                                             uint32_t cap_prov, const float* __restrict__ score_planes, ScoreLayout sl,
                                             uint32_t* __restrict__ counts, CornerData* __restrict__ corners,
                                             uint32_t cap, float* __restrict__ out_scores) {
    const uint32_t f = blockIdx.z;
    const uint32_t n = min(prov_counts[f], cap_prov);
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {  // uniform per workgroup
        const uint32_t i = i0 + threadIdx.x;
        bool keep = false;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        float s = 0.0f;
        if (i < n) {
            rec = *reinterpret_cast<const uint4*>(&prov[(size_t)f * cap_prov + i]);
            s = prov_scores[(size_t)f * cap_prov + i];
            const float* plane = score_planes + (size_t)f * sl.stride + sl.off[rec.w];
            const uint32_t pitch = sl.pitch[rec.w];
            keep = true;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    if (dx == 0 && dy == 0) continue;
                    const float t = plane[(size_t)((int)rec.y + 1 + dy) * pitch + (size_t)((int)rec.x + 1 + dx)];
                    const bool later = dy > 0 || (dy == 0 && dx > 0);
                    if (t > 0.0f && (t > s || (t == s && !later))) keep = false;
                }
        }
        const uint32_t idx = append_corners(keep, rec.x, rec.y, rec.z, rec.w, counts + f, corners + (size_t)f * cap, cap);
        if (out_scores && idx != ~0u) out_scores[(size_t)f * cap + idx] = s;
    }
}

// "intended" mode IM-8 (not in the reference, which keeps whichever records win the atomic, Q9/Q10): when a
// frame has more candidates than `cap`, keep the `cap` best -- larger score first, ties by smaller
// (octave, y, x).  Keys are unique 64-bit integers (score bits : ~position), so the cap-th largest key is found
// by an 8-pass MSB radix select over LDS histograms; one workgroup per frame.  counts[f] = candidates before
// the cut (raw counter of the detector / NMS).
__global__ __launch_bounds__(1024) void k_topk(const uint32_t* __restrict__ in_counts,
                                               const CornerData* __restrict__ in, const float* __restrict__ in_scores,
                                               uint32_t cap_in, uint32_t* __restrict__ counts,
                                               CornerData* __restrict__ out, uint32_t cap) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ uint32_t sel_want, out_n;
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const uint32_t raw = in_counts[f], n = min(raw, cap_in);
    const CornerData* src = in + (size_t)f * cap_in;
    const float* sc = in_scores + (size_t)f * cap_in;
    CornerData* dst = out + (size_t)f * cap;
    if (tid == 0u) counts[f] = raw;
    if (n <= cap) {
        for (uint32_t i = tid; i < n; i += 1024u)
            *reinterpret_cast<uint4*>(&dst[i]) = *reinterpret_cast<const uint4*>(&src[i]);
        return;
    }
    auto key_of = [&](uint32_t i) -> unsigned long long {
        const uint4 r = *reinterpret_cast<const uint4*>(&src[i]);
        const uint32_t pos = (r.w << 28) | (r.y << 14) | r.x;  // x, y < 2^14 (checked at create)
        return ((unsigned long long)__float_as_uint(sc[i]) << 32) | (unsigned long long)(0xffffffffu - pos);
    };
    if (tid == 0u) {
        sel_prefix = 0ull;
        sel_want = cap;
        out_n = 0u;
    }
    unsigned long long known = 0ull;  // mask of the key bits fixed so far
    for (int pass = 7; pass >= 0; pass--) {
        if (tid < 256u) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = sel_prefix;
        for (uint32_t i = tid; i < n; i += 1024u) {
            const unsigned long long k = key_of(i);
            if ((k & known) == prefix) atomicAdd(&hist[(uint32_t)(k >> (8 * pass)) & 255u], 1u);
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
    const unsigned long long kth = sel_prefix;  // exactly `cap` keys are >= kth
    for (uint32_t i = tid; i < n; i += 1024u) {
        if (key_of(i) >= kth) {
            const uint32_t idx = atomicAdd(&out_n, 1u);
            if (idx < cap) *reinterpret_cast<uint4*>(&dst[idx]) = *reinterpret_cast<const uint4*>(&src[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6  brief.wgsl:20-68.  One wave64 per keypoint: lane l evaluates tests l, 64+l, 128+l, 192+l;
// each ballot yields two of the eight u32 words (bit i of word k = test 32k+i, brief.wgsl:47,63).
// cos/sin come from the 3142-entry table of CRD-10 (the angle code is an integer 0..3141).
// grid: (waves_x, 1, frames); each wave strides over the frame's keypoints.
// ---------------------------------------------------------------------------------------------
struct BriefTables {
    const uint32_t* pattern;  // 256 x packed (ax, ay, bx, by) int8
    const float* cos_tab;     // ORB_ANGLE_STEPS_FULL entries (the reference's codes use the first ORB_ANGLE_STEPS)
    const float* sin_tab;
    // The pattern rotated by every angle code, for the wave-per-keypoint kernels (k_rot_table, orb_kernels_brief.h):
    // rot[code * 64 + lane] = 8 x int16 = BYTE offsets of points a, b of tests lane, 64 + lane, 128 + lane, 192 + lane
    // into a window of f16 texels around the keypoint (2 * (ry * pitch + rx); pitch: k_brief_nf's patch, k_brief_i's window).
    const uint4* rot;
};

__device__ __forceinline__ float level_load(const uint16_t* lvl, uint32_t w, uint32_t h, int x, int y, uint32_t oob = kOobZero) {
    if (x < 0 || y < 0 || x >= (int)w || y >= (int)h) {
        if (oob == kOobZero) return 0.0f;  // CRD-6
        x = oob_index(x, (int)w, oob), y = oob_index(y, (int)h, oob);
    }
    return from_half(bits_half(lvl[(size_t)y * w + x]));
}

// (s1, s2) = (st, -st): the reference's R(-theta); (-st, st): "intended" mode IM-6, R(+theta).
__device__ __forceinline__ bool brief_test(uint32_t packed, float ct, float s1, float s2, int px, int py,
                                           const uint16_t* lvl, uint32_t w, uint32_t h, uint32_t oob = kOobZero, uint32_t fp = 0u) {
    const float ax = (float)(int8_t)(packed & 255u), ay = (float)(int8_t)((packed >> 8) & 255u);
    const float bx = (float)(int8_t)((packed >> 16) & 255u), by = (float)(int8_t)(packed >> 24);
    // mat2x2f(ct,-st, st,ct) * p (column-major): (ct*x + st*y, -st*x + ct*y)   brief.wgsl:38-54; CRD-13: rot_form(fp)
    float rax, ray, rbx, rby;
    rotate_fp(ct, s1, s2, ax, ay, rot_form(fp), &rax, &ray);
    rotate_fp(ct, s1, s2, bx, by, rot_form(fp), &rbx, &rby);
    const float va = level_load(lvl, w, h, (int)rax + px, (int)ray + py, oob);  // vec2i() truncates, brief.wgsl:56-57
    const float vb = level_load(lvl, w, h, (int)rbx + px, (int)rby + py, oob);
    return va > vb;  // brief.wgsl:62
}

__global__ __launch_bounds__(256) void k_brief(const uint16_t* __restrict__ blur, Pyramid pyr,
                                               const uint32_t* __restrict__ counts,
                                               const CornerData* __restrict__ corners, uint32_t cap,
                                               CornerDescriptor* __restrict__ descriptors, BriefTables tab,
                                               uint32_t intended, uint32_t oob = kOobZero, uint32_t fp = 0u, uint32_t angle_bins = 0u) {
    const uint32_t f = blockIdx.z;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t steps = intended ? ORB_ANGLE_STEPS_FULL : ORB_ANGLE_STEPS;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t n = min(counts[f], cap);
    const CornerData* kp = corners + (size_t)f * cap;
    uint32_t* out = reinterpret_cast<uint32_t*>(descriptors + (size_t)f * cap);
    const uint32_t p0 = tab.pattern[lane], p1 = tab.pattern[64u + lane], p2 = tab.pattern[128u + lane],
                   p3 = tab.pattern[192u + lane];
    for (uint32_t k = wave; k < n; k += n_waves) {
        const uint4 rec = *reinterpret_cast<const uint4*>(&kp[k]);  // x, y, angle, octave
        const uint32_t oct = rec.w;
        uint64_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        if (oct < pyr.depth) {
            const uint32_t code = binned_angle_code(min(rec.z, steps - 1u), intended ? angle_bins : 0u);  // IM-6b: the bin's centre code
            const float ct = tab.cos_tab[code], sn = tab.sin_tab[code];
            const float st = intended ? -sn : sn, nst = -st;
            const uint32_t w = pyr.w[oct], h = pyr.h[oct];
            const uint16_t* lvl = blur + (size_t)f * pyr.stride + pyr.off[oct];
            const int px = (int)rec.x, py = (int)rec.y;
            b0 = __ballot(brief_test(p0, ct, st, nst, px, py, lvl, w, h, oob, fp));
            b1 = __ballot(brief_test(p1, ct, st, nst, px, py, lvl, w, h, oob, fp));
            b2 = __ballot(brief_test(p2, ct, st, nst, px, py, lvl, w, h, oob, fp));
            b3 = __ballot(brief_test(p3, ct, st, nst, px, py, lvl, w, h, oob, fp));
        }
        if (lane < 8u) {
            const uint64_t src = lane < 2u ? b0 : (lane < 4u ? b1 : (lane < 6u ? b2 : b3));
            out[(size_t)k * 8u + lane] = (uint32_t)(src >> ((lane & 1u) * 32u));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Brute-force Hamming matcher (not in the reference).  Pair p = (frame p, frame p+1).
// A wave holds kMatchQ x 64 query descriptors in registers (8 VGPRs each); the candidate is the same for every lane,
// so its descriptor is read with one scalar load (8 SGPRs, the next one issued before the current is used) and
// shared by the kMatchQ queries of a lane: per pair 8 x (xor, popcount) + key + max/min/min, no LDS.
// grid: (n_pairs, ceil(cap / (64 * kMatchQ))), block 64.  The pair is the fast grid index: workgroups past a frame's
// keypoint count exit at once, and with the chunk as the fast index their regular pattern (15 busy, 17 idle, ...)
// lands on the same CUs every time -- half the chip then idles (measured: 4.8 ms against 2.3 ms).
// ---------------------------------------------------------------------------------------------
struct MatchRecord {
    uint32_t index;
    uint32_t dist;  // distance | second << 16
};
constexpr int kMatchQ = 2;
constexpr int kMatchG = 3;

__global__ __launch_bounds__(64) void k_match(const uint32_t* __restrict__ counts,
                                              const CornerDescriptor* __restrict__ descriptors, uint32_t cap,
                                              MatchRecord* __restrict__ matches) {
    const uint32_t pair = blockIdx.x, lane = threadIdx.x;
    const uint32_t na = min(counts[pair], cap), nb = min(counts[pair + 1u], cap);
    const uint32_t i0 = blockIdx.y * (64u * kMatchQ);
    if (i0 >= na) return;  // uniform
    const uint4* qa = reinterpret_cast<const uint4*>(descriptors + (size_t)pair * cap);
    const uint4* qb = reinterpret_cast<const uint4*>(descriptors + (size_t)(pair + 1u) * cap);
    uint4 a0[kMatchQ], a1[kMatchQ];
    uint32_t k1[kMatchQ], k2[kMatchQ];
#pragma unroll
    for (int q = 0; q < kMatchQ; q++) {
        const uint32_t i = min(i0 + (uint32_t)q * 64u + lane, na - 1u);
        a0[q] = qa[2u * i];
        a1[q] = qa[2u * i + 1u];
        k1[q] = k2[q] = 0xffffffffu;
    }
    // key = distance << 23 | candidate index (distance <= 256, index < 2^23): the two smallest keys are the best match
    // with ties to the smallest index, and the runner-up
    // candidates in groups of kMatchG: the scalar loads of the next group are issued before the current group is
    // used (a scalar load under load takes well over a thousand cycles; one wait per group instead of per candidate)
    uint4 n0[kMatchG], n1[kMatchG];
#pragma unroll
    for (int g = 0; g < kMatchG; g++) {
        const uint32_t jc = nb ? min((uint32_t)g, nb - 1u) : 0u;
        n0[g] = nb ? qb[2u * jc] : make_uint4(0u, 0u, 0u, 0u);
        n1[g] = nb ? qb[2u * jc + 1u] : make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t j0 = 0; j0 < nb; j0 += kMatchG) {
        uint4 b0[kMatchG], b1[kMatchG];
#pragma unroll
        for (int g = 0; g < kMatchG; g++) {
            b0[g] = n0[g];
            b1[g] = n1[g];
        }
#pragma unroll
        for (int g = 0; g < kMatchG; g++) {
            const uint32_t jn = min(j0 + kMatchG + (uint32_t)g, nb - 1u);
            n0[g] = qb[2u * jn];
            n1[g] = qb[2u * jn + 1u];
        }
#pragma unroll
        for (int g = 0; g < kMatchG; g++) {
            const uint32_t j = j0 + (uint32_t)g;
            if (j >= nb) break;  // uniform
#pragma unroll
            for (int q = 0; q < kMatchQ; q++) {
                uint32_t d = __builtin_popcount(a0[q].x ^ b0[g].x);
                d += __builtin_popcount(a0[q].y ^ b0[g].y);
                d += __builtin_popcount(a0[q].z ^ b0[g].z);
                d += __builtin_popcount(a0[q].w ^ b0[g].w);
                d += __builtin_popcount(a1[q].x ^ b1[g].x);
                d += __builtin_popcount(a1[q].y ^ b1[g].y);
                d += __builtin_popcount(a1[q].z ^ b1[g].z);
                d += __builtin_popcount(a1[q].w ^ b1[g].w);
                const uint32_t key = (d << 23) | j;
                k2[q] = min(k2[q], max(k1[q], key));
                k1[q] = min(k1[q], key);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kMatchQ; q++) {
        const uint32_t i = i0 + (uint32_t)q * 64u + lane;
        if (i < na) {
            MatchRecord r;
            r.index = k1[q] == 0xffffffffu ? 0xffffffffu : (k1[q] & 0x7fffffu);
            r.dist = (k1[q] == 0xffffffffu ? 0xffffu : (k1[q] >> 23)) | ((k2[q] == 0xffffffffu ? 0xffffu : (k2[q] >> 23)) << 16);
            matches[(size_t)pair * cap + i] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Synthetic frames: frame i of the launch uses seed0 + i.  One thread = 4 pixels.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t synth_pixel(uint32_t x, uint32_t y, uint32_t W, uint32_t H, uint32_t seed,
                                                uint32_t flags) {
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    if (flags & 1u) {
        c0 = W > 1u ? 255u * x / (W - 1u) : 0u;
        c1 = H > 1u ? 255u * y / (H - 1u) : 0u;
        c2 = (W + H > 2u) ? 255u * (x + y) / (W + H - 2u) : 0u;
    }
    if (flags & 4u) {
        const uint32_t cx = x / 128u, cy = y / 64u;
        const uint32_t hsh = syn_rnd(seed, 2u, cy * ((W + 127u) / 128u) + cx);
        if (hsh & 1u) {
            const uint32_t rise = 4u + ((hsh >> 1) & 15u) % 13u, run = 2u * rise;
            const uint32_t ox = ((hsh >> 8) & 255u) % (128u - run), oy = ((hsh >> 16) & 255u) % (64u - rise);
            const int u = (int)x - (int)(cx * 128u + ox), v = (int)y - (int)(cy * 64u + oy);
            if (u >= 0 && v >= 0 && u < (int)run && v < (int)rise) {
                const uint32_t uu = (hsh & 32u) ? run - 1u - (uint32_t)u : (uint32_t)u;
                const uint32_t vv = (hsh & 64u) ? rise - 1u - (uint32_t)v : (uint32_t)v;
                if (vv * run <= uu * rise) c0 = c1 = c2 = (hsh >> 24) & 255u;
            }
        }
    }
    if (flags & 2u) {
        const uint32_t cx = x / 32u, cy = y / 32u;
        const uint32_t hsh = syn_rnd(seed, 1u, cy * ((W + 31u) / 32u) + cx);
        if (hsh & 1u) {
            const uint32_t s = 1u + ((hsh >> 1) & 3u);
            const uint32_t bx = cx * 32u + 1u + ((hsh >> 4) & 255u) % (31u - s);
            const uint32_t by = cy * 32u + 1u + ((hsh >> 12) & 255u) % (31u - s);
            if (x >= bx && x < bx + s && y >= by && y < by + s) c0 = c1 = c2 = 128u + ((hsh >> 20) & 127u);
        }
    }
    if (flags & 8u) {
        const uint32_t n = syn_rnd(seed, 3u, y * W + x);
        c0 = (3u * c0 + (n & 255u)) / 4u;
        c1 = (3u * c1 + ((n >> 8) & 255u)) / 4u;
        c2 = (3u * c2 + ((n >> 16) & 255u)) / 4u;
    }
    return c0 | (c1 << 8) | (c2 << 16) | 0xff000000u;
}

__global__ __launch_bounds__(256) void k_synth(uint8_t* __restrict__ frames, size_t frame_bytes, uint32_t W,
                                               uint32_t H, uint32_t seed0, uint32_t flags) {
    const uint32_t y = blockIdx.y, f = blockIdx.z;
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= W) return;
    const uint32_t px = synth_pixel(x, y, W, H, seed0 + f, flags & 15u);
    if (flags & 16u) {  // ORB_SYN_Y8: one byte per pixel, the integer BT.601 luma of the RGBA recipe
        const uint32_t r = px & 255u, g = (px >> 8) & 255u, b = (px >> 16) & 255u;
        frames[(size_t)f * frame_bytes + (size_t)y * W + x] = (uint8_t)((77u * r + 150u * g + 29u * b + 128u) >> 8);
    } else {
        uint32_t* row = reinterpret_cast<uint32_t*>(frames + (size_t)f * frame_bytes + (size_t)y * W * 4u);
        row[x] = px;
    }
}

// scalar probes for the tests (CRD-3, CRD-9 on the device)
__global__ void k_probe_f16(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = half_bits(to_half(src[i]));
}
__global__ void k_probe_angle(const float* __restrict__ cy, const float* __restrict__ cx, uint32_t* __restrict__ dst,
                              size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = angle_code(cy[i], cx[i]);
}

}  // namespace orb
