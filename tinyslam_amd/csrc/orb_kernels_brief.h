// orb_kernels_brief.h -- rotated BRIEF-256 (brief.wgsl:20-68) for the fused literal pipeline, one THREAD per keypoint.
//
// The literal blur is one value per row for every column below qa (88 % of the width, k_front phase C), so for a
// keypoint with 18 <= x < qa - 18 ("flat", 88 % of them) a sample is the row constant of the rotated point's row and
// only the y component of brief.wgsl:50-57's rotation matters.  A workgroup takes 256 consecutive
// keypoints of a frame's final list (the band lists back to back, angle code 0 first): it stages the frame's slot prefix and row
// constants in LDS (2.3 KB at 720p, zero rows around every level: texels outside the level read 0, CRD-6), finds each
// lane's band slot by binary search, copies the record to the final list and, for flat keypoints, runs the 256 tests
// as straight-line code: the pattern's coordinates are literals, so a test is two products and a sum per point (equal
// ones shared inside a word), a truncation, two ds_read_u16, one subtraction and one v_alignbit that shifts its sign
// into the descriptor word.  Neighbouring lanes mostly come from the same band, so the rows a wave reads in one
// instruction lie within ~52 consecutive halfs (fewer than 32 dwords): conflict-free reads with broadcasts.
// Keypoints that are not flat (within 18 px of the left border or of the stored tail of the plane) are left to a launch
// of k_brief_rows in its non-flat-only mode.
// Against the wave-per-keypoint form (k_brief_rows: 85 VALU + 58 SALU per keypoint): ~31 vector instructions per
// keypoint, next to no scalar work, no barrier between waves.
#pragma once
#include "orb_kernels_fused.h"

namespace orb {

constexpr int kBriefTThreads = 256;  // four waves share the staging of the frame's prefix and row constants; nothing couples them afterwards
constexpr int kBriefTWaves = 6;       // waves per SIMD the register allocation aims at
constexpr uint32_t kBriefTMaxSlots = 8192;  // seg_before of a frame is staged in LDS
constexpr uint32_t kBriefTMaxRows = 12288;  // rows (all levels) + 36 per level, staged in LDS as f16

struct BriefTGeom {
    uint32_t n_slots, seg_cap;
    uint32_t n_classes;             // lists per band slot (FrontGeom::n_classes): 2 = angle code 0 / the rest
    uint32_t flat_end[kMaxLevels];  // Q - 18 (Q = FrontGeom::blur_q: every column below it holds the row constant)
    uint32_t qa[kMaxLevels];
    uint32_t row_base[kMaxLevels];  // index of row 0 of the level in the padded LDS row array
    uint32_t rows_padded;           // entries of that array
    uint32_t oob;                   // OrbOptions::oob_policy: what a sample outside the level reads (brief.wgsl:59-60; kOobZero: 0)
};

__host__ __device__ inline uint32_t brieft_lds_bytes(const BriefTGeom& g) {
    return (g.n_slots * g.n_classes + 1u) * 4u + ((g.rows_padded + 1u) & ~1u) * 2u;
}

// pattern as compile-time constants (orb_tables.h is generated from brief.wgsl:70-327)
__device__ __forceinline__ constexpr int pat_ax(int j) { return ORB_BRIEF_PATTERN[4 * j + 0]; }
__device__ __forceinline__ constexpr int pat_ay(int j) { return ORB_BRIEF_PATTERN[4 * j + 1]; }
__device__ __forceinline__ constexpr int pat_bx(int j) { return ORB_BRIEF_PATTERN[4 * j + 2]; }
__device__ __forceinline__ constexpr int pat_by(int j) { return ORB_BRIEF_PATTERN[4 * j + 3]; }

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (row shifts inside the rows of 16, then the two row
// broadcasts): six adds in registers where a __shfl_up chain is six LDS round trips.  All lanes must be active.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}

// Stores into the pinned host staging arrays of the single-frame call (k_brief_one).  Plain vector stores: the workgroup's
// system-scope release in front of its completion count is what makes them visible to the polling host.  (Measured and
// dropped: stores with the scope bits set, sc0 sc1 -- written through one by one, k_brief_one 20 -> 113 us at 720p --, and
// __hip_atomic_store at system scope, issued lane by lane as 8-byte PCIe packets: 740 us.)
__device__ __forceinline__ void store_host_u32(uint32_t* p, uint32_t v) { *p = v; }
__device__ __forceinline__ void store_host_16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// word = (word << 1) | (a > b), a and b non-negative f16 bit patterns (they order like the values, brief.wgsl:62)
__device__ __forceinline__ uint32_t push_gt(uint32_t word, uint32_t a, uint32_t b) {
    return __builtin_amdgcn_alignbit(word, b - a, 31);  // bit 31 of (b - a) is set iff a > b
}

// Body of k_brief_t for the workgroup that takes keypoints [chunk * 256, chunk * 256 + 256) of `frame` (the batch kernel:
// blockIdx; the single-frame kernel k_brief_one: its own chunk).  lds_raw: brieft_lds_bytes() of dynamic LDS.
// The frame's blur row constants into LDS, 18 zero rows in front of every level and 26 behind it (BriefTGeom::row_base).
__device__ __forceinline__ void brief_t_stage_rows(uint16_t* rows, const uint16_t* __restrict__ src, const Pyramid& pyr,
                                                   const BriefTGeom& bg, uint32_t tid, uint32_t n_threads = kBriefTThreads) {
    uint32_t m = 0;
    for (uint32_t i = tid; i < bg.rows_padded; i += n_threads) {
        while (m + 1u < pyr.depth && i + (uint32_t)kBriefHalo >= bg.row_base[m + 1u]) m++;  // level whose padded range holds i
        const uint32_t y = i - bg.row_base[m];  // wraps for the zero rows in front of the level
        uint16_t v = 0;  // CRD-6; with an out-of-level policy the padding rows repeat a row of the level (clamp: the nearest, umin: the last)
        if (y < pyr.h[m]) v = src[pyr.row_off[m] + y];
        else if (bg.oob != kOobZero) v = src[pyr.row_off[m] + (uint32_t)oob_index((int)y, (int)pyr.h[m], bg.oob)];
        rows[i] = v;
    }
}

// CHUNK: keypoints per workgroup (the first CHUNK threads take one each; all 256 stage).  STAGED: the caller has already
// put the list prefix and the row constants into lds_raw (k_brief_one does, next to its own prefix scan).
// ROT: the form matrix * vector takes under the adapter's shader compiler (rot_form(), CRD-13): 0 = both products and the sum rounded
// (CRD-10, the default); 1 = y' = fma(ct, y, -st*x); 2 = y' = fma(-st, x, ct*y) -- a product fewer per point either way.
template <int CHUNK = kBriefTThreads, bool STAGED = false, int ROT = 0>
__device__ __forceinline__ void brief_t_body(uint8_t* lds_raw, uint32_t frame, uint32_t chunk, const uint16_t* __restrict__ blur_rowc,
                                             const Pyramid& pyr, const BriefTGeom& bg, const uint32_t* __restrict__ seg_counts,
                                             const uint32_t* __restrict__ seg_before, const CornerData* __restrict__ segments,
                                             CornerData* __restrict__ corners, uint32_t cap,
                                             CornerDescriptor* __restrict__ descriptors, const BriefTables& tab,
                                             CornerData* __restrict__ host_corners = nullptr,
                                             CornerDescriptor* __restrict__ host_descriptors = nullptr) {
    // host_corners / host_descriptors (k_brief_one): every record and descriptor is also written to these (pinned host) arrays
    const uint32_t n_ent = bg.n_slots * bg.n_classes;  // lists of a frame, in final order: class-major, slot-minor
    uint32_t* const before = reinterpret_cast<uint32_t*>(lds_raw);                    // [n_ent + 1]
    uint16_t* const rows = reinterpret_cast<uint16_t*>(before + n_ent + 1u);          // [rows_padded]
    __shared__ uint32_t lv[kMaxLevels][2];  // flat_end, row_base (a run-time index into kernel arguments is a global load)

    const uint32_t tid = threadIdx.x;
    const size_t sbase = (size_t)frame * n_ent;
    // the last list (last class, last slot): seg_before is [class][slot], seg_counts is [slot][class]
    uint32_t stored_total;
    if (STAGED) {
        if (tid < pyr.depth) lv[tid][0] = bg.flat_end[tid], lv[tid][1] = bg.row_base[tid];
        __syncthreads();
        stored_total = before[n_ent];
    } else {
        stored_total = seg_before[sbase + n_ent - 1u] + min(seg_counts[sbase + n_ent - 1u], bg.seg_cap);
    }
    const uint32_t n_frame = min(stored_total, cap);
    const uint32_t k0 = chunk * (uint32_t)CHUNK;
    if (k0 >= n_frame) return;  // uniform for the workgroup

    if (!STAGED) {
        // ---- stage the frame's list prefix and row constants (zeros around every level)
        for (uint32_t s = tid; s < n_ent; s += kBriefTThreads) before[s] = seg_before[sbase + s];
        if (tid == 0) before[n_ent] = stored_total;
        if (tid < pyr.depth) lv[tid][0] = bg.flat_end[tid], lv[tid][1] = bg.row_base[tid];
        brief_t_stage_rows(rows, blur_rowc + (size_t)frame * pyr.row_stride, pyr, bg, tid);
        __syncthreads();
    }

    // ---- this thread's keypoint: slot by binary search in the prefix, record from the band segment
    const uint32_t k = k0 + tid;
    if (tid >= (uint32_t)CHUNK || k >= n_frame) return;
    uint32_t lo = 0, hi = n_ent;  // largest e with before[e] <= k (empty lists repeat the value: take the last)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (before[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t cls = lo >= bg.n_slots ? 1u : 0u, slot = lo - cls * bg.n_slots;
    const uint4 rec = *reinterpret_cast<const uint4*>(
        &segments[(((size_t)frame * bg.n_slots + slot) * bg.n_classes + cls) * bg.seg_cap + (k - before[lo])]);
    *reinterpret_cast<uint4*>(&corners[(size_t)frame * cap + k]) = rec;  // final list = the lists back to back
    if (host_corners) store_host_16(&host_corners[k], rec);
    const uint32_t lvl = min(rec.w, pyr.depth - 1u);
    if (!(rec.x >= (uint32_t)kBriefHalo && rec.x < lv[lvl][0])) return;  // not flat: k_brief_nf takes it

    uint32_t d[8];
    uint4* const o = reinterpret_cast<uint4*>(descriptors + (size_t)frame * cap + k);
    if (bg.n_classes == 2u && k0 + (tid & ~63u) + 64u <= before[bg.n_slots]) {
        // Every keypoint of this wave comes from a first list: angle code 0, R = I (brief.wgsl:50-57 with theta = 0; half
        // of all keypoints, Q7).  The sample rows are the pattern's y coordinates: compile-time offsets, and hipcc keeps
        // the 26 distinct rows in registers -- a test is one subtraction and one v_alignbit.
        const uint16_t* base0 = rows + lv[lvl][1] + rec.y - kBriefHalo;  // non-negative immediate offsets
#pragma unroll
        for (int wd = 0; wd < 8; wd++) {
            uint32_t acc = 0;
#pragma unroll
            for (int i = 31; i >= 0; i--) {
                const int j = wd * 32 + i;
                acc = push_gt(acc, base0[pat_ay(j) + kBriefHalo], base0[pat_by(j) + kBriefHalo]);
            }
            d[wd] = acc;
        }
        o[0] = make_uint4(d[0], d[1], d[2], d[3]);
        o[1] = make_uint4(d[4], d[5], d[6], d[7]);
        if (host_descriptors) {
            store_host_16(reinterpret_cast<uint4*>(host_descriptors + k), make_uint4(d[0], d[1], d[2], d[3]));
            store_host_16(reinterpret_cast<uint4*>(host_descriptors + k) + 1, make_uint4(d[4], d[5], d[6], d[7]));
        }
        return;
    }
    const uint32_t code = min(rec.z, (uint32_t)(ORB_ANGLE_STEPS - 1));
    const float ct = tab.cos_tab[code], st = tab.sin_tab[code], nst = -st;  // CRD-10 table (code 0: ct = 1, st = 0)
    const uint16_t* base = rows + lv[lvl][1] + rec.y;  // the keypoint's own row
#pragma unroll
    for (int wd = 0; wd < 8; wd++) {
        // The pattern's coordinates are literals, so equal products and sums are shared.  Sharing them across the whole
        // descriptor keeps hundreds of values alive (255 VGPRs, two waves per SIMD): ct and -st are made opaque once per
        // word, which confines the sharing to the word's 64 points.
        float ctw = ct, nstw = nst;
        asm volatile("" : "+v"(ctw), "+v"(nstw));
        uint32_t acc = 0;
#pragma unroll
        for (int i = 31; i >= 0; i--) {
            const int j = wd * 32 + i;
            // mat2x2f(ct,-st, st,ct) * p (column-major): y' = -st*x + ct*y, every product and the sum rounded (CRD-10)
            float ray, rby;
            if constexpr (ROT == 1) {
                const float a2 = nstw * (float)pat_ax(j), b2 = nstw * (float)pat_bx(j);
                ray = __builtin_fmaf(ctw, (float)pat_ay(j), a2), rby = __builtin_fmaf(ctw, (float)pat_by(j), b2);
            } else if constexpr (ROT == 2) {
                const float a3 = ctw * (float)pat_ay(j), b3 = ctw * (float)pat_by(j);
                ray = __builtin_fmaf(nstw, (float)pat_ax(j), a3), rby = __builtin_fmaf(nstw, (float)pat_bx(j), b3);
            } else {
                const float a2 = nstw * (float)pat_ax(j), a3 = ctw * (float)pat_ay(j);
                const float b2 = nstw * (float)pat_bx(j), b3 = ctw * (float)pat_by(j);
                ray = a2 + a3, rby = b2 + b3;
            }
            acc = push_gt(acc, base[(int)ray], base[(int)rby]);  // vec2i() truncates
        }
        d[wd] = acc;
    }
    o[0] = make_uint4(d[0], d[1], d[2], d[3]);
    o[1] = make_uint4(d[4], d[5], d[6], d[7]);
    if (host_descriptors) {
        store_host_16(reinterpret_cast<uint4*>(host_descriptors + k), make_uint4(d[0], d[1], d[2], d[3]));
        store_host_16(reinterpret_cast<uint4*>(host_descriptors + k) + 1, make_uint4(d[4], d[5], d[6], d[7]));
    }
}

template <int kWavesPerSimd, int ROT = 0>
__global__ __launch_bounds__(kBriefTThreads, kWavesPerSimd) void k_brief_t(const uint16_t* __restrict__ blur_rowc, Pyramid pyr, BriefTGeom bg,
                                                                const uint32_t* __restrict__ seg_counts,
                                                                const uint32_t* __restrict__ seg_before,
                                                                const CornerData* __restrict__ segments,
                                                                CornerData* __restrict__ corners, uint32_t cap,
                                                                CornerDescriptor* __restrict__ descriptors, BriefTables tab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    // The frame is the FAST grid index: chunks past a frame's keypoint count exit at once, and with the chunk as the fast
    // index their regular pattern (15 busy, 17 idle, ...) lands every busy workgroup on the same half of the CUs.
    brief_t_body<kBriefTThreads, false, ROT>(lds_raw, blockIdx.x, blockIdx.y, blur_rowc, pyr, bg, seg_counts, seg_before, segments, corners, cap, descriptors, tab);
}

// The keypoints k_brief_t leaves (not flat, 12 %: a sample may come from the stored tail of the blur plane, or lie left
// of the level).  A workgroup scans 256 consecutive keypoints of a frame's final list, compacts the ones that are not
// flat (ballots, no atomics) and deals them to its four waves, one wave per keypoint: lane l evaluates tests l, 64+l,
// 128+l, 192+l (brief.wgsl:47,63,67).
// Scattered 2-byte gathers from the plane are what this path used to spend its time on: the vector L1 looks up one
// cache line per cycle, and 512 samples of one keypoint are 512 look-ups although they all fall into the same 37 x 37
// patch.  So the wave first builds that patch in LDS -- 37 rows x 48 columns from the 8-aligned column left of x - 18,
// 16-byte pieces, each piece composed of 0 (outside the level, CRD-6), the row constant (columns < qa) or the stored
// texels (columns >= qa): ~120 line look-ups -- and then samples it with ds_read_u16.  Levels of odd width (2-byte
// aligned rows) keep the direct gathers.
constexpr int kNfPatchCols = 48, kNfPatchRows = 2 * kBriefHalo + 1;                       // halfs, rows
constexpr int kNfPatchHalfs = kNfPatchRows * kNfPatchCols;

// CHUNK: keypoints per workgroup (k_brief_one takes 64 so that a frame's ~450 such keypoints spread over 60 workgroups
// instead of 15 -- a wave works through its share one memory round trip after the other).  n_known: the frame's stored
// count when the caller has it (else ~0u: read from the prefix).
// ---------------------------------------------------------------------------------------------
// The rotated pattern of every angle code (BriefTables::rot), built once per program.  A wave that describes one keypoint
// used to rotate its lanes' eight points itself -- 16 products, 8 sums and 8 truncations per lane and keypoint, a
// third of k_brief_nf's vector instructions and two thirds of k_brief_i's -- although the result depends on nothing
// but the angle code: one block per code evaluates brief.wgsl:50-57 (or IM-6's R(+theta)) with the kernels' own
// arithmetic (every product and sum rounded on its own, truncation) and stores each lane's eight points as byte offsets
// into the consumer's window.  3142 (6284) codes x 1 KB.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rot_table(const uint32_t* __restrict__ pattern, const float* __restrict__ cos_tab,
                                                  const float* __restrict__ sin_tab, int pitch, int intended, uint4* __restrict__ out,
                                                  uint32_t fp = 0u, uint32_t angle_bins = 0u) {
    // angle_bins (IM-6b, intended mode): entry b is the pattern rotated by the centre code of angle bin b, (b * 6284 + 3142) / bins
    const uint32_t code = angle_bins ? (blockIdx.x * 6284u + 3142u) / angle_bins : blockIdx.x, lane = threadIdx.x;
    const float ct = cos_tab[code], st = sin_tab[code], nst = -st;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t pk = pattern[64u * (uint32_t)e + lane];
        const float pax = (float)(int8_t)(pk & 255u), pay = (float)(int8_t)((pk >> 8) & 255u);
        const float pbx = (float)(int8_t)((pk >> 16) & 255u), pby = (float)(int8_t)(pk >> 24);
        float rax, ray, rbx, rby;
        if (intended) {  // R(+theta) p = (ct*x - st*y, st*x + ct*y), IM-6
            const float a0 = ct * pax, a1 = nst * pay, a2 = st * pax, a3 = ct * pay;
            const float b0 = ct * pbx, b1 = nst * pby, b2 = st * pbx, b3 = ct * pby;
            rax = a0 + a1, ray = a2 + a3, rbx = b0 + b1, rby = b2 + b3;
        } else {  // mat2x2f(ct,-st, st,ct) * p (column-major): (ct*x + st*y, -st*x + ct*y), brief.wgsl:38-54; CRD-13: rot_form(fp)
            rotate_fp(ct, st, nst, pax, pay, rot_form(fp), &rax, &ray);
            rotate_fp(ct, st, nst, pbx, pby, rot_form(fp), &rbx, &rby);
        }
        const int oa = 2 * ((int)ray * pitch + (int)rax), ob = 2 * ((int)rby * pitch + (int)rbx);  // vec2i() truncates
        w[e] = ((uint32_t)oa & 0xffffu) | ((uint32_t)ob << 16);
    }
    out[(size_t)blockIdx.x * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ int rot_a(uint32_t w) { return (int)(int16_t)(w & 0xffffu); }  // byte offset of point a
__device__ __forceinline__ int rot_b(uint32_t w) { return (int)w >> 16; }                  // ... of point b

// NW: waves of the workgroup (k_brief_nf: 4; k_brief_one: 8, so that a chunk's keypoints that are not flat take one turn).
// OOB: the program has an out-of-level policy other than "zero" (BriefTGeom::oob says which): pieces and samples outside
// the level take a texel of the level instead of 0.  A template flag so that the default's code stays what it was.
template <int CHUNK = 256, int NW = 4, bool OOB = false>
__device__ __forceinline__ void brief_nf_body(uint32_t frame, uint32_t chunk, uint32_t n_known, const uint16_t* __restrict__ blur,
                                              const uint16_t* __restrict__ blur_rowc, const Pyramid& pyr, const BriefTGeom& bg,
                                              const uint32_t* __restrict__ seg_counts, const uint32_t* __restrict__ seg_before,
                                              const CornerData* __restrict__ corners, uint32_t cap,
                                              CornerDescriptor* __restrict__ descriptors, const BriefTables& tab,
                                              CornerDescriptor* __restrict__ host_descriptors = nullptr) {
    static_assert(CHUNK <= 256 && NW >= 4, "the scan runs on the first four waves");
    __shared__ __attribute__((aligned(16))) uint16_t patches[NW][kNfPatchHalfs];
    __shared__ uint4 recs[256];
    __shared__ uint16_t list[256];
    __shared__ uint32_t wave_n[4];
    // per-level geometry: indexing the by-value kernel arguments with a run-time level is a global load from the
    // kernarg segment (a memory round trip in front of every keypoint); from LDS it is 64 cycles
    __shared__ uint32_t lv[kMaxLevels][6];  // w, h, qa, row_off, off, flat_end
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < pyr.depth) {
        lv[tid][0] = pyr.w[tid], lv[tid][1] = pyr.h[tid], lv[tid][2] = bg.qa[tid], lv[tid][3] = pyr.row_off[tid];
        lv[tid][4] = pyr.off[tid], lv[tid][5] = bg.flat_end[tid];
    }
    const uint32_t n_ent = bg.n_slots * bg.n_classes;
    const size_t sbase = (size_t)frame * n_ent;
    const uint32_t n_frame = n_known != ~0u ? min(n_known, cap)
                                            : min(seg_before[sbase + n_ent - 1u] + min(seg_counts[sbase + n_ent - 1u], bg.seg_cap), cap);
    const uint32_t k0 = chunk * (uint32_t)CHUNK;
    if (k0 >= n_frame) return;  // uniform for the workgroup
    __syncthreads();
    // ---- scan: which of this chunk's keypoints are not flat (k_brief_t has written the final list)
    const uint32_t kk = k0 + tid;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    bool mine = false;
    if (tid < (uint32_t)CHUNK && kk < n_frame) {
        rec = *reinterpret_cast<const uint4*>(&corners[(size_t)frame * cap + kk]);
        const uint32_t l = min(rec.w, pyr.depth - 1u);
        mine = !(rec.x >= (uint32_t)kBriefHalo && rec.x < lv[l][5]);
    }
    const uint64_t m = __ballot(mine);
    if (lane == 0u && wave < 4u) wave_n[wave] = (uint32_t)__popcll(m);
    if (tid < 256u) recs[tid] = rec;
    __syncthreads();
    uint32_t start = 0, n_nf = 0;
#pragma unroll
    for (uint32_t w2 = 0; w2 < 4u; w2++) {
        if (w2 < wave) start += wave_n[w2];
        n_nf += wave_n[w2];
    }
    if (mine) list[start + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    __syncthreads();
    if (wave >= n_nf) return;

    uint32_t* out_desc = reinterpret_cast<uint32_t*>(descriptors + (size_t)frame * cap + k0);
    uint16_t* const patch = patches[wave];
    for (uint32_t i = wave; i < n_nf; i += (uint32_t)NW) {
        // the whole wave works on one keypoint: everything derived from its record is wave-uniform -- say so
        // (readfirstlane), so that it lives in scalar registers and the arithmetic on it runs on the scalar unit
        const uint32_t idx = __builtin_amdgcn_readfirstlane((uint32_t)list[i]);
        uint4 r = recs[idx];  // x, y, angle, octave
        r.x = __builtin_amdgcn_readfirstlane(r.x), r.y = __builtin_amdgcn_readfirstlane(r.y);
        r.z = __builtin_amdgcn_readfirstlane(r.z), r.w = __builtin_amdgcn_readfirstlane(r.w);
        const uint32_t lvl = min(r.w, pyr.depth - 1u);
        const int w = (int)__builtin_amdgcn_readfirstlane(lv[lvl][0]), h = (int)__builtin_amdgcn_readfirstlane(lv[lvl][1]);
        const int qa = (int)__builtin_amdgcn_readfirstlane(lv[lvl][2]);
        const uint16_t* rowc = blur_rowc + (size_t)frame * pyr.row_stride + __builtin_amdgcn_readfirstlane(lv[lvl][3]);
        const uint16_t* plane = blur + (size_t)frame * pyr.stride + __builtin_amdgcn_readfirstlane(lv[lvl][4]);
        // this lane's eight rotated points (brief.wgsl:50-57), from the table of the keypoint's angle code: issued here, used
        // behind the patch fill
        const uint4 tt = tab.rot[(size_t)min(r.z, (uint32_t)(ORB_ANGLE_STEPS - 1)) * 64u + lane];
        const uint32_t tw[4] = {tt.x, tt.y, tt.z, tt.w};
        uint64_t bal[4];
        if ((w & 1) == 0) {
            // ---- patch in LDS: rows y-18..y+18, columns c0..c0+47 with c0 = (x - 18) rounded down to 8.  A piece
            //      (8 columns) lies entirely left of the level (zeros), below qa (the row constant) or in the stored
            //      tail (one 16-byte load); the four pieces of a lane are loaded back to back from addresses that are
            //      always valid (the plane's first texels where none is needed) and composed afterwards.
            const int c0 = ((int)r.x - kBriefHalo) & ~7;
            constexpr int kPiecesPerRow = kNfPatchCols / 8, kPieces = kNfPatchRows * kPiecesPerRow;
            constexpr int kRounds = (kPieces + 63) / 64;
            uint4 tv[kRounds];
            uint32_t rcv[kRounds];
            int kind[kRounds], where[kRounds];  // 0: zeros, 1: row constant, 2: loaded, 3: the level ends inside the piece
#pragma unroll
            for (int rr = 0; rr < kRounds; rr++) {
                const int p = (int)lane + 64 * rr;
                const int pr = (int)(((float)p + 0.5f) * (1.0f / (float)kPiecesPerRow));
                const int pc = p - pr * kPiecesPerRow;
                const int gy0 = (int)r.y - kBriefHalo + pr, cx = c0 + 8 * pc;
                const int gy = OOB ? oob_index(gy0, h, bg.oob) : gy0;  // OOB: a row outside the level reads a row of the level
                const bool in = p < kPieces && gy >= 0 && gy < h && cx >= 0 && cx < w;
                kind[rr] = !in ? 0 : (cx < qa ? 1 : (cx + 8 <= w ? 2 : 3));
                where[rr] = p < kPieces ? pr * kNfPatchCols + 8 * pc : -1;
                rcv[rr] = rowc[min(max(gy, 0), h - 1)];
                if (OOB && p < kPieces && !(cx >= 0 && cx < w)) {
                    // the whole piece lies left (cx <= -8) or right (cx >= w) of the level: one texel of row gy repeated -- its first
                    // (clamp, left) or its last one (right; umin: left as well); a column below qa is the row constant
                    const int xm = oob_index(cx, w, bg.oob);
                    if (xm >= qa) rcv[rr] = plane[(size_t)(uint32_t)(__mul24(gy, w) + xm)];
                    kind[rr] = 1;
                }
                // 4-byte aligned: even width, cx a multiple of 8
                tv[rr] = *reinterpret_cast<const uint4*>(plane + (kind[rr] == 2 ? (size_t)(uint32_t)(__mul24(gy, w) + cx) : (size_t)0));
            }
#pragma unroll
            for (int rr = 0; rr < kRounds; rr++) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (kind[rr] == 1) {
                    const uint32_t c2 = rcv[rr] | (rcv[rr] << 16);
                    v = make_uint4(c2, c2, c2, c2);
                } else if (kind[rr] == 2) {
                    v = tv[rr];
                } else if (kind[rr] == 3) {  // width not a multiple of 8: the last piece of a row, texel by texel
                    const int p = (int)lane + 64 * rr;
                    const int pr = (int)(((float)p + 0.5f) * (1.0f / (float)kPiecesPerRow));
                    const int gy0 = (int)r.y - kBriefHalo + pr, cx = c0 + 8 * (p - pr * kPiecesPerRow);
                    const int gy = OOB ? oob_index(gy0, h, bg.oob) : gy0;
                    const uint16_t* src = plane + (size_t)(uint32_t)(__mul24(gy, w) + cx);
                    uint32_t t[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) t[q] = cx + q < w ? (uint32_t)src[q] : (OOB ? (uint32_t)src[w - 1 - cx] : 0u);  // OOB: the row's last texel (cx >= qa here)
                    v = make_uint4(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16));
                }
                if (where[rr] >= 0) *reinterpret_cast<uint4*>(&patch[where[rr]]) = v;
            }
            // the patch is filled with 16-byte stores and sampled as halfs by OTHER lanes of this wave: order the two
            // (LDS is in order within a wave; the fence keeps the compiler from moving the differently typed accesses)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int xo = (int)r.x - c0;  // 18..25
            const uint8_t* const centre = reinterpret_cast<const uint8_t*>(patch + kBriefHalo * kNfPatchCols + xo);  // the keypoint's texel
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t va = *reinterpret_cast<const uint16_t*>(centre + rot_a(tw[e]));
                const uint32_t vb = *reinterpret_cast<const uint16_t*>(centre + rot_b(tw[e]));
                bal[e] = __ballot(va > vb);  // non-negative f16: bit patterns order like the values (brief.wgsl:62)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next keypoint's fill overwrites what was just sampled
            __builtin_amdgcn_wave_barrier();
        } else {
            // ---- odd width: rows of the plane are only 2-byte aligned; gather sample by sample
            const int gy = (int)r.y - kBriefHalo + (int)lane;  // lanes 0..36 are the patch rows
            const uint32_t rowv = (gy >= 0 && gy < h && lane < 37u) ? (uint32_t)rowc[gy] : 0u;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // the table holds 2 * (dy * kNfPatchCols + dx) with |dx| < kNfPatchCols / 2: take it apart again
                const int oa = rot_a(tw[e]) >> 1, ob = rot_b(tw[e]) >> 1;
                const int dya = (oa + kNfPatchCols / 2 + 64 * kNfPatchCols) / kNfPatchCols - 64, dxa = oa - dya * kNfPatchCols;
                const int dyb = (ob + kNfPatchCols / 2 + 64 * kNfPatchCols) / kNfPatchCols - 64, dxb = ob - dyb * kNfPatchCols;
                const int xa = (int)r.x + dxa, ya = (int)r.y + dya;
                const int xb = (int)r.x + dxb, yb = (int)r.y + dyb;
                uint32_t va = (uint32_t)__shfl((int)rowv, dya + kBriefHalo);  // 0 when the row is outside the level
                uint32_t vb = (uint32_t)__shfl((int)rowv, dyb + kBriefHalo);
                const bool ina = xa >= 0 && xa < w && ya >= 0 && ya < h;
                const bool inb = xb >= 0 && xb < w && yb >= 0 && yb < h;
                if (!ina)
                    va = OOB ? blur_sample_mapped(plane, rowc, w, h, qa, xa, ya, bg.oob) : 0u;
                else if (xa >= qa)
                    va = plane[(size_t)(uint32_t)(__mul24(ya, w) + xa)];
                if (!inb)
                    vb = OOB ? blur_sample_mapped(plane, rowc, w, h, qa, xb, yb, bg.oob) : 0u;
                else if (xb >= qa)
                    vb = plane[(size_t)(uint32_t)(__mul24(yb, w) + xb)];
                bal[e] = __ballot(va > vb);
            }
        }
        if (lane < 8u) {
            const uint64_t src = lane < 2u ? bal[0] : (lane < 4u ? bal[1] : (lane < 6u ? bal[2] : bal[3]));
            out_desc[(size_t)idx * 8u + lane] = (uint32_t)(src >> ((lane & 1u) * 32u));
            if (host_descriptors) store_host_u32(reinterpret_cast<uint32_t*>(host_descriptors + k0) + (size_t)idx * 8u + lane, (uint32_t)(src >> ((lane & 1u) * 32u)));
        }
    }
}

constexpr int kBriefNfChunk = 64;  // keypoints of the final list a workgroup scans
template <bool OOB = false>
__global__ __launch_bounds__(256) void k_brief_nf(const uint16_t* __restrict__ blur, const uint16_t* __restrict__ blur_rowc,
                                                  Pyramid pyr, BriefTGeom bg, const uint32_t* __restrict__ seg_counts,
                                                  const uint32_t* __restrict__ seg_before,
                                                  const CornerData* __restrict__ corners, uint32_t cap,
                                                  CornerDescriptor* __restrict__ descriptors, BriefTables tab) {
    brief_nf_body<kBriefNfChunk, 4, OOB>(blockIdx.x, blockIdx.y, ~0u, blur, blur_rowc, pyr, bg, seg_counts, seg_before, corners, cap, descriptors, tab);  // frame = fast index, as above
}

// ---------------------------------------------------------------------------------------------
// The reference's only call shape is ONE frame per blocking call (orb.rs:469-557).  Launched one after the other, the
// slot prefix, the two BRIEF kernels and the packing of the results into host staging are four dependent launches of
// 6-19 us each for a few microseconds of work.  k_brief_one is the four in one launch for a single frame: every
// workgroup (kBriefOneChunk keypoints of the final list each; 256 threads) derives the prefix of the frame's lists
// itself, straight into the LDS layout of k_brief_t -- 2 x 68 lists at 720p: every count is loaded once, together with
// the row constants, wave 0 scans --, then runs k_brief_t's and k_brief_nf's bodies on its chunk, which write their
// records and descriptors into the (pinned, device-visible) host staging arrays as well; workgroup 0 also writes the raw
// counter there (orb.rs:550-556).  Latency, not throughput, is what it is built for: small chunks, so that the
// keypoints that are not flat (one memory round trip each, per wave) spread over many workgroups.
// ---------------------------------------------------------------------------------------------
constexpr int kBriefOneChunk = 64;
// the pinned counter block of the single-frame call: word 0 = the raw counter (orb.rs:550-556), word kSingleDoneWord = the completion
// sequence number, in a cache line of its own (words 1..10: k_brief_one's stamps in the diagnostic build)
constexpr int kSingleCountWords = 32, kSingleDoneWord = 16;
constexpr int kBriefOneThreads = 512;  // the flat keypoints take the first wave, the others one wave each: eight of those per turn
template <int ROT = 0>  // brief_t_body's
__global__ __launch_bounds__(kBriefOneThreads) void k_brief_one(const uint16_t* __restrict__ blur, const uint16_t* __restrict__ blur_rowc, Pyramid pyr,
                                                   BriefTGeom bg, const uint32_t* __restrict__ seg_counts,
                                                   uint32_t* __restrict__ seg_before, const CornerData* __restrict__ segments,
                                                   uint32_t* __restrict__ counts, CornerData* __restrict__ corners, uint32_t cap,
                                                   CornerDescriptor* __restrict__ descriptors, BriefTables tab,
                                                   uint32_t* __restrict__ host_count, CornerData* __restrict__ host_corners,
                                                   CornerDescriptor* __restrict__ host_descriptors, uint32_t* __restrict__ done_count,
                                                   uint32_t seq) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t tid = threadIdx.x, chunk = blockIdx.x;
    const uint32_t n_ent = bg.n_slots * bg.n_classes;
#ifdef TINYORB_STAMPS  // diagnostic build: 100 MHz wall clock of thread 0 of workgroups 0 and 24 at the kernel's joints, read with orb_single_stamps()
    uint64_t st[6];
    int n_st = 0;
#define BRIEF_ONE_STAMP() do { if (n_st < 6) st[n_st++] = wall_clock64(); } while (0)
#else
#define BRIEF_ONE_STAMP() do { } while (0)
#endif
    BRIEF_ONE_STAMP();
    uint32_t* const before = reinterpret_cast<uint32_t*>(lds_raw);             // k_brief_t's layout: [n_ent + 1] ...
    uint16_t* const rows = reinterpret_cast<uint16_t*>(before + n_ent + 1u);   // ... then [rows_padded]
    // raw counts in final order (class-major) into before[], all loads of the workgroup in flight together
    for (uint32_t e = tid; e < n_ent; e += (uint32_t)kBriefOneThreads) {
        const uint32_t cls = e / bg.n_slots, sl = e - cls * bg.n_slots;
        before[e] = seg_counts[sl * bg.n_classes + cls];
    }
    brief_t_stage_rows(rows, blur_rowc, pyr, bg, tid, (uint32_t)kBriefOneThreads);
    __syncthreads();
    BRIEF_ONE_STAMP();
    if (tid < 64u) {  // k_slot_prefix, by wave 0 of every workgroup: exclusive prefix of the stored counts, in place
        uint32_t carry = 0, total = 0;
        for (uint32_t e0 = 0; e0 < n_ent; e0 += 64u) {
            const uint32_t e = e0 + tid;
            const uint32_t raw = e < n_ent ? before[e] : 0u;
            const uint32_t stored = min(raw, bg.seg_cap);
            const uint32_t incl = wave_inclusive_sum(stored);  // DPP: no LDS round trip per step
            if (e < n_ent) before[e] = carry + incl - stored;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            total += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(raw), 63);
        }
        if (tid == 0u) {
            before[n_ent] = carry;  // stored keypoints of the frame
            if (chunk == 0u) {
                counts[0] = total;
                store_host_u32(host_count, total);
            }
        }
    }
    BRIEF_ONE_STAMP();
    // (brief_t_body's first barrier publishes the prefix)
    // both bodies write their records and descriptors to the device lists AND to host staging (orb.rs:537-547): no copy pass
    brief_t_body<kBriefOneChunk, true, ROT>(lds_raw, 0u, chunk, blur_rowc, pyr, bg, seg_counts, seg_before, segments, corners, cap,
                                       descriptors, tab, host_corners, host_descriptors);
    __syncthreads();  // the chunk's records are in the final list: brief_nf_body reads them
    BRIEF_ONE_STAMP();
    const uint32_t n_stored = before[n_ent];
    brief_nf_body<kBriefOneChunk, kBriefOneThreads / 64>(0u, chunk, n_stored, blur, blur_rowc, pyr, bg, seg_counts, seg_before, corners, cap, descriptors, tab,
                                  host_descriptors);
    // ---- completion, for a host that polls instead of synchronising the stream.  Every wave's stores are ordered in front of
    // the barrier (workgroup-scope release), then ONE lane of the workgroup releases at system scope -- buffer_wbl2 sc0 sc1: what
    // the workgroup wrote leaves this XCD's L2 for the host -- before the workgroup counts itself done; the last workgroup
    // publishes the sequence number, in a cache line of its own, and clears the count for the next call.  (All 512 threads
    // fencing: k_brief_one 22 -> 28 us, and 37 / 57 us with 256 / 512 workgroups.  No system-scope release per workgroup, only the
    // last one's: 3 us faster and wrong -- the other XCDs' L2s are not written back by it, and one call in some thousands
    // returned before the counter had arrived.)
    BRIEF_ONE_STAMP();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's loads and stores are done
    __syncthreads();
    BRIEF_ONE_STAMP();
#ifdef TINYORB_STAMPS
    if (tid == 0u && (chunk == 0u || chunk == 24u))
        for (int i = 0; i < 5; i++) host_count[(chunk ? 6 : 1) + i] = (uint32_t)(st[i + 1] - st[i]);
#endif
    if (tid == 0u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: buffer_wbl2 sc0 sc1 ...
        __builtin_amdgcn_s_waitcnt(0);                 // ... and its completion (hipcc leaves the wait out in front of a relaxed atomic)
        // The count is an acquire-release read-modify-write at agent scope: every workgroup's system-scope release above is
        // ordered in front of its increment, and the workgroup that reads gridDim.x - 1 acquires all the others' -- the
        // protocol is right by the memory model, not only by what gfx950 does (one lane per workgroup: nothing measurable).
        const uint32_t prev = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == gridDim.x) {
            __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_count + kSingleDoneWord, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace orb
