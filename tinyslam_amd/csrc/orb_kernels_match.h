// orb_kernels_match.h -- the Hamming matcher of consecutive frames (not in the reference) on the matrix
// cores, in two forms: int8 (below, the first) and block-scaled fp4 (at the end of the file: the default, half the MFMAs and bytes).  All-pairs Hamming distance IS a matrix product: with a descriptor's 256 bits written as 256 signed bytes
// s = +127 (bit set) / -127 (bit clear),  sum_k s_a[k] * s_b[k] = 127^2 * (256 - 2 * popcount(a ^ b))  -- exact in integers.
//
//   k_desc_expand   every stored descriptor of the batch as 256 bytes of +-127 (two 16-byte stores per thread)
//   k_match_mfma    a wave owns 16 * kMatchRowTiles queries of frame f (their A fragments stay in registers) and walks the
//                   candidates of frame f + 1 in tiles of 16: four v_mfma_i32_16x16x64_i8 per row tile give 256 dot products,
//                   and the accumulator they start from is the REST OF THE KEY: with K = 127^2 it is preset to
//                   256 K + (K - 1 - candidate), so the instruction's result is (512 - 2 * distance) * K + (K - 1 - candidate)
//                   -- distance first, then the smaller candidate index -- and costs the vector unit TWO instructions, the
//                   median and the maximum of (best, runner-up, key) for the two largest keys, where k_match's loop spends
//                   8 x (xor, popcount) + 4.  (A power of two as the byte would make the key a shift, but 64 * 64 = 2^12 leaves
//                   the index 12 bits; 127^2 = 16129 leaves it 16128 values.)  The 16 lanes that hold a row's columns merge
//                   their pairs once, at the end.
// A and B fragments are loaded by the same rule (lane l: descriptor row l & 15, bytes 64 t + 16 (l >> 4) ... + 15 of k tile t),
// so whatever order the instruction gives the k index inside a lane group, both operands agree on it; only the row / column
// maps matter, and those are the documented ones (A row = B column = l & 15; C column = l & 15, row = 4 (l >> 4) + register).
#pragma once
#include "orb_kernels_staged.h"

namespace orb {

constexpr uint32_t kMatchK = 127u * 127u;                 // key = (512 - 2 * distance) * K + (K - 1 - candidate index)
constexpr uint32_t kMatchMaxCap = kMatchK - 1u;           // candidate indices 0 .. K - 2: a real key is never 0
#ifndef TINYORB_MATCH_ROWTILES
#define TINYORB_MATCH_ROWTILES 4
#endif
constexpr int kMatchRowTiles = TINYORB_MATCH_ROWTILES;    // row tiles of 16 queries per wave
#ifndef TINYORB_MATCH_WAVES
#define TINYORB_MATCH_WAVES (16 / TINYORB_MATCH_ROWTILES)
#endif
constexpr int kMatchWaves = TINYORB_MATCH_WAVES;          // waves per workgroup (default: 256 queries per workgroup)
#ifndef TINYORB_MATCH_MINWAVES
#define TINYORB_MATCH_MINWAVES 1
#endif
constexpr int kMatchQueriesPerWg = 16 * kMatchRowTiles * kMatchWaves;
typedef int v4i_t __attribute__((ext_vector_type(4)));

// grid (ceil(cap / 32), n_frames), block 256: thread -> (descriptor, word): 32 bits -> 32 bytes
__global__ __launch_bounds__(256) void k_desc_expand(const uint32_t* __restrict__ counts, const CornerDescriptor* __restrict__ descriptors,
                                                     uint32_t cap, uint8_t* __restrict__ desc8) {
    const uint32_t frame = blockIdx.y, n = min(counts[frame], cap);
    const uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3), wd = threadIdx.x & 7u;
    if (i >= n) return;
    const uint32_t bits = reinterpret_cast<const uint32_t*>(descriptors + (size_t)frame * cap + i)[wd];
    uint32_t out[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {  // four bits -> four bytes: +127 (0x7f) where the bit is set, -127 (0x81) where it is clear
        const uint32_t nib = (bits >> (4 * q)) & 15u;
        const uint32_t y = (nib * 0x00204081u) & 0x01010101u;  // bit i of the nibble in the low bit of byte i (no two terms share a position)
        out[q] = 0x81818181u - (y << 1);                         // 0x81 - 2 per set byte: no borrow between bytes
    }
    uint4* dst = reinterpret_cast<uint4*>(desc8 + ((size_t)frame * cap + i) * 256u + wd * 32u);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// grid (n_pairs, ceil(cap / 256)), block 256 (the pair is the fast index, as in k_match: chunks past a frame's count exit at once).
// Candidates come through LDS in chunks of 64 (16 KB, rows 272 bytes apart: the 64 lanes' 16-byte fragment reads then spread
// evenly over the banks), loaded once per workgroup -- every wave loading its own fragments from the L1 asked the texture
// path for 4 KB per wave and tile, four times the same bytes, and ran at a third of this form's rate -- and double-buffered:
// the next chunk's loads are in flight while this one is multiplied, one barrier per chunk.
#ifndef TINYORB_MATCH_CHUNK
#define TINYORB_MATCH_CHUNK 64
#endif
constexpr int kMatchChunk = TINYORB_MATCH_CHUNK;  // candidates per LDS stage (4 column tiles)
constexpr int kMatchRowBytes = 256 + 32;   // LDS row stride (dword offset 72 r + 4 g: conflict-free by the rule found for k_match_fp4, kMatch4RowBytes)
__global__ __launch_bounds__(64 * kMatchWaves) void k_match_mfma(const uint32_t* __restrict__ counts, const uint8_t* __restrict__ desc8,
                                                                 uint32_t cap, MatchRecord* __restrict__ matches) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][kMatchChunk * kMatchRowBytes];
    const uint32_t pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t na = min(counts[pair], cap), nb = min(counts[pair + 1u], cap);
    const uint32_t qw = blockIdx.y * (uint32_t)kMatchQueriesPerWg;
    if (qw >= na) return;  // uniform for the workgroup
    const uint32_t q0 = qw + wave * (16u * kMatchRowTiles);
    const bool wave_live = q0 < na;  // a wave without queries still stages candidates and meets the barriers
    const uint32_t rc = lane & 15u, g = lane >> 4;  // row of A / column of B, and the lane group
    const uint8_t* const qa = desc8 + (size_t)pair * cap * 256u + 16u * g;
    const uint8_t* const qb = desc8 + (size_t)(pair + 1u) * cap * 256u;
    v4i_t a[kMatchRowTiles][4];
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++) {
        const uint32_t row = min(q0 + 16u * (uint32_t)m + rc, na - 1u);  // rows past the frame's count repeat its last one; never stored
#pragma unroll
        for (int t = 0; t < 4; t++) a[m][t] = *reinterpret_cast<const v4i_t*>(qa + (size_t)row * 256u + 64u * (uint32_t)t);
    }
    uint32_t best[kMatchRowTiles][4], second[kMatchRowTiles][4];
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
        for (int i = 0; i < 4; i++) best[m][i] = second[m][i] = 0u;  // 0 = nothing: a real key is never 0 (its low part is K - 1 - index > 0)

    // staging: a chunk is 1024 pieces of 16 bytes, piece p = bytes 16 (p & 15) .. of candidate row p >> 4; a thread takes pieces
    // tid + NT k.  (Written with explicit registers: an array captured by a lambda went through scratch memory.)
    constexpr uint32_t NT = 64u * (uint32_t)kMatchWaves, NP = (uint32_t)(kMatchChunk * 16) / NT;  // threads, pieces per thread
    static_assert(NP >= 1u && NP <= 8u && NP * NT == (uint32_t)(kMatchChunk * 16), "staging: whole pieces per thread");
    const uint32_t prow = tid >> 4, pcol = 16u * (tid & 15u);   // piece k: row prow + (NT / 16) k
    uint8_t* const put0 = &stage[0][prow * (uint32_t)kMatchRowBytes + pcol];
    constexpr uint32_t kRowStep = NT / 16u, kPutStep = kRowStep * (uint32_t)kMatchRowBytes, kBufBytes = (uint32_t)(kMatchChunk * kMatchRowBytes);
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;  // (named registers: an array went through scratch memory)
#define MATCH_FETCH(J0) \
    do { \
        const uint8_t* const fb_ = qb + pcol; \
        if (NP > 0u) p0 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 0u * kRowStep, nb - 1u) * 256u); \
        if (NP > 1u) p1 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 1u * kRowStep, nb - 1u) * 256u); \
        if (NP > 2u) p2 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 2u * kRowStep, nb - 1u) * 256u); \
        if (NP > 3u) p3 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 3u * kRowStep, nb - 1u) * 256u); \
        if (NP > 4u) p4 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 4u * kRowStep, nb - 1u) * 256u); \
        if (NP > 5u) p5 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 5u * kRowStep, nb - 1u) * 256u); \
        if (NP > 6u) p6 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 6u * kRowStep, nb - 1u) * 256u); \
        if (NP > 7u) p7 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 7u * kRowStep, nb - 1u) * 256u); \
    } while (0)
#define MATCH_PUT(BUF) \
    do { \
        uint8_t* const pb_ = put0 + (uint32_t)(BUF) * kBufBytes; \
        if (NP > 0u) *reinterpret_cast<uint4*>(pb_ + 0u * kPutStep) = p0; \
        if (NP > 1u) *reinterpret_cast<uint4*>(pb_ + 1u * kPutStep) = p1; \
        if (NP > 2u) *reinterpret_cast<uint4*>(pb_ + 2u * kPutStep) = p2; \
        if (NP > 3u) *reinterpret_cast<uint4*>(pb_ + 3u * kPutStep) = p3; \
        if (NP > 4u) *reinterpret_cast<uint4*>(pb_ + 4u * kPutStep) = p4; \
        if (NP > 5u) *reinterpret_cast<uint4*>(pb_ + 5u * kPutStep) = p5; \
        if (NP > 6u) *reinterpret_cast<uint4*>(pb_ + 6u * kPutStep) = p6; \
        if (NP > 7u) *reinterpret_cast<uint4*>(pb_ + 7u * kPutStep) = p7; \
    } while (0)
    // one column tile: 4 k steps x 4 row tiles with four independent accumulators (a dependent chain of MFMAs waits out each
    // one's latency), then three vector instructions per result.  MASK: the frame's last, partial tile -- its missing columns get key 0.
    auto tile = [&](auto mask_tag, const uint8_t* src, uint32_t jt) {
        constexpr bool MASK = decltype(mask_tag)::value;
        v4i_t b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) b[t] = *reinterpret_cast<const v4i_t*>(src + 64 * t);
        const uint32_t col = jt + rc;
        const int c0 = (int)(256u * kMatchK + (kMatchK - 1u) - col);  // what the dot products are added to: the key's other two terms
        const v4i_t cin = {c0, c0, c0, c0};
        v4i_t acc[kMatchRowTiles];
#pragma unroll
        for (int m = 0; m < kMatchRowTiles; m++) acc[m] = cin;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int m = 0; m < kMatchRowTiles; m++) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m][t], b[t], acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t key = (uint32_t)acc[m][i];
                if (MASK) key = col < nb ? key : 0u;
                // the two largest of {best >= second, key}: the runner-up is their median.  (Plain C, not asm: an asm statement
                // that reads an MFMA result gets no wait states in front of it -- a first version read stale registers.)
                const uint32_t b0 = best[m][i], s0 = second[m][i];
                second[m][i] = max(min(b0, key), min(max(b0, key), s0));  // = med3(b0, s0, key): hipcc emits v_med3_u32
                best[m][i] = max(b0, key);
            }
    };
    if (nb) {
        MATCH_FETCH(0u);
        MATCH_PUT(0);
    }
    __syncthreads();
    // whole chunks without a branch around the tiles, the partial chunk behind the loop (see k_match_fp4)
    const uint8_t* const src_base = &stage[0][rc * (uint32_t)kMatchRowBytes + 16u * g];
    const uint32_t n_whole = nb / (uint32_t)kMatchChunk, n_chunks = (nb + (uint32_t)kMatchChunk - 1u) / (uint32_t)kMatchChunk;
    uint32_t c = 0;
    for (; c < n_whole; c++) {
        const uint32_t j0 = c * (uint32_t)kMatchChunk;
        const bool more = c + 1u < n_chunks;
        if (more) MATCH_FETCH(j0 + (uint32_t)kMatchChunk);
        const uint8_t* const src0 = src_base + (c & 1u) * kBufBytes;
#pragma unroll
        for (int tt = 0; tt < kMatchChunk / 16; tt++) tile(std::false_type{}, src0 + tt * 16 * kMatchRowBytes, j0 + 16u * (uint32_t)tt);
        if (more) MATCH_PUT((c & 1u) ^ 1u);
        __syncthreads();
    }
    if (c < n_chunks) {  // the partial chunk: whole tiles, then one masked tile
        const uint32_t j0 = c * (uint32_t)kMatchChunk;
        const uint8_t* const src0 = src_base + (c & 1u) * kBufBytes;
        for (uint32_t tt = 0; j0 + 16u * tt < nb; tt++) {
            const uint32_t jt = j0 + 16u * tt;
            if (jt + 16u <= nb)
                tile(std::false_type{}, src0 + tt * (uint32_t)(16 * kMatchRowBytes), jt);
            else
                tile(std::true_type{}, src0 + tt * (uint32_t)(16 * kMatchRowBytes), jt);
        }
    }
#undef MATCH_FETCH
#undef MATCH_PUT
    if (!wave_live) return;
    // the sixteen lanes of a group hold one row's columns: merge their (best, second) pairs
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t b1 = best[m][i], s1 = second[m][i];
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) {
                const uint32_t b2 = (uint32_t)__shfl_xor((int)b1, sh), s2 = (uint32_t)__shfl_xor((int)s1, sh);
                s1 = max(min(b1, b2), max(s1, s2));
                b1 = max(b1, b2);
            }
            const uint32_t q = q0 + 16u * (uint32_t)m + 4u * g + (uint32_t)i;
            if (rc == 0u && q < na) {
                MatchRecord r;
                const uint32_t v1 = b1 / kMatchK, v2 = s1 / kMatchK;  // 512 - 2 * distance
                r.index = b1 ? kMatchK - 1u - (b1 - v1 * kMatchK) : 0xffffffffu;
                const uint32_t d1 = b1 ? (512u - v1) >> 1 : 0xffffu;
                const uint32_t d2 = s1 ? (512u - v2) >> 1 : 0xffffu;
                r.dist = d1 | (d2 << 16);
                matches[(size_t)pair * cap + q] = r;
            }
        }
}


// ---------------------------------------------------------------------------------------------
// The same matcher on the block-scaled fp4 form (v_mfma_scale_f32_16x16x128_f8f6f4, E2M1 operands: +-1 is exact, 0x2 / 0xa):
// 128 bytes per descriptor instead of 256, two MFMAs per row tile and column tile instead of four, each at the int8 form's
// cycles.  The block scales (E8M0, one per lane's 32 elements) are 2^7 on both sides, so a product is +-2^14 = +-K: the
// accumulator is preset to 256 K + (K - 1 - candidate) as before, every value is an integer below 2^24 and exact in binary32,
// and the two largest keys are kept with v_med3_u32 / v_max_u32 on their bit patterns.  A and B fragments are loaded by one rule (lane l: row l & 15,
// bytes 64 t + 16 (l >> 4) .. + 15 of k step t), so the order of the 32 elements inside a lane's block cannot matter.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kMatch4K = 1u << 14;
constexpr uint32_t kMatch4MaxCap = kMatch4K - 1u;
#ifndef TINYORB_MATCH4_ROWBYTES
#define TINYORB_MATCH4_ROWBYTES (128 + 32)
#endif
// LDS row stride of a staged candidate.  Lane (r = l & 15, g = l >> 4) reads 16 bytes at r * stride + 16 g: with a stride of 128 + 16 bytes
// SQ_LDS_BANK_CONFLICT read 36 % of SQ_LDS_IDX_ACTIVE (7.3 cycles per LDS instruction); with 128 + 32 it reads 0 (4.7 cycles) -- the dword
// offset 40 r + 4 g covers the 32 banks once per 4 rows x 2 groups, where 36 r + 4 g puts (r, g + 1) on (r + 1, g)'s banks.  176, 192, 208
// and 272 conflict like 144 (profiles/r05_match_experiments.txt).  The kernel's time does not move (it waits for issue slots, not for LDS).
constexpr int kMatch4RowBytes = TINYORB_MATCH4_ROWBYTES;
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

// grid (ceil(cap / 32), n_frames), block 256: thread -> (descriptor, word): 32 bits -> 32 nibbles (16 bytes)
__global__ __launch_bounds__(256) void k_desc_expand4(const uint32_t* __restrict__ counts, const CornerDescriptor* __restrict__ descriptors,
                                                      uint32_t cap, uint8_t* __restrict__ desc4) {
    const uint32_t frame = blockIdx.y, n = min(counts[frame], cap);
    const uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3), wd = threadIdx.x & 7u;
    if (i >= n) return;
    const uint32_t bits = reinterpret_cast<const uint32_t*>(descriptors + (size_t)frame * cap + i)[wd];
    uint32_t out[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {  // eight bits -> eight nibbles: 0x2 (+1) where the bit is set, 0xa (-1) where it is clear
        const uint32_t b = (bits >> (8 * q)) & 255u;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v |= ((b >> k) & 1u) << (4 * k + 3);  // bit k at the top of nibble k
        out[q] = 0xaaaaaaaau ^ v;
    }
    *reinterpret_cast<uint4*>(desc4 + ((size_t)frame * cap + i) * 128u + wd * 16u) = make_uint4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(64 * kMatchWaves, TINYORB_MATCH_MINWAVES) void k_match_fp4(const uint32_t* __restrict__ counts, const uint8_t* __restrict__ desc4,
                                                                uint32_t cap, MatchRecord* __restrict__ matches) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][kMatchChunk * kMatch4RowBytes];
    const uint32_t pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t na = min(counts[pair], cap), nb = min(counts[pair + 1u], cap);
    const uint32_t qw = blockIdx.y * (uint32_t)kMatchQueriesPerWg;
    if (qw >= na) return;  // uniform for the workgroup
    const uint32_t q0 = qw + wave * (16u * kMatchRowTiles);
    const bool wave_live = q0 < na;
    const uint32_t rc = lane & 15u, g = lane >> 4;
    const uint8_t* const qa = desc4 + (size_t)pair * cap * 128u + 16u * g;
    const uint8_t* const qb = desc4 + (size_t)(pair + 1u) * cap * 128u;
    v4i_t a[kMatchRowTiles][2];
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++) {
        const uint32_t row = min(q0 + 16u * (uint32_t)m + rc, na - 1u);
#pragma unroll
        for (int t = 0; t < 2; t++) a[m][t] = *reinterpret_cast<const v4i_t*>(qa + (size_t)row * 128u + 64u * (uint32_t)t);
    }
    // The keys are kept as the BIT PATTERNS of the (non-negative) binary32 results, which order like the values: on floats every
    // v_max_f32 / v_med3_f32 is preceded by a canonicalising v_max_f32 x, x, x of the MFMA result (hipcc cannot know it is no
    // signalling NaN) -- 158 v_max_f32 per 64 results instead of 64.
    uint32_t best[kMatchRowTiles][4], second[kMatchRowTiles][4];
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
        for (int i = 0; i < 4; i++) best[m][i] = second[m][i] = 0u;  // +0.0 = nothing: a real key is >= 1.0

    // staging: a chunk is 64 rows x 8 pieces of 16 bytes = 512 pieces, piece p = bytes 16 (p & 7) .. of candidate row p >> 3
    constexpr uint32_t NT = 64u * (uint32_t)kMatchWaves, NP = (uint32_t)(kMatchChunk * 8) / NT;
    static_assert(NP >= 1u && NP <= 8u && NP * NT == (uint32_t)(kMatchChunk * 8), "staging: whole pieces per thread");
    const uint32_t prow = tid >> 3, pcol = 16u * (tid & 7u);
    uint8_t* const put0 = &stage[0][prow * (uint32_t)kMatch4RowBytes + pcol];
    constexpr uint32_t kRowStep = NT / 8u, kPutStep = kRowStep * (uint32_t)kMatch4RowBytes, kBufBytes = (uint32_t)(kMatchChunk * kMatch4RowBytes);
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;  // (named registers: an array went through scratch memory)
#define MATCH4_FETCH(J0) \
    do { \
        const uint8_t* const fb_ = qb + pcol; \
        if (NP > 0u) p0 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 0u * kRowStep, nb - 1u) * 128u); \
        if (NP > 1u) p1 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 1u * kRowStep, nb - 1u) * 128u); \
        if (NP > 2u) p2 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 2u * kRowStep, nb - 1u) * 128u); \
        if (NP > 3u) p3 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 3u * kRowStep, nb - 1u) * 128u); \
        if (NP > 4u) p4 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 4u * kRowStep, nb - 1u) * 128u); \
        if (NP > 5u) p5 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 5u * kRowStep, nb - 1u) * 128u); \
        if (NP > 6u) p6 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 6u * kRowStep, nb - 1u) * 128u); \
        if (NP > 7u) p7 = *reinterpret_cast<const uint4*>(fb_ + (size_t)min((J0) + prow + 7u * kRowStep, nb - 1u) * 128u); \
    } while (0)
#define MATCH4_PUT(BUF) \
    do { \
        uint8_t* const pb_ = put0 + (uint32_t)(BUF) * kBufBytes; \
        if (NP > 0u) *reinterpret_cast<uint4*>(pb_ + 0u * kPutStep) = p0; \
        if (NP > 1u) *reinterpret_cast<uint4*>(pb_ + 1u * kPutStep) = p1; \
        if (NP > 2u) *reinterpret_cast<uint4*>(pb_ + 2u * kPutStep) = p2; \
        if (NP > 3u) *reinterpret_cast<uint4*>(pb_ + 3u * kPutStep) = p3; \
        if (NP > 4u) *reinterpret_cast<uint4*>(pb_ + 4u * kPutStep) = p4; \
        if (NP > 5u) *reinterpret_cast<uint4*>(pb_ + 5u * kPutStep) = p5; \
        if (NP > 6u) *reinterpret_cast<uint4*>(pb_ + 6u * kPutStep) = p6; \
        if (NP > 7u) *reinterpret_cast<uint4*>(pb_ + 7u * kPutStep) = p7; \
    } while (0)
    const int scale = 127 + 7;  // E8M0: 2^7 for every block of both operands
    auto tile = [&](auto mask_tag, const uint8_t* src, uint32_t jt) {
        constexpr bool MASK = decltype(mask_tag)::value;
        v8i_t b[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const v4i_t q = *reinterpret_cast<const v4i_t*>(src + 64 * t);
            b[t] = v8i_t{q[0], q[1], q[2], q[3], 0, 0, 0, 0};
        }
        const uint32_t col = jt + rc;
        const float c0 = (float)(256u * kMatch4K + (kMatch4K - 1u) - col);
        const v4f_t cin = {c0, c0, c0, c0};
        v4f_t acc[kMatchRowTiles];
#pragma unroll
        for (int m = 0; m < kMatchRowTiles; m++) acc[m] = cin;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int m = 0; m < kMatchRowTiles; m++) {
                const v8i_t am = {a[m][t][0], a[m][t][1], a[m][t][2], a[m][t][3], 0, 0, 0, 0};
                acc[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(am, b[t], acc[m], 4, 4, 0, scale, 0, scale);
            }
#pragma unroll
        for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t key = __float_as_uint(acc[m][i]);
                if (MASK) key = col < nb ? key : 0u;
                const uint32_t b0 = best[m][i], s0 = second[m][i];
                second[m][i] = max(min(b0, key), min(max(b0, key), s0));  // = med3(b0, s0, key): hipcc emits v_med3_u32
                best[m][i] = max(b0, key);
            }
    };
    if (nb) {
        MATCH4_FETCH(0u);
        MATCH4_PUT(0);
    }
    __syncthreads();
    // The loop over whole chunks has NO branch around the tiles: a wave without queries (wave_live false: its rows repeat the frame's last
    // one, nothing of it is stored) multiplies like the others, and the frame's last, partial chunk is handled behind the loop.  With the
    // partial chunk and `if (wave_live)` inside the loop hipcc kept the 2 x 16 (best, runner-up) registers of the two paths in different
    // places and copied them back every iteration: 32 v_mov per chunk next to its 128 useful vector instructions (6.4 vector instructions
    // per MFMA in the counters where the tile's arithmetic needs 4; 5.3 without them, 0.623 -> 0.59 ms per 255 pairs).
    const uint8_t* const src_base = &stage[0][rc * (uint32_t)kMatch4RowBytes + 16u * g];
    const uint32_t n_whole = nb / (uint32_t)kMatchChunk, n_chunks = (nb + (uint32_t)kMatchChunk - 1u) / (uint32_t)kMatchChunk;
    uint32_t c = 0;
    for (; c < n_whole; c++) {
        const uint32_t j0 = c * (uint32_t)kMatchChunk;
        const bool more = c + 1u < n_chunks;
        if (more) MATCH4_FETCH(j0 + (uint32_t)kMatchChunk);
        const uint8_t* const src0 = src_base + (c & 1u) * kBufBytes;
#pragma unroll
        for (int tt = 0; tt < kMatchChunk / 16; tt++) tile(std::false_type{}, src0 + tt * 16 * kMatch4RowBytes, j0 + 16u * (uint32_t)tt);
        if (more) MATCH4_PUT((c & 1u) ^ 1u);
        __syncthreads();
    }
    if (c < n_chunks) {  // the partial chunk: whole tiles, then one masked tile
        const uint32_t j0 = c * (uint32_t)kMatchChunk;
        const uint8_t* const src0 = src_base + (c & 1u) * kBufBytes;
        for (uint32_t tt = 0; j0 + 16u * tt < nb; tt++) {
            const uint32_t jt = j0 + 16u * tt;
            if (jt + 16u <= nb)
                tile(std::false_type{}, src0 + tt * (uint32_t)(16 * kMatch4RowBytes), jt);
            else
                tile(std::true_type{}, src0 + tt * (uint32_t)(16 * kMatch4RowBytes), jt);
        }
    }
#undef MATCH4_FETCH
#undef MATCH4_PUT
    if (!wave_live) return;
#pragma unroll
    for (int m = 0; m < kMatchRowTiles; m++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t b1 = best[m][i], s1 = second[m][i];
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) {
                const uint32_t b2 = (uint32_t)__shfl_xor((int)b1, sh), s2 = (uint32_t)__shfl_xor((int)s1, sh);
                s1 = max(min(b1, b2), max(s1, s2));
                b1 = max(b1, b2);
            }
            const uint32_t q = q0 + 16u * (uint32_t)m + 4u * g + (uint32_t)i;
            if (rc == 0u && q < na) {
                MatchRecord r;
                const uint32_t k1 = (uint32_t)__uint_as_float(b1), k2 = (uint32_t)__uint_as_float(s1);  // exact: integers below 2^24
                const uint32_t v1 = k1 >> 14, v2 = k2 >> 14;           // 512 - 2 * distance
                r.index = k1 ? kMatch4K - 1u - (k1 & (kMatch4K - 1u)) : 0xffffffffu;
                const uint32_t d1 = k1 ? (512u - v1) >> 1 : 0xffffu;
                const uint32_t d2 = k2 ? (512u - v2) >> 1 : 0xffffu;
                r.dist = d1 | (d2 << 16);
                matches[(size_t)pair * cap + q] = r;
            }
        }
}

}  // namespace orb
