// orb_kernels_collate.h -- results of a batch, packed for transport.
//
// The reference copies counter + corners + descriptors to host staging after every frame (orb.rs:537-565).  The
// batched pipeline keeps per-frame slabs of max_features records on the device (random access per frame for the
// matcher); k_compact packs the STORED records of all frames back to back, in frame order, with their offsets, into
// one destination that may be device memory (payload of the multi-GPU collate) or pinned host memory written
// straight over PCIe (bulk read-back without a size round trip through the host).
#pragma once
#include "orb_kernels_staged.h"

namespace orb {

constexpr uint32_t kCompactChunk = 1024;  // records per workgroup

// grid (n_frames, chunks_per_frame), 256 threads; the frame is the fast index so that the chunks past a frame's count
// (which exit at once) do not leave half of the XCDs without work.
//   counts[f]             raw per-frame counters of the batch (orb.rs:550-556)
//   corners/descriptors   [n_frames][cap] slabs
//   out_counts[f]         raw counter again (may be null)
//   out_offsets[f]        exclusive prefix of min(counts, cap); [n_frames] = total stored records (may be null)
//   out_c / out_d         [capacity] packed records; records past `capacity` are dropped (the total still says so)
__global__ __launch_bounds__(256) void k_compact(const uint32_t* __restrict__ counts, const CornerData* __restrict__ corners,
                                                 const CornerDescriptor* __restrict__ descriptors, uint32_t cap,
                                                 uint32_t n_frames, uint32_t* __restrict__ out_counts,
                                                 unsigned long long* __restrict__ out_offsets, CornerData* __restrict__ out_c,
                                                 CornerDescriptor* __restrict__ out_d, unsigned long long capacity) {
    __shared__ unsigned long long wave_sum[4];
    const uint32_t frame = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    // offset of this frame = stored records of the frames before it (<= a few thousand counts, L2 hits)
    unsigned long long part = 0;
    for (uint32_t f = tid; f < frame; f += 256u) part += (unsigned long long)min(counts[f], cap);
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) part += __shfl_xor(part, sh);
    if ((tid & 63u) == 0u) wave_sum[tid >> 6] = part;
    __syncthreads();
    const unsigned long long base = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    const uint32_t raw = counts[frame];
    const uint32_t stored = min(raw, cap);
    if (chunk == 0u && tid == 0u) {
        if (out_counts) out_counts[frame] = raw;
        if (out_offsets) {
            out_offsets[frame] = base;
            if (frame + 1u == n_frames) out_offsets[n_frames] = base + stored;
        }
    }
    const uint32_t r0 = chunk * kCompactChunk;
    if (r0 >= stored) return;
    const uint32_t n = min(stored - r0, kCompactChunk);
    const uint4* src_c = reinterpret_cast<const uint4*>(corners + (size_t)frame * cap + r0);
    const uint4* src_d = reinterpret_cast<const uint4*>(descriptors + (size_t)frame * cap + r0);
    uint4* dst_c = reinterpret_cast<uint4*>(out_c + base + r0);
    uint4* dst_d = reinterpret_cast<uint4*>(out_d + base + r0);
    unsigned long long room = capacity > base + r0 ? capacity - (base + r0) : 0ull;  // records that still fit
    const uint32_t m = (uint32_t)(room < n ? room : n);
    for (uint32_t i = tid; i < m; i += 256u) dst_c[i] = src_c[i];          // 16 B per record
    for (uint32_t i = tid; i < 2u * m; i += 256u) dst_d[i] = src_d[i];     // 32 B per record
}

// ---------------------------------------------------------------------------------------------
// Transport records: what crosses xGMI in the multi-GPU collate.  A CornerData is four u32 of which 37 bits are used
// (x, y < 65536; angle code < 6284; octave < 8): on the wire a keypoint is 2 + 8 words instead of 4 + 8 (40 B, -17 %),
// packed back to back in frame order (no padding to the fullest frame).  Lossless: k_unpack_transport restores the
// reference's two record layouts (orb.rs:10-23) on the receiving GPU.
// ---------------------------------------------------------------------------------------------
struct TransportRecord {
    uint32_t xy;  // x | y << 16
    uint32_t ao;  // angle | octave << 16
    uint32_t d[8];
};
static_assert(sizeof(TransportRecord) == ORB_TRANSPORT_RECORD_BYTES, "transport record layout");

// grid (n_frames, chunks_per_frame), 256 threads, frame = fast index (as k_compact).  out_offsets[f] = first record of
// frame f, [n_frames] = total; records past `capacity` are dropped (the total still says so).
__global__ __launch_bounds__(256) void k_compact_transport(const uint32_t* __restrict__ counts, const CornerData* __restrict__ corners,
                                                           const CornerDescriptor* __restrict__ descriptors, uint32_t cap,
                                                           uint32_t n_frames, unsigned long long* __restrict__ out_offsets,
                                                           TransportRecord* __restrict__ out, unsigned long long capacity) {
    __shared__ unsigned long long wave_sum[4];
    const uint32_t frame = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    unsigned long long part = 0;
    for (uint32_t f = tid; f < frame; f += 256u) part += (unsigned long long)min(counts[f], cap);
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) part += __shfl_xor(part, sh);
    if ((tid & 63u) == 0u) wave_sum[tid >> 6] = part;
    __syncthreads();
    const unsigned long long base = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    const uint32_t stored = min(counts[frame], cap);
    if (chunk == 0u && tid == 0u && out_offsets) {
        out_offsets[frame] = base;
        if (frame + 1u == n_frames) out_offsets[n_frames] = base + stored;
    }
    const uint32_t r0 = chunk * kCompactChunk;
    if (r0 >= stored) return;
    const uint32_t n = min(stored - r0, kCompactChunk);
    const unsigned long long room = capacity > base + r0 ? capacity - (base + r0) : 0ull;
    const uint32_t m = (uint32_t)(room < n ? room : n);
    const uint4* src_c = reinterpret_cast<const uint4*>(corners + (size_t)frame * cap + r0);
    const uint4* src_d = reinterpret_cast<const uint4*>(descriptors + (size_t)frame * cap + r0);
    uint2* dst = reinterpret_cast<uint2*>(out + base + r0);  // 40-byte records: 8-byte aligned
    for (uint32_t i = tid; i < m; i += 256u) {
        const uint4 c = src_c[i], d0 = src_d[2u * i], d1 = src_d[2u * i + 1u];
        uint2* o = dst + 5u * i;
        o[0] = make_uint2((c.x & 0xffffu) | (c.y << 16), (c.z & 0xffffu) | (c.w << 16));
        o[1] = make_uint2(d0.x, d0.y);
        o[2] = make_uint2(d0.z, d0.w);
        o[3] = make_uint2(d1.x, d1.y);
        o[4] = make_uint2(d1.z, d1.w);
    }
}

// Up to kUnpackSegments runs of transport records (one per sending GPU) -> the two record arrays, each run at its own
// place.  grid (chunks, n_segments), 256 threads, one record per thread.
constexpr int kUnpackSegments = 16;
struct UnpackGeom {
    unsigned long long src_first[kUnpackSegments], count[kUnpackSegments], dst_first[kUnpackSegments];
};
__global__ __launch_bounds__(256) void k_unpack_transport(const TransportRecord* __restrict__ in, UnpackGeom g,
                                                          CornerData* __restrict__ corners, CornerDescriptor* __restrict__ descriptors) {
    const uint32_t seg = blockIdx.y;
    const unsigned long long n = g.count[seg];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const uint2* r = reinterpret_cast<const uint2*>(in + g.src_first[seg] + i);
        const uint2 h = r[0], a = r[1], b = r[2], c = r[3], d = r[4];
        const unsigned long long o = g.dst_first[seg] + i;
        *reinterpret_cast<uint4*>(corners + o) = make_uint4(h.x & 0xffffu, h.x >> 16, h.y & 0xffffu, h.y >> 16);
        uint4* dd = reinterpret_cast<uint4*>(descriptors + o);
        dd[0] = make_uint4(a.x, a.y, b.x, b.y);
        dd[1] = make_uint4(c.x, c.y, d.x, d.y);
    }
}

}  // namespace orb
