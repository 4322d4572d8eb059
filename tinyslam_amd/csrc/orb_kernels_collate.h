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

}  // namespace orb
