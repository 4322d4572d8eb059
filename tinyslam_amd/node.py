"""Batched-frame sharding over the GPUs of one node and the final collate.

The reference has no multi-device code (single wgpu device, orb.rs:47-51).  Frames are
independent, so the path shards with NO data-path collective: rank g extracts the contiguous
frame range `shard_range(F, G, g)` on its own GPU.  The only exchange is the collate of the
results to rank 0: per-frame counts (all_gather, 4 B/frame) and the records -- as 40-byte
transport records packed back to back (`collate_transport_to_root`, what bench.py runs), or as the
keypoint/descriptor slabs trimmed to the largest per-frame count (`collate_to_root`).  With backend
"nccl" this is RCCL over xGMI; the CPU tests run the same code over gloo.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of frame indices owned by `rank` (collated output is in frame order)."""
    lo = (n_frames * rank) // world
    hi = (n_frames * (rank + 1)) // world
    return lo, hi


class DeviceArray:
    """Zero-copy view of a raw device allocation for torch (`torch.as_tensor(DeviceArray(...))`)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr,
                                         "version": 2, "strides": None}


def as_tensor(ptr: int, shape, typestr: str, device) -> torch.Tensor:
    return torch.as_tensor(DeviceArray(ptr, shape, typestr), device=device)


def collate_to_root(counts: torch.Tensor, corners: torch.Tensor, descriptors: torch.Tensor, cap: int,
                    dst: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Gathers every rank's results on `dst`.

    counts       (B,)        int32   raw per-frame counters of this rank's frames
    corners      (B, cap, 4) int32   CornerData slabs (x, y, angle, octave)
    descriptors  (B, cap, 8) int32   CornerDescriptor slabs
    Every rank must pass the same B (pad the last shard).  Returns on dst a tuple
    (counts_all (G*B,), corners_all (G*B, mx, 4), descriptors_all (G*B, mx, 8)) in frame order,
    where mx = the largest stored count of any frame on any rank; other ranks get None.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        mx = int(torch.clamp(counts, max=cap).max().item()) if counts.numel() else 0
        mx = max(mx, 1)  # same shape rule as the gathered case: at least one (possibly unused) record per frame
        return counts, corners[:, :mx].contiguous(), descriptors[:, :mx].contiguous()
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    counts_all = torch.cat(all_counts)
    mx = int(torch.clamp(counts_all, max=cap).max().item()) if counts_all.numel() else 0
    mx = max(mx, 1)
    # one payload per rank: corners and descriptors of a frame side by side -> a single gather, received straight
    # into the root's frame-ordered buffer (no concatenation or re-split afterwards: the returned corner and
    # descriptor tensors are views of it)
    payload = torch.cat([corners[:, :mx], descriptors[:, :mx]], dim=2).contiguous()  # (B, mx, 12)
    merged = None
    bufs = None
    if rank == dst:
        merged = torch.empty((world * payload.shape[0],) + tuple(payload.shape[1:]), dtype=payload.dtype,
                             device=payload.device)
        bufs = list(merged.chunk(world, dim=0))  # contiguous slices, one per rank
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return counts_all, merged[:, :, :4], merged[:, :, 4:]


TRANSPORT_WORDS = 10  # ORB_TRANSPORT_RECORD_BYTES / 4: {x | y << 16, angle | octave << 16, descriptor[8]}


def collate_transport_to_root(counts: torch.Tensor, records: torch.Tensor, cap: int, dst: int = 0,
                              group: Optional[dist.ProcessGroup] = None):
    """The collate on 40-byte transport records (orb_batch_pack_transport): what crosses the links is every rank's
    stored records back to back -- no padding to the fullest frame, 10 words per keypoint instead of 12.

    counts    (B,)      int32   raw per-frame counters of this rank's frames (same B on every rank)
    records   (>= S, 10) int32   this rank's transport records, the first S = sum(min(counts, cap)) rows valid
    Returns on dst (counts_all (G*B,), totals [S_0 .. S_{G-1}], merged (G, S_max, 10)) where rank r's records are
    merged[r, :totals[r]], in frame order; other ranks get None.  One all_gather of the counters (they size the
    gather), one gather of equal-size slices padded to the largest rank total (ranks hold the same number of frames,
    their totals differ by a per cent or two)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        total = int(torch.clamp(counts, max=cap).sum().item()) if counts.numel() else 0
        return counts, [total], records[:max(total, 1)].unsqueeze(0)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    counts_all = torch.cat(all_counts)
    totals = [int(t) for t in torch.clamp(counts_all, max=cap).view(world, -1).sum(dim=1).tolist()]
    s_max = max(max(totals), 1)
    if records.shape[0] >= s_max:
        payload = records[:s_max]  # contiguous leading slice
    else:  # a buffer sized for this rank alone: pad to the common size
        payload = records.new_zeros((s_max, records.shape[1]))
        payload[:records.shape[0]] = records
    merged = None
    bufs = None
    if rank == dst:
        merged = torch.empty((world, s_max, records.shape[1]), dtype=records.dtype, device=records.device)
        bufs = list(merged.unbind(0))  # contiguous slices, one per rank
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return counts_all, totals, merged
