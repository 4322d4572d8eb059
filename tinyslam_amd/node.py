"""Batched-frame sharding over the GPUs of one node and the final collate.

The reference has no multi-device code (single wgpu device, orb.rs:47-51).  Frames are
independent, so the path shards with NO data-path collective: rank g extracts the contiguous
frame range `shard_range(F, G, g)` on its own GPU.  The only exchange is the collate of the
results to rank 0: per-frame counts (all_gather, 4 B/frame) and the records -- as 40-byte
transport records packed back to back (`collate_transport_to_root`, what bench.py runs), or as the
keypoint/descriptor slabs trimmed to the largest per-frame count (`collate_to_root`).  With backend
"nccl" this is RCCL over xGMI; the CPU tests run the same code over gloo.
"""
import time
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


class TransportCollator:
    """The transport collate without a host stall per batch and without a byte too many (bench.py, N > 1).

    What sizes the exchange of a batch is its per-frame counters, and they are on the device.  Reading them there and
    then (`collate_transport_to_root`: all_gather + `.tolist()`) stalls the Python thread once per half-millisecond
    batch; sizing the exchange from the batch before (round 3: the largest rank total plus a quarter) never stalls but
    over-sends -- and into rank 0, where the links of a node meet, bytes are what the collate costs.  So the two halves
    of a batch's collate are one batch apart:

        submit(k)    all_gather of batch k's counters, their asynchronous copy to pinned host memory; batch k's transport
                     records stay where they are (the caller's buffer of output set k % slots);
        exchange(k)  one batch later, when the counters have long arrived: every rank reads the same totals S_0..S_{G-1}
                     and the records move in ONE all_to_all_single with split sizes -- rank r sends exactly S_r records to
                     `dst` and nothing to anybody else, `dst` receives them back to back in rank order (with the "nccl"
                     backend that is one group of ncclSend/ncclRecv, every peer on its own xGMI link into `dst`; the
                     root's own records take the same call, a copy inside its HBM).

        if ticket is not None:
            info = col.exchange(ticket)             # batch k-1: exact sizes, enqueue only
        ticket = col.submit(slot, counts, records)  # batch k

    `bytes_exchanged` counts what `dst` received: the sum of S_r x 40 bytes, exactly.  Buffers are allocated once; a
    slot's records and landing area are reused `slots` batches later, in stream order."""

    def __init__(self, n_frames: int, cap: int, device, dst: int = 0, group: Optional[dist.ProcessGroup] = None,
                 slots: int = 2):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.B, self.cap, self.dst, self.group = n_frames, cap, dst, group
        self.s_cap = n_frames * cap          # records a rank can hold
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        self.counts_all = [torch.empty(self.world * n_frames, dtype=torch.int32, device=self.device) for _ in range(slots)]
        self.h_counts = [torch.empty(self.world * n_frames, dtype=torch.int32, pin_memory=pin) for _ in range(slots)]
        self.copied = [None] * slots         # event behind the counters' copy to the host
        self.records = [None] * slots        # the rank's transport records of the batch in this slot
        self.merged = [None] * slots         # root: (world * s_cap, 10) landing area per slot, allocated on first use
        self.bytes_exchanged = 0
        self.wait_s = 0.0                    # host time spent waiting for a batch's counters in exchange() (the rest of a collate's host time is enqueueing)

    def submit(self, slot: int, counts: torch.Tensor, records: torch.Tensor):
        """Enqueues the first half of batch `slot`'s collate on the current stream: all_gather of the counters and their
        copy to pinned host memory.  `records` (this rank's transport records, the first sum(min(counts, cap)) rows
        valid) must stay untouched until `exchange` has been enqueued for the returned ticket."""
        chunks = list(self.counts_all[slot].chunk(self.world))
        dist.all_gather(chunks, counts, group=self.group)
        self.h_counts[slot].copy_(self.counts_all[slot], non_blocking=True)
        if self.device.type == "cuda":
            self.copied[slot] = torch.cuda.Event()
            self.copied[slot].record()
        self.records[slot] = records
        return slot

    def exchange(self, ticket):
        """Second half, one batch later: reads the counters (on the host by now), and enqueues the exact-size exchange.
        Returns {"slot", "counts_all" (numpy, G*B raw counters in frame order), "totals" [S_0..S_{G-1}], "first"
        (exclusive prefix of the totals, G+1 entries), "merged" (root: (sum S, 10) view, rank r's records are rows
        first[r]..first[r+1]; other ranks None)}."""
        slot = ticket
        if self.copied[slot] is not None:
            t0 = time.perf_counter()
            self.copied[slot].synchronize()  # one batch behind: normally set long ago -- wait_s says whether it was
            self.wait_s += time.perf_counter() - t0
        counts_all = self.h_counts[slot].numpy().copy()
        totals = [int(t) for t in counts_all.clip(max=self.cap).reshape(self.world, -1).sum(axis=1)]
        first = [0]
        for t in totals:
            first.append(first[-1] + t)
        records = self.records[slot]
        words = records.shape[1]
        mine = totals[self.rank]
        if records.shape[0] < mine:
            raise ValueError("rank %d holds %d transport records, its counters say %d" % (self.rank, records.shape[0], mine))
        in_splits = [mine if r == self.dst else 0 for r in range(self.world)]
        if self.rank == self.dst:
            if self.merged[slot] is None:
                self.merged[slot] = torch.empty((self.world * self.s_cap, words), dtype=records.dtype, device=records.device)
            out = self.merged[slot][:first[-1]]
            out_splits = totals
            self.bytes_exchanged += first[-1] * words * records.element_size()
        else:
            out = records.new_empty((0, words))
            out_splits = [0] * self.world
        dist.all_to_all_single(out, records[:mine], output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        return {"slot": slot, "counts_all": counts_all, "totals": totals, "first": first,
                "merged": out if self.rank == self.dst else None}
