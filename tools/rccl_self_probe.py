"""First contact with RCCL on a one-GPU box, outside pytest and under a timeout of the caller's:
    timeout -k 10 180 python tools/rccl_self_probe.py
1. orb_node_* with TINYORB_NODE_LOOPBACK=2 (one-rank communicator, ncclSend/ncclRecv to itself), n = 1 and n = 3;
2. torch.distributed "nccl" with one rank: all_gather + all_to_all_single with split sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TINYORB_NODE_LOOPBACK"] = "2"
import numpy as np  # noqa: E402
from tinyslam_amd import orb  # noqa: E402

W, H, B = 320, 240, 4
for ranks in (1, 3):
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=2048, max_batch=B)
    with orb.OrbNode(cfg, [0] * ranks) as node:
        F = B * ranks
        ptrs = [node.program(r).synth_frames_device(B, 100 + r * B) for r in range(ranks)]
        t0 = time.perf_counter()
        node.extract_batch(ptrs, F)
        counts, offsets, _, _ = node.collate(F)
        dt = time.perf_counter() - t0
        with orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_features=2048, max_batch=B)).init() as p:
            want = np.concatenate([(p.extract_batch_device(p.synth_frames_device(B, 100 + r * B), B), p.batch_counts(B))[1]
                                   for r in range(ranks)])
        print("node ranks=%d backend=%s rccl_pairs=%d first job %.1f ms counts_ok=%s total=%d" % (
            ranks, node.exchange_backend(), node.rccl_pairs(), dt * 1e3, bool(np.array_equal(counts, want)), int(offsets[-1])), flush=True)

import socket  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
c = torch.arange(8, dtype=torch.int32, device=dev)
out = [torch.empty_like(c)]
dist.all_gather(out, c)
rec = torch.arange(100 * 10, dtype=torch.int32, device=dev).reshape(100, 10)
land = torch.empty((37, 10), dtype=torch.int32, device=dev)
dist.all_to_all_single(land, rec[:37], output_split_sizes=[37], input_split_sizes=[37])
torch.cuda.synchronize()
print("torch nccl world 1: all_gather ok=%s all_to_all_single ok=%s" % (bool(torch.equal(out[0], c)), bool(torch.equal(land, rec[:37]))), flush=True)
dist.destroy_process_group()
