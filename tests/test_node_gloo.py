"""N > 1 host logic on CPU: frame sharding and the collate to rank 0, world_size 2 over gloo.

The compute of each rank is stood in for by the oracle (tests may use it); the code under test is
tinyslam_amd.node -- the same functions bench.py runs over RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, CAP, THR = 96, 64, 256, 20.0 / 255.0
N_FRAMES, SEED0 = 6, 300


def test_shard_range_partitions_contiguously():
    from tinyslam_amd import node
    for n, g in [(2048, 8), (256, 1), (10, 4), (3, 8), (7, 2)]:
        ranges = [node.shard_range(n, g, r) for r in range(g)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and a <= b
        assert max(b - a for a, b in ranges) - min(b - a for a, b in ranges) <= 1
    assert [node.shard_range(2048, 8, r) for r in (0, 7)] == [(0, 256), (1792, 2048)]


def _frame_result(oracle, seed):
    r = oracle.extract(oracle.synth_frame(W, H, seed), depth=2, threshold=THR, max_features=CAP)
    n = len(r["corners"])
    kp = np.zeros((CAP, 4), dtype=np.int32)
    ds = np.zeros((CAP, 8), dtype=np.int32)
    kp[:n] = np.stack([r["corners"][k] for k in ("x", "y", "angle", "octave")], 1).astype(np.int32)
    ds[:n] = r["descriptors"].view(np.int32)
    return r["total"], kp, ds


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orb_oracle
    from tinyslam_amd import node
    lo, hi = node.shard_range(N_FRAMES, world, rank)
    res = [_frame_result(orb_oracle, SEED0 + i) for i in range(lo, hi)]
    counts = torch.tensor([r[0] for r in res], dtype=torch.int32)
    corners = torch.from_numpy(np.stack([r[1] for r in res]))
    desc = torch.from_numpy(np.stack([r[2] for r in res]))
    out = node.collate_to_root(counts, corners, desc, CAP)
    if rank == 0:
        c, k, d = out
        np.savez(out_path, counts=c.numpy(), corners=k.numpy(), desc=d.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_collate_world2_equals_single_process(oracle, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "collated.npz")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    assert got["counts"].shape == (N_FRAMES,)
    mx = got["corners"].shape[1]
    for i in range(N_FRAMES):
        total, kp, ds = _frame_result(oracle, SEED0 + i)
        assert got["counts"][i] == total
        n = min(total, CAP)
        assert n <= mx
        assert np.array_equal(got["corners"][i, :n], kp[:n])
        assert np.array_equal(got["desc"][i, :n], ds[:n])
    assert mx == min(int(got["counts"].max()), CAP)


def test_collate_world1_is_a_trim():
    from tinyslam_amd import node
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        counts = torch.tensor([3, 5], dtype=torch.int32)
        corners = torch.arange(2 * 8 * 4, dtype=torch.int32).reshape(2, 8, 4)
        desc = torch.arange(2 * 8 * 8, dtype=torch.int32).reshape(2, 8, 8)
        c, k, d = node.collate_to_root(counts, corners, desc, 8)
        assert k.shape == (2, 5, 4) and d.shape == (2, 5, 8) and torch.equal(c, counts)
        assert torch.equal(k, corners[:, :5])
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _to_transport(kp, ds, n):
    """NumPy statement of the 40-byte transport record (include/tinyorb.h): {x | y << 16, angle | octave << 16, d[8]}."""
    rec = np.zeros((n, 10), dtype=np.int64)
    rec[:, 0] = kp[:n, 0] | (kp[:n, 1] << 16)
    rec[:, 1] = kp[:n, 2] | (kp[:n, 3] << 16)
    rec[:, 2:] = ds[:n].view(np.uint32)
    return rec.astype(np.uint32).view(np.int32)


def _worker_transport(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orb_oracle
    from tinyslam_amd import node
    lo, hi = node.shard_range(N_FRAMES, world, rank)
    res = [_frame_result(orb_oracle, SEED0 + i) for i in range(lo, hi)]
    counts = torch.tensor([r[0] for r in res], dtype=torch.int32)
    recs = [_to_transport(r[1], r[2], min(r[0], CAP)) for r in res]
    slack = np.full((7, 10), -1, dtype=np.int32)  # rows past the rank's total must not matter
    records = torch.from_numpy(np.concatenate(recs + [slack]))
    out = node.collate_transport_to_root(counts, records, CAP)
    if rank == 0:
        c, totals, merged = out
        np.savez(out_path, counts=c.numpy(), totals=np.array(totals), merged=merged.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_transport_collate_world2_equals_single_process(oracle, tmp_path):
    """collate_transport_to_root over gloo: rank r's records arrive as merged[r, :totals[r]], in frame order."""
    out_path = str(tmp_path / "collated_t.npz")
    mp.spawn(_worker_transport, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = np.load(out_path)
    from tinyslam_amd import node
    assert got["counts"].shape == (N_FRAMES,) and got["merged"].shape[0] == 2 and got["merged"].shape[2] == node.TRANSPORT_WORDS
    assert got["merged"].shape[1] == max(int(got["totals"].max()), 1)
    frame = 0
    for r in range(2):
        lo, hi = node.shard_range(N_FRAMES, 2, r)
        at = 0
        for i in range(lo, hi):
            total, kp, ds = _frame_result(oracle, SEED0 + i)
            assert got["counts"][frame] == total
            n = min(total, CAP)
            assert np.array_equal(got["merged"][r, at:at + n], _to_transport(kp, ds, n)), (r, i)
            at += n
            frame += 1
        assert at == got["totals"][r]


def test_transport_collate_world1():
    from tinyslam_amd import node
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        counts = torch.tensor([3, 9, 0], dtype=torch.int32)
        records = torch.arange(20 * 10, dtype=torch.int32).reshape(20, 10)
        c, totals, merged = node.collate_transport_to_root(counts, records, 8)  # 9 is cut to the capacity 8
        assert totals == [11] and merged.shape == (1, 11, 10) and torch.equal(merged[0], records[:11]) and torch.equal(c, counts)
    finally:
        dist.destroy_process_group()


def _collator_batches(world, B):
    """Five batches of very different sizes (an empty rank, a full one, a frame over the capacity)."""
    rng = np.random.RandomState(7)
    return [rng.randint(0, 4, size=(world, B)), rng.randint(0, 3, size=(world, B)), np.full((world, B), 9),
            np.stack([np.zeros(B, dtype=np.int64)] + [rng.randint(0, 5, size=B) for _ in range(world - 1)]),
            rng.randint(0, 9, size=(world, B))]


def _collator_records(k, rank, total, rows):
    """Transport records of batch k on `rank`: word 0 numbers them, word 9 mixes batch, rank and index; -1 behind them."""
    rec = torch.full((rows, 10), -1, dtype=torch.int32)
    idx = torch.arange(total, dtype=torch.int32)
    rec[:total, 0] = idx + 1000 * k + 100 * rank
    rec[:total, 9] = idx * 7 + k * 31 + rank
    return rec


def _worker_collator(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinyslam_amd import node
    B, cap = 3, 8
    col = node.TransportCollator(B, cap, "cpu")
    all_counts = _collator_batches(world, B)
    bufs = [None, None]  # the caller's two transport buffers, reused every other batch as in bench.py
    log, pending = [], None
    for k, cnt in enumerate(all_counts):
        if pending is not None:  # batch k-1: its counters went to the host a whole batch ago; its records move now
            info = col.exchange(pending)
            log.append((k - 1, info, None if info["merged"] is None else info["merged"].numpy().copy(), col.bytes_exchanged))
        mine = torch.tensor(cnt[rank], dtype=torch.int32)
        total = int(np.minimum(cnt[rank], cap).sum())
        bufs[k & 1] = _collator_records(k, rank, total, B * cap)  # overwrites batch k-2's records: they have been exchanged
        pending = col.submit(k & 1, mine, bufs[k & 1])
    info = col.exchange(pending)
    log.append((len(all_counts) - 1, info, None if info["merged"] is None else info["merged"].numpy().copy(), col.bytes_exchanged))
    if rank == 0:
        out = {}
        for k, info, merged, nbytes in log:
            out["counts%d" % k] = info["counts_all"]
            out["totals%d" % k] = np.array(info["totals"])
            out["first%d" % k] = np.array(info["first"])
            out["merged%d" % k] = merged
            out["bytes%d" % k] = np.array(nbytes)
        np.savez(out_path, **out)
    else:
        assert all(m is None for _, _, m, _ in log) and col.bytes_exchanged == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exact_lagged_transport_collator(tmp_path, world):
    """TransportCollator over gloo: the counters of batch k are read one batch late and size its exchange EXACTLY -- rank 0
    receives sum(S_r) records, back to back in rank order, record for record what the ranks held, whatever batches k-1
    and k+1 looked like (empty ranks, full ranks, a frame over the capacity); buffers are reused across batches."""
    out_path = str(tmp_path / "collator.npz")
    mp.spawn(_worker_collator, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    B, cap = 3, 8
    all_counts = _collator_batches(world, B)
    bytes_so_far = 0
    for k, cnt in enumerate(all_counts):
        assert np.array_equal(got["counts%d" % k], cnt.reshape(-1))
        totals = np.minimum(cnt, cap).sum(axis=1)
        assert np.array_equal(got["totals%d" % k], totals)
        assert np.array_equal(got["first%d" % k], np.concatenate([[0], np.cumsum(totals)]))
        merged = got["merged%d" % k]
        assert merged.shape == (int(totals.sum()), 10)  # not a record more
        bytes_so_far += int(totals.sum()) * 40
        assert int(got["bytes%d" % k]) == bytes_so_far  # exactly sum(S_r) x 40 bytes per batch
        at = 0
        for r in range(world):
            want = _collator_records(k, r, int(totals[r]), int(totals[r])).numpy()
            assert np.array_equal(merged[at:at + int(totals[r])], want), (k, r)
            at += int(totals[r])


def test_exact_collator_world1():
    """A group of one rank: the exchange is a send to itself (what `bench.py --gpus 1 --force-collate` runs over RCCL)."""
    from tinyslam_amd import node
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        col = node.TransportCollator(3, 8, "cpu")
        rec = _collator_records(0, 0, 14, 24)
        t = col.submit(0, torch.tensor([3, 11, 3], dtype=torch.int32), rec)  # 11 is cut to the capacity 8
        info = col.exchange(t)
        assert info["totals"] == [14] and info["first"] == [0, 14] and torch.equal(info["merged"], rec[:14])
        assert col.bytes_exchanged == 14 * 40
        with pytest.raises(ValueError):  # fewer records than the counters say: refused, never padded
            col.exchange(col.submit(1, torch.tensor([8, 8, 8], dtype=torch.int32), rec[:10]))
    finally:
        dist.destroy_process_group()
