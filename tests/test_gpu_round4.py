"""GPU parity, round 4: the RCCL code executed on a one-GPU box (orb_node_* with a one-rank communicator sending to
itself; bench.py's ranks path with a process group of one rank), the implementation-defined switches (out-of-level loads,
sampler weight precision) against the oracle, the asynchronous single-frame upload.  Bit for bit, as everywhere."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THR = 20.0 / 255.0


def _sorted(corners, desc):
    order = np.lexsort((corners["x"], corners["y"], corners["octave"]))
    return corners[order], desc[order]


def _assert_frame_equal(oracle, ref, total, corners, desc):
    assert total == ref["total"]
    c, d = _sorted(corners, desc)
    rc, rd = oracle.sort_keypoints(ref["corners"], ref["descriptors"])
    assert len(c) == len(rc)
    for k in ("octave", "y", "x", "angle"):
        assert np.array_equal(c[k], rc[k]), k
    assert np.array_equal(d, rd), "descriptors differ"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("ranks", [1, 3])
def test_node_collate_through_rccl_on_one_device(tinyorb, oracle, monkeypatch, ranks):
    """TINYORB_NODE_LOOPBACK=2: the node's exchange runs through librccl on this one GPU -- dlopen, ncclCommInitAll over
    the device, one group of ncclSend + ncclRecv per job with the exact payload sizes (every rank sends to the one-rank
    communicator's own rank; with one rank its own records take that way instead of k_compact), k_unpack_transport behind
    it on the exchange stream.  Four jobs of different sizes streamed with two outstanding, an empty shard among them;
    every frame against the oracle."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "2")
    W, H, CAP, B = 320, 240, 2048, 4
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=CAP, hierarchy_depth=2, initial_threshold=THR, max_batch=B)
    sizes = [4 * ranks, 2 * ranks + 1, 1, 3 * ranks][: 4]
    jobs = [np.stack([oracle.synth_frame(W, H, 5000 + 20 * j + i) for i in range(n)]) for j, n in enumerate(sizes)]

    def check(job, counts, offsets, kp, desc):
        for i in range(sizes[job]):
            ref = oracle.extract(jobs[job][i], depth=2, threshold=THR)
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])

    with tinyorb.OrbNode(cfg, [0] * ranks) as node:
        assert node.exchange_backend() == "rccl-self"
        for k in range(len(jobs)):
            node.extract_batch_host(jobs[k])
            if node.pending() == 2:
                node.collate_begin()
                counts, offsets, _, _ = node.collate_end(sizes[k - 1])
                kp, desc = node.read_collated(int(offsets[-1]))
                check(k - 1, counts, offsets, kp, desc)
        counts, offsets, _, _ = node.collate(sizes[-1])
        kp, desc = node.read_collated(int(offsets[-1]))
        check(len(jobs) - 1, counts, offsets, kp, desc)
        # every job moved at least one rank's records through ncclSend/ncclRecv
        assert node.rccl_pairs() >= len(jobs), node.rccl_pairs()


@pytest.mark.timeout(600)
def test_bench_ranks_path_collates_through_nccl_world1():
    """`bench.py --gpus 1 --force-collate`: the code the driver's N > 1 runs take -- process group "nccl" (RCCL),
    all_gather of the counters, the exact-size all_to_all_single of the transport records (a send to itself in a group of
    one), the expansion on rank 0 -- executed on one GPU; rank 0's frames must come out of it unchanged and the bytes
    moved must be exactly 40 per stored keypoint."""
    env = dict(os.environ, TINYORB_QUIET="1")
    env.pop("TINYORB_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collate", "--frames", "32", "--steps", "4",
           "--warmup", "1", "--repeats", "2", "--cpu-sample", "0", "--no-single-frame", "--preheat-ms", "20"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    col = rec["collate"]
    assert col["backend"] == "nccl" and col["form"] == "transport"
    assert col["root_check"] is True and col["exact"] is True
    kp = rec["keypoints_per_frame"] * 32
    assert abs(col["bytes_gathered_per_batch"] - 40 * kp) < 1e-6 * 40 * kp + 1
    assert rec["n_gpus"] == 1 and rec["value"] > 0
