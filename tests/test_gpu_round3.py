"""GPU parity, round 3: BASELINE.json configs[4] at its own size through the eight-rank code path, the pipelined
node-level collate (orb_node_collate_begin / _end), a Y8 node, the non-blocking pinned ingest.  Everything is compared
with the CPU oracle or with the plain batched call, bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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


def _read_dev_records(prog, tinyorb, c_ptr, d_ptr, total):
    kp = prog.copy_to_host(c_ptr, total * 16).view(tinyorb.CORNER_DTYPE)
    desc = prog.copy_to_host(d_ptr, total * 32).view(np.uint32).reshape(total, 8)
    return kp, desc


def test_node_pipeline_three_jobs_in_flight(tinyorb, oracle, monkeypatch):
    """extract(k) / collate_begin / collate_end as a stream of jobs on three loopback ranks: two jobs outstanding at a
    time, output sets, wire buffers, pinned counters and events reused across six jobs of different sizes, a third
    extract refused, and a result that is still intact after the next TWO jobs were extracted (three result buffers)."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "1")
    W, H, CAP, B = 320, 240, 2048, 4
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=CAP, hierarchy_depth=2, initial_threshold=THR, max_batch=B)
    sizes = [10, 12, 3, 7, 12, 1]
    jobs = [np.stack([oracle.synth_frame(W, H, 2000 + 20 * j + i) for i in range(n)]) for j, n in enumerate(sizes)]
    refs = {}

    def check(job, counts, offsets, kp, desc):
        frames = jobs[job]
        assert len(counts) == len(frames)
        for i in range(len(frames)):
            key = (job, i)
            if key not in refs:
                refs[key] = oracle.extract(frames[i], depth=2, threshold=THR)
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            _assert_frame_equal(oracle, refs[key], int(counts[i]), kp[lo:hi], desc[lo:hi])

    with tinyorb.OrbNode(cfg, [0, 0, 0]) as node:
        p0 = node.program(0)
        held = None  # (job, counts, offsets, device pointers): re-read two extracts later
        for k in range(len(jobs)):
            node.extract_batch_host(jobs[k])
            if node.pending() == 2:
                if k == 2:
                    with pytest.raises(tinyorb.OrbError):  # a third outstanding job is refused
                        node.extract_batch_host(jobs[k])
                node.collate_begin()  # job k - 1: its exchange overlaps job k's kernels
                counts, offsets, c_ptr, d_ptr = node.collate_end(sizes[k - 1])
                total = int(offsets[-1])
                kp, desc = _read_dev_records(p0, tinyorb, c_ptr, d_ptr, total)
                check(k - 1, counts, offsets, kp, desc)
                if held is not None:
                    # job k-2's result buffer: jobs k-1 and k were extracted since (the third extract would reuse it)
                    j, hc, ho, hcp, hdp = held
                    assert j == k - 2
                    kp2, desc2 = _read_dev_records(p0, tinyorb, hcp, hdp, int(ho[-1]))
                    check(j, hc, ho, kp2, desc2)
                held = (k - 1, counts, offsets, c_ptr, d_ptr)
        counts, offsets, c_ptr, d_ptr = node.collate_end(sizes[-1])  # begins the last job's exchange itself
        kp, desc = node.read_collated(int(offsets[-1]))
        check(len(jobs) - 1, counts, offsets, kp, desc)
        assert node.pending() == 0
        with pytest.raises(tinyorb.OrbError):
            node.collate_end(1)


def test_node_y8_shards_follow_the_frame_size(tinyorb, oracle, monkeypatch):
    """A node of ORB_FLAG_INPUT_Y8 programs takes W*H bytes per frame: rank r's shard starts r * shard * W*H bytes into
    the host array (round 2 sliced it at 4*W*H and read past the caller's buffer)."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "1")
    W, H, F = 320, 240, 7
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=4096, hierarchy_depth=2, initial_threshold=THR, max_batch=4,
                            flags=tinyorb.ORB_FLAG_INPUT_Y8)
    frames = np.stack([oracle.synth_frame_y8(W, H, 700 + i) for i in range(F)])
    assert frames.shape == (F, H, W)
    with tinyorb.OrbNode(cfg, [0, 0]) as node:
        node.extract_batch_host(frames)
        counts, offsets, _, _ = node.collate(F)
        kp, desc = node.read_collated(int(offsets[F]))
        for i in range(F):
            ref = oracle.extract_y8(frames[i], depth=2, threshold=THR)
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])


def test_pinned_ingest_does_not_block(tinyorb, oracle):
    """orb_extract_batch_pinned: frames in pinned memory, chunked upload (40 frames = three chunks), asynchronous; two
    batches back to back reuse the input slab in order."""
    W, H, B = 160, 120, 40
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=4096, hierarchy_depth=2, initial_threshold=THR, max_batch=B)
    with tinyorb.OrbProgram(cfg).init() as prog:
        pin = tinyorb.PinnedArray((2, B, H, W, 4), np.uint8)
        for j in range(2):
            for i in range(B):
                pin.array[j, i] = oracle.synth_frame(W, H, 3000 + 100 * j + i)
        for j in range(2):
            prog.extract_batch_pinned(pin.array[j], B)
            prog.upload_sync()
            counts = prog.batch_counts(B)
            for i in (0, 15, 16, 31, 32, 39):
                ref = oracle.extract(pin.array[j, i], depth=2, threshold=THR)
                corners, desc = prog.batch_read(i, int(counts[i]))
                _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
        pin.close()


def test_configs4_full_size_eight_ranks(tinyorb, oracle, monkeypatch):
    """BASELINE.json configs[4] at its own size: 2048 device-generated 1280x720 frames (seeds 1000..3047) sharded over
    EIGHT ranks of 256 frames through orb_node_* -- transport records, exact offsets, expansion on rank 0 -- with the
    eight ranks on this one GPU (TINYORB_NODE_LOOPBACK=1: the exchange runs as device copies instead of RCCL, which no
    one-GPU box can run; everything else is the code an 8-GPU node executes).  Checked: per-frame counters equal the
    plain 256-frame batches of a single program, offsets exact, frame order, no duplicate (octave, y, x) in any frame,
    and the frames either side of rank boundaries bit-equal to the oracle."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "1")
    W, H, CAP, B, RANKS = 1280, 720, 8192, 256, 8
    F = B * RANKS
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=CAP, hierarchy_depth=2, initial_threshold=THR, max_batch=B)
    with tinyorb.OrbNode(cfg, [0] * RANKS) as node:
        ptrs = []
        for r in range(RANKS):
            lo, hi = node.shard(F, r)
            assert (lo, hi) == (r * B, (r + 1) * B)
            ptrs.append(node.program(r).synth_frames_device(B, 1000 + lo))
        node.extract_batch(ptrs, F)
        counts, offsets, c_ptr, d_ptr = node.collate(F)
        total = int(offsets[F])
        stored = np.minimum(counts, CAP).astype(np.uint64)
        assert np.array_equal(offsets, np.concatenate([[0], np.cumsum(stored)]).astype(np.uint64))
        kp, desc = node.read_collated(total)
        assert len(kp) == total and 1000 * F < total < CAP * F
        # the same frames through one plain program, 256 at a time: counters and (for two batches) every record
        with tinyorb.OrbProgram(cfg).init() as prog:
            for r in range(RANKS):
                dev = prog.synth_frames_device(B, 1000 + r * B)
                prog.extract_batch_device(dev, B)
                c1 = prog.batch_counts(B)
                assert np.array_equal(c1, counts[r * B:(r + 1) * B]), r
                if r in (0, RANKS - 1):
                    hb = prog.batch_read_all(B)
                    lo, hi = int(offsets[r * B]), int(offsets[(r + 1) * B])
                    a, b = hb.corners[:hi - lo], hb.descriptors[:hi - lo].reshape(-1, 8)
                    # both are packed in frame order; inside a frame the records of a band list are appended in the order
                    # the waves finish (fast.wgsl:123-141 leaves it open too): compare as sorted per frame
                    fr = np.repeat(np.arange(B), np.minimum(c1, CAP).astype(np.int64))
                    oa = np.lexsort((a["x"], a["y"], a["octave"], fr))
                    kb, db = kp[lo:hi], desc[lo:hi].reshape(-1, 8)
                    ob = np.lexsort((kb["x"], kb["y"], kb["octave"], fr))
                    assert np.array_equal(a[oa], kb[ob]) and np.array_equal(b[oa], db[ob]), r
                    hb.close()
        # no keypoint twice in a frame (a compaction race would show here), octaves and coordinates in range
        frame_of = np.repeat(np.arange(F, dtype=np.int64), stored.astype(np.int64))
        assert kp["octave"].max() <= 1
        key = ((frame_of * 2 + kp["octave"].astype(np.int64)) * 1024 + kp["y"].astype(np.int64)) * 2048 + kp["x"].astype(np.int64)
        assert len(np.unique(key)) == total
        # against the oracle (CPU-generated frames, the restatement frame-parallel over the host's threads): both sides of every rank
        # boundary and every eighth frame in between -- 270 of the 2048 (round 5; six spot frames before)
        pick = sorted(set(range(0, F, 8)) | {r * B + d for r in range(RANKS) for d in (0, B - 1)})
        frames = np.stack([oracle.synth_frame(W, H, 1000 + f) for f in pick])
        totals, rc, rd = oracle.extract_batch(frames, depth=2, threshold=THR, max_features=CAP, n_threads=min(16, os.cpu_count() or 1))
        for k, f in enumerate(pick):
            n = min(int(totals[k]), CAP)
            lo, hi = int(offsets[f]), int(offsets[f + 1])
            _assert_frame_equal(oracle, dict(total=int(totals[k]), corners=rc[k, :n], descriptors=rd[k, :n]), int(counts[f]), kp[lo:hi], desc[lo:hi])


@pytest.mark.parametrize("intended", [False, True])
def test_rotated_pattern_table_equals_numpy(tinyorb, intended):
    """k_rot_table (brief.wgsl:50-57 / IM-6 evaluated once per angle code instead of once per keypoint): every entry of
    the program's table against NumPy binary32 arithmetic -- cos/sin of code / 1000 correctly rounded (CRD-10), every
    product and sum rounded on its own, truncation -- for all 3142 (6284) codes x 256 tests x 2 points."""
    from oracle import orb_numpy as on
    flags = (tinyorb.ORB_FLAG_INTENDED | tinyorb.ORB_FLAG_NMS) if intended else 0
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), max_features=1024, hierarchy_depth=2, initial_threshold=THR,
                            flags=flags, fast_arc=9 if intended else 0)
    with tinyorb.OrbProgram(cfg) as prog:
        table, pitch = prog.rot_table()
    codes = table.shape[0]
    assert codes == (6284 if intended else 3142) and pitch == (368 if intended else 48)
    F = np.float32
    theta = np.arange(codes, dtype=np.float32) / F(1000.0)
    ct = np.cos(theta.astype(np.float64)).astype(np.float32)
    st = np.sin(theta.astype(np.float64)).astype(np.float32)
    for j in range(256):
        ax, ay, bx, by = (F(v) for v in on.PATTERN[j])
        if intended:
            rax, ray = ct * ax + (-st) * ay, st * ax + ct * ay
            rbx, rby = ct * bx + (-st) * by, st * bx + ct * by
        else:
            rax, ray = ct * ax + st * ay, (-st) * ax + ct * ay
            rbx, rby = ct * bx + st * by, (-st) * bx + ct * by
        oa = 2 * (np.trunc(ray).astype(np.int64) * pitch + np.trunc(rax).astype(np.int64))
        ob = 2 * (np.trunc(rby).astype(np.int64) * pitch + np.trunc(rbx).astype(np.int64))
        assert np.array_equal(table[:, j & 63, j >> 6, 0].astype(np.int64), oa), "test %d point a" % j
        assert np.array_equal(table[:, j & 63, j >> 6, 1].astype(np.int64), ob), "test %d point b" % j


def test_single_frame_staging_is_complete_when_the_call_returns(tinyorb, oracle):
    """orb_extract_corners returns when the polled completion word arrives, not after a stream synchronisation: everything
    k_brief_one wrote to the pinned staging arrays must be there by then.  A large frame (19 500 keypoints, 1024 workgroups
    over all XCDs), 60 calls in a row, every result against the first (which is checked against the oracle).  The staging
    arrays are written with plain stores; what makes them visible is the protocol behind them (orb_kernels_brief.h, the end of
    k_brief_one): one lane per workgroup releases at system scope, then counts the workgroup done with an acquire-release
    add at agent scope, and the workgroup that completes the count publishes the sequence number the host polls.  With only
    the last workgroup releasing, one call in some thousands returned before the counter had arrived."""
    W, H, cap = 2048, 2200, 1 << 16
    rgba = oracle.synth_frame(W, H, 31)
    ref = oracle.extract(rgba, depth=2, threshold=THR, max_features=cap)
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=2, initial_threshold=THR)
    with tinyorb.OrbProgram(cfg) as prog:
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
        c0, d0 = _sorted(corners, desc)
        for it in range(60):
            t, c, d = prog.extract(rgba)
            assert t == total, "call %d: counter %d" % (it, t)
            c, d = _sorted(c, d)
            assert np.array_equal(c, c0) and np.array_equal(d, d0), "call %d: staging arrays incomplete" % it
