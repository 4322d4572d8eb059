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
    # round 5: one line separates kernel scaling from the collate -- the same repeats with the results left sharded -- and the
    # host's time inside a collate is split into waiting (the lagged counters' event) and enqueueing
    assert rec["value_sharded"] >= 0.9 * rec["value"] and rec["ms_per_step_sharded"] > 0
    assert abs(col["host_wait_ms_per_batch"] + col["host_enqueue_ms_per_batch"] - col["host_ms_per_batch"]) < 1e-9
    assert 0 <= col["host_wait_ms_per_batch"] <= col["host_ms_per_batch"] and col["host_enqueue_ms_per_batch"] > 0


# ---------------------------------------------------------------------------------------------
# The implementation-defined switches (include/tinyorb.h OrbOptions::oob_policy / sampler_weight_bits): the committed
# fixtures straight against the GPU, then shapes x policies against the oracle -- fused and staged kernels, the batch and
# the single-frame entries, Y8 input, the wave-per-keypoint BRIEF fallback.
# ---------------------------------------------------------------------------------------------
import glob
import hashlib

_IMPL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "impl", "*.npz")))


def _impl_program(tinyorb, W, H, depth, oob, wbits, flags=0, max_batch=1, cap=8192, thr=THR):
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=depth, initial_threshold=thr,
                            max_batch=max_batch, flags=flags, oob_policy=oob, sampler_weight_bits=wbits)
    return tinyorb.OrbProgram(cfg).init()


@pytest.mark.parametrize("path", _IMPL, ids=[os.path.basename(p) for p in _IMPL])
@pytest.mark.parametrize("flags", [0, 1])
def test_impl_switch_fixture_on_gpu(tinyorb, path, flags):
    """No oracle code runs here: the fixture holds the expected keypoints, descriptors and blur planes."""
    g = np.load(path)
    W, H, depth, seed, syn_flags, cap, oob, wbits = (int(v) for v in g["params"])
    with _impl_program(tinyorb, W, H, depth, oob, wbits, flags=flags, cap=cap, thr=float(g["threshold"])) as prog:
        assert prog.pipeline() == ("staged" if flags else "fused")
        dev = prog.synth_frames_device(1, seed, syn_flags)
        rgba = prog.copy_to_host(dev, W * H * 4)
        assert hashlib.sha256(rgba.tobytes()).hexdigest() == str(g["rgba_sha256"])
        prog.extract_batch_device(dev, 1)
        total = int(prog.batch_counts(1)[0])
        assert total == int(g["total"])
        corners, desc = prog.batch_read(0, total)
        c, d = _sorted(corners, desc)
        assert np.array_equal(np.stack([c[k] for k in ("x", "y", "angle", "octave")], 1), g["corners"])
        assert np.array_equal(d, g["descriptors"])
        for m in range(depth):
            b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m)
            assert hashlib.sha256(b.tobytes()).hexdigest() == str(g["blur_sha256"][m])


@pytest.mark.parametrize("W,H,depth,seed", [(320, 240, 3, 45), (333, 211, 4, 41), (1282, 96, 2, 17), (200, 97, 3, 7), (640, 360, 2, 3),
                                            (44, 36, 3, 12), (1241, 376, 3, 14), (2052, 80, 2, 6), (4100, 72, 2, 5)])
@pytest.mark.parametrize("oob,wbits", [("clamp", 0), ("umin", 8), ("zero", 8), ("clamp", 4)])
def test_impl_switches_match_oracle(tinyorb, oracle, W, H, depth, seed, oob, wbits):
    """Every out-of-level policy and a sampler weight precision on shapes that take different kernels (bands of 64..8 rows,
    odd widths and halvings: k_mip and the general level-0 variant, column tiles), fused and staged, through the batch
    entry and through the reference's six calls."""
    rgba = oracle.synth_frame(W, H, seed)
    ref = oracle.extract(rgba, depth=depth, threshold=THR, planes=True, oob=oob, weight_bits=wbits)
    dims, _ = oracle.level_dims(W, H, depth)
    for flags in (0, 1):
        with _impl_program(tinyorb, W, H, depth, tinyorb.OOB_POLICIES[oob], wbits, flags=flags) as prog:
            total, corners, desc = prog.extract(rgba)  # write_input_image + extract_corners + the two reads
            _assert_frame_equal(oracle, ref, total, corners, desc)
            for m, (w, h, off) in enumerate(dims):
                b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m)
                assert np.array_equal(b.ravel(), ref["blur"][off:off + w * h]), "blur level %d" % m


def test_impl_switches_batch_y8_and_brief_fallback(tinyorb, oracle, monkeypatch):
    """A batch of Y8 frames under clamp + 8-bit weights; then the same RGBA frames with the wave-per-keypoint BRIEF kernel
    forced (k_brief_rows, the fallback for frames whose row constants do not fit k_brief_t's LDS staging)."""
    W, H, B = 320, 240, 5
    y8 = np.stack([oracle.synth_frame_y8(W, H, 800 + i) for i in range(B)])
    with _impl_program(tinyorb, W, H, 3, tinyorb.ORB_OOB_CLAMP, 8, flags=tinyorb.ORB_FLAG_INPUT_Y8, max_batch=B) as prog:
        prog.extract_batch_host(y8)
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract_y8(y8[i], depth=3, threshold=THR, oob="clamp", weight_bits=8)
            _assert_frame_equal(oracle, ref, int(counts[i]), *prog.batch_read(i, min(int(counts[i]), 8192)))
    monkeypatch.setenv("TINYORB_BRIEF_ROWS", "1")
    frames = np.stack([oracle.synth_frame(W, H, 45 + i) for i in range(B)])
    for oob in ("umin", "clamp"):
        with _impl_program(tinyorb, W, H, 3, tinyorb.OOB_POLICIES[oob], 0, max_batch=B) as prog:
            prog.extract_batch_host(frames)
            counts = prog.batch_counts(B)
            for i in range(B):
                ref = oracle.extract(frames[i], depth=3, threshold=THR, oob=oob)
                _assert_frame_equal(oracle, ref, int(counts[i]), *prog.batch_read(i, min(int(counts[i]), 8192)))


def test_impl_switches_are_refused_with_the_extensions(tinyorb):
    """The switches follow the reference's adapter; the build's own extensions have no reference behaviour to follow."""
    for kw in (dict(flags=tinyorb.ORB_FLAG_INTENDED), dict(flags=tinyorb.ORB_FLAG_NMS), dict(fast_arc=9)):
        for sw in (dict(oob_policy=tinyorb.ORB_OOB_CLAMP), dict(sampler_weight_bits=8)):
            with pytest.raises(tinyorb.OrbError) as e:
                tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), **kw, **sw)).init()
            assert e.value.code == tinyorb.ORB_EINVAL
    for sw in (dict(oob_policy=3), dict(sampler_weight_bits=24)):
        with pytest.raises(tinyorb.OrbError):
            tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), **sw)).init()


_REF_DUMPS = sorted(d for d in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "reference_dump*"))
                    if os.path.exists(os.path.join(d, "total.npy")))


@pytest.mark.skipif(not _REF_DUMPS, reason="no dump of the reference (tests/golden/reference_dump*/): parity unpinned")
@pytest.mark.parametrize("dump", _REF_DUMPS or [None])
def test_reference_dump_on_gpu(tinyorb, dump):
    """The GPU path against the REFERENCE's own output (rust/dump_config0), with no oracle code in between: counter, sorted
    keypoints and descriptor bits, under the default switches."""
    total = int(np.load(os.path.join(dump, "total.npy")))
    corners = np.load(os.path.join(dump, "corners.npy")).astype(np.uint32).reshape(-1, 4)
    desc = np.load(os.path.join(dump, "descriptors.npy")).astype(np.uint32).reshape(-1, 8)
    W, H, depth, seed, syn_flags, cap = (int(v) for v in np.load(os.path.join(dump, "params.npy")))
    order = np.lexsort((corners[:, 0], corners[:, 1], corners[:, 3]))
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=depth, initial_threshold=THR)
    with tinyorb.OrbProgram(cfg).init() as prog:
        dev = prog.synth_frames_device(1, seed, syn_flags)
        prog.extract_batch_device(dev, 1)
        got_total = int(prog.batch_counts(1)[0])
        assert got_total == total
        c, d = _sorted(*prog.batch_read(0, min(total, cap)))
        assert np.array_equal(np.stack([c[k] for k in ("x", "y", "angle", "octave")], 1), corners[order])
        assert np.array_equal(d, desc[order])


# ---------------------------------------------------------------------------------------------
# orb_write_input_image_pinned: the reference's non-blocking upload (orb.rs:567-583), one image ahead
# ---------------------------------------------------------------------------------------------
def test_pinned_write_runs_one_image_ahead(tinyorb, oracle):
    """A camera loop: frame k + 1 goes up on the copy stream while frame k is extracted.  Images come out in the order they
    were written, each bit-equal to the oracle; a third write ahead is refused; a blocking write overwrites the newest
    waiting image (the last write wins, as in the reference); with nothing written, extract works on the last image again."""
    W, H, N = 640, 360, 7
    frames = [oracle.synth_frame(W, H, 900 + i) for i in range(N)]
    refs = [oracle.extract(f, depth=2, threshold=THR) for f in frames]
    pins = [tinyorb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=2, initial_threshold=THR)

    def result(prog):
        total = prog.extract_corners()
        n = min(total, 8192)
        return total, prog.read_corners(np.zeros(n, dtype=tinyorb.CORNER_DTYPE)), prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32))

    with tinyorb.OrbProgram(cfg).init() as prog:
        pins[0].array[:] = frames[0]
        prog.write_input_image_pinned(pins[0].array)
        for k in range(N):
            if k + 1 < N:
                prog.upload_sync()  # the pinned array of frame k - 1 ... k is free again
                pins[(k + 1) & 1].array[:] = frames[k + 1]
                prog.write_input_image_pinned(pins[(k + 1) & 1].array)  # under the kernels of frame k
            _assert_frame_equal(oracle, refs[k], *result(prog))
        _assert_frame_equal(oracle, refs[N - 1], *result(prog))  # nothing written since: the last image again
        # two ahead is the limit
        prog.write_input_image_pinned(pins[0].array)
        prog.write_input_image_pinned(pins[1].array)
        with pytest.raises(tinyorb.OrbError) as e:
            prog.write_input_image_pinned(pins[0].array)
        assert e.value.code == tinyorb.ORB_ESTATE
        prog.upload_sync()
        # a blocking write replaces the newest waiting image (pins[1]'s) and may be repeated
        prog.write_input_image(frames[2])
        prog.write_input_image(frames[3])
        first = pins[0].array.copy()
        _assert_frame_equal(oracle, oracle.extract(first, depth=2, threshold=THR), *result(prog))  # pins[0]'s image
        _assert_frame_equal(oracle, refs[3], *result(prog))  # then the overwritten slot: frame 3
    for pn in pins:
        pn.close()


def test_node_sharded_results_stay_on_their_devices(tinyorb, oracle, monkeypatch):
    """orb_node_set_results(SHARDED): no exchange -- every rank packs its own records on its own device.  Three loopback ranks,
    jobs streamed with two outstanding; rank by rank the records equal the oracle's, offsets keep counting across ranks,
    switching back to the collated form gives the same bytes in one array, and changing the mode with a job outstanding is
    refused."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "1")
    W, H, CAP, B, RANKS = 320, 240, 2048, 4, 3
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=CAP, hierarchy_depth=2, initial_threshold=THR, max_batch=B)
    sizes = [11, 5, 12]
    jobs = [np.stack([oracle.synth_frame(W, H, 7000 + 20 * j + i) for i in range(n)]) for j, n in enumerate(sizes)]
    with tinyorb.OrbNode(cfg, [0] * RANKS) as node:
        node.set_results(True)
        assert node.exchange_backend() == "copies"  # what an exchange WOULD use; none happens
        p0 = node.program(0)
        for k, frames in enumerate(jobs):
            node.extract_batch_host(frames)
            if k == 0:
                with pytest.raises(tinyorb.OrbError):
                    node.set_results(False)  # a job is outstanding
            counts, offsets, c_ptr, d_ptr = node.collate(sizes[k])
            assert c_ptr is None and d_ptr is None
            frame = 0
            for r in range(RANKS):
                lo, hi = node.shard(sizes[k], r)
                nf, nr, c_r, d_r = node.shard_result(r)
                assert nf == hi - lo and nr == int(offsets[hi]) - int(offsets[lo])
                kp = p0.copy_to_host(c_r, nr * 16).view(tinyorb.CORNER_DTYPE) if nr else np.zeros(0, tinyorb.CORNER_DTYPE)
                ds = p0.copy_to_host(d_r, nr * 32).view(np.uint32).reshape(nr, 8) if nr else np.zeros((0, 8), np.uint32)
                base = int(offsets[lo])
                for f in range(lo, hi):
                    ref = oracle.extract(frames[f], depth=2, threshold=THR, max_features=CAP)
                    a, b = int(offsets[f]) - base, int(offsets[f + 1]) - base
                    _assert_frame_equal(oracle, ref, int(counts[f]), kp[a:b], ds[a:b])
                    frame += 1
            assert frame == sizes[k]
            all_kp, all_ds = node.read_collated(int(offsets[-1]))  # rank by rank = frame order
        node.set_results(False)
        node.extract_batch_host(jobs[-1])
        counts2, offsets2, c_ptr, d_ptr = node.collate(sizes[-1])
        assert c_ptr and np.array_equal(counts2, counts) and np.array_equal(offsets2, offsets)
        kp2, ds2 = node.read_collated(int(offsets2[-1]))
        # inside a frame the records of a band list are appended in the order the waves finish: compare frame by frame, sorted
        for f in range(sizes[-1]):
            a, b = int(offsets[f]), int(offsets[f + 1])
            ca, da = _sorted(all_kp[a:b], all_ds[a:b])
            cb, db = _sorted(kp2[a:b], ds2[a:b])
            assert np.array_equal(ca, cb) and np.array_equal(da, db)


@pytest.mark.parametrize("W,H,depth", [(640, 480, 2), (640, 480, 4), (1280, 720, 3), (2560, 360, 3), (1241, 376, 3), (752, 480, 3), (1288, 200, 2),
                                       (644, 300, 2), (1920, 264, 4)])
def test_batch_programs_with_large_upper_level_bands(tinyorb, oracle, W, H, depth):
    """A batch program runs a level >= 1 on 1024 threads with sixteen-pixel pre-test items where a band of which two fit a CU
    holds 18 k pixels or more (k_front<false, ..., kFrontThreadsLNBig>; chosen per level at create): 64-row bands of 320 columns,
    32 x 640, 16 x 1280 and the widths in between, deeper pyramids, a level that does not halve exactly.  Three frames per batch,
    every frame against the oracle (planes of the levels >= 1 included); a dense frame exercises the detector's rounds there."""
    B = 3
    frames = np.stack([oracle.synth_frame(W, H, 8100 + i) for i in range(B)])
    rng = np.random.default_rng(W + H)
    frames[2, ..., :3] = ((rng.random((H, W, 1)) < 0.2) * 255).astype(np.uint8)  # salt and pepper: queues overflow
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=1 << 16, hierarchy_depth=depth, initial_threshold=THR, max_batch=B)
    dims, _ = oracle.level_dims(W, H, depth)
    with tinyorb.OrbProgram(cfg).init() as prog:
        assert prog.pipeline() == "fused"
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract(frames[i], depth=depth, threshold=THR, max_features=1 << 16, planes=True)
            _assert_frame_equal(oracle, ref, int(counts[i]), *prog.batch_read(i, min(int(counts[i]), 1 << 16)))
            for m, (w, h, off) in enumerate(dims):
                if m > 0:
                    g = prog.read_plane(tinyorb.ORB_PLANE_GRAY, m, frame=i)
                    assert np.array_equal(g.ravel(), ref["gray"][off:off + w * h]), "gray level %d of frame %d" % (m, i)
                b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m, frame=i)
                assert np.array_equal(b.ravel(), ref["blur"][off:off + w * h]), "blur level %d of frame %d" % (m, i)


# ---------------------------------------------------------------------------------------------
# SURVEY.md 8b, threading: "different handles may be used from different threads"
# ---------------------------------------------------------------------------------------------
@pytest.mark.timeout(300)
def test_programs_on_different_threads(tinyorb, oracle):
    """Four host threads, each with a program of its own (different sizes, modes and entry points: the single-frame calls,
    the pinned upload, a batch) created, used and destroyed concurrently -- ctypes releases the GIL inside every call, so the
    library's calls really overlap.  Every result bit-equal to the oracle; no call fails."""
    import threading
    jobs = [  # (W, H, depth, flags, kind)
        (640, 360, 2, 0, "single"),
        (320, 240, 3, 0, "pinned"),
        (512, 256, 2, 0, "batch"),
        (636, 200, 2, tinyorb.ORB_FLAG_STAGED, "single"),
    ]
    N = 6
    work = []
    for j, (W, H, depth, flags, kind) in enumerate(jobs):
        frames = [oracle.synth_frame(W, H, 1200 + 10 * j + i) for i in range(N)]
        work.append((frames, [oracle.extract(f, depth=depth, threshold=THR) for f in frames]))
    errors = []
    start = threading.Barrier(len(jobs))

    def run(j):
        W, H, depth, flags, kind = jobs[j]
        frames, refs = work[j]
        try:
            start.wait()
            for _ in range(2):  # create / destroy also overlap
                cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=depth, initial_threshold=THR, flags=flags,
                                        max_batch=N if kind == "batch" else 1)
                with tinyorb.OrbProgram(cfg).init() as prog:
                    if kind == "batch":
                        for _ in range(3):
                            prog.extract_batch_host(np.stack(frames))
                            counts = prog.batch_counts(N)
                            for i in range(N):
                                c, d = prog.batch_read(i, min(int(counts[i]), 8192))
                                _assert_frame_equal(oracle, refs[i], int(counts[i]), c, d)
                        continue
                    pin = tinyorb.PinnedArray((H, W, 4), np.uint8) if kind == "pinned" else None
                    for i in range(N):
                        if pin is not None:
                            pin.array[:] = frames[i]
                            prog.write_input_image_pinned(pin.array)
                        else:
                            prog.write_input_image(frames[i])
                        total = prog.extract_corners()
                        n = min(total, 8192)
                        _assert_frame_equal(oracle, refs[i], total, prog.read_corners(np.zeros(n, dtype=tinyorb.CORNER_DTYPE)),
                                            prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32)))
                        if pin is not None:
                            prog.upload_sync()
                    if pin is not None:
                        pin.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((j, repr(e)))

    threads = [threading.Thread(target=run, args=(j,)) for j in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# (CRD-13, OrbOptions::fp_contract: tests/test_gpu_round5.py -- since round 5 the fused kernels carry every form of it)
