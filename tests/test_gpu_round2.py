"""GPU parity, round 2: the surfaces round 1 left untested (chunked host ingest, a non-Python caller of the C ABI,
orb_corner_level0_xy, wide literal frames, two programs alive at once, octaves without a FAST dispatch) and the new
entries (bulk read-back, node-level API).  Everything is compared with the CPU oracle, bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THR = 20.0 / 255.0


def _program(tinyorb, W, H, depth=2, max_features=8192, max_batch=1, flags=0, thr=THR, fast_arc=0):
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=max_features, hierarchy_depth=depth,
                            initial_threshold=thr, max_batch=max_batch, flags=flags, fast_arc=fast_arc)
    return tinyorb.OrbProgram(cfg).init()


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


def _compile(src, out):
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src),
                           "-L" + os.path.join(ROOT, "tinyslam_amd"), "-ltinyorb",
                           "-Wl,-rpath," + os.path.join(ROOT, "tinyslam_amd"), "-o", out])


def test_chunked_host_ingest_equals_per_frame_oracle(tinyorb, oracle):
    """orb_extract_batch_host with more frames than one upload chunk (16) and a ragged last chunk: the copy stream, the
    per-chunk events and run_fused_range with f0 > 0 (orb_api.hip)."""
    W, H, B = 160, 120, 40
    frames = np.stack([oracle.synth_frame(W, H, 500 + i) for i in range(B)])
    keep = frames.copy()
    with _program(tinyorb, W, H, 2, max_batch=B) as prog:
        assert prog.pipeline() == "fused"
        prog.extract_batch_host(frames)
        assert np.array_equal(frames, keep), "the caller's frames were modified"
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract(frames[i], depth=2, threshold=THR)
            corners, desc = prog.batch_read(i, int(counts[i]))
            _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
        # the bulk read-back of the same batch says the same (pinned destination, written by the device)
        hb = prog.batch_read_all(B)
        assert np.array_equal(hb.counts, counts)
        assert np.array_equal(hb.offsets, np.concatenate([[0], np.cumsum(np.minimum(counts, 8192))]).astype(np.uint64))
        for i in (0, 15, 16, 17, 39):
            c, d = hb.frame(i)
            c2, d2 = prog.batch_read(i, int(counts[i]))
            assert np.array_equal(c, c2) and np.array_equal(d, d2)
        hb.close()


def test_pack_and_fetch_equal_read_all(tinyorb, oracle):
    """The two-step streaming read-back (orb_batch_pack + orb_batch_fetch: exact-size DMA copies) against the one-call
    form and the per-frame reads, on both output sets."""
    W, H, B = 320, 240, 6
    with _program(tinyorb, W, H, 2, max_batch=B, max_features=1024, flags=tinyorb.ORB_FLAG_DOUBLE_OUTPUT) as prog:
        outs = []
        for s_, seed in ((0, 400), (1, 420)):
            prog.batch_select_output(s_)
            dev = prog.synth_frames_device(B, seed)
            prog.extract_batch_device(dev, B)
            prog.batch_pack(B)
            outs.append((s_, seed))
        for s_, seed in outs:
            hb = tinyorb.HostBatch(B, B * 1024)
            prog.batch_fetch(s_, hb)
            prog.stream_sync()
            prog.batch_select_output(s_)
            counts = prog.batch_counts(B)
            assert np.array_equal(hb.counts, counts)
            assert np.array_equal(hb.offsets, np.concatenate([[0], np.cumsum(np.minimum(counts, 1024))]).astype(np.uint64))
            for i in range(B):
                c, d = hb.frame(i)
                c2, d2 = prog.batch_read(i, int(min(counts[i], 1024)))
                assert np.array_equal(c, c2) and np.array_equal(d, d2)
                ref = oracle.extract(oracle.synth_frame(W, H, seed + i), depth=2, threshold=THR, max_features=1024)
                if counts[i] <= 1024:
                    _assert_frame_equal(oracle, ref, int(counts[i]), c, d)
            hb.close()
        with pytest.raises(tinyorb.OrbError):  # a set that was never packed
            with _program(tinyorb, 64, 48, 2, max_batch=2) as p2:
                p2.batch_fetch(0, tinyorb.HostBatch(2, 16))


def test_read_all_capacity_and_pageable_buffers(tinyorb, oracle):
    W, H, B, cap = 160, 120, 5, 64  # cap below the per-frame count: stored = min(raw, cap)
    frames = np.stack([oracle.synth_frame(W, H, 900 + i) for i in range(B)])
    with _program(tinyorb, W, H, 2, max_batch=B, max_features=cap) as prog:
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(B)
        assert counts.max() > cap  # the raw counter is reported (orb.rs:550-556), records are capped
        stored = np.minimum(counts, cap)
        total = int(stored.sum())
        # pageable numpy destinations: pinned for the duration of the call
        c_counts = np.zeros(B, np.uint32)
        c_off = np.zeros(B + 1, np.uint64)
        c_kp = np.zeros(total, tinyorb.CORNER_DTYPE)
        c_d = np.zeros((total, 8), np.uint32)
        L = prog._lib
        rc = L.orb_batch_read_all(prog._handle(), B, c_counts.ctypes.data, c_off.ctypes.data, c_kp.ctypes.data, c_d.ctypes.data,
                                  total, None)
        assert rc == 0
        assert np.array_equal(c_counts, counts) and int(c_off[B]) == total
        for i in range(B):
            kp, d = prog.batch_read(i, int(stored[i]))
            lo, hi = int(c_off[i]), int(c_off[i + 1])
            assert np.array_equal(c_kp[lo:hi], kp) and np.array_equal(c_d[lo:hi], d)
        # a destination that is too small: records past the capacity are dropped, the total still reports them
        small = total - int(stored[-1]) - 3
        s_kp = np.full(small + 8, 0xEE, np.uint8).view(np.uint8)
        s_kp = np.zeros(small + 8, tinyorb.CORNER_DTYPE)
        s_kp["x"] = 0xEEEEEEEE
        s_d = np.zeros((small + 8, 8), np.uint32)
        rc = L.orb_batch_read_all(prog._handle(), B, None, c_off.ctypes.data, s_kp.ctypes.data, s_d.ctypes.data, small, None)
        assert rc == 0 and int(c_off[B]) == total
        assert np.array_equal(s_kp[:small], c_kp[:small]) and np.all(s_kp["x"][small:] == 0xEEEEEEEE)


def test_c_caller_matches_oracle(tinyorb, oracle, tmp_path):
    """examples/minimal.c, compiled here with gcc against include/tinyorb.h and libtinyorb.so: a non-Python caller of
    the six reference methods (stands in for the Rust shim, which cannot be built in this image)."""
    exe = str(tmp_path / "minimal")
    _compile("minimal.c", exe)
    # 1. the built-in checker frame
    dump = str(tmp_path / "dump.bin")
    out = subprocess.run([exe, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    i = np.arange(640 * 480)
    v = np.where(((i % 640) // 3 % 5 == 0) & ((i // 640) // 3 % 5 == 0), 255, 0).astype(np.uint8)
    rgba = np.stack([v, v, v, np.full_like(v, 255)], axis=1).reshape(480, 640, 4)
    for frame_path, frame in ((None, rgba), (str(tmp_path / "frame.rgba"), oracle.synth_frame(640, 480, 1))):
        if frame_path:
            frame.tofile(frame_path)
            out = subprocess.run([exe, dump, frame_path], capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr
        raw = open(dump, "rb").read()
        total, n = np.frombuffer(raw[:8], np.uint32)
        kp = np.frombuffer(raw[8:8 + 16 * n], tinyorb.CORNER_DTYPE)
        desc = np.frombuffer(raw[8 + 16 * n:8 + 48 * n], np.uint32).reshape(n, 8)
        ref = oracle.extract(frame, depth=2, threshold=THR, max_features=4096)
        assert int(total) == ref["total"] and ("%d keypoints (fused pipeline)" % total) in out.stdout
        if total <= 4096:
            _assert_frame_equal(oracle, ref, int(total), kp, desc)
        else:  # more corners than max_features: which ones are kept is unspecified (Q9/Q10); every kept one must be real
            full = oracle.extract(frame, depth=2, threshold=THR, max_features=1 << 20)
            want = {(int(c["octave"]), int(c["y"]), int(c["x"])): (int(c["angle"]), d.tobytes())
                    for c, d in zip(full["corners"], full["descriptors"])}
            assert n == 4096
            for c, d in zip(kp, desc):
                assert want[(int(c["octave"]), int(c["y"]), int(c["x"]))] == (int(c["angle"]), d.tobytes())


def test_c_node_caller_one_device(tinyorb, oracle, tmp_path):
    """examples/node_batch.c: orb_node_* (n = 1: RCCL communicator of one rank, degenerate collate) against the
    single-device bulk read-back, and both against the oracle."""
    exe = str(tmp_path / "node_batch")
    _compile("node_batch.c", exe)
    W, H, F = 160, 120, 7
    frames = np.stack([oracle.synth_frame(W, H, 300 + i) for i in range(F)])
    fpath, opath = str(tmp_path / "frames.rgba"), str(tmp_path / "out.bin")
    frames.tofile(fpath)
    out = subprocess.run([exe, fpath, str(W), str(H), str(F), opath, "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "node collate == single-device read-back" in out.stdout
    raw = open(opath, "rb").read()
    assert np.frombuffer(raw[:4], np.uint32)[0] == F
    counts = np.frombuffer(raw[4:4 + 4 * F], np.uint32)
    offsets = np.frombuffer(raw[4 + 4 * F:4 + 4 * F + 8 * (F + 1)], np.uint64)
    base = 4 + 4 * F + 8 * (F + 1)
    total = int(offsets[F])
    kp = np.frombuffer(raw[base:base + 16 * total], tinyorb.CORNER_DTYPE)
    desc = np.frombuffer(raw[base + 16 * total:base + 48 * total], np.uint32).reshape(total, 8)
    for i in range(F):
        ref = oracle.extract(frames[i], depth=2, threshold=THR, max_features=2048)
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        assert hi - lo == min(ref["total"], 2048)
        _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])


def test_node_api_one_device_equals_plain_batch(tinyorb, oracle):
    """orb_node_* through the Python mirror: device-resident shards, collate, read-back; bit-equal to the plain batched
    call on the same frames."""
    W, H, F = 320, 240, 6
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=4096, hierarchy_depth=2, initial_threshold=THR, max_batch=F)
    with tinyorb.OrbNode(cfg, [0]) as node:
        assert node.device_count() == 1 and node.shard(F, 0) == (0, F)
        prog = node.program(0)
        dev = prog.synth_frames_device(F, 700)
        node.extract_batch([dev], F)
        counts, offsets, c_dev, d_dev = node.collate(F)
        kp, desc = node.read_collated(int(offsets[F]))
        # the node's own program still holds the padded slabs of the same batch
        for i in range(F):
            c2, d2 = prog.batch_read(i, int(min(counts[i], 4096)))
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            assert np.array_equal(kp[lo:hi], c2) and np.array_equal(desc[lo:hi], d2)
            ref = oracle.extract(oracle.synth_frame(W, H, 700 + i), depth=2, threshold=THR, max_features=4096)
            _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])
        # a second job on the same node (buffers are reused), from host frames this time
        frames = np.stack([oracle.synth_frame(W, H, 800 + i) for i in range(4)])
        node.extract_batch_host(frames)
        counts, offsets, _, _ = node.collate(4)
        kp, desc = node.read_collated(int(offsets[4]))
        for i in range(4):
            ref = oracle.extract(frames[i], depth=2, threshold=THR, max_features=4096)
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])
    with pytest.raises(tinyorb.OrbError):
        tinyorb.OrbNode(cfg, [0, 0]).init()  # a device listed twice


def test_corner_level0_xy_matches_python_mirror(tinyorb):
    L = tinyorb.load_library()
    rng = np.random.default_rng(3)
    c = np.zeros(512, tinyorb.CORNER_DTYPE)
    c["x"] = rng.integers(0, 4000, 512)
    c["y"] = rng.integers(0, 3000, 512)
    c["octave"] = rng.integers(0, 10, 512)
    wx, wy = tinyorb.level0_xy(c)
    for i in range(512):
        x0, y0 = ctypes.c_float(), ctypes.c_float()
        L.orb_corner_level0_xy(c[i:i + 1].ctypes.data, ctypes.byref(x0), ctypes.byref(y0))
        assert np.float32(x0.value) == wx[i] and np.float32(y0.value) == wy[i]
    # octave 0 is the identity, octave m scales the pixel centre by 2^m
    one = np.zeros(1, tinyorb.CORNER_DTYPE)
    one["x"], one["y"], one["octave"] = 7, 9, 2
    x0, y0 = ctypes.c_float(), ctypes.c_float()
    L.orb_corner_level0_xy(one.ctypes.data, ctypes.byref(x0), ctypes.byref(y0))
    assert (x0.value, y0.value) == (29.5, 37.5)


@pytest.mark.parametrize("W,H,depth", [(3840, 96, 2), (2560, 64, 3), (4096, 40, 2), (2052, 100, 2), (4100, 24, 1), (2050, 48, 2),
                                       (1920, 120, 3), (5120, 40, 2), (8200, 24, 2), (1442, 70, 2)])
def test_wide_literal_frames(tinyorb, oracle, W, H, depth):
    """Literal mode on levels too wide for two full-width bands per CU: they run on column tiles (k_front<..., TILED>) --
    tile 0 doing the band's blur with three grey columns fetched from beyond its own --, also on widths that are not a
    multiple of 4 (the general level-0 variant), past 4096 (round 2 sent those to the per-stage kernels) and where the
    next level runs on full-width bands again.  The fused pipeline takes all of them, and the results are the oracle's."""
    rgba = oracle.synth_frame(W, H, 21)
    ref = oracle.extract(rgba, depth=depth, threshold=THR, planes=True)
    with _program(tinyorb, W, H, depth) as prog:
        assert prog.pipeline() == "fused" and prog.pipeline_note() == ""
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m)
            assert np.array_equal(b.ravel(), ref["blur"][off:off + w * h]), "blur level %d" % m
            if m > 0:
                g = prog.read_plane(tinyorb.ORB_PLANE_GRAY, m)
                assert np.array_equal(g.ravel(), ref["gray"][off:off + w * h]), "gray level %d" % m


def test_wide_batch_and_y8(tinyorb, oracle):
    """The 8-row-band kernels in batched mode (several frames, XCD swizzle on) and with one-byte-per-pixel input."""
    W, H, B = 2560, 72, 8
    with _program(tinyorb, W, H, 2, max_batch=B) as prog:
        assert prog.pipeline() == "fused"
        dev = prog.synth_frames_device(B, 900)
        prog.extract_batch_device(dev, B)
        counts = prog.batch_counts(B)
        for i in (0, 3, 7):
            ref = oracle.extract(oracle.synth_frame(W, H, 900 + i), depth=2, threshold=THR)
            corners, desc = prog.batch_read(i, int(counts[i]))
            _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
    y8 = oracle.synth_frame_y8(3072, 64, 33)
    ref = oracle.extract_y8(y8, depth=2, threshold=THR)
    with _program(tinyorb, 3072, 64, 2, flags=tinyorb.ORB_FLAG_INPUT_Y8) as prog:
        assert prog.pipeline() == "fused"
        total, corners, desc = prog.extract(y8)
        _assert_frame_equal(oracle, ref, total, corners, desc)


def test_two_programs_of_different_size_alive(tinyorb, oracle):
    """The dynamic-LDS attribute of k_front belongs to the function, not to a program: a small program created while a
    large one is alive must not break the large one's launches."""
    big_f = oracle.synth_frame(1280, 96, 5)
    small_f = oracle.synth_frame(64, 48, 6)
    big_ref = oracle.extract(big_f, depth=2, threshold=THR)
    small_ref = oracle.extract(small_f, depth=2, threshold=THR)
    with _program(tinyorb, 1280, 96, 2) as big:
        t, c, d = big.extract(big_f)
        _assert_frame_equal(oracle, big_ref, t, c, d)
        with _program(tinyorb, 64, 48, 2) as small:
            for _ in range(2):
                t, c, d = small.extract(small_f)
                _assert_frame_equal(oracle, small_ref, t, c, d)
                t, c, d = big.extract(big_f)
                _assert_frame_equal(oracle, big_ref, t, c, d)


@pytest.mark.parametrize("flags,arc", [(4, 0), (0, 9), (4, 9), (0, 0)])
def test_octaves_without_a_fast_dispatch(tinyorb, oracle, flags, arc):
    """64x48 with depth 7: level 6 is 1 texel wide and 0 rows of dispatch (48 >> 6 == 0, orb.rs:511-519).  The fused
    arc/NMS pipeline, the staged one and the oracle must agree (the earlier levels' counts must survive)."""
    W, H, depth = 64, 48, 7
    rgba = oracle.synth_frame(W, H, 17).copy()
    rgba[:, :, :3] //= 4  # dim background + bright blobs inside the 16-px guard, so that FAST-12 fires on this tiny frame
    for (x, y, s) in ((22, 21, 2), (30, 25, 3), (40, 22, 1), (26, 27, 2), (36, 26, 3), (44, 24, 2)):
        rgba[y:y + s, x:x + s, :3] = 255
    nms = bool(flags & 4)
    if nms or arc:
        ref = oracle.extract_ex(rgba, depth=depth, threshold=THR, arc=arc or 12, nms=nms)
    else:
        ref = oracle.extract(rgba, depth=depth, threshold=THR)
    assert ref["total"] > 0
    results = []
    for staged in (0, 1):
        with _program(tinyorb, W, H, depth, flags=flags | staged, fast_arc=arc) as prog:
            total, corners, desc = prog.extract(rgba)
            _assert_frame_equal(oracle, ref, total, corners, desc)
            results.append(total)
    assert results[0] == results[1]


# ---------------------------------------------------------------------------------------------
# Y8 input variant (ORB_FLAG_INPUT_Y8; SURVEY.md 8f rank 3, the reference's roadmap README.md:42)
# ---------------------------------------------------------------------------------------------
import glob
import hashlib

_Y8_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "y8", "*.npz")))


@pytest.mark.parametrize("W,H,depth,seed", [(64, 48, 2, 3), (160, 120, 3, 1), (1280, 96, 2, 4), (332, 202, 4, 9),
                                            (2052, 40, 1, 6), (200, 97, 3, 7), (1241, 376, 3, 10), (333, 77, 3, 11),
                                            (1282, 50, 2, 12), (9, 9, 2, 13)])
@pytest.mark.parametrize("staged", [0, 1])
def test_y8_matches_oracle(tinyorb, oracle, W, H, depth, seed, staged):
    y8 = oracle.synth_frame_y8(W, H, seed)
    ref = oracle.extract_y8(y8, depth=depth, threshold=THR, planes=True)
    with _program(tinyorb, W, H, depth, flags=tinyorb.ORB_FLAG_INPUT_Y8 | staged) as prog:
        assert prog.pipeline() == ("staged" if staged else "fused")  # any width, any halving (the general level-0 variant)
        total, corners, desc = prog.extract(y8)
        _assert_frame_equal(oracle, ref, total, corners, desc)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            if m > 0 or prog.pipeline() == "staged":
                g = prog.read_plane(tinyorb.ORB_PLANE_GRAY, m)
                assert np.array_equal(g.ravel(), ref["gray"][off:off + w * h]), "gray level %d" % m
            b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m)
            assert np.array_equal(b.ravel(), ref["blur"][off:off + w * h]), "blur level %d" % m
        # an RGBA-sized buffer is refused: a Y8 program takes W*H bytes
        with pytest.raises(tinyorb.OrbError):
            prog.write_input_image(np.zeros((H, W, 4), np.uint8))


@pytest.mark.parametrize("path", _Y8_GOLDEN, ids=[os.path.basename(p) for p in _Y8_GOLDEN])
@pytest.mark.parametrize("staged", [0, 1])
def test_y8_golden_fixture_on_gpu(tinyorb, path, staged):
    g = np.load(path)
    W, H, depth, seed, syn_flags, cap = (int(v) for v in g["params"])
    with _program(tinyorb, W, H, depth, max_features=cap, flags=tinyorb.ORB_FLAG_INPUT_Y8 | staged,
                  thr=float(g["threshold"])) as prog:
        dev = prog.synth_frames_device(1, seed, syn_flags)  # a Y8 program generates one byte per pixel (integer luma)
        y8 = prog.copy_to_host(dev, W * H)
        assert hashlib.sha256(y8.tobytes()).hexdigest() == str(g["y8_sha256"])
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


def test_y8_batch_and_rejections(tinyorb, oracle):
    W, H, B = 320, 240, 5
    with _program(tinyorb, W, H, 2, max_batch=B, flags=tinyorb.ORB_FLAG_INPUT_Y8) as prog:
        frames = np.stack([oracle.synth_frame_y8(W, H, 60 + i) for i in range(B)])
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract_y8(frames[i], depth=2, threshold=THR)
            corners, desc = prog.batch_read(i, int(counts[i]))
            _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
    # the opt-in detectors have no Y8 definition to check against: refused, not guessed
    for kw in (dict(flags=tinyorb.ORB_FLAG_INPUT_Y8 | tinyorb.ORB_FLAG_INTENDED),
               dict(flags=tinyorb.ORB_FLAG_INPUT_Y8 | tinyorb.ORB_FLAG_NMS), dict(flags=tinyorb.ORB_FLAG_INPUT_Y8, fast_arc=9)):
        with pytest.raises(tinyorb.OrbError):
            _program(tinyorb, W, H, 2, **kw)


def test_forced_wave_per_keypoint_brief(tinyorb, oracle, monkeypatch):
    """TINYORB_BRIEF_ROWS=1 keeps k_brief_rows (the fallback of frames too large for k_brief_t's LDS staging) in use: it
    must still agree with the oracle."""
    monkeypatch.setenv("TINYORB_BRIEF_ROWS", "1")
    for (W, H, depth, seed) in ((640, 480, 2, 1), (332, 202, 5, 11), (1280, 96, 2, 4)):
        rgba = oracle.synth_frame(W, H, seed)
        ref = oracle.extract(rgba, depth=depth, threshold=THR)
        with _program(tinyorb, W, H, depth) as prog:
            total, corners, desc = prog.extract(rgba)
            _assert_frame_equal(oracle, ref, total, corners, desc)


def test_transport_records_round_trip(tinyorb, oracle):
    """orb_batch_pack_transport -> orb_unpack_transport (the multi-GPU collate's wire format: 40-byte records back to
    back) reproduces the per-frame reads, on both output sets, with a capacity cut (count > max_features), an empty
    frame, two runs placed out of order (as the records of two ranks would be) and a destination that is too small."""
    import torch
    W, H, B, CAP = 320, 240, 6, 256
    dev = torch.device("cuda", 0)
    with _program(tinyorb, W, H, 2, max_batch=B, max_features=CAP, flags=tinyorb.ORB_FLAG_DOUBLE_OUTPUT) as prog:
        for s_, seed in ((0, 500), (1, 530)):
            prog.batch_select_output(s_)
            frames = np.stack([oracle.synth_frame(W, H, seed + i) for i in range(B)])
            frames[2] = 90  # a flat frame: no keypoints
            frames[4] = frames[4] // 3 + 80  # low contrast: few keypoints (the others are around the capacity)
            prog.extract_batch_host(frames)
            prog.batch_sync()
        for s_ in (0, 1):
            prog.batch_select_output(s_)
            counts = prog.batch_counts(B)
            stored = np.minimum(counts, CAP).astype(np.int64)
            assert counts[2] == 0 and (counts > CAP).any() and (counts[counts > 0] <= CAP).any()
            total = int(stored.sum())
            rec = torch.full((total + 5, 10), -1, dtype=torch.int32, device=dev)
            offs = torch.zeros(B + 1, dtype=torch.int64, device=dev)
            prog.batch_pack_transport(s_, B, rec.data_ptr(), total + 5, offs.data_ptr())
            prog.stream_sync()
            torch.cuda.synchronize()
            first = np.concatenate([[0], np.cumsum(stored)])
            assert np.array_equal(offs.cpu().numpy(), first)
            assert (rec[total:] == -1).all()
            # two runs (frames 0..2 and 3..5), the second placed first in the destination
            cut = int(first[3])
            corners = torch.zeros((total, 4), dtype=torch.int32, device=dev)
            desc = torch.zeros((total, 8), dtype=torch.int32, device=dev)
            prog.unpack_transport(rec.data_ptr(), [0, cut], [cut, total - cut], [total - cut, 0], corners.data_ptr(), desc.data_ptr())
            prog.stream_sync()
            torch.cuda.synchronize()
            kc, kd = corners.cpu().numpy(), desc.cpu().numpy().view(np.uint32)
            for i in range(B):
                n = int(stored[i])
                at = int(first[i]) + (total - cut if i < 3 else -cut)
                c2, d2 = prog.batch_read(i, n)
                got = kc[at:at + n]
                for j, k in enumerate(("x", "y", "angle", "octave")):
                    assert np.array_equal(got[:, j].astype(np.uint32), c2[k]), (i, k)
                assert np.array_equal(kd[at:at + n], d2), i
            # a destination that holds fewer records than there are: filled to its end, the offsets still tell
            small = torch.full((max(cut, 1), 10), -1, dtype=torch.int32, device=dev)
            prog.batch_pack_transport(s_, B, small.data_ptr(), cut, offs.data_ptr())
            prog.stream_sync()
            torch.cuda.synchronize()
            assert torch.equal(small[:cut], rec[:cut]) and int(offs[B]) == total
        with pytest.raises(tinyorb.OrbError):
            prog.batch_pack_transport(0, B + 1, rec.data_ptr(), 1)
    with _program(tinyorb, 96, 64, 2, max_batch=2) as p2:  # no second output set: set 0 packs (the node API's case), set 1 does not exist
        frames = np.stack([oracle.synth_frame(96, 64, 610 + i) for i in range(2)])
        p2.extract_batch_host(frames)
        p2.batch_sync()
        counts = p2.batch_counts(2)
        total = int(counts.sum())
        rec = torch.zeros((max(total, 1), 10), dtype=torch.int32, device=dev)
        offs = torch.zeros(3, dtype=torch.int64, device=dev)
        p2.batch_pack_transport(0, 2, rec.data_ptr(), max(total, 1), offs.data_ptr())
        p2.stream_sync()
        torch.cuda.synchronize()
        assert offs.tolist() == [0, int(counts[0]), total]
        c0, _ = p2.batch_read(0, int(counts[0]))
        assert np.array_equal((rec[:int(counts[0]), 0].cpu().numpy() & 0xffff).astype(np.uint32), c0["x"])
        with pytest.raises(tinyorb.OrbError):
            p2.batch_pack_transport(1, 1, rec.data_ptr(), 1)


@pytest.mark.parametrize("ranks,F", [(2, 7), (3, 10), (4, 3)])
def test_node_api_several_ranks_on_one_device(tinyorb, oracle, monkeypatch, ranks, F):
    """The n > 1 data path of orb_node_* -- uneven shards (an empty one with 4 ranks and 3 frames), 40-byte transport
    records from ranks >= 1, exact offsets, expansion behind rank 0's own records -- on one GPU: with
    TINYORB_NODE_LOOPBACK=1 the exchange runs as device copies (RCCL refuses a device listed twice), everything else is
    the code an 8-GPU node runs.  Frame order and every record against the oracle."""
    monkeypatch.setenv("TINYORB_NODE_LOOPBACK", "1")
    W, H, CAP = 320, 240, 300  # some frames overflow the capacity: the stored records are then cap per frame
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=CAP, hierarchy_depth=2, initial_threshold=THR, max_batch=F)
    frames = np.stack([oracle.synth_frame(W, H, 900 + i) for i in range(F)])
    frames[1] = 77  # an empty frame inside rank 0's shard
    with tinyorb.OrbNode(cfg, [0] * ranks) as node:
        assert node.device_count() == ranks
        shards = [node.shard(F, r) for r in range(ranks)]
        assert shards[0][0] == 0 and shards[-1][1] == F and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
        for job in range(2):  # buffers are reused by a second job
            node.extract_batch_host(frames if job == 0 else frames[::-1].copy())
            counts, offsets, _, _ = node.collate(F)
            kp, desc = node.read_collated(int(offsets[F]))
            src = frames if job == 0 else frames[::-1]
            for i in range(F):
                ref = oracle.extract(src[i], depth=2, threshold=THR, max_features=1 << 16)
                assert int(counts[i]) == ref["total"]
                lo, hi = int(offsets[i]), int(offsets[i + 1])
                assert hi - lo == min(ref["total"], CAP)
                if ref["total"] <= CAP:
                    _assert_frame_equal(oracle, ref, int(counts[i]), kp[lo:hi], desc[lo:hi])
                else:  # which records survive the cut is not defined: each stored one must be a keypoint of the frame
                    table = {(int(k["octave"]), int(k["y"]), int(k["x"])): (int(k["angle"]), d.tobytes())
                             for k, d in zip(ref["corners"], ref["descriptors"])}
                    assert all(table.get((int(k["octave"]), int(k["y"]), int(k["x"]))) == (int(k["angle"]), d.tobytes())
                               for k, d in zip(kp[lo:hi], desc[lo:hi]))
