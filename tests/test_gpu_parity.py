"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer/index work (keypoint x, y, angle code, octave) and descriptor bits must be identical
(Hamming distance 0); intermediate binary16 planes are compared as bit patterns.  Output order
is unspecified in the reference (atomic append), so lists are compared after
sorting by (octave, y, x) (CRD-11).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR = 20.0 / 255.0


def _program(tinyorb, W, H, depth=2, max_features=8192, max_batch=1, flags=0, thr=THR):
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=max_features, hierarchy_depth=depth,
                            initial_threshold=thr, max_batch=max_batch, flags=flags)
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
    ham = np.unpackbits((d ^ rd).view(np.uint8), axis=1).sum(axis=1)
    assert int(ham.max(initial=0)) == 0, "descriptor Hamming distance %d" % int(ham.max())
    # compaction sanity: no duplicate (octave, x, y)
    keys = np.stack([c["octave"], c["y"], c["x"]], 1)
    assert len(np.unique(keys, axis=0)) == len(keys)


@pytest.mark.parametrize("W,H,depth,seed", [(64, 48, 2, 3), (160, 120, 3, 1), (200, 97, 3, 7), (640, 480, 2, 1),
                                            (1284, 250, 4, 5), (332, 202, 5, 11), (2048, 64, 2, 4), (2052, 40, 1, 6),
                                            (36, 40, 2, 8), (12, 10, 1, 9), (8, 8, 1, 10), (44, 36, 3, 12), (1920, 56, 2, 13),
                                            (1241, 376, 3, 14), (1226, 370, 2, 15), (333, 77, 3, 16), (1282, 96, 2, 17),
                                            (1280, 97, 2, 18), (2049, 40, 2, 19), (9, 9, 2, 20)])
@pytest.mark.parametrize("flags", [1, 0])
def test_single_frame_matches_oracle(tinyorb, oracle, W, H, depth, seed, flags):
    rgba = oracle.synth_frame(W, H, seed)
    ref = oracle.extract(rgba, depth=depth, threshold=THR, planes=True)
    with _program(tinyorb, W, H, depth, flags=flags) as prog:
        staged = bool(flags & tinyorb.ORB_FLAG_STAGED)
        fused_ok = 8 <= W <= 4096  # any width (rows 4-byte aligned) and any halving: the general level-0 variant
        assert prog.pipeline() == ("fused" if fused_ok and not staged else "staged")
        total, corners, desc = prog.extract(rgba)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            assert prog.level_size(m) == (w, h)
            if m > 0 or prog.pipeline() == "staged":  # the fused path keeps the level-0 grey plane in LDS only
                g = prog.read_plane(tinyorb.ORB_PLANE_GRAY, m)
                assert np.array_equal(g.ravel(), ref["gray"][off:off + w * h]), "gray level %d" % m
            b = prog.read_plane(tinyorb.ORB_PLANE_BLUR, m)
            assert np.array_equal(b.ravel(), ref["blur"][off:off + w * h]), "blur level %d" % m
        _assert_frame_equal(oracle, ref, total, corners, desc)


def test_config2_config3_720p(tinyorb, oracle):
    """BASELINE.json configs[1] and [2]: one 1280x720 frame, keypoints then descriptors."""
    rgba = oracle.synth_frame(1280, 720, 2)
    ref = oracle.extract(rgba, depth=2, threshold=THR)
    assert ref["total"] > 1000
    with _program(tinyorb, 1280, 720, 2) as prog:
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)


def test_device_scalar_math_matches_oracle(tinyorb, oracle):
    """CRD-3 (f32->f16 RNE, subnormals) and CRD-9 (canonical atan2 -> milliradian code) on the GPU."""
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2**32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    special = np.array([0x00000000, 0x80000000, 0x33000000, 0x33000001, 0x337fffff, 0x33800000, 0x387fc000,
                        0x38800000, 0x477fe000, 0x477ff000, 0x47800000, 0x7f800000, 0x3f800000, 0x3f801000,
                        0x3f803000, 0x38000000, 0x37ffffff, 0x00000001, 0x007fffff], dtype=np.uint32)
    unit = rng.random(1 << 20, dtype=np.float32)
    tiny = (rng.random(1 << 18, dtype=np.float32) * np.float32(2.0 ** -14)).astype(np.float32)
    src = np.concatenate([bits.view(np.float32), special.view(np.float32), unit, tiny])
    src = src[~np.isnan(src)]
    with _program(tinyorb, 64, 48) as prog:
        got = prog.device_f32_to_f16(src)
        want = src.astype(np.float16).view(np.uint16)
        assert np.array_equal(got, want)
        sample = src[:4096]
        assert [oracle.f32_to_f16(v) for v in sample] == got[:4096].tolist()

        n = 1 << 18
        cy = (rng.random(n, dtype=np.float32) * 40 - 8).astype(np.float32)
        cx = (rng.random(n, dtype=np.float32) * 40 - 20).astype(np.float32)
        cy[:64] = 0
        cx[:32] = 0
        cy[64:128] = np.abs(cx[64:128])
        got = prog.device_angle_code(cy, cx)
        from oracle import orb_numpy
        ang = orb_numpy.atan2f(cy, cx)
        want = np.where((cy < 0) | (ang < 0), np.float32(0), np.trunc(ang * np.float32(1000.0))).astype(np.uint32)
        assert np.array_equal(got, want)
        idx = rng.integers(0, n, 2048)
        assert [oracle.angle_code(cy[i], cx[i]) for i in idx] == got[idx].tolist()


def test_batch_equals_singles(tinyorb, oracle):
    """batch(N frames) == [single(frame)], frames generated on the device."""
    W, H, B = 320, 240, 6
    with _program(tinyorb, W, H, 2, max_batch=B) as prog:
        dev = prog.synth_frames_device(B, 100)
        frames = prog.copy_to_host(dev, B * W * H * 4).reshape(B, H, W, 4)
        for i in range(B):
            assert np.array_equal(frames[i], oracle.synth_frame(W, H, 100 + i)), "device generator frame %d" % i
        prog.extract_batch_device(dev, B)
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract(frames[i], depth=2, threshold=THR)
            corners, desc = prog.batch_read(i, int(counts[i]))
            _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)


def test_capacity_overflow_reports_raw_count(tinyorb, oracle):
    rgba = oracle.synth_frame(320, 240, 9)
    ref = oracle.extract(rgba, depth=2, threshold=THR)
    cap = ref["total"] // 2
    with _program(tinyorb, 320, 240, 2, max_features=cap) as prog:
        total, corners, desc = prog.extract(rgba)
        assert total == ref["total"] and len(corners) == cap
        # every stored record is a genuine detection with the right descriptor
        full = {(int(c["octave"]), int(c["y"]), int(c["x"])): (int(c["angle"]), d.tobytes())
                for c, d in zip(ref["corners"], ref["descriptors"])}
        for c, d in zip(corners, desc):
            assert full[(int(c["octave"]), int(c["y"]), int(c["x"]))] == (int(c["angle"]), d.tobytes())


# ---------------------------------------------------------------------------------------------
# committed golden fixtures, straight against the GPU (no oracle code runs in this test)
# ---------------------------------------------------------------------------------------------
import glob
import hashlib
import os

_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", _GOLDEN, ids=[os.path.basename(p) for p in _GOLDEN])
@pytest.mark.parametrize("flags", [0, 1])
def test_golden_fixture_on_gpu(tinyorb, path, flags):
    g = np.load(path)
    W, H, depth, seed, syn_flags, cap = (int(v) for v in g["params"])
    with _program(tinyorb, W, H, depth, max_features=cap, flags=flags, thr=float(g["threshold"])) as prog:
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


# ---------------------------------------------------------------------------------------------
# BASELINE.json full size (256 x 1280x720): size-independent properties + spot checks
# ---------------------------------------------------------------------------------------------
def test_full_size_batch_properties(tinyorb, oracle):
    W, H, B, cap = 1280, 720, 256, 8192
    with _program(tinyorb, W, H, 2, max_features=cap, max_batch=B) as fused, \
            _program(tinyorb, W, H, 2, max_features=cap, max_batch=8, flags=tinyorb.ORB_FLAG_STAGED) as staged:
        assert fused.pipeline() == "fused" and staged.pipeline() == "staged"
        dev = fused.synth_frames_device(B // 2, 1000)
        # second half of the batch = the same 128 frames again: duplicates must give identical results
        fused.synth_frames_device(B // 2, 1000, frames_dev_ptr=dev + (B // 2) * W * H * 4)
        fused.extract_batch_device(dev, B)
        counts = fused.batch_counts(B)
        assert np.array_equal(counts[:B // 2], counts[B // 2:])
        assert counts.min() > 1000 and counts.max() <= cap
        p = None
        for i in (0, 1, 77, 127):
            ca, da = _sorted(*fused.batch_read(i, int(counts[i])))
            cb, db = _sorted(*fused.batch_read(i + B // 2, int(counts[i])))
            assert np.array_equal(ca, cb) and np.array_equal(da, db)
            # structural invariants of the reference's detector (fast.wgsl:77, orb.rs:509-519)
            assert np.all(ca["octave"] < 2)
            assert np.all((ca["x"] > 16) & (ca["y"] > 16) & (ca["x"] < W - 16) & (ca["y"] < H - 16))
            lvl1 = ca["octave"] == 1
            assert np.all(ca["x"][lvl1] < 640) and np.all(ca["y"][lvl1] < 360)
            assert np.all(ca["angle"] <= 3141)
            keys = np.stack([ca["octave"], ca["y"], ca["x"]], 1)
            assert len(np.unique(keys, axis=0)) == len(keys)
            # idempotence: the staged pipeline on the same device-resident frame gives the same answer
            if i < 2:
                staged.extract_batch_device(dev + i * W * H * 4, 1)
                n = int(staged.batch_counts(1)[0])
                cs, ds = _sorted(*staged.batch_read(0, n))
                assert n == counts[i] and np.array_equal(cs, ca) and np.array_equal(ds, da)
        # a second run over the same input reproduces every count (no state leaks between batches)
        fused.extract_batch_device(dev, B)
        assert np.array_equal(fused.batch_counts(B), counts)
        # EVERY distinct frame of the batch against the oracle, record by record (round 5; two spot frames before): the restatement runs
        # frame-parallel over the host's threads on the frames copied back from the device
        half = B // 2
        frames = fused.copy_to_host(dev, half * W * H * 4).reshape(half, H, W, 4)
        totals, rc, rd = oracle.extract_batch(frames, depth=2, threshold=THR, max_features=cap, n_threads=min(16, os.cpu_count() or 1))
        assert np.array_equal(totals, counts[:half])
        hb = fused.batch_read_all(B)
        off = np.concatenate([[0], np.cumsum(np.minimum(counts, cap).astype(np.int64))])
        for i in range(half):
            ref = dict(total=int(totals[i]), corners=rc[i, :min(int(totals[i]), cap)], descriptors=rd[i, :min(int(totals[i]), cap)])
            lo, hi = int(off[i]), int(off[i + 1])
            _assert_frame_equal(oracle, ref, int(counts[i]), hb.corners[lo:hi], hb.descriptors[lo:hi].reshape(-1, 8))
        hb.close()


def test_threshold_and_reuse(tinyorb, oracle):
    """set_threshold (orb.rs:585) takes effect on the next extract; a program is reusable across frames."""
    W, H = 320, 240
    a, b = oracle.synth_frame(W, H, 31), oracle.synth_frame(W, H, 32)
    with _program(tinyorb, W, H, 2) as prog:
        for rgba, thr in [(a, THR), (b, THR), (a, 0.04), (a, 0.2), (b, 0.0)]:
            prog.set_threshold(thr)
            total, corners, desc = prog.extract(rgba)
            ref = oracle.extract(rgba, depth=2, threshold=thr, max_features=8192)
            assert total == ref["total"]
            if total <= 8192:
                _assert_frame_equal(oracle, ref, total, corners, desc)


def test_pathological_dense_frame(tinyorb, oracle):
    """Salt-and-pepper noise: far more pre-test survivors than the LDS queue holds, and more corners
    than max_features -> exercises the queue-overflow path and the capacity clamp."""
    rng = np.random.default_rng(12)
    W, H = 256, 128
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    rgba[..., :3] = (rng.random((H, W, 1)) < 0.08) * 255
    rgba[..., 3] = 255
    ref = oracle.extract(rgba, depth=2, threshold=THR, max_features=1 << 16)
    assert ref["total"] > 1500
    with _program(tinyorb, W, H, 2, max_features=1 << 16) as prog:
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
    with _program(tinyorb, W, H, 2, max_features=1000) as prog:
        total, corners, desc = prog.extract(rgba)
        assert total == ref["total"] and len(corners) == 1000
        full = {(int(c["octave"]), int(c["y"]), int(c["x"])): (int(c["angle"]), d.tobytes())
                for c, d in zip(ref["corners"], ref["descriptors"])}
        seen = set()
        for c, d in zip(corners, desc):
            key = (int(c["octave"]), int(c["y"]), int(c["x"]))
            assert full[key] == (int(c["angle"]), d.tobytes()) and key not in seen
            seen.add(key)


def test_host_batch_and_single_frame_api_agree(tinyorb, oracle):
    W, H, B = 320, 240, 4
    frames = np.stack([oracle.synth_frame(W, H, 60 + i) for i in range(B)])
    with _program(tinyorb, W, H, 2, max_batch=B) as prog:
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(B)
        batch = [_sorted(*prog.batch_read(i, int(counts[i]))) for i in range(B)]
        for i in range(B):
            total, corners, desc = prog.extract(frames[i])
            c, d = _sorted(corners, desc)
            assert total == counts[i] and np.array_equal(c, batch[i][0]) and np.array_equal(d, batch[i][1])


def test_double_output_sets(tinyorb, oracle):
    """ORB_FLAG_DOUBLE_OUTPUT: two independent output sets (batch k+1 computes while batch k is collated)."""
    W, H, B = 320, 240, 3
    with _program(tinyorb, W, H, 2, max_batch=B, flags=tinyorb.ORB_FLAG_DOUBLE_OUTPUT) as prog:
        dev_a = prog.synth_frames_device(B, 500)
        frames_a = prog.copy_to_host(dev_a, B * W * H * 4).reshape(B, H, W, 4).copy()
        prog.batch_select_output(0)
        prog.extract_batch_device(dev_a, B)
        prog.batch_sync()
        prog.synth_frames_device(B, 600)  # same slab, new frames
        frames_b = prog.copy_to_host(dev_a, B * W * H * 4).reshape(B, H, W, 4).copy()
        prog.batch_select_output(1)
        prog.extract_batch_device(dev_a, B)
        for slot, frames in ((0, frames_a), (1, frames_b)):
            prog.batch_select_output(slot)
            counts = prog.batch_counts(B)
            for i in range(B):
                ref = oracle.extract(frames[i], depth=2, threshold=THR)
                corners, desc = prog.batch_read(i, int(counts[i]))
                _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
    with _program(tinyorb, W, H, 2) as prog:
        with pytest.raises(tinyorb.OrbError):
            prog.batch_select_output(1)


# ---------------------------------------------------------------------------------------------
# opt-in extensions: FAST arc length and 3x3 NMS.  Not in the reference;
# the definitions are the build's own (oracle/orb_oracle.h) -- GPU vs C oracle, bit for bit.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arc,nms", [(9, False), (9, True), (12, True), (10, False), (16, True), (12, False)])
@pytest.mark.parametrize("W,H,depth", [(320, 240, 2), (332, 202, 4), (1284, 96, 3), (642, 120, 2)])
@pytest.mark.parametrize("staged", [False, True])
def test_arc_length_and_nms_extensions(tinyorb, oracle, arc, nms, W, H, depth, staged):
    rgba = oracle.synth_frame(W, H, 77)
    ref = oracle.extract_ex(rgba, depth=depth, threshold=THR, max_features=1 << 15, arc=arc, nms=nms)
    planes = oracle.extract(rgba, depth=depth, threshold=THR, planes=True)  # grey and blur do not depend on the detector
    flags = (tinyorb.ORB_FLAG_NMS if nms else 0) | (tinyorb.ORB_FLAG_STAGED if staged else 0)
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=1 << 15, hierarchy_depth=depth, initial_threshold=THR,
                            flags=flags, fast_arc=arc)
    with tinyorb.OrbProgram(cfg) as prog:
        plain = arc == 12 and not nms  # the reference's own detector: band kernels, any width; else the tile kernels (RGBA quads)
        fused_ok = plain or (W % 4 == 0 and W % 2 == 0 and H % 2 == 0)
        assert prog.pipeline() == ("fused" if fused_ok and not staged else "staged")
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_BLUR, m).ravel(), planes["blur"][off:off + w * h])
            plain_fused = arc == 12 and not nms and prog.pipeline() == "fused"  # keeps the level-0 grey plane in LDS only
            if m > 0 or not plain_fused:
                assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_GRAY, m).ravel(), planes["gray"][off:off + w * h])
    if arc == 9 and not nms and (W, H) == (320, 240) and not staged:
        assert ref["total"] > oracle.extract(rgba, depth=depth, threshold=THR)["total"]  # FAST-9 finds more than FAST-12


def test_extensions_dense_frame_on_tiles(tinyorb, oracle):
    """Bright dots everywhere: the tile queues of the fused detector overflow, its direct path must agree."""
    W, H = 640, 128
    rng = np.random.default_rng(6)
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    rgba[..., :3] = (rng.random((H, W, 1)) < 0.2) * 255
    rgba[..., 3] = 255
    for nms in (False, True):
        ref = oracle.extract_ex(rgba, depth=2, threshold=THR, max_features=1 << 16, arc=9, nms=nms)
        assert ref["total"] > 4000
        cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=1 << 16, hierarchy_depth=2, initial_threshold=THR,
                                flags=tinyorb.ORB_FLAG_NMS if nms else 0, fast_arc=9)
        with tinyorb.OrbProgram(cfg) as prog:
            assert prog.pipeline() == "fused"
            total, corners, desc = prog.extract(rgba)
            _assert_frame_equal(oracle, ref, total, corners, desc)


def test_bad_arc_is_rejected(tinyorb):
    with pytest.raises(tinyorb.OrbError):
        tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(64, 48), fast_arc=8)).init()


# ---------------------------------------------------------------------------------------------
# "intended" mode (ORB_FLAG_INTENDED; definitions IM-1..IM-8 in oracle/orb_oracle.h).
# Not in the reference -- GPU vs the build's C oracle, bit for bit, planes included.
# ---------------------------------------------------------------------------------------------
def _gray_plane(tinyorb, prog, m):
    """Grey plane of level m, or None where the fused intended pipeline keeps it in LDS only (level 0, when every level is an
    exact half: k_front_i blurs its own tile, nothing reads the plane).  Any other refusal is an error."""
    try:
        return prog.read_plane(tinyorb.ORB_PLANE_GRAY, m)
    except tinyorb.OrbError:
        assert m == 0 and prog.pipeline() == "fused"
        return None


def _intended_program(tinyorb, W, H, depth, cap, arc, nms, max_batch=1, staged=False):
    flags = tinyorb.ORB_FLAG_INTENDED | (tinyorb.ORB_FLAG_NMS if nms else 0) | (tinyorb.ORB_FLAG_STAGED if staged else 0)
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=depth, initial_threshold=THR,
                            max_batch=max_batch, flags=flags, fast_arc=arc)
    return tinyorb.OrbProgram(cfg).init()


@pytest.mark.parametrize("W,H,depth,arc,nms,cap", [(160, 120, 2, 0, False, 8192), (320, 240, 3, 9, True, 8192),
                                                   (200, 97, 3, 12, False, 8192), (332, 202, 4, 10, True, 100),
                                                   (640, 480, 2, 9, False, 777), (36, 40, 2, 9, False, 64),
                                                   (70, 34, 1, 9, True, 64), (1280, 720, 2, 9, True, 8192),
                                                   (1284, 250, 4, 9, True, 8192), (648, 100, 3, 16, True, 8192),
                                                   (322, 202, 2, 9, False, 8192), (1920, 120, 2, 11, True, 50),
                                                   (320, 240, 2, 9, False, 8), (640, 360, 1, 9, True, 3)])
@pytest.mark.parametrize("staged", [False, True])
def test_intended_mode_matches_oracle(tinyorb, oracle, W, H, depth, arc, nms, cap, staged):
    rgba = oracle.synth_frame(W, H, 90 + depth)
    ref = oracle.extract_intended(rgba, depth=depth, threshold=THR, max_features=cap, arc=arc or 9, nms=nms, planes=True)
    with _intended_program(tinyorb, W, H, depth, cap, arc, nms, staged=staged) as prog:
        assert prog.pipeline() == ("fused" if W % 4 == 0 and not staged else "staged")
        total, corners, desc = prog.extract(rgba)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            g = _gray_plane(tinyorb, prog, m)
            assert g is None or np.array_equal(g.ravel(), ref["gray"][off:off + w * h])
            assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_BLUR, m).ravel(), ref["blur"][off:off + w * h])
        assert total == ref["total"]
        n = min(total, cap)
        c, d = _sorted(corners[:n], desc[:n])
        rc, rd = oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        for k in ("octave", "y", "x", "angle"):
            assert np.array_equal(c[k], rc[k]), k
        assert np.array_equal(d, rd)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,depth,nms", [(320, 240, 3, True), (1284, 250, 4, False), (36, 40, 2, False)])
def test_intended_blur_kernel_of_its_own(tinyorb, oracle, monkeypatch, W, H, depth, nms):
    """TINYORB_I_GAUSS_KERNEL=1: the fused intended pipeline with k_gauss over a stored grey plane instead of k_front_i's
    phase G (the round-2 form, kept for A/B measurements): same planes, same keypoints."""
    monkeypatch.setenv("TINYORB_I_GAUSS_KERNEL", "1")
    rgba = oracle.synth_frame(W, H, 77)
    ref = oracle.extract_intended(rgba, depth=depth, threshold=THR, max_features=8192, arc=9, nms=nms, planes=True)
    with _intended_program(tinyorb, W, H, depth, 8192, 9, nms) as prog:
        assert prog.pipeline() == "fused"
        total, corners, desc = prog.extract(rgba)
        dims, _ = oracle.level_dims(W, H, depth)
        for m, (w, h, off) in enumerate(dims):
            assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_GRAY, m).ravel(), ref["gray"][off:off + w * h])
            assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_BLUR, m).ravel(), ref["blur"][off:off + w * h])
        _assert_frame_equal(oracle, ref, total, corners, desc)


_INTENDED = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "intended", "*.npz")))


@pytest.mark.parametrize("path", _INTENDED, ids=[os.path.basename(p) for p in _INTENDED])
@pytest.mark.parametrize("staged", [False, True])
def test_intended_golden_fixture_on_gpu(tinyorb, path, staged):
    g = np.load(path)
    W, H, depth, seed, syn_flags, cap, arc, nms = (int(v) for v in g["params"])
    with _intended_program(tinyorb, W, H, depth, cap, arc, bool(nms), staged=staged) as prog:
        dev = prog.synth_frames_device(1, seed, syn_flags)
        prog.extract_batch_device(dev, 1)
        total = int(prog.batch_counts(1)[0])
        assert total == int(g["total"])
        corners, desc = prog.batch_read(0, min(total, cap))
        c, d = _sorted(corners, desc)
        assert np.array_equal(np.stack([c[k] for k in ("x", "y", "angle", "octave")], 1), g["corners"])
        assert np.array_equal(d, g["descriptors"])
        for m in range(depth):
            gp = _gray_plane(tinyorb, prog, m)
            assert gp is None or hashlib.sha256(gp.tobytes()).hexdigest() == str(g["gray_sha256"][m])
            assert hashlib.sha256(prog.read_plane(tinyorb.ORB_PLANE_BLUR, m).tobytes()).hexdigest() == str(g["blur_sha256"][m])


def test_intended_batch_with_top_k(tinyorb, oracle):
    """Batched call: every frame gets its own cut; frames below the capacity are untouched."""
    W, H, cap = 320, 240, 300
    frames = np.stack([oracle.synth_frame(W, H, 200 + i, flags) for i, flags in enumerate((15, 15, 7, 1, 15))])
    with _intended_program(tinyorb, W, H, 2, cap, 9, True, max_batch=5) as prog:
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(5)
        for i in range(5):
            ref = oracle.extract_intended(frames[i], depth=2, threshold=THR, max_features=cap, arc=9, nms=True)
            assert int(counts[i]) == ref["total"]
            n = min(ref["total"], cap)
            c, d = _sorted(*prog.batch_read(i, n))
            rc, rd = oracle.sort_keypoints(ref["corners"], ref["descriptors"])
            assert np.array_equal(np.stack([c[k] for k in ("x", "y", "angle", "octave")], 1),
                                  np.stack([rc[k] for k in ("x", "y", "angle", "octave")], 1))
            assert np.array_equal(d, rd)
        assert int(counts.max()) > cap and int(counts.min()) < cap


def test_intended_pathological_dense_tile(tinyorb, oracle):
    """A frame dense enough to overflow the tile queues of the fused kernel: the direct path must give the same list."""
    W, H = 640, 128
    rng = np.random.default_rng(5)
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    rgba[..., :3] = (rng.random((H, W, 1)) < 0.2) * 255  # bright dots: a tenth of all pixels are FAST-9 corners
    rgba[..., 3] = 255
    for nms in (False, True):
        ref = oracle.extract_intended(rgba, depth=2, threshold=THR, max_features=1 << 16, arc=9, nms=nms)
        assert ref["total"] > 4000
        with _intended_program(tinyorb, W, H, 2, 1 << 16, 9, nms) as prog:
            assert prog.pipeline() == "fused"
            total, corners, desc = prog.extract(rgba)
            assert total == ref["total"]
            c, d = _sorted(corners, desc)
            rc, rd = oracle.sort_keypoints(ref["corners"], ref["descriptors"])
            for k in ("octave", "y", "x", "angle"):
                assert np.array_equal(c[k], rc[k]), k
            assert np.array_equal(d, rd)


# ---------------------------------------------------------------------------------------------
# Hamming matcher between consecutive frames (not in the reference): GPU vs NumPy brute force
# ---------------------------------------------------------------------------------------------
_MATCH_FORMS = {"fp4": {}, "i8": {"TINYORB_MATCH_I8": "1"}, "valu": {"TINYORB_MATCH_VALU": "1"}}


def _match_form(monkeypatch, form):
    """fp4: the default (k_desc_expand4 + k_match_fp4); i8: the int8 matrix-core form (k_desc_expand + k_match_mfma); valu: k_match."""
    for k in ("TINYORB_MATCH_I8", "TINYORB_MATCH_VALU"):
        monkeypatch.delenv(k, raising=False)
    for k, v in _MATCH_FORMS[form].items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("form", list(_MATCH_FORMS))
def test_match_consecutive_frames(tinyorb, oracle, monkeypatch, form):
    from oracle import orb_numpy
    _match_form(monkeypatch, form)
    W, H, cap = 320, 240, 600
    base = oracle.synth_frame(W + 8, H + 6, 300)
    frames = np.stack([np.ascontiguousarray(base[dy:dy + H, dx:dx + W]) for dx, dy in ((0, 0), (3, 2), (8, 6))]
                      + [np.zeros((H, W, 4), np.uint8)])  # three shifted views of one scene, then an empty frame
    with _program(tinyorb, W, H, 2, max_features=cap, max_batch=4) as prog:
        prog.extract_batch_host(frames)
        counts = np.minimum(prog.batch_counts(4), cap)
        assert counts[0] > 200 and counts[3] == 0
        desc = [prog.batch_read(f, int(counts[f]))[1] for f in range(4)]
        prog.match_consecutive(4)
        for f in range(3):
            got = prog.match_read(f, int(counts[f]))
            idx, dist, second = orb_numpy.match(desc[f], desc[f + 1])
            assert np.array_equal(got["index"], idx) and np.array_equal(got["distance"], dist)
            assert np.array_equal(got["second"], second)
        m01 = prog.match_read(0, int(counts[0]))
        assert float(np.mean(m01["distance"] < 40)) > 0.1  # a shifted scene: a good share of close matches
        assert (prog.match_read(2, int(counts[2]))["index"] == tinyorb.ORB_MATCH_NONE).all()
        with pytest.raises(tinyorb.OrbError):
            prog.match_consecutive(5)


@pytest.mark.parametrize("form", list(_MATCH_FORMS))
def test_match_frames_of_different_sizes(tinyorb, oracle, monkeypatch, form):
    """Counts that are no multiple of a tile (16 candidates, 64 queries per wave, 256 per workgroup), more queries than one
    workgroup takes, a frame with a single keypoint (no runner-up) and a capacity cut: every record against the NumPy brute force."""
    from oracle import orb_numpy
    _match_form(monkeypatch, form)
    W, H, cap = 640, 480, 1200
    frames = np.stack([oracle.synth_frame(W, H, 41, 7), oracle.synth_frame(W, H, 41, 3), oracle.synth_frame(W, H, 41, 5),
                       np.zeros((H, W, 4), np.uint8), oracle.synth_frame(W, H, 44)])
    frames[3, 200:203, 300:303] = 255  # one 3x3 blob: a handful of keypoints at most
    with _program(tinyorb, W, H, 2, max_features=cap, max_batch=5) as prog:
        prog.extract_batch_host(frames)
        counts = np.minimum(prog.batch_counts(5), cap)
        assert counts[0] == cap and 256 < counts[1] < cap and counts[1] % 16 and 0 < counts[3] < 16, counts  # a cut frame, a ragged one, a tiny one
        desc = [prog.batch_read(f, int(counts[f]))[1] for f in range(5)]
        prog.match_consecutive(5)
        for f in range(4):
            got = prog.match_read(f, int(counts[f]))
            idx, dist, second = orb_numpy.match(desc[f], desc[f + 1])
            assert np.array_equal(got["index"], idx), f
            assert np.array_equal(got["distance"], dist) and np.array_equal(got["second"], second), f


def test_large_frames_take_the_fused_pipelines(tinyorb, oracle):
    """Frames beyond 2^22 pixels (4 K): both fused pipelines, against the oracle."""
    W, H = 2048, 2200
    rgba = oracle.synth_frame(W, H, 31)
    ref = oracle.extract(rgba, depth=2, threshold=THR, max_features=1 << 16)
    with _program(tinyorb, W, H, 2, max_features=1 << 16) as prog:
        assert prog.pipeline() == "fused"
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
    W, H = 3840, 2160  # literal mode at 4K: the 8-row-band kernels
    rgba = oracle.synth_frame(W, H, 33)
    ref = oracle.extract(rgba, depth=3, threshold=THR, max_features=1 << 17)
    with _program(tinyorb, W, H, 3, max_features=1 << 17) as prog:
        assert prog.pipeline() == "fused"
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)
    rgba = oracle.synth_frame(W, H, 32)
    ref = oracle.extract_intended(rgba, depth=3, threshold=THR, max_features=1 << 17, arc=9, nms=True)
    with _intended_program(tinyorb, W, H, 3, 1 << 17, 9, True) as prog:
        assert prog.pipeline() == "fused"
        total, corners, desc = prog.extract(rgba)
        _assert_frame_equal(oracle, ref, total, corners, desc)


def test_intended_full_size_batch(tinyorb, oracle):
    """BASELINE.json's frame size in the intended mode: a 64-frame batch on both output sets, fused against staged on
    every frame's counter, the oracle on a few frames, matches between consecutive frames, a second run."""
    W, H, B, cap = 1280, 720, 64, 8192
    flags = tinyorb.ORB_FLAG_INTENDED | tinyorb.ORB_FLAG_NMS
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=2, initial_threshold=THR,
                            max_batch=B, flags=flags | tinyorb.ORB_FLAG_DOUBLE_OUTPUT, fast_arc=9)
    cfg_s = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), max_features=cap, hierarchy_depth=2, initial_threshold=THR,
                              max_batch=8, flags=flags | tinyorb.ORB_FLAG_STAGED, fast_arc=9)
    with tinyorb.OrbProgram(cfg) as fused, tinyorb.OrbProgram(cfg_s) as staged:
        assert fused.pipeline() == "fused" and staged.pipeline() == "staged"
        dev = fused.synth_frames_device(B, 5000)
        fused.extract_batch_device(dev, B)
        counts = fused.batch_counts(B)
        assert counts.min() > 3000 and counts.max() <= cap
        for i0 in range(0, B, 8):
            staged.extract_batch_device(dev + i0 * W * H * 4, 8)
            assert np.array_equal(staged.batch_counts(8), counts[i0:i0 + 8])
        for i in (0, 37, 63):
            frame = fused.copy_to_host(dev + i * W * H * 4, W * H * 4).reshape(H, W, 4)
            ref = oracle.extract_intended(frame, depth=2, threshold=THR, max_features=cap, arc=9, nms=True)
            corners, desc = fused.batch_read(i, int(counts[i]))
            _assert_frame_equal(oracle, ref, int(counts[i]), corners, desc)
            assert int(corners["angle"].max()) > 3141 and int(corners["angle"].max()) <= 6283
        fused.match_consecutive(B)
        m = fused.match_read(10, int(counts[10]))
        assert (m["index"] < counts[11]).all() and (m["distance"] <= m["second"]).all()
        # the other output set, then the first again: same counters, nothing leaks between batches or sets
        fused.batch_select_output(1)
        fused.extract_batch_device(dev, B)
        assert np.array_equal(fused.batch_counts(B), counts)
        c1, d1 = _sorted(*fused.batch_read(37, int(counts[37])))
        fused.batch_select_output(0)
        c0, d0 = _sorted(*fused.batch_read(37, int(counts[37])))
        assert np.array_equal(c0, c1) and np.array_equal(d0, d1)


def test_match_capacity_beyond_the_matrix_core_key(tinyorb, oracle):
    """max_features above 16 383 (the index range of the matrix-core matchers' key): orb_match_consecutive takes the vector-unit
    kernel by itself; at 16 383 it is still the fp4 form.  Same records either way."""
    from oracle import orb_numpy
    W, H = 320, 240
    frames = np.stack([oracle.synth_frame(W, H, 71), oracle.synth_frame(W, H, 72)])
    for cap in (16383, 16500):
        with _program(tinyorb, W, H, 2, max_features=cap, max_batch=2) as prog:
            prog.extract_batch_host(frames)
            counts = np.minimum(prog.batch_counts(2), cap)
            desc = [prog.batch_read(f, int(counts[f]))[1] for f in range(2)]
            prog.match_consecutive(2)
            got = prog.match_read(0, int(counts[0]))
            idx, dist, second = orb_numpy.match(desc[0], desc[1])
            assert np.array_equal(got["index"], idx) and np.array_equal(got["distance"], dist) and np.array_equal(got["second"], second)
