"""GPU parity, round 5: the arithmetic a shader compiler may choose (CRD-13, OrbOptions::fp_contract -- per-stage contraction
into fused multiply-adds, reduction order of dot() and matrix * vector) carried by the FUSED kernels as well as the per-stage
ones, against the restatement under the same setting.  Bit for bit, as everywhere."""
import os

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


def _oracle_kw(fp):
    return dict(contract=fp & 7, dot_order=(fp >> 3) & 1)


# luminance / blur / rotation alone, all three, last-term-first alone (what Mesa gives hardware without an fma), and with them
FP_FORMS = [1, 2, 4, 7, 8, 9, 12, 15]
# 640x480 seed 1: a frame on which the default and the contracted arithmetic differ in an angle code; 333x211: the general level-0
# variant (rows not quad-aligned, level 1 by the bilinear blit); 2304x40 / 2306x40: column tiles, aligned and general
SHAPES = [(640, 480, 2, 1), (320, 240, 3, 7), (333, 211, 2, 13), (1280, 720, 2, 2), (2304, 40, 2, 5), (2306, 40, 2, 6)]


@pytest.mark.parametrize("W,H,depth,seed", SHAPES)
@pytest.mark.parametrize("fp", FP_FORMS)
@pytest.mark.parametrize("flags", [0, 1])
def test_fp_forms_match_the_oracle_fused_and_staged(tinyorb, oracle, W, H, depth, seed, fp, flags):
    """Every form of OrbOptions::fp_contract on the fused kernels (flags 0) and on the per-stage ones (ORB_FLAG_STAGED) equals the
    restatement under the same setting: planes, counter, keypoints, angle codes, descriptors -- through the six calls of a
    one-frame program (k_front_pair / k_brief_one where the shape allows) and through a batch (k_front per level, k_brief_t,
    k_brief_nf)."""
    frame = oracle.synth_frame(W, H, seed, 15)
    ref = oracle.extract(frame, depth=depth, threshold=THR, planes=True, **_oracle_kw(fp))
    staged = bool(flags & tinyorb.ORB_FLAG_STAGED)
    for max_batch in (1, 3):
        cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=depth, initial_threshold=THR, fp_contract=fp, max_batch=max_batch,
                                flags=flags)
        with tinyorb.OrbProgram(cfg).init() as prog:
            assert prog.pipeline() == ("staged" if staged else "fused") and prog.pipeline_note() == ""
            if max_batch == 1:
                prog.write_input_image(frame)
                total = prog.extract_corners()
                n = min(total, 8192)
                _assert_frame_equal(oracle, ref, total, prog.read_corners(np.zeros(n, dtype=tinyorb.CORNER_DTYPE)),
                                    prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32)))
            else:
                other = oracle.synth_frame(W, H, seed + 100, 15)
                prog.extract_batch_host(np.stack([frame, other, frame]))
                counts = prog.batch_counts(3)
                ref2 = oracle.extract(other, depth=depth, threshold=THR, **_oracle_kw(fp))
                for i, r in enumerate((ref, ref2, ref)):
                    c, d = prog.batch_read(i, min(int(counts[i]), 8192))
                    _assert_frame_equal(oracle, r, int(counts[i]), c, d)
                prog.extract_batch_host(np.stack([frame]))  # planes of frame 0
            dims, _ = oracle.level_dims(W, H, depth)
            for m, (w, h, off) in enumerate(dims):
                if m > 0 or staged:  # the fused path keeps the level-0 grey plane in LDS only
                    assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_GRAY, m).ravel(), ref["gray"][off:off + w * h]), "gray level %d" % m
                assert np.array_equal(prog.read_plane(tinyorb.ORB_PLANE_BLUR, m).ravel(), ref["blur"][off:off + w * h]), "blur level %d" % m


def test_the_forms_are_different_functions(oracle):
    """The test above could not pass on the wrong form: on its frames the restatement's planes differ between the default and each
    stage's contraction, and between the two reduction orders (the f16 store hides most, not all, of the binary32 differences); on
    (640x480, seed 1) an angle code differs as well."""
    frame = oracle.synth_frame(1280, 720, 2, 15)
    base = oracle.extract(frame, depth=2, threshold=THR, planes=True)
    for fp, plane in ((1, "gray"), (8, "gray"), (9, "gray"), (2, "blur")):
        r = oracle.extract(frame, depth=2, threshold=THR, planes=True, **_oracle_kw(fp))
        assert np.any(r[plane] != base[plane]), fp
    a = oracle.extract(frame, depth=2, threshold=THR, planes=True, **_oracle_kw(1))
    b = oracle.extract(frame, depth=2, threshold=THR, planes=True, **_oracle_kw(9))
    assert np.any(a["gray"] != b["gray"])
    f2 = oracle.synth_frame(640, 480, 1, 15)
    c0, _ = oracle.sort_keypoints(*[oracle.extract(f2, depth=2, threshold=THR)[k] for k in ("corners", "descriptors")])
    c1, _ = oracle.sort_keypoints(*[oracle.extract(f2, depth=2, threshold=THR, contract=7)[k] for k in ("corners", "descriptors")])
    assert not np.array_equal(c0["angle"], c1["angle"])
    # the rotation: over all 3142 codes and every pattern point the three forms truncate differently in ONE place -- code 2214, where
    # (cos, sin) rounds to (-0.6, 0.8) and the points (-3, 4), (6, -8), (8, 6) rotate onto integers -- a 3-4-5 triangle
    diff = set()
    for (x, y) in ((-3, 4), (6, -8), (8, 6), (8, -3)):
        r = [tuple(int(v) for v in oracle.brief_rotate(2214, x, y, ct, do)) for ct, do in ((0, 0), (4, 0), (4, 1))]
        if len(set(r)) > 1:
            diff.add((x, y))
    assert diff == {(-3, 4), (6, -8), (8, 6)}


@pytest.mark.parametrize("fp", [4, 12])
def test_rotated_pattern_table_follows_the_rotation_form(tinyorb, fp):
    """k_rot_table under OrbOptions::fp_contract (k_brief_nf's and the odd-width gathers' source of rotated points): every entry against
    the NumPy restatement's rotate() in the same form -- all 3142 codes x 256 tests x 2 points, code 2214 (where the forms part)
    among them."""
    from oracle import orb_numpy as on
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), max_features=1024, hierarchy_depth=2, initial_threshold=THR, fp_contract=fp)
    with tinyorb.OrbProgram(cfg) as prog:
        table, pitch = prog.rot_table()
    with tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), max_features=1024, hierarchy_depth=2, initial_threshold=THR)) as prog:
        plain, _ = prog.rot_table()
    assert table.shape[0] == 3142 and pitch == 48
    F = np.float32
    theta = np.arange(3142, dtype=np.float32) / F(1000.0)
    ct = np.cos(theta.astype(np.float64)).astype(np.float32)
    st = np.sin(theta.astype(np.float64)).astype(np.float32)
    for j in range(256):
        ax, ay, bx, by = (F(v) for v in on.PATTERN[j])
        rax, ray = on.rotate(ct, st, ax, ay, 1, (fp >> 3) & 1)
        rbx, rby = on.rotate(ct, st, bx, by, 1, (fp >> 3) & 1)
        oa = 2 * (np.trunc(ray).astype(np.int64) * pitch + np.trunc(rax).astype(np.int64))
        ob = 2 * (np.trunc(rby).astype(np.int64) * pitch + np.trunc(rbx).astype(np.int64))
        assert np.array_equal(table[:, j & 63, j >> 6, 0].astype(np.int64), oa), "test %d point a" % j
        assert np.array_equal(table[:, j & 63, j >> 6, 1].astype(np.int64), ob), "test %d point b" % j
    where = np.argwhere(table != plain)
    assert len(where) > 0 and set(where[:, 0].tolist()) == {2214}  # the one code at which a fused product changes a truncation


@pytest.mark.parametrize("oob,wbits,fp", [("clamp", 8, 15), ("umin", 0, 7), ("umin", 8, 12), ("zero", 8, 3)])
@pytest.mark.parametrize("flags", [0, 1])
def test_fp_forms_with_the_other_switches(tinyorb, oracle, oob, wbits, fp, flags):
    """fp_contract together with an out-of-level policy and a sampler weight precision (the OOBK instances of k_front, k_brief_nf<OOB>,
    the rotated-pattern table): three levels of a frame with an odd level, fused and staged."""
    W, H, depth = 322, 242, 3
    frame = oracle.synth_frame(W, H, 21, 15)
    ref = oracle.extract(frame, depth=depth, threshold=THR, planes=True, oob=oob, weight_bits=wbits, **_oracle_kw(fp))
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=depth, initial_threshold=THR, fp_contract=fp, max_batch=2, flags=flags,
                            oob_policy=tinyorb.OOB_POLICIES[oob], sampler_weight_bits=wbits)
    with tinyorb.OrbProgram(cfg).init() as prog:
        prog.extract_batch_host(np.stack([frame, frame]))
        counts = prog.batch_counts(2)
        for i in range(2):
            c, d = prog.batch_read(i, min(int(counts[i]), 8192))
            _assert_frame_equal(oracle, ref, int(counts[i]), c, d)
        prog.write_input_image(frame)
        total = prog.extract_corners()
        n = min(total, 8192)
        _assert_frame_equal(oracle, ref, total, prog.read_corners(np.zeros(n, dtype=tinyorb.CORNER_DTYPE)),
                            prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32)))


@pytest.mark.parametrize("fp", [4, 12, 15])
def test_fp_forms_on_the_brief_fallback_kernel(tinyorb, oracle, monkeypatch, fp):
    """TINYORB_BRIEF_ROWS=1 sends every keypoint through k_brief_rows (the wave-per-keypoint kernel frames too large for k_brief_t's
    staging take): it rotates at run time and must follow the rotation's form too."""
    monkeypatch.setenv("TINYORB_BRIEF_ROWS", "1")
    W, H = 640, 480
    frame = oracle.synth_frame(W, H, 2, 15)
    ref = oracle.extract(frame, depth=2, threshold=THR, **_oracle_kw(fp))
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=2, initial_threshold=THR, fp_contract=fp, max_batch=2)
    with tinyorb.OrbProgram(cfg).init() as prog:
        prog.extract_batch_host(np.stack([frame, frame]))
        counts = prog.batch_counts(2)
        c, d = prog.batch_read(1, min(int(counts[1]), 8192))
        _assert_frame_equal(oracle, ref, int(counts[1]), c, d)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fp", [7, 15])
def test_fp_contract_batch_of_256(tinyorb, oracle, fp):
    """256 distinct frames through the fused kernels under a contracting compiler, EVERY frame against the restatement (320x240), and
    BASELINE configs[3]'s shape -- 256 x 1280x720 generated on the device -- fused against the per-stage kernels frame by frame, with
    the first four also against the restatement."""
    W, H, B = 320, 240, 256
    frames = np.stack([oracle.synth_frame(W, H, 9000 + i, 15) for i in range(B)])
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=2, initial_threshold=THR, fp_contract=fp, max_batch=B, max_features=4096)
    with tinyorb.OrbProgram(cfg).init() as prog:
        assert prog.pipeline() == "fused"
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(B)
        for i in range(B):
            ref = oracle.extract(frames[i], depth=2, threshold=THR, max_features=4096, **_oracle_kw(fp))
            c, d = prog.batch_read(i, min(int(counts[i]), 4096))
            _assert_frame_equal(oracle, ref, int(counts[i]), c, d)
    W, H = 1280, 720
    out = {}
    for flags in (0, tinyorb.ORB_FLAG_STAGED):
        cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=2, initial_threshold=THR, fp_contract=fp, max_batch=B, flags=flags)
        with tinyorb.OrbProgram(cfg).init() as prog:
            dev = prog.synth_frames_device(B, 1000, 15)
            prog.extract_batch_device(dev, B)
            counts = prog.batch_counts(B)
            out[flags] = [(int(counts[i]),) + _sorted(*prog.batch_read(i, min(int(counts[i]), 8192))) for i in range(B)]
    for i in range(B):
        a, b = out[0][i], out[tinyorb.ORB_FLAG_STAGED][i]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), i
    for i in range(4):
        ref = oracle.extract(oracle.synth_frame(W, H, 1000 + i, 15), depth=2, threshold=THR, **_oracle_kw(fp))
        _assert_frame_equal(oracle, ref, out[0][i][0], out[0][i][1], out[0][i][2])


def test_fp_contract_is_refused_with_the_extensions(tinyorb):
    """The switch follows the reference's shader compiler: it exists for the reference's detector on RGBA input only, and is a mask of
    four bits."""
    for kw in (dict(flags=tinyorb.ORB_FLAG_INTENDED), dict(flags=tinyorb.ORB_FLAG_NMS), dict(fast_arc=9), dict(flags=tinyorb.ORB_FLAG_INPUT_Y8),
               dict(fp_contract=16)):
        kw.setdefault("fp_contract", tinyorb.ORB_FP_CONTRACT_ALL)
        with pytest.raises(tinyorb.OrbError) as e:
            tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), **kw)).init()
        assert e.value.code == tinyorb.ORB_EINVAL
    with tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), fp_contract=tinyorb.ORB_FP_LAST_TERM_FIRST)).init() as prog:
        assert prog.pipeline() == "fused"


# ---------------------------------------------------------------------------------------------
# intended mode, IM-6b (OrbOptions::angle_bins): descriptors rotated by the centre of the keypoint's angle bin
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bins", [30, 1024, 6284, 8])
@pytest.mark.parametrize("flags_extra", [0, 1])
def test_intended_angle_bins_match_the_oracle(tinyorb, oracle, bins, flags_extra):
    """ORB_FLAG_INTENDED with angle_bins on the fused kernels (k_brief_i reads a table of one rotated pattern per BIN) and on the
    per-stage ones (k_brief rotates by the bin's centre code): keypoints, angles (still milliradian codes) and descriptors equal the
    restatement's under the same option -- a batch of three 640x480 frames and the six calls; 6284 bins reproduce the unbinned
    descriptors, 30 bins change nearly all of them."""
    W, H, depth = 640, 480, 2
    frames = np.stack([oracle.synth_frame(W, H, 40 + i, 15) for i in range(3)])
    flags = tinyorb.ORB_FLAG_INTENDED | tinyorb.ORB_FLAG_NMS | (tinyorb.ORB_FLAG_STAGED if flags_extra else 0)
    cfg = tinyorb.OrbConfig(tinyorb.Extent3d(W, H), hierarchy_depth=depth, initial_threshold=THR, flags=flags, fast_arc=9, max_batch=3,
                            angle_bins=bins)
    refs = [oracle.extract_intended(f, depth=depth, threshold=THR, arc=9, nms=True, angle_bins=bins) for f in frames]
    with tinyorb.OrbProgram(cfg).init() as prog:
        assert prog.pipeline() == ("staged" if flags_extra else "fused")
        prog.extract_batch_host(frames)
        counts = prog.batch_counts(3)
        for i in range(3):
            c, d = prog.batch_read(i, min(int(counts[i]), 8192))
            _assert_frame_equal(oracle, refs[i], int(counts[i]), c, d)
        prog.write_input_image(frames[1])
        total = prog.extract_corners()
        n = min(total, 8192)
        _assert_frame_equal(oracle, refs[1], total, prog.read_corners(np.zeros(n, dtype=tinyorb.CORNER_DTYPE)),
                            prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32)))
    plain = oracle.extract_intended(frames[0], depth=depth, threshold=THR, arc=9, nms=True)
    changed = int((refs[0]["descriptors"] != plain["descriptors"]).any(1).sum())
    assert np.array_equal(refs[0]["corners"], plain["corners"])  # positions and reported angles do not depend on the bins
    assert (changed == 0) if bins == 6284 else (changed > len(plain["descriptors"]) // (3 if bins == 1024 else 2))


def test_angle_bins_is_an_option_of_the_intended_mode(tinyorb):
    for kw in (dict(angle_bins=1024), dict(angle_bins=4, flags=tinyorb.ORB_FLAG_INTENDED), dict(angle_bins=7000, flags=tinyorb.ORB_FLAG_INTENDED)):
        with pytest.raises(tinyorb.OrbError) as e:
            tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), **kw)).init()
        assert e.value.code == tinyorb.ORB_EINVAL
    with tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(320, 240), flags=tinyorb.ORB_FLAG_INTENDED, fast_arc=9, angle_bins=1024)).init() as prog:
        table, pitch = prog.rot_table()
        assert table.shape[0] == 1024
