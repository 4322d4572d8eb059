#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU restatement (oracle/orb_oracle.c),
cross-checked here against the independent NumPy restatement (oracle/orb_numpy.py).

PARITY UNPINNED: the reference ships no vectors and cannot run in this environment, so these are
outputs of the BUILD's own oracle (BASELINE.json configs[0] "ground-truth dump through
the reference wgpu/CPU-adapter path" is replaced by this).  A fixture is data only: generator
parameters, SHA-256 of the generated RGBA frame and of the intermediate planes, the sorted keypoints
and their descriptors.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orb_numpy, orb_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
THR = np.float32(20.0 / 255.0)
# name, W, H, depth, seed, flags
CASES = [
    ("g64x48_d2", 64, 48, 2, 3, 15),
    ("g160x120_d3", 160, 120, 3, 1, 15),
    ("g200x97_d3_odd", 200, 97, 3, 7, 15),
    ("g640x480_d2_config1", 640, 480, 2, 1, 7),   # BASELINE.json configs[0]: 640x480 gradient frame (+blobs+wedges)
    ("g332x202_d5", 332, 202, 5, 11, 15),
]


# "intended" mode (definitions IM-1..IM-8 in oracle/orb_oracle.h): name, W, H, depth, seed,
# flags, max_features, arc, nms.  Kept in tests/golden/intended/.
INTENDED_CASES = [
    ("i160x120_d2_arc9", 160, 120, 2, 21, 15, 8192, 9, 0),
    ("i320x240_d3_arc9_nms", 320, 240, 3, 22, 15, 8192, 9, 1),
    ("i332x202_d4_arc12_top200", 332, 202, 4, 23, 15, 200, 12, 0),
    ("i640x480_d2_arc9_nms_top500", 640, 480, 2, 24, 15, 500, 9, 1),
]


# Y8 input variant (ORB_FLAG_INPUT_Y8; not in the reference's code, see oracle/orb_oracle.c): name, W, H, depth, seed,
# flags.  The frame is the integer BT.601 luma of the synthetic RGBA frame.  Kept in tests/golden/y8/.
Y8_CASES = [
    ("y160x120_d3", 160, 120, 3, 31, 15),
    ("y640x480_d2", 640, 480, 2, 32, 15),
]


# The implementation-defined switches (oracle/orb_oracle.h, orc_impl_t; include/tinyorb.h, OrbOptions::oob_policy /
# sampler_weight_bits): name, W, H, depth, seed, flags, oob, weight bits.  Frames chosen so that the switch changes the
# result (keypoints at octaves >= 1 near the level's edge, samples that leave the level -- for s45 one with a negative
# coordinate, where clamp and umin part --, lerp weights of the blur and of the odd-sized blit).  Kept in tests/golden/impl/.
IMPL_CASES = [
    ("m320x240_d3_s45_clamp_w0", 320, 240, 3, 45, 15, "clamp", 0),
    ("m320x240_d3_s45_umin_w0", 320, 240, 3, 45, 15, "umin", 0),
    ("m320x240_d3_s45_zero_w8", 320, 240, 3, 45, 15, "zero", 8),
    ("m320x240_d3_s45_umin_w8", 320, 240, 3, 45, 15, "umin", 8),
    ("m333x211_d4_s41_zero_w8", 333, 211, 4, 41, 15, "zero", 8),
    ("m333x211_d4_s41_clamp_w8", 333, 211, 4, 41, 15, "clamp", 8),
    ("m640x480_d2_s41_clamp_w0", 640, 480, 2, 41, 15, "clamp", 0),
    ("m640x480_d2_config1_umin_w8", 640, 480, 2, 1, 7, "umin", 8),  # BASELINE.json configs[0]'s frame under the other reading
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    for name, W, H, depth, seed, flags in CASES:
        rgba = orb_oracle.synth_frame(W, H, seed, flags)
        assert np.array_equal(rgba, orb_numpy.synth_frame(W, H, seed, flags))
        ref = orb_oracle.extract(rgba, depth=depth, threshold=THR, max_features=8192, planes=True)
        alt = orb_numpy.extract(rgba, depth=depth, threshold=THR, max_features=8192)
        kc = np.stack([ref["corners"][k] for k in ("x", "y", "angle", "octave")], 1)
        assert ref["total"] == alt["total"] and np.array_equal(kc, alt["corners"])
        assert np.array_equal(ref["descriptors"], alt["descriptors"])
        dims, _ = orb_oracle.level_dims(W, H, depth)
        gray_sha, blur_sha = [], []
        for m, (w, h, off) in enumerate(dims):
            g, b = ref["gray"][off:off + w * h], ref["blur"][off:off + w * h]
            assert np.array_equal(g, alt["gray"][m].ravel()) and np.array_equal(b, alt["blur"][m].ravel())
            gray_sha.append(sha(g))
            blur_sha.append(sha(b))
        corners, desc = orb_oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            params=np.array([W, H, depth, seed, flags, 8192], dtype=np.int64), threshold=THR,
            rgba_sha256=sha(rgba), gray_sha256=np.array(gray_sha), blur_sha256=np.array(blur_sha),
            total=np.int64(ref["total"]),
            corners=np.stack([corners[k] for k in ("x", "y", "angle", "octave")], 1).astype(np.uint32),
            descriptors=desc.astype(np.uint32))
        print(name, "total", ref["total"])
    os.makedirs(os.path.join(HERE, "impl"), exist_ok=True)
    for name, W, H, depth, seed, flags, oob, wbits in IMPL_CASES:
        rgba = orb_oracle.synth_frame(W, H, seed, flags)
        ref = orb_oracle.extract(rgba, depth=depth, threshold=THR, max_features=8192, planes=True, oob=oob, weight_bits=wbits)
        alt = orb_numpy.extract(rgba, depth=depth, threshold=THR, max_features=8192, oob=oob, weight_bits=wbits)
        base = orb_oracle.extract(rgba, depth=depth, threshold=THR, max_features=8192)
        kc = np.stack([ref["corners"][k] for k in ("x", "y", "angle", "octave")], 1)
        assert ref["total"] == alt["total"] and np.array_equal(kc, alt["corners"])
        assert np.array_equal(ref["descriptors"], alt["descriptors"])
        assert ref["total"] != base["total"] or not np.array_equal(ref["descriptors"], base["descriptors"]), "the switch changes nothing here"
        dims, _ = orb_oracle.level_dims(W, H, depth)
        blur_sha = []
        for m, (w, h, off) in enumerate(dims):
            b = ref["blur"][off:off + w * h]
            assert np.array_equal(b, alt["blur"][m].ravel())
            blur_sha.append(sha(b))
        corners, desc = orb_oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        np.savez_compressed(
            os.path.join(HERE, "impl", name + ".npz"),
            params=np.array([W, H, depth, seed, flags, 8192, orb_oracle.OOB_POLICIES[oob], wbits], dtype=np.int64), threshold=THR,
            rgba_sha256=sha(rgba), blur_sha256=np.array(blur_sha), total=np.int64(ref["total"]),
            total_default=np.int64(base["total"]),
            corners=np.stack([corners[k] for k in ("x", "y", "angle", "octave")], 1).astype(np.uint32),
            descriptors=desc.astype(np.uint32))
        print(name, "total", ref["total"], "(default switches:", base["total"], ")")
    os.makedirs(os.path.join(HERE, "y8"), exist_ok=True)
    for name, W, H, depth, seed, flags in Y8_CASES:
        y8 = orb_oracle.synth_frame_y8(W, H, seed, flags)
        ref = orb_oracle.extract_y8(y8, depth=depth, threshold=THR, max_features=8192, planes=True)
        alt = orb_numpy.extract(y8, depth=depth, threshold=THR, max_features=8192, y8=True)
        kc = np.stack([ref["corners"][k] for k in ("x", "y", "angle", "octave")], 1)
        assert ref["total"] == alt["total"] and np.array_equal(kc, alt["corners"])
        assert np.array_equal(ref["descriptors"], alt["descriptors"])
        dims, _ = orb_oracle.level_dims(W, H, depth)
        gray_sha, blur_sha = [], []
        for m, (w, h, off) in enumerate(dims):
            g, b = ref["gray"][off:off + w * h], ref["blur"][off:off + w * h]
            assert np.array_equal(g, alt["gray"][m].ravel()) and np.array_equal(b, alt["blur"][m].ravel())
            gray_sha.append(sha(g))
            blur_sha.append(sha(b))
        corners, desc = orb_oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        np.savez_compressed(
            os.path.join(HERE, "y8", name + ".npz"),
            params=np.array([W, H, depth, seed, flags, 8192], dtype=np.int64), threshold=THR,
            y8_sha256=sha(y8), gray_sha256=np.array(gray_sha), blur_sha256=np.array(blur_sha),
            total=np.int64(ref["total"]),
            corners=np.stack([corners[k] for k in ("x", "y", "angle", "octave")], 1).astype(np.uint32),
            descriptors=desc.astype(np.uint32))
        print(name, "total", ref["total"])
    os.makedirs(os.path.join(HERE, "intended"), exist_ok=True)
    for name, W, H, depth, seed, flags, cap, arc, nms in INTENDED_CASES:
        rgba = orb_oracle.synth_frame(W, H, seed, flags)
        ref = orb_oracle.extract_intended(rgba, depth=depth, threshold=THR, max_features=cap, arc=arc, nms=bool(nms),
                                          planes=True)
        alt = orb_numpy.extract_intended(rgba, depth=depth, threshold=THR, max_features=cap, arc=arc, use_nms=bool(nms))
        corners, desc = orb_oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        kc = np.stack([corners[k] for k in ("x", "y", "angle", "octave")], 1).astype(np.uint32)
        order = np.lexsort((alt["corners"][:, 0], alt["corners"][:, 1], alt["corners"][:, 3]))
        assert ref["total"] == alt["total"] and np.array_equal(kc, alt["corners"][order])
        assert np.array_equal(desc, alt["descriptors"][order])
        dims, _ = orb_oracle.level_dims(W, H, depth)
        gray_sha, blur_sha = [], []
        for m, (w, h, off) in enumerate(dims):
            g, b = ref["gray"][off:off + w * h], ref["blur"][off:off + w * h]
            assert np.array_equal(g, alt["gray"][m].ravel()) and np.array_equal(b, alt["blur"][m].ravel())
            gray_sha.append(sha(g))
            blur_sha.append(sha(b))
        np.savez_compressed(
            os.path.join(HERE, "intended", name + ".npz"),
            params=np.array([W, H, depth, seed, flags, cap, arc, nms], dtype=np.int64), threshold=THR,
            rgba_sha256=sha(rgba), gray_sha256=np.array(gray_sha), blur_sha256=np.array(blur_sha),
            total=np.int64(ref["total"]), corners=kc, descriptors=desc.astype(np.uint32))
        print(name, "total", ref["total"], "stored", len(kc))


if __name__ == "__main__":
    main()
