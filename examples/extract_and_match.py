#!/usr/bin/env python3
"""Minimal tour of the Python mirror of `tinyslam::orb` (needs an MI355X; build first: python -m tinyslam_amd.build).

    python examples/extract_and_match.py            # the reference's algorithm (default)
    python examples/extract_and_match.py intended   # the opt-in repaired algorithm, FAST-9 + NMS
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tinyslam_amd import orb

W, H = 640, 480
intended = len(sys.argv) > 1 and sys.argv[1] == "intended"
cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=4096, hierarchy_depth=2, initial_threshold=20.0 / 255.0, max_batch=2,
                    flags=(orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS) if intended else 0, fast_arc=9 if intended else 0)

with orb.OrbProgram(cfg) as prog:  # == OrbProgram { config, .. }.init() in the reference (orb.rs:107)
    # --- the reference's six calls, one frame at a time -------------------------------------------------------
    dev = prog.synth_frames_device(2, seed0=7)                      # two synthetic RGBA frames on the device
    frames = prog.copy_to_host(dev, 2 * W * H * 4).reshape(2, H, W, 4)
    prog.write_input_image(frames[0])                               # orb.rs:567
    prog.set_threshold(20.0 / 255.0)                                # orb.rs:585
    total = prog.extract_corners()                                  # orb.rs:469 (raw counter)
    n = min(total, cfg.max_features)
    corners = prog.read_corners(np.zeros(n, dtype=orb.CORNER_DTYPE))       # orb.rs:559
    descriptors = prog.read_descriptors(np.zeros((n, 8), dtype=np.uint32))  # orb.rs:563
    print("%s pipeline, frame 0: %d keypoints, first: %s" % (prog.pipeline(), total, corners[0]))
    x0, y0 = orb.level0_xy(corners)
    print("octave-1 keypoints in level-0 pixels:", np.stack([x0, y0], 1)[corners["octave"] == 1][:3])

    # --- batched mode + matching between consecutive frames ---------------------------------------------------
    prog.extract_batch_host(frames)
    counts = np.minimum(prog.batch_counts(2), cfg.max_features)
    prog.match_consecutive(2)
    m = prog.match_read(0, int(counts[0]))
    good = m["distance"] < 0.8 * m["second"]                        # ratio test on the two best Hamming distances
    print("frame 0 -> 1: %d of %d keypoints pass the ratio test (the two synthetic frames are unrelated scenes)"
          % (int(good.sum()), int(counts[0])))
