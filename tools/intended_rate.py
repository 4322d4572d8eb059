"""Throughput of the opt-in "intended" mode (ORB_FLAG_INTENDED, staged pipeline) on the bench workload, for DESIGN.md.
Never bench.py's `value`: the headline metric is the reference's literal algorithm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
B, W, H = 256, 1280, 720
for name, flags, arc in (("intended FAST-9", orb.ORB_FLAG_INTENDED, 9),
                         ("intended FAST-9 + NMS", orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS, 9),
                         ("intended FAST-12 + NMS", orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS, 12),
                         ("literal FAST-9 + NMS", orb.ORB_FLAG_NMS, 9),
                         ("literal FAST-9", 0, 9),
                         ("literal FAST-12 + NMS", orb.ORB_FLAG_NMS, 12),
                         ("literal FAST-9 + NMS staged", orb.ORB_FLAG_NMS | orb.ORB_FLAG_STAGED, 9),
                         ("literal staged", orb.ORB_FLAG_STAGED, 0)):
    prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=B, flags=flags, fast_arc=arc)).init()
    dev = prog.synth_frames_device(B, 1000)
    for _ in range(2):
        prog.extract_batch_device(dev, B)
    prog.batch_sync()
    prog.profile_enable(True)
    prog.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        prog.extract_batch_device(dev, B)
    prog.batch_sync()
    dt = (time.perf_counter() - t0) / 5
    counts = prog.batch_counts(B)
    prof = {k: round(v[0] / 5, 3) for k, v in prog.profile().items()}
    print("%-28s %.2f ms per %d frames = %.0f frames/s; raw keypoints/frame mean %.0f max %d; ms per batch by kernel: %s"
          % (name, dt * 1e3, B, B / dt, counts.mean(), counts.max(), prof))
    prog.close()
