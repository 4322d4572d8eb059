"""Rate of the Hamming matcher (orb_match_consecutive) on the bench workload, for DESIGN.md.  Never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
B, W, H = 256, 1280, 720
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=B)).init()
dev = prog.synth_frames_device(B, 1000)
prog.extract_batch_device(dev, B)
prog.batch_sync()
counts = np.minimum(prog.batch_counts(B), prog.config.max_features).astype(np.int64)
prog.match_consecutive(B)
prog.batch_sync()
prog.profile_enable(True)
prog.profile_reset()
t0 = time.perf_counter()
for _ in range(5):
    prog.match_consecutive(B)
prog.batch_sync()
dt = (time.perf_counter() - t0) / 5
pairs = float((counts[:-1] * counts[1:]).sum())
print("match: %.2f ms per %d frame pairs (%.0f x %.0f descriptors each) = %.2e descriptor pairs/s; kernel %s"
      % (dt * 1e3, B - 1, counts.mean(), counts.mean(), pairs / dt, {k: round(v[0] / 5, 3) for k, v in prog.profile().items()}))
