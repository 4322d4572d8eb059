"""Literal mode at larger frame sizes (default 4K, 3840x2160): the fused kernels against the per-stage kernels.
usage (GPU box): python tools/wide_rate.py [W H batch depth [max_features]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyslam_amd import orb

W, H, B, D = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (3840, 2160, 32, 3)
CAP = int(sys.argv[5]) if len(sys.argv) >= 6 else (8192 if W * H <= 1280 * 960 else 1 << 16)  # bench.py's capacity where it is enough
for name, flags in (("fused", 0), ("staged", orb.ORB_FLAG_STAGED)):
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=CAP, hierarchy_depth=D, initial_threshold=20.0 / 255.0, max_batch=B, flags=flags)
    with orb.OrbProgram(cfg) as prog:
        dev = prog.synth_frames_device(B, 1000)
        for _ in range(3):
            prog.extract_batch_device(dev, B)
        prog.batch_sync()
        prog.profile_enable(True); prog.profile_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            prog.extract_batch_device(dev, B)
        prog.batch_sync()
        dt = (time.perf_counter() - t0) / 10
        prof = {k: round(v[0] / 10, 4) for k, v in prog.profile().items()}
        counts = prog.batch_counts(B)
        print("%dx%d %-8s %s: %.3f ms per %d frames = %.0f frames/s = %.2f Gpixel/s, %.0f keypoints/frame, ms per batch %s"
              % (W, H, name, prog.pipeline(), dt * 1e3, B, B / dt, B * W * H / dt / 1e9, counts.mean(), prof), flush=True)
