"""Fused literal pipeline over frame sizes and depths (frames/s and ms per kernel); TINYORB_LIB selects the build (A/B on one box).
usage (GPU box): python tools/size_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyslam_amd import orb
CASES = [(320, 240, 1024, 3), (640, 480, 512, 2), (640, 480, 512, 4), (752, 480, 512, 3), (960, 540, 256, 3), (1280, 720, 256, 2), (1280, 720, 256, 4),
         (1241, 376, 512, 3), (1920, 1080, 128, 3), (2560, 1440, 64, 3), (3840, 2160, 32, 3)]
for W, H, B, D in CASES:
    cap = 8192 if W * H <= 1280 * 960 else 1 << 16
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=cap, hierarchy_depth=D, initial_threshold=20.0 / 255.0, max_batch=B)
    with orb.OrbProgram(cfg) as prog:
        dev = prog.synth_frames_device(B, 1000)
        for _ in range(3):
            prog.extract_batch_device(dev, B)
        prog.batch_sync()
        t0 = time.perf_counter()
        for _ in range(10):
            prog.extract_batch_device(dev, B)
        prog.batch_sync()
        dt = (time.perf_counter() - t0) / 10
        prog.profile_enable(True); prog.profile_reset()
        for _ in range(5):
            prog.extract_batch_device(dev, B)
        prog.batch_sync()
        prof = {k: round(v[0] / 5, 4) for k, v in prog.profile().items()}
        print("%4dx%-4d d%d x%-4d %.3f ms = %8.0f frames/s = %6.1f Gpixel/s  %s" % (W, H, D, B, dt * 1e3, B / dt, B * W * H / dt / 1e9, prof), flush=True)
