"""Host-inclusive rate (frames arriving in pageable host memory -> results on the device), for DESIGN.md.
Never bench.py's `value`.  usage (GPU box): python tools/ingest_rate.py [y8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
B, W, H = 256, 1280, 720
Y8 = len(sys.argv) > 1 and sys.argv[1] == "y8"
bpp = 1 if Y8 else 4
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=B, flags=orb.ORB_FLAG_INPUT_Y8 if Y8 else 0)).init()
dev = prog.synth_frames_device(B, 1000)  # a Y8 program generates one byte per pixel
frames = prog.copy_to_host(dev, B * W * H * bpp).reshape((B, H, W) if Y8 else (B, H, W, 4))
prog.extract_batch_device(dev, B); prog.batch_sync()
ref = prog.batch_counts(B)
for it in range(3):
    t0 = time.perf_counter()
    prog.extract_batch_host(frames)
    prog.batch_sync()
    dt = time.perf_counter() - t0
    assert np.array_equal(prog.batch_counts(B), ref)
    print("host-inclusive (%s): %.1f ms per %d frames = %.0f frames/s (%.1f GB/s of input)" % ("Y8" if Y8 else "RGBA", dt * 1e3, B, B / dt, B * W * H * bpp / dt / 1e9))
