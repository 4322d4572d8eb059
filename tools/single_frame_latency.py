"""Single-frame latency through the reference-shaped API (write_input_image / extract_corners / read_*).
usage (GPU box): python tools/single_frame_latency.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (1280, 720)
print("%dx%d" % (W, H))
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
dev = prog.synth_frames_device(1, 1000)
frame = prog.copy_to_host(dev, W * H * 4)
corners = np.zeros(8192, dtype=orb.CORNER_DTYPE)
desc = np.zeros((8192, 8), dtype=np.uint32)
for name, fn in (("write_input_image", lambda: prog.write_input_image(frame)),
                 ("extract_corners", lambda: prog.extract_corners()),
                 ("read_corners+descriptors", lambda: (prog.read_corners(corners), prog.read_descriptors(desc)))):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print("%-26s %8.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))
prog.profile_enable(True); prog.profile_reset()
for _ in range(20):
    prog.extract_corners()
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in prog.profile().items()}, "us per kernel launch")
