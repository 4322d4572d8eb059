import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
import bench
print("standalone:", bench.single_frame_latency(orb, dict(max_features=8192, hierarchy_depth=2, initial_threshold=20/255.0, device=0)))
W, H = 1280, 720
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
dev = prog.synth_frames_device(1, 1000)
frame = prog.copy_to_host(dev, W * H * 4)
pins = [orb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
for p in pins:
    p.array[:] = frame.reshape(H, W, 4)
corners = np.zeros(8192, dtype=orb.CORNER_DTYPE)
desc = np.zeros((8192, 8), dtype=np.uint32)
def run(tag, reads, n=200):
    k = [0]
    prog.write_input_image_pinned(pins[0].array)
    tw = te = tr = 0.0
    for i in range(n + 10):
        if i == 10:
            tw = te = tr = 0.0
        a = time.perf_counter()
        prog.write_input_image_pinned(pins[(k[0] + 1) & 1].array); k[0] += 1
        b = time.perf_counter()
        prog.extract_corners()
        c = time.perf_counter()
        if reads:
            prog.read_corners(corners); prog.read_descriptors(desc)
        d = time.perf_counter()
        tw += b - a; te += c - b; tr += d - c
    prog.extract_corners()
    print("%-12s write %.1f extract %.1f reads %.1f us" % (tag, tw / n * 1e6, te / n * 1e6, tr / n * 1e6), flush=True)
run("no reads", False)
run("with reads", True)
run("no reads", False)
