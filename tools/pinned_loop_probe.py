"""Where the time of the one-image-ahead camera loop goes (orb_write_input_image_pinned + extract_corners), 1280x720.
usage (GPU box): python tools/pinned_loop_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
W, H = 1280, 720
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
dev = prog.synth_frames_device(1, 1000)
frame = prog.copy_to_host(dev, W * H * 4)
pins = [orb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
for p in pins:
    p.array[:] = frame.reshape(H, W, 4)
corners = np.zeros(8192, dtype=orb.CORNER_DTYPE)
desc = np.zeros((8192, 8), dtype=np.uint32)
N = 200
def timeit(fn, n=N, warm=10):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6
print("extract alone                      %7.1f us" % timeit(prog.extract_corners))
print("blocking write                     %7.1f us" % timeit(lambda: prog.write_input_image(frame)))
prog.extract_corners()  # drain the waiting image
def w_sync():
    prog.write_input_image_pinned(pins[0].array); prog.upload_sync(); prog.extract_corners()
print("pinned write + upload_sync + extract %5.1f us" % timeit(w_sync))
t_w, t_e = [], []
k = [0]
prog.write_input_image_pinned(pins[0].array)
def loop():
    a = time.perf_counter()
    prog.write_input_image_pinned(pins[(k[0] + 1) & 1].array)
    b = time.perf_counter()
    prog.extract_corners()
    c = time.perf_counter()
    t_w.append(b - a); t_e.append(c - b); k[0] += 1
print("ahead loop                         %7.1f us" % timeit(loop))
print("  write_pinned call %.1f us, extract call %.1f us (means of the last %d)" % (np.mean(t_w[-N:]) * 1e6, np.mean(t_e[-N:]) * 1e6, N))
prog.upload_sync()
