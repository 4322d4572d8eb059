"""The one-image-ahead loop against the number of HIP streams created before the program's (hardware-queue assignment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
W, H = 1280, 720
extra = []
for n_extra in range(0, 7):
    if n_extra:
        extra.append(orb.OrbProgram(orb.OrbConfig(orb.Extent3d(64, 48), max_batch=1)).init())  # one more stream in the process
    prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
    dev = prog.synth_frames_device(1, 1000)
    frame = prog.copy_to_host(dev, W * H * 4)
    pins = [orb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
    for p in pins:
        p.array[:] = frame.reshape(H, W, 4)
    k = [0]
    prog.write_input_image_pinned(pins[0].array)
    def loop():
        prog.write_input_image_pinned(pins[(k[0] + 1) & 1].array)
        prog.extract_corners()
        k[0] += 1
    for _ in range(10):
        loop()
    t0 = time.perf_counter()
    for _ in range(200):
        loop()
    print("streams before: %d  ahead loop %.1f us" % (n_extra, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
    prog.upload_sync()
    prog.close()
