"""What in bench.py's process makes the one-image-ahead loop slow?  Stages added one by one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
W, H = 1280, 720
def ahead_loop(tag):
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
    print("%-40s ahead loop %.1f us" % (tag, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
    prog.upload_sync()
    prog.close()
ahead_loop("plain")
import torch
ahead_loop("torch imported")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
x = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
ahead_loop("torch tensor allocated")
big = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=256, flags=orb.ORB_FLAG_DOUBLE_OUTPUT)).init()
frames_t = torch.empty(256 * W * H * 4, dtype=torch.uint8, device=dev)
big.synth_frames_device(256, 1000, frames_dev_ptr=frames_t.data_ptr())
ahead_loop("big program + 0.9 GB torch tensor")
for _ in range(20):
    big.extract_batch_device(frames_t.data_ptr(), 256)
big.batch_sync()
ahead_loop("after 20 batches")
t = torch.tensor([1.0], dtype=torch.float64, device=dev); float(t.item()); torch.cuda.synchronize()
ahead_loop("after torch .item() + synchronize")
big.profile_enable(True); big.profile_reset()
for _ in range(5):
    big.extract_batch_device(frames_t.data_ptr(), 256)
big.batch_sync(); big.profile(); big.profile_enable(False)
ahead_loop("after a profiled pass (events)")
c = big.batch_counts(256)
ahead_loop("after batch_counts (hipMemcpy D2H)")
