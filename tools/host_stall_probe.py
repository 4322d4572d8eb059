"""In bench.py's process the one-image-ahead loop is sometimes 3x slower and the time sits in read_corners / read_descriptors (a host
memcpy out of the pinned staging arrays).  Control: the same loop with a plain numpy copy of the same size in place of the reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tinyslam_amd import orb
W, H = 1280, 720
big = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=256, flags=orb.ORB_FLAG_DOUBLE_OUTPUT)).init()
dev = torch.device("cuda", 0)
frames_t = torch.empty(256 * W * H * 4, dtype=torch.uint8, device=dev)
big.synth_frames_device(256, 1000, frames_dev_ptr=frames_t.data_ptr())
for _ in range(200):
    big.extract_batch_device(frames_t.data_ptr(), 256)
big.batch_sync()
if "--cpu" in sys.argv:
    from oracle import orb_oracle
    fr = np.stack([orb_oracle.synth_frame(W, H, 1000 + i) for i in range(32)])
    orb_oracle.extract_batch(fr, depth=2, max_features=8192, n_threads=16)
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
d = prog.synth_frames_device(1, 1000)
frame = prog.copy_to_host(d, W * H * 4)
pins = [orb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
for p in pins:
    p.array[:] = frame.reshape(H, W, 4)
corners = np.zeros(8192, dtype=orb.CORNER_DTYPE)
desc = np.zeros((8192, 8), dtype=np.uint32)
src_a, dst_a = np.ones(8192 * 12, dtype=np.uint32), np.zeros(8192 * 12, dtype=np.uint32)
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
        if reads == "staging":
            prog.read_corners(corners); prog.read_descriptors(desc)
        elif reads == "control":
            np.copyto(dst_a, src_a)
        elif reads == "count":
            prog.read_corners(corners[:3800]); prog.read_descriptors(desc[:3800])
        dd = time.perf_counter()
        tw += b - a; te += c - b; tr += dd - c
    prog.extract_corners()
    print("%-10s write %.1f extract %.1f reads %.1f us" % (tag, tw / n * 1e6, te / n * 1e6, tr / n * 1e6), flush=True)
for rep in range(3):
    run("staging", "staging"); run("control", "control"); run("count", "count"); run("none", None)
