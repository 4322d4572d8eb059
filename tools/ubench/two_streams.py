"""Two batches in flight (two programs, two streams) against one: does overlapping the tails of consecutive batches pay?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tinyslam_amd import orb
B, W, H = 256, 1280, 720
dev = torch.device("cuda:0")
progs = [orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=B)).init() for _ in range(3)]
frames = progs[0].synth_frames_device(B, 1000)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def run(n_inflight, steps):
    for k in range(steps):
        i = k % n_inflight
        progs[i].extract_batch_device(frames, B, stream=streams[i].cuda_stream)
for n in (1, 2, 3, 1, 2, 3):
    run(n, 200); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(n, 200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("in flight %d: %.4f ms per batch = %.0f frames/s" % (n, dt * 1e3, B / dt))
