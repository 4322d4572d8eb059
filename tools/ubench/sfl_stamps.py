import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tinyslam_amd import orb
W, H = 1280, 720
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_batch=1)).init()
dev = prog.synth_frames_device(1, 1000)
frame = prog.copy_to_host(dev, W * H * 4)
prog.write_input_image(frame)
for _ in range(8):
    prog.extract_corners()
