import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tinyslam_amd import orb
from oracle import orb_oracle as oo
W, H, cap = 2048, 2200, 1 << 16
rgba = oo.synth_frame(W, H, 31)
cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=cap, hierarchy_depth=2, initial_threshold=20.0 / 255.0)
bad = 0
with orb.OrbProgram(cfg) as prog:
    t0, c0, d0 = prog.extract(rgba)
    o = np.lexsort((c0["x"], c0["y"], c0["octave"])); c0, d0 = c0[o], d0[o]
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        t, c, d = prog.extract(rgba)
        if t != t0:
            bad += 1; continue
        o = np.lexsort((c["x"], c["y"], c["octave"]))
        if not (np.array_equal(c[o], c0) and np.array_equal(d[o], d0)):
            bad += 1
print("first total", t0, "bad calls", bad)
