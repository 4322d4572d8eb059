import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TINYORB_STAMPS"] = "1"
import numpy as np
from tinyslam_amd import orb
B = 256
prog = orb.OrbProgram(orb.OrbConfig(orb.Extent3d(1280, 720), max_batch=B)).init()
dev = prog.synth_frames_device(B, 1000)
for _ in range(3):
    prog.extract_batch_device(dev, B)
prog.batch_sync()
st = prog.debug_stamps(512).astype(np.float64)
names = ["issue", "describe", "bar_after_describe", "bar_reset", "commit", "bar_after_commit"]
tot = st.sum(1)
print("per-workgroup total cycles: mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-20s mean %9.0f cycles  %5.1f%%" % (n, st[:, i].mean(), 100 * st[:, i].sum() / tot.sum()))
