"""Phase shares of k_front / k_brief_tiles from in-kernel cycle stamps.
Needs a DIAGNOSTIC build:  TINYORB_BUILD_STAMPS=1 python -m tinyslam_amd.build --force  (rebuild without it afterwards)."""
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
raw = np.zeros(32, dtype=np.uint64)
prog._check(prog._lib.orb_debug_stamps(prog._handle(), raw.ctypes.data, raw.size))
front = raw.astype(np.float64)
fn = ["A stage", "bar", "B1 pretest", "bar", "S1 diag", "bar", "S2 ring", "bar", "S3 angle", "bar", "C0 mip", "C blur", "bar"]
for base, name in ((0, "k_front<L0>"), (16, "k_front<LN>")):
    tot = front[base:base + 13].sum()
    print(name, "wave-0 cycles per launch set: %.3e" % tot)
    for i, n in enumerate(fn):
        print("   %-12s %5.1f%%" % (n, 100 * front[base + i] / max(tot, 1)))
