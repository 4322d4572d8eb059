import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import orb_oracle as o
from tinyslam_amd import orb
W, H, depth = 640, 480, 2
frame = o.synth_frame(W, H, 1, 15)
for fp in (0, 4, 2):
    ref = o.extract(frame, depth=depth, threshold=20 / 255., planes=True, contract=fp & 7, dot_order=fp >> 3)
    for mb in (1, 3):
        cfg = orb.OrbConfig(orb.Extent3d(W, H), hierarchy_depth=depth, initial_threshold=20 / 255., fp_contract=fp, max_batch=mb)
        with orb.OrbProgram(cfg).init() as prog:
            if mb == 1:
                prog.write_input_image(frame); prog.extract_corners()
            else:
                prog.extract_batch_host(np.stack([frame]))
            dims, _ = o.level_dims(W, H, depth)
            for m, (w, h, off) in enumerate(dims):
                b = prog.read_plane(orb.ORB_PLANE_BLUR, m)
                r = ref["blur"][off:off + w * h].reshape(h, w)
                d = np.argwhere(b != r)
                print("fp", fp, "max_batch", mb, "level", m, "diffs", len(d), d[:6].tolist(), [(int(b[y, x]), int(r[y, x])) for y, x in d[:6]])
