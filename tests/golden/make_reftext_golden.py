#!/usr/bin/env python3
"""Generates tests/golden/reftext/*.npz: what the REFERENCE'S SHADER TEXT yields when tests/wgsl_interp.py executes it
(tests/test_reference_text.py: the passes of orb.rs:469-557 around the interpreted shaders, the implementation-defined points bound to
SURVEY.md's CRD decisions).  Needs the reference checkout (/root/reference/src/shaders/*.wgsl, read as text at run time); the fixtures are
DATA -- the input frame, every grey and blur level as binary16 bit patterns, the raw counter, the sorted keypoints and their descriptors --
and travel to machines that have no checkout (the GPU box), where tests/test_oracle.py and tests/test_gpu_reftext.py compare the restatement
and the kernels with them.  Not a pin to an adapter: DESIGN.md section 2.

    python tests/golden/make_reftext_golden.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_text as rt  # noqa: E402
from oracle import orb_oracle  # noqa: E402

THR = np.float32(20.0 / 255.0)
CAP = 512
# name, W, H, depth, seed, generator flags
CASES = [
    ("t64x48_d2", 64, 48, 2, 5, 15),
    ("t80x56_d2", 80, 56, 2, 9, 15),
    ("t128x96_d3", 128, 96, 3, 13, 15),
    ("t160x112_d2", 160, 112, 2, 17, 7),
]


def main():
    orb_oracle.build()
    os.makedirs(os.path.join(HERE, "reftext"), exist_ok=True)
    for name, W, H, depth, seed, flags in CASES:
        t0 = time.time()
        rgba = rt.frame_with_corners(orb_oracle, W, H, seed, flags)
        gray, blur, total, c, d = rt.run_reference_text(orb_oracle, rgba, depth, THR, CAP)
        assert total <= CAP
        order = np.lexsort((c[:, 0], c[:, 1], c[:, 3]))
        out = dict(rgba=rgba, depth=np.uint32(depth), threshold=THR, max_features=np.uint32(CAP), total=np.uint32(total),
                   corners=c[order].astype(np.uint32), descriptors=d[order].astype(np.uint32))
        for m in range(depth):
            out["gray%d" % m] = gray[m].a.astype(np.float16).view(np.uint16)
            out["blur%d" % m] = blur[m].a.astype(np.float16).view(np.uint16)
        np.savez_compressed(os.path.join(HERE, "reftext", name + ".npz"), **out)
        print("%s: %d keypoints (%s per octave, %d with a non-zero angle), %.0f s"
              % (name, total, np.bincount(c[:, 3], minlength=depth).tolist(), int((c[:, 2] > 0).sum()), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
