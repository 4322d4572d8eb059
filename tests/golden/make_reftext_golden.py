#!/usr/bin/env python3
"""Generates tests/golden/reftext/*.npz: what the REFERENCE'S SHADER TEXT yields when tests/wgsl_interp.py executes it
(tests/test_reference_text.py: the passes of orb.rs:469-557 around the interpreted shaders, the implementation-defined points bound to
SURVEY.md's CRD decisions).  Needs the reference checkout (/root/reference/src/shaders/*.wgsl, read as text at run time); the fixtures are
DATA -- the input frame, every grey and blur level as binary16 bit patterns, the raw counter, the sorted keypoints and their descriptors --
and travel to machines that have no checkout (the GPU box), where tests/test_oracle.py and tests/test_gpu_reftext.py compare the restatement
and the kernels with them.  Not a pin to an adapter: DESIGN.md section 2.

    python tests/golden/make_reftext_golden.py            # the four small frames, one minute
    python tests/golden/make_reftext_golden.py config0    # BASELINE.json configs[0]'s 640x480 frame (nine minutes: the interpreter runs
                                                          # 1.9 million fragment and 0.4 million compute invocations)
    python tests/golden/make_reftext_golden.py config1    # configs[1]'s 1280x720 frame (half an hour)
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


# BASELINE.json configs[0]: one 640x480 frame, generator gradient + blobs + wedges, seed 1, depth 2, max_features 8192 -- the frame itself is not
# stored (1.2 MB): the fixture holds its generator parameters and SHA-256, tests rebuild it with oracle.synth_frame
CONFIG0 = ("t640x480_d2_config0", 640, 480, 2, 1, 7)
# BASELINE.json configs[1] / [2]: one 1280x720 frame, + noise, seed 2 -- the benchmark's frame size.  Half an hour of interpreter; the grey levels
# (noise: incompressible) are stored as SHA-256 only.
CONFIG1 = ("t1280x720_d2_config1", 1280, 720, 2, 2, 15)
# the second frame tools/pin_oracle.py asks a maintainer to dump (`frame <out> 2 15`): 640x480 with noise -- it tells the sampler's weight bits apart
NOISY0 = ("t640x480_d2_noisy", 640, 480, 2, 2, 15)


def main():
    orb_oracle.build()
    os.makedirs(os.path.join(HERE, "reftext"), exist_ok=True)
    big = {"config0": CONFIG0, "config1": CONFIG1, "noisy0": NOISY0}.get(sys.argv[1] if len(sys.argv) > 1 else "")
    config0 = big is not None
    for name, W, H, depth, seed, flags in ([big] if big else CASES):
        t0 = time.time()
        cap = 8192 if config0 else CAP
        rgba = orb_oracle.synth_frame(W, H, seed, flags) if config0 else rt.frame_with_corners(orb_oracle, W, H, seed, flags)
        gray, blur, total, c, d = rt.run_reference_text(orb_oracle, rgba, depth, THR, cap)
        assert total <= cap
        order = np.lexsort((c[:, 0], c[:, 1], c[:, 3]))
        out = dict(depth=np.uint32(depth), threshold=THR, max_features=np.uint32(cap), total=np.uint32(total),
                   corners=c[order].astype(np.uint32), descriptors=d[order].astype(np.uint32))
        if config0:
            import hashlib
            out.update(synth=np.array([W, H, seed, flags], dtype=np.uint32), rgba_sha256=np.array(hashlib.sha256(rgba.tobytes()).hexdigest()))
        else:
            out.update(rgba=rgba)
        for m in range(depth):
            g16 = gray[m].a.astype(np.float16).view(np.uint16)
            if W * H > 400000 or name.endswith("_noisy"):
                import hashlib
                out["gray%d_sha256" % m] = np.array(hashlib.sha256(np.ascontiguousarray(g16).tobytes()).hexdigest())
            else:
                out["gray%d" % m] = g16
            out["blur%d" % m] = blur[m].a.astype(np.float16).view(np.uint16)
        np.savez_compressed(os.path.join(HERE, "reftext", name + ".npz"), **out)
        print("%s: %d keypoints (%s per octave, %d with a non-zero angle), %.0f s"
              % (name, total, np.bincount(c[:, 3], minlength=depth).tolist(), int((c[:, 2] > 0).sum()), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
