#!/usr/bin/env python3
"""Differential fuzz of the restatement against the reference's shader text EXECUTED (tests/wgsl_interp.py): random small frames (noise,
salt-and-pepper, blocks, gradients), sizes, depths, thresholds and switches (out-of-level policy, sampler weight bits, contractions, reduction
order, negative-angle conversion); every plane and record compared with oracle/orb_oracle.c under the same switches.  Needs the reference
checkout.  TEST INFRASTRUCTURE.      python tools/fuzz_reference_text.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_text as rt  # noqa: E402
from oracle import orb_oracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orb_oracle.build()
bad, kp, t0 = 0, 0, time.time()
for case in range(n_cases):
    depth = int(rng.choice([1, 2, 2, 3]))
    q = 1 << (depth - 1)
    W, H = int(rng.integers(36 // q + 1, 80 // q)) * q * 2 // 2, int(rng.integers(36 // q + 1, 64 // q)) * q
    W, H = max(W - W % (2 * q), 36 + (36 % (2 * q))), max(H - H % (2 * q), 36 + (36 % (2 * q)))  # every level above the last has even sizes (exact 2x2 blit)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        rgba = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    elif kind == 1:
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        rgba[..., :3] = ((rng.random((H, W, 1)) < rng.choice([0.03, 0.1, 0.3])) * rng.integers(100, 256)).astype(np.uint8)
    elif kind == 2:
        rgba = orb_oracle.synth_frame(W, H, int(rng.integers(0, 1 << 20)), 15)
        for _ in range(12):
            s = int(rng.integers(1, 5)); x, y = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 4))
            rgba[y:y + s, x:x + s, :3] = rng.integers(0, 256, size=3)
    else:
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        b = int(rng.integers(2, 7))
        blocks = rng.integers(0, 256, size=((H + b - 1) // b, (W + b - 1) // b, 3), dtype=np.uint8)
        rgba[..., :3] = np.kron(blocks, np.ones((b, b, 1), dtype=np.uint8))[:H, :W]
    rgba[..., 3] = 255
    thr = np.float32(rng.choice([5, 12, 20, 40, 80]) / 255.0)
    sw = {}
    if rng.random() < 0.6:
        sw = dict(oob=str(rng.choice(["zero", "clamp", "umin"])), weight_bits=int(rng.choice([0, 0, 8, 4])), contract=int(rng.integers(0, 8)),
                  dot_order=int(rng.integers(0, 2)), neg_angle=str(rng.choice(["zero", "zero", "wrap", "ones"])))
    cap = 4096
    got = rt.run_reference_text(orb_oracle, rgba, depth, thr, cap, **sw)
    if got[2] > cap:
        continue
    diff = rt.differences(orb_oracle, rgba, depth, thr, cap, got, **sw)
    kp += got[2]
    if diff:
        bad += 1
        print("MISMATCH", dict(W=W, H=H, depth=depth, kind=kind, thr=float(thr), sw=sw, diff=diff), flush=True)
    if case % 10 == 9:
        print("case %d, %d mismatches, %d keypoints so far, %.0f s" % (case + 1, bad, kp, time.time() - t0), flush=True)
print("done: %d cases, %d keypoints, %d mismatches" % (n_cases, kp, bad))
sys.exit(1 if bad else 0)
