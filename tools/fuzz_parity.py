"""One-off randomized parity sweep on the GPU: random sizes, depths, thresholds, modes, arcs, capacities -- every result
against the C oracle (tests/ hold the fixed cases).  usage: python tools/fuzz_parity.py [n_cases] [seed] [wide]
("wide": every frame 1284..4156 columns wide -- the levels that mix 8-, 16- and 32-row bands and column tiles;
 "xwide": 1284..9196 columns, low frames: column tiles at several levels, widths past 4096)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
from oracle import orb_oracle as oo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WIDE = len(sys.argv) > 3 and sys.argv[3] in ("wide", "xwide")
XWIDE = len(sys.argv) > 3 and sys.argv[3] == "xwide"


def sort(c, d):
    o = np.lexsort((c["x"], c["y"], c["octave"]))
    return c[o], d[o]


bad = 0
t0 = time.time()
for case in range(n_cases):
    W = int(rng.integers(2, 180)) * 4 if rng.random() < 0.8 else int(rng.integers(9, 700))
    if rng.random() < 0.12:
        W = int(rng.integers(180, 1040)) * 4  # up to 4156 wide: 16-row bands, 8-row bands (2049..4096) and beyond
    if WIDE:
        W = int(rng.integers(321, 1040)) * 4
    H = int(rng.integers(8, 400)) if W <= 720 else int(rng.integers(8, 160))
    if WIDE:
        H = int(rng.integers(24, 260))
    if XWIDE:
        W = int(rng.integers(321, 2300)) * 4 + (int(rng.integers(0, 4)) if rng.random() < 0.2 else 0)
        H = int(rng.integers(20, 90))
    depth = int(rng.integers(1, 7))
    thr = float(np.float32(rng.choice([5, 12, 20, 40, 80]) / 255.0))
    intended = bool(rng.random() < 0.3)
    nms = bool(rng.random() < 0.35)
    arc = int(rng.choice([0, 0, 12, 9, 10, 13, 16]))
    y8 = (not intended) and (not nms) and arc in (0, 12) and bool(rng.random() < 0.3)  # ORB_FLAG_INPUT_Y8: the literal detector only
    # the implementation-defined switches (OrbOptions::oob_policy / sampler_weight_bits): the reference's detector only
    plain = (not intended) and (not nms) and arc in (0, 12)
    oob = str(rng.choice(["zero", "zero", "clamp", "umin"])) if plain else "zero"
    wbits = int(rng.choice([0, 0, 8, 4, 12])) if plain else 0
    # OrbOptions::fp_contract (round 5): which stages' products and sums are fused, and the reduction order -- RGBA, the reference's detector
    fp = int(rng.integers(0, 16)) if plain and not y8 and rng.random() < 0.5 else 0
    fpk = dict(contract=fp & 7, dot_order=fp >> 3)
    cap = int(rng.choice([8192, 8192, 300, 40, 5]))
    staged = bool(rng.random() < 0.25)
    syn = int(rng.choice([15, 15, 7, 14, 9]))
    seed = int(rng.integers(0, 1 << 30))
    rgba = oo.synth_frame(W, H, seed, syn)
    if rng.random() < 0.15:  # salt-and-pepper: dense corners
        rgba[..., :3] = ((rng.random((H, W, 1)) < rng.choice([0.05, 0.2])) * 255).astype(np.uint8)
    if y8:
        rgba = np.ascontiguousarray(rgba[..., 1])  # any byte plane will do as a Y8 frame
    flags = (orb.ORB_FLAG_INTENDED if intended else 0) | (orb.ORB_FLAG_NMS if nms else 0) | (orb.ORB_FLAG_STAGED if staged else 0) \
        | (orb.ORB_FLAG_INPUT_Y8 if y8 else 0)
    if y8:
        ref = oo.extract_y8(rgba, depth=depth, threshold=thr, max_features=cap, oob=oob, weight_bits=wbits)
    elif intended:
        ref = oo.extract_intended(rgba, depth=depth, threshold=thr, max_features=cap, arc=arc or 9, nms=nms)
    elif nms or (arc not in (0, 12)):
        ref = oo.extract_ex(rgba, depth=depth, threshold=thr, max_features=cap, arc=arc or 12, nms=nms)
    else:
        ref = oo.extract(rgba, depth=depth, threshold=thr, max_features=cap, oob=oob, weight_bits=wbits, **fpk)
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=cap, hierarchy_depth=depth, initial_threshold=thr, flags=flags, fast_arc=arc,
                        oob_policy=orb.OOB_POLICIES[oob], sampler_weight_bits=wbits, fp_contract=fp)
    with orb.OrbProgram(cfg) as prog:
        total, corners, desc = prog.extract(rgba)
        pipe = prog.pipeline()
    ok = total == ref["total"]
    exact_subset = intended or total <= cap  # the literal algorithm does not say which keypoints survive an overflow
    if ok and exact_subset:
        c, d = sort(corners, desc)
        rc, rd = oo.sort_keypoints(ref["corners"], ref["descriptors"])
        ok = len(c) == len(rc) and all(np.array_equal(c[k], rc[k]) for k in ("x", "y", "angle", "octave")) and np.array_equal(d, rd)
    elif ok:
        full = oo.extract_y8(rgba, depth=depth, threshold=thr, max_features=1 << 20, oob=oob, weight_bits=wbits) if y8 else \
            oo.extract(rgba, depth=depth, threshold=thr, max_features=1 << 20, oob=oob, weight_bits=wbits, **fpk) if not (nms or arc not in (0, 12)) else \
            oo.extract_ex(rgba, depth=depth, threshold=thr, max_features=1 << 20, arc=arc or 12, nms=nms)
        table = {(int(k["octave"]), int(k["y"]), int(k["x"])): (int(k["angle"]), dd.tobytes()) for k, dd in zip(full["corners"], full["descriptors"])}
        ok = len(corners) == cap and all(table.get((int(k["octave"]), int(k["y"]), int(k["x"]))) == (int(k["angle"]), dd.tobytes())
                                         for k, dd in zip(corners, desc))
    if not ok:
        bad += 1
        print("MISMATCH", dict(W=W, H=H, depth=depth, thr=thr, intended=intended, nms=nms, arc=arc, cap=cap, staged=staged, y8=y8, oob=oob, wbits=wbits, fp=fp,
                               syn=syn, seed=seed, pipe=pipe, total=total, ref_total=ref["total"]), flush=True)
    if case % 20 == 19:
        print("case %d, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("done: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
