"""Randomized check of the three matchers (orb_match_consecutive on the matrix cores -- fp4 and int8 -- and on the vector unit, every
pair of every case against the NumPy brute force): random frame sizes, thresholds, capacities and batch sizes, so that the
counts fall on and off every tile boundary (16 candidates, 32 queries per wave, 256 per workgroup, 64-candidate chunks).
usage: python tools/fuzz_match.py [n_cases] [seed]      TEST INFRASTRUCTURE (imports oracle/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
from oracle import orb_oracle as oo, orb_numpy

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(n_cases):
    W, H = int(rng.integers(16, 120)) * 4, int(rng.integers(40, 300))
    n = int(rng.integers(2, 7))
    cap = int(rng.choice([17, 64, 255, 256, 257, 600, 1000, 4096, 8192, 16128, 16383]))
    thr = float(np.float32(rng.choice([8, 12, 20, 40]) / 255.0))
    flags = int(rng.choice([15, 7, 5, 13]))
    frames = np.stack([oo.synth_frame(W, H, int(rng.integers(1, 10 ** 6)), flags) for _ in range(n)])
    if rng.random() < 0.3:
        frames[int(rng.integers(0, n))] = 0  # an empty frame somewhere
    ok = True
    for form in ({}, {"TINYORB_MATCH_I8": "1"}, {"TINYORB_MATCH_VALU": "1"}):  # fp4 (default), int8, vector unit -- each against the
        for k in ("TINYORB_MATCH_I8", "TINYORB_MATCH_VALU"):                     # brute force on its OWN program's descriptors: the order of
            os.environ.pop(k, None)                                                # a frame's records differs from one program to the next
        os.environ.update(form)
        with orb.OrbProgram(orb.OrbConfig(orb.Extent3d(W, H), max_features=cap, hierarchy_depth=2, initial_threshold=thr, max_batch=n)).init() as prog:
            prog.extract_batch_host(frames)
            counts = np.minimum(prog.batch_counts(n), cap)
            prog.match_consecutive(n)
            desc = [prog.batch_read(f, int(counts[f]))[1] for f in range(n)]
            for f in range(n - 1):
                got = prog.match_read(f, int(counts[f]))
                idx, dist, second = orb_numpy.match(desc[f], desc[f + 1])
                ok = ok and np.array_equal(got["index"], idx) and np.array_equal(got["distance"], dist) and np.array_equal(got["second"], second)
    if not ok:
        bad += 1
        print("MISMATCH case", case, W, H, n, cap, thr, flags, [int(c) for c in counts])
print("done: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
