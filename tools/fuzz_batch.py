"""Randomized parity sweep of the BATCHED entries on the GPU (tools/fuzz_parity.py covers the single-frame calls): random
frame sizes, depths, thresholds, modes and batch sizes (1..40: with and without the XCD swizzle, chunked host ingest
above 16 frames), read back three ways (per frame, orb_batch_read_all, orb_batch_pack + orb_batch_fetch) -- every frame
against the C oracle.  usage: python tools/fuzz_batch.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyslam_amd import orb
from oracle import orb_oracle as oo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def sort(c, d):
    o = np.lexsort((c["x"], c["y"], c["octave"]))
    return c[o], d[o]


def same(c, d, ref):
    c, d = sort(np.array(c), np.array(d))
    rc, rd = oo.sort_keypoints(ref["corners"], ref["descriptors"])
    return len(c) == len(rc) and all(np.array_equal(c[k], rc[k]) for k in ("x", "y", "angle", "octave")) and np.array_equal(d, rd)


bad = 0
frames_done = 0
t0 = time.time()
for case in range(n_cases):
    W = int(rng.integers(6, 120)) * 4 if rng.random() < 0.85 else int(rng.integers(25, 300))
    if rng.random() < 0.05:
        W = int(rng.integers(513, 600)) * 4  # 8-row bands
    if rng.random() < 0.15:
        W = int(rng.integers(160, 660)) * 4  # 640 .. 2636 columns: levels >= 1 whose bands hold 18 k pixels (1024-thread k_front<false>)
    H = int(rng.integers(24, 200)) if W <= 600 else int(rng.integers(24, 64))
    depth = int(rng.integers(1, 5))
    thr = float(np.float32(rng.choice([12, 20, 40]) / 255.0))
    mode = rng.choice(["literal", "literal", "y8", "arc", "intended"])
    nms = mode in ("arc", "intended") and bool(rng.random() < 0.5)
    arc = int(rng.choice([9, 10, 13])) if mode == "arc" else 0
    staged = bool(rng.random() < 0.15)
    cap = int(rng.choice([8192, 2048]))  # large enough: the literal algorithm does not define which records survive a cut
    n = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 24, 33, 40]))
    seeds = [int(s) for s in rng.integers(0, 1 << 30, n)]
    y8 = mode == "y8"
    fp = int(rng.integers(0, 16)) if mode == "literal" and rng.random() < 0.6 else 0  # OrbOptions::fp_contract (round 5)
    frames = [oo.synth_frame_y8(W, H, s) if y8 else oo.synth_frame(W, H, s) for s in seeds]
    flags = (orb.ORB_FLAG_INTENDED if mode == "intended" else 0) | (orb.ORB_FLAG_NMS if nms else 0) \
        | (orb.ORB_FLAG_STAGED if staged else 0) | (orb.ORB_FLAG_INPUT_Y8 if y8 else 0)
    refs = []
    for f in frames:
        if y8:
            refs.append(oo.extract_y8(f, depth=depth, threshold=thr, max_features=cap))
        elif mode == "intended":
            refs.append(oo.extract_intended(f, depth=depth, threshold=thr, max_features=cap, arc=9, nms=nms))
        elif mode == "arc":
            refs.append(oo.extract_ex(f, depth=depth, threshold=thr, max_features=cap, arc=arc, nms=nms))
        else:
            refs.append(oo.extract(f, depth=depth, threshold=thr, max_features=cap, contract=fp & 7, dot_order=fp >> 3))
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=cap, hierarchy_depth=depth, initial_threshold=thr, flags=flags,
                        fast_arc=arc, max_batch=n, fp_contract=fp)
    ok = True
    why = ""
    with orb.OrbProgram(cfg) as prog:
        pipe = prog.pipeline()
        prog.extract_batch_host(np.stack(frames))
        prog.batch_sync()
        counts = prog.batch_counts(n)
        if any(r["total"] > cap for r in refs):
            ok = all(int(counts[i]) == refs[i]["total"] for i in range(n))  # overflow: only the counters are defined
            why = "counts (overflow)"
        else:
            for i in range(n):
                if int(counts[i]) != refs[i]["total"]:
                    ok, why = False, "count of frame %d" % i
                    break
            way = int(rng.integers(0, 3))
            if ok and way == 0:
                for i in range(n):
                    c, d = prog.batch_read(i, int(counts[i]))
                    if not same(c, d, refs[i]):
                        ok, why = False, "batch_read frame %d" % i
                        break
            elif ok:
                total = int(sum(min(int(c), cap) for c in counts))
                hb = orb.HostBatch(n, max(total, 1))
                if way == 1:
                    prog.batch_read_all(n, out=hb)
                else:
                    prog.batch_pack(n)
                    prog.batch_fetch(0, hb)
                    prog.stream_sync()
                if not np.array_equal(hb.counts, counts) or int(hb.offsets[n]) != total:
                    ok, why = False, "packed counts/offsets (way %d)" % way
                for i in range(n):
                    if not ok:
                        break
                    c, d = hb.frame(i)
                    if not same(c, d, refs[i]):
                        ok, why = False, "packed frame %d (way %d)" % (i, way)
                hb.close()
    frames_done += n
    if not ok:
        bad += 1
        print("MISMATCH", why, dict(W=W, H=H, depth=depth, thr=thr, mode=str(mode), nms=nms, arc=arc, cap=cap, staged=staged, n=n,
                                    seeds=seeds[:4], pipe=pipe), flush=True)
    if case % 20 == 19:
        print("case %d (%d frames), %d mismatches, %.0f s" % (case + 1, frames_done, bad, time.time() - t0), flush=True)
print("done: %d cases, %d frames, %d mismatches" % (n_cases, frames_done, bad))
sys.exit(1 if bad else 0)
