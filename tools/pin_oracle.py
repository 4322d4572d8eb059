#!/usr/bin/env python3
"""Pins the CPU restatement to the reference (BASELINE.json configs[0]) -- the half of the job that can be done here.

    python tools/pin_oracle.py frame <out.rgba> [seed flags]
                                                  a 640x480 synthetic frame as raw RGBA8.  Default: configs[0]'s (seed 1, flags 7 =
                                                  gradient + blobs + wedges: tests/golden/g640x480_d2_config1.npz).  That frame is
                                                  smooth: it tells the out-of-level policies zero / clamp apart, but neither the
                                                  sampler weight precision nor clamp from umin -- dump `2 15` (with noise) as well.
    python tools/pin_oracle.py check <dump dir>   compares a dump of the reference itself (rust/dump_config0, run on a machine
                                                  with cargo and a Vulkan adapter) with oracle/orb_oracle.c under EVERY setting of
                                                  the implementation-defined switches (orc_impl_t: out-of-level policy x sampler
                                                  weight bits x the stages a shader compiler contracts x the order in which it
                                                  reduces dot() / matrix * vector x the rounding of the R16Float stores: 192
                                                  settings) and reports which ones the adapter may follow, or how far the nearest
                                                  one is and where the differences sit.

Exit codes of `check`:
    0   pinned: some setting reproduces counter, keypoints, angle codes and every descriptor bit, and the kernels carry it
        (include/tinyorb.h, OrbOptions: oob_policy, sampler_weight_bits, fp_contract) -- the line names the options to pass;
    5   pinned, but only by settings that exist in the restatement alone (f16_round = 1: R16Float stores that round toward zero);
    3   pinned up to the adapter's atan2: angle codes off by one milliradian whose descriptors, recomputed at the dumped angle,
        are the dump's (CRD-9: no switch can follow a driver's atan2);
    6   pinned up to the adapter's sin / cos (and possibly its atan2): every descriptor bit that differs at an agreeing angle
        belongs to a test one of whose rotated coordinates (brief.wgsl:50-57) lies within --sincos-tol of a non-zero integer, and
        truncating that coordinate to the other side gives the dump's bit;
    7   pinned on the keypoints with non-negative angles only: the adapter's `u32(angle * 1000.0)` (fast.wgsl:153) does not
        saturate a NEGATIVE angle to 0 (SPIR-V leaves that conversion undefined; SURVEY.md Q7 assumed what GPUs do) but wraps it
        modulo 2^32 or returns all ones -- a CPU adapter (lavapipe / llvmpipe on x86-64).  The codes themselves are checked
        against the restatement under that policy (orc_impl_t::neg_angle); the descriptors of those keypoints are rotated by
        an angle of millions of radians, whose cos / sin are the adapter's own, and are NOT compared.  The kernels saturate;
    1   not pinned.

TEST INFRASTRUCTURE (it imports oracle/): never part of the product.
"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orb_oracle  # noqa: E402

W, H, DEPTH, SEED, FLAGS, CAP = 640, 480, 2, 1, 7, 8192
THR = np.float32(20.0 / 255.0)
# What an adapter's own sin / cos may move a rotated coordinate by and still be a sin / cos matter: WGSL allows sin and cos an
# ABSOLUTE error of 2^-11 on [-pi, pi] (x |p| <= 26 per coordinate: 0.013 -- too wide to be evidence of anything), hardware
# special-function units deliver about 2^-21: 26 * 2^-21 = 1.2e-5, plus a few places of the products and the sum.
SINCOS_TOL = 2e-5

Setting = collections.namedtuple("Setting", "oob weight_bits contract dot_order f16_round")
DEFAULT = Setting("zero", 0, 0, 0, 0)
SETTINGS = [Setting(oob, wb, ct, do, rz) for rz in (0, 1) for ct in range(8) for do in (0, 1) for oob in ("zero", "clamp", "umin")
            for wb in (0, 8)]
_CT_NAMES = {0: "none", 1: "luminance", 2: "blur", 4: "rotation"}


def setting(oob="zero", weight_bits=0, contract=0, dot_order=0, f16_round=0):
    return Setting(oob, weight_bits, contract, dot_order, f16_round)


def describe(s):
    ct = "+".join(_CT_NAMES[b] for b in (1, 2, 4) if s.contract & b) or "none"
    return "oob=%s weight_bits=%d contract=%s dot_order=%s f16_round=%s" % (
        s.oob, s.weight_bits, ct, "last-first" if s.dot_order else "first-first", "rtz" if s.f16_round else "rte")


def options_for(s):
    """The OrbOptions (include/tinyorb.h) that make the kernels follow setting s; None when they cannot (f16_round)."""
    if s.f16_round:
        return None
    return "oob_policy=ORB_OOB_%s, sampler_weight_bits=%d, fp_contract=%d%s" % (
        s.oob.upper(), s.weight_bits, s.contract, " | ORB_FP_LAST_TERM_FIRST" if s.dot_order else "")


def dump_params(d):
    """(seed, flags) of the frame a dump was taken on (params.npy: W, H, depth, seed, flags, max_features)."""
    p = [int(v) for v in np.load(os.path.join(d, "params.npy"))]
    assert p[:3] == [W, H, DEPTH] and p[5] == CAP, "the dump harness runs 640x480, depth 2, max_features 8192"
    return p[3], p[4]


def load_dump(d):
    """(total, corners (n, 4) u32 sorted by (octave, y, x), descriptors (n, 8) u32 in the same order)."""
    total = int(np.load(os.path.join(d, "total.npy")))
    corners = np.load(os.path.join(d, "corners.npy")).astype(np.uint32).reshape(-1, 4)
    desc = np.load(os.path.join(d, "descriptors.npy")).astype(np.uint32).reshape(-1, 8)
    assert len(corners) == len(desc) == min(total, CAP)
    order = np.lexsort((corners[:, 0], corners[:, 1], corners[:, 3]))
    return total, corners[order], desc[order]


_CACHE = {}


def oracle_result(s=DEFAULT, seed=SEED, flags=FLAGS, planes=False, neg_angle="zero"):
    """(total, corners (n, 4) sorted, descriptors (n, 8)[, blur pyramid]) of the restatement under setting s (kept per process:
    `check` asks for 192 of them, the tests for the same ones again).  neg_angle: see neg_angle_policy()."""
    s = Setting(*s)
    key = (s, seed, flags, neg_angle)
    if key not in _CACHE:
        rgba = orb_oracle.synth_frame(W, H, seed, flags)
        ref = orb_oracle.extract(rgba, depth=DEPTH, threshold=THR, max_features=CAP, oob=s.oob, weight_bits=s.weight_bits, planes=True,
                                 contract=s.contract, dot_order=s.dot_order, f16_round=s.f16_round, neg_angle=neg_angle)
        c, d = orb_oracle.sort_keypoints(ref["corners"], ref["descriptors"])
        _CACHE[key] = (ref["total"], np.stack([c[k] for k in ("x", "y", "angle", "octave")], 1).astype(np.uint32), d.astype(np.uint32),
                       ref["blur"])
    out = _CACHE[key]
    return (out[0], out[1].copy(), out[2].copy(), out[3]) if planes else (out[0], out[1].copy(), out[2].copy())


def descriptors_at(blur, corners_xyao, s=DEFAULT):
    """The restatement's descriptors of the given (x, y, angle, octave) rows over its own blur pyramid: what brief.wgsl:20-68
    yields at an angle code that somebody else's atan2 produced."""
    s = Setting(*s)
    c = np.zeros(len(corners_xyao), dtype=orb_oracle.CORNER_DTYPE)
    for i, k in enumerate(("x", "y", "angle", "octave")):
        c[k] = corners_xyao[:, i]
    return orb_oracle.brief(blur, W, H, DEPTH, c, oob=s.oob, contract=s.contract, dot_order=s.dot_order)


def neg_angle_policy(corners):
    """What the dumping adapter's `u32(angle * 1000.0)` made of NEGATIVE angles, read off the dumped codes (atan2 lies in
    [-pi, pi]: a code above 3142 can only be a negative angle that was not saturated): "zero" (no such code: Q7, every GPU),
    "ones" (all of them 0xffffffff) or "wrap" (2^32 - m, m <= 3142).  None: codes that are none of these."""
    big = corners[:, 2][corners[:, 2] > 3142]
    if big.size == 0:
        return "zero"
    if (big == 0xFFFFFFFF).all():
        return "ones"
    if (big >= np.uint32(0x100000000 - 3142 - 1)).all():
        return "wrap"
    return None


_PATTERN = None


def pattern():
    global _PATTERN
    if _PATTERN is None:
        from oracle import orb_numpy
        _PATTERN = orb_numpy.PATTERN
    return _PATTERN


def _level_load(blur, lvl, x, y, oob):
    """textureLoad of one blur level under an out-of-level policy (the restatement's level_load_p), as a binary16 bit pattern."""
    dims, _ = orb_oracle.level_dims(W, H, DEPTH)
    w, h, off = dims[lvl]
    if x < 0 or y < 0 or x >= w or y >= h:
        if oob == "zero":
            return 0
        if oob == "clamp":
            x, y = min(max(x, 0), w - 1), min(max(y, 0), h - 1)
        else:
            x = w - 1 if (x < 0 or x >= w) else x
            y = h - 1 if (y < 0 or y >= h) else y
    return int(blur[off + y * w + x])


def sincos_explains(blur, kp, test, want_bit, s, tol=SINCOS_TOL):
    """Can an adapter's own sin / cos explain bit `test` of keypoint kp = (x, y, angle, octave) being want_bit?  The test's four
    rotated coordinates (brief.wgsl:50-57, before vec2i() truncates them) are taken from the restatement at the keypoint's angle;
    those within `tol` of a NON-ZERO integer (the only places where truncation toward zero changes) may fall on either side.
    Returns (explained, smallest distance to such an integer among the coordinates that had to move, or None)."""
    x, y, code, octv = (int(v) for v in kp)
    if code == 0:  # cos 0 = 1 and sin 0 = 0 in every implementation: R = I, nothing to round
        return False, None
    ax, ay, bx, by = (int(v) for v in pattern()[test])
    ra = orb_oracle.brief_rotate(code, ax, ay, s.contract, s.dot_order)
    rb = orb_oracle.brief_rotate(code, bx, by, s.contract, s.dot_order)
    coords = [float(ra[0]), float(ra[1]), float(rb[0]), float(rb[1])]
    alts = []
    for r in coords:
        n = round(r)
        t = int(r)  # trunc
        opts = [(t, 0.0)]
        if n != 0 and abs(r - n) <= tol:
            # the other side of the integer n: a value just below |n| truncates to n -+ 1's side
            other = n if t != n else (n - 1 if n > 0 else n + 1)
            opts.append((other, abs(r - n)))
        alts.append(opts)
    best = None
    for i0 in alts[0]:
        for i1 in alts[1]:
            for i2 in alts[2]:
                for i3 in alts[3]:
                    va = _level_load(blur, octv, x + i0[0], y + i1[0], s.oob)
                    vb = _level_load(blur, octv, x + i2[0], y + i3[0], s.oob)
                    if int(va > vb) == want_bit:  # non-negative binary16: the patterns order like the values
                        need = max([o[1] for o in (i0, i1, i2, i3)])
                        if best is None or need < best:
                            best = need
    return (best is not None), best


def compare(dump, ref, blur=None, s=DEFAULT, sincos_tol=SINCOS_TOL):
    """Differences between a dump and one oracle setting, by kind.  With the setting's blur pyramid: the keypoints whose angle
    code differs by one milliradian are described again at the DUMP's angle -- if those descriptors are the dump's, the only
    thing the adapter does differently there is atan2 (CRD-9: a driver's atan2 may round the other way next to an integer
    milliradian; WGSL allows it thousands of ulp), and the setting is `exact_up_to_atan2`.  The descriptor bits that still differ
    (at agreeing angles, and at the dumped angle where it is off by one) are then put to sincos_explains(): if every one of them
    has a rotated coordinate within sincos_tol of an integer whose other side gives the dump's bit, the setting is
    `exact_up_to_sincos`."""
    s = Setting(*s)
    t0, c0, d0 = dump
    t1, c1, d1 = ref[:3]

    def keys(c):  # (octave, y, x) as one integer; the rows are sorted by it and unique
        return (c[:, 3].astype(np.int64) << 40) | (c[:, 1].astype(np.int64) << 20) | c[:, 0].astype(np.int64)

    def unkey(k):
        return [(int(v >> 40), int((v >> 20) & 0xfffff), int(v & 0xfffff)) for v in k]

    k0, k1 = keys(c0), keys(c1)
    _, i0, i1 = np.intersect1d(k0, k1, assume_unique=True, return_indices=True)
    only_dump, only_oracle = unkey(np.setdiff1d(k0, k1, assume_unique=True)), unkey(np.setdiff1d(k1, k0, assume_unique=True))
    da = np.abs(c0[i0, 2].astype(np.int64) - c1[i1, 2].astype(np.int64))
    xor = d0[i0] ^ d1[i1]
    unsat = (c0[i0, 2] > 3142) | (c1[i1, 2] > 3142)  # a negative angle that u32() did not saturate (neg_angle_policy): the code is
    xor[unsat] = 0                                   # compared, the descriptor -- rotated by millions of radians -- is not
    nb = np.unpackbits(xor.view(np.uint8), axis=1).sum(1).astype(np.int64) if len(i0) else np.zeros(0, dtype=np.int64)
    angle_off1, angle_other = int((da == 1).sum()), int((da > 1).sum())
    bits, kp_with_bits, bits_same_angle = int(nb.sum()), int((nb > 0).sum()), int(nb[da == 0].sum())
    off1 = [int(i) for i in i0[(da == 1) & ~unsat]]  # dump rows whose angle code is the oracle's +- 1
    left = [(int(i0[j]), xor[j]) for j in np.flatnonzero((da == 0) & (nb > 0))]  # bits differ although the angle codes agree
    bits_at_dump_angle = None
    if blur is not None and off1:
        again = descriptors_at(blur, c0[off1], s)
        xs = again ^ d0[off1]
        bits_at_dump_angle = int(np.unpackbits(xs.view(np.uint8)).sum())
        left += [(i, x) for i, x in zip(off1, xs) if x.any()]
    same_sets = t0 == t1 and not only_dump and not only_oracle
    # sin / cos: only worth asking when everything else fits and a handful of bits are left
    n_left = sum(int(np.unpackbits(x.view(np.uint8)).sum()) for _, x in left)
    sincos_ok, sincos_need, sincos_unexplained = None, None, None
    if blur is not None and same_sets and angle_other == 0 and 0 < n_left <= 4096:
        sincos_ok, sincos_need, sincos_unexplained = True, 0.0, 0
        for i, x in left:
            for word in range(8):
                m = int(x[word])
                while m:
                    bit = (m & -m).bit_length() - 1
                    m &= m - 1
                    want = (int(d0[i, word]) >> bit) & 1
                    ok, need = sincos_explains(blur, c0[i], 32 * word + bit, want, s, sincos_tol)
                    if ok:
                        sincos_need = max(sincos_need, need)
                    else:
                        sincos_ok = False
                        sincos_unexplained += 1
    return {"descriptors_not_compared": int(unsat.sum()), "total_dump": t0, "total_oracle": t1, "only_in_dump": only_dump, "only_in_oracle": only_oracle,
            "angle_off_by_1": int(angle_off1), "angle_off_by_more": int(angle_other), "descriptor_bits": int(bits),
            "keypoints_with_bit_differences": int(kp_with_bits),
            "descriptor_bits_where_angles_agree": int(bits_same_angle),
            "descriptor_bits_at_the_dumped_angle": bits_at_dump_angle,  # over the keypoints whose angle is off by one (None: not computed)
            "exact": same_sets and angle_off1 == 0 and angle_other == 0 and bits == 0,
            "exact_up_to_atan2": same_sets and angle_other == 0 and bits_same_angle == 0
                                 and (angle_off1 == 0 or bits_at_dump_angle == 0),
            "exact_up_to_sincos": bool(sincos_ok),   # ... and up to atan2 where angle codes are off by one
            "sincos_bits": n_left if sincos_ok is not None else None, "sincos_unexplained": sincos_unexplained,
            "sincos_max_distance": sincos_need}


def check(d, settings=None, sincos_tol=SINCOS_TOL):
    """(dump, {Setting: differences}, [settings that reproduce the dump exactly]).  The restatement runs under the negative-angle
    policy the dump itself shows (neg_angle_policy; "zero" when its codes fit none)."""
    dump = load_dump(d)
    seed, flags = dump_params(d)
    neg = neg_angle_policy(dump[1]) or "zero"
    results = {}
    for s in (settings or SETTINGS):
        ref = oracle_result(s, seed, flags, planes=True, neg_angle=neg)
        results[s] = compare(dump, ref, blur=ref[3], s=s, sincos_tol=sincos_tol)
    exact = [s for s in results if results[s]["exact"]]
    return dump, results, exact


def _axes(settings):
    """Which values of each switch occur among `settings` (to say what a frame does not tell apart)."""
    return {f: sorted({getattr(s, f) for s in settings}, key=str) for f in Setting._fields}


def main(argv):
    tol = SINCOS_TOL
    if "--sincos-tol" in argv:
        i = argv.index("--sincos-tol")
        tol = float(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    if len(argv) in (3, 5) and argv[1] == "frame":
        seed, flags = (int(argv[3]), int(argv[4])) if len(argv) == 5 else (SEED, FLAGS)
        rgba = orb_oracle.synth_frame(W, H, seed, flags)
        rgba.tofile(argv[2])
        print("wrote %d bytes (%dx%d RGBA8, seed %d, flags %d) to %s -- pass `%d %d` to dump_config0 as well"
              % (rgba.size, W, H, seed, flags, argv[2], seed, flags))
        return 0
    if len(argv) == 3 and argv[1] == "check":
        dump, results, exact = check(argv[2], sincos_tol=tol)
        neg = neg_angle_policy(dump[1])
        if neg != "zero":
            n_big = int((dump[1][:, 2] > 3142).sum())
            print("NEGATIVE ANGLES ARE NOT SATURATED by this adapter: %d of %d dumped angle codes exceed 3142 (%s).  SURVEY.md Q7 / CRD-9 "
                  "assumed u32() of a negative float gives 0, as GPUs do; SPIR-V leaves it undefined.  Those codes are compared with "
                  "the restatement under neg_angle=%s; the descriptors of those keypoints are not compared.\n"
                  % (n_big, len(dump[1]), {"wrap": "2^32 - m: the conversion wraps", "ones": "all 0xffffffff",
                                           None: "neither 2^32 - m nor all ones: not a policy the restatement knows"}[neg], neg or "zero"))
        # one line per setting would be 192 lines: the default arithmetic per (oob, weight bits), then whatever is exact or near
        for s, r in results.items():
            if (s.contract, s.dot_order, s.f16_round) != (0, 0, 0) and not (r["exact"] or r["exact_up_to_atan2"] or r["exact_up_to_sincos"]):
                continue
            print("%s: %s  total %d vs %d, keypoints only in dump %d / only in oracle %d, angle codes off by one %d / by more %d, "
                  "descriptor bits %d in %d keypoints"
                  % (describe(s), "EXACT" if r["exact"] else "differs", r["total_dump"], r["total_oracle"], len(r["only_in_dump"]),
                     len(r["only_in_oracle"]), r["angle_off_by_1"], r["angle_off_by_more"], r["descriptor_bits"],
                     r["keypoints_with_bit_differences"]))
        if exact:
            carried = [s for s in exact if not s.f16_round]
            ax = _axes(carried or exact)
            print("\nPINNED on this frame: %d of %d settings of the implementation-defined switches reproduce the dump bit for bit." % (len(exact), len(results)))
            print("What this frame leaves open among them: " + "; ".join("%s in %s" % (f, v) for f, v in ax.items() if len(v) > 1) or "nothing")
            if neg != "zero":
                print("ON THE KEYPOINTS WITH NON-NEGATIVE ANGLES ONLY (%d of %d): the other descriptors were rotated by an angle of millions "
                      "of radians and were not compared.  The kernels saturate negative angles (what GPUs do); to reproduce THIS adapter "
                      "a kernel switch for the conversion and that adapter's cos / sin at 4e6 rad would be owed -- more likely the "
                      "adapter is a CPU rasteriser (lavapipe) and not the one to pin to.  Options for the rest: %s"
                      % (len(dump[1]) - results[exact[0]]["descriptors_not_compared"], len(dump[1]), options_for((carried or exact)[0])))
                return 7
            if not carried:
                print("ONLY with R16Float stores that round toward zero (f16_round = 1): the restatement follows, the kernels round to "
                      "nearest even (v_cvt_f16_f32) -- a kernel switch is owed.  Settings: %s" % [describe(s) for s in exact[:4]])
                return 5
            if DEFAULT in exact:
                print("The defaults (OrbOptions zero-initialised) are among them.")
            else:
                print("The defaults are NOT among them.  Pass: %s" % options_for(carried[0]))
            if len(carried) > 1:
                print("Several settings agree on this frame -- it does not tell them apart; dump the noisy frame as well "
                      "(`frame <out> 2 15`) and intersect.")
            return 0
        near = [s for s in results if results[s]["exact_up_to_atan2"]]
        if near:
            s = (([t for t in near if t == DEFAULT] or [t for t in near if not t.f16_round]) or near)[0]
            r = results[s]
            print("\nPINNED UP TO atan2 (CRD-9) on this frame under %d settings, e.g. %s: counter, keypoint set and every "
                  "descriptor at an agreeing angle are identical; %d angle codes differ by ONE milliradian, and at the dumped angle the "
                  "restatement's descriptors of those keypoints are the dump's, bit for bit.  The adapter's atan2 rounds the other way "
                  "next to an integer milliradian there -- no switch can follow a driver's atan2; compare angles with a tolerance of "
                  "one code and descriptors at the reference's angle.  Options: %s" % (len(near), describe(s), r["angle_off_by_1"], options_for(s)))
            return 3
        near = [s for s in results if results[s]["exact_up_to_sincos"]]
        if near:
            s = min(near, key=lambda t: (t.f16_round, t != DEFAULT, results[t]["sincos_bits"]))
            r = results[s]
            print("\nPINNED UP TO sin / cos (brief.wgsl:36-37) on this frame under %d settings, e.g. %s: counter and keypoint set are "
                  "identical, angle codes agree%s, and each of the %d descriptor bits that differ belongs to a test with a rotated "
                  "coordinate within %.2g of a non-zero integer (the largest distance needed: %.3g) whose truncation to the other side "
                  "gives the dump's bit.  WGSL allows an adapter's sin and cos an absolute error of 2^-11; no switch can follow them -- "
                  "compare descriptors with those bits masked.  Options: %s"
                  % (len(near), describe(s), " up to one milliradian (atan2, CRD-9)" if r["angle_off_by_1"] else "", r["sincos_bits"], tol,
                     r["sincos_max_distance"], options_for(s)))
            return 6
        best = min(results, key=lambda s: (len(results[s]["only_in_dump"]) + len(results[s]["only_in_oracle"]),
                                           results[s]["descriptor_bits"] + results[s]["angle_off_by_1"], s.f16_round, s != DEFAULT))
        r = results[best]
        print("\nNOT PINNED.  Nearest: %s.  Where to look:" % describe(best))
        if r["only_in_dump"] or r["only_in_oracle"]:
            print("  keypoints that differ (octave, y, x): dump-only %s oracle-only %s -- at octaves >= 1 within 3 px of the level's "
                  "right/bottom edge this is the out-of-level policy (CRD-6)" % (r["only_in_dump"][:8], r["only_in_oracle"][:8]))
        if r["angle_off_by_1"]:
            print("  %d angle codes differ by one milliradian: the adapter's atan2 against the canonical one (CRD-9)" % r["angle_off_by_1"])
        if r["descriptor_bits"]:
            print("  %d descriptor bits in %d keypoints: sampler arithmetic in the blur's varying columns (CRD-5), or samples that "
                  "leave the level (CRD-6)%s" % (r["descriptor_bits"], r["keypoints_with_bit_differences"],
                                               "; %d of them have no rotated coordinate near an integer (not sin / cos)" % r["sincos_unexplained"]
                                               if r["sincos_unexplained"] else ""))
        return 1
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv))
