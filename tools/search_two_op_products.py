#!/usr/bin/env python3
"""Which of the luminance's channel products can be had in TWO binary32 operations instead of three?

The non-contracted luminance (grayscale.wgsl:36 under CRD-1 / CRD-2; IM-1 with 0.299) needs, per channel, the DOUBLY rounded value
    T_w[b] = fl(fl(b / 255) * w)          b = 0..255
which the kernels compute in three operations (byte/255 exactly as fma(b, hi, fl(b * lo)), then the product with w).  This script looks
for constants that give the same 256 values in two operations, in three families:
    F1  fma(b, m, fl(b * e))              (the one tinyslam_amd/csrc/orb_kernels_front.h uses for w = 0.587f)
    F2  fl(fma(b, m1, c1) * m2)
    F3  fma(fl(b * m1), m2, c2)
and checks every hit with exact rational arithmetic.  Result (round 5): F1 has solutions for 0.587f only; none of the three families
has one for 0.229f, 0.114f or 0.299f (m within +-20000 ulps for F1, +-40 ulps and 49 offsets over ten scales for F2 / F3).

    python tools/search_two_op_products.py [quick]

No GPU, no oracle: NumPy and fractions only."""
import sys
from fractions import Fraction

import numpy as np

F = np.float32
B = np.arange(256, dtype=np.float32)
BD = B.astype(np.float64)
BH = (B / F(255.0)).astype(np.float32)


def rne(fr):
    """A rational to the nearest binary32, ties to even."""
    x = np.float32(float(fr))
    cands = [np.nextafter(x, F(-np.inf)), x, np.nextafter(x, F(np.inf))]
    d = [abs(Fraction(float(c)) - fr) for c in cands]
    best = min(d)
    win = [c for c, dd in zip(cands, d) if dd == best]
    return win[0] if len(win) == 1 else [c for c in win if (int(np.float32(c).view(np.uint32)) & 1) == 0][0]


def exact_f1(m, e, T):
    t = (B * e).astype(np.float32)
    return all(rne(Fraction(i) * Fraction(float(m)) + Fraction(float(t[i]))) == T[i] for i in range(256))


def ulps(x, n):
    bits = int(np.float32(x).view(np.uint32))
    return np.arange(bits - n, bits + n + 1).astype(np.uint32).view(np.float32)


def search_f1(w, span):
    """fma(b, m, fl(b * e)): for every m the interval of e that every byte's rounding interval allows, then candidates inside it."""
    T = (BH * w).astype(np.float32)
    Td = T.astype(np.float64)
    up, dn = np.nextafter(T, F(np.inf)).astype(np.float64), np.nextafter(T, F(-np.inf)).astype(np.float64)
    hi_b, lo_b = (Td + up) / 2, (Td + dn) / 2
    found = []
    for m in ulps(F(np.float64(w) / 255.0), span):
        lo = ((lo_b[1:] - BD[1:] * np.float64(m)) / BD[1:]).max()
        hi = ((hi_b[1:] - BD[1:] * np.float64(m)) / BD[1:]).min()
        if lo * (1 - np.sign(lo) * 2e-7) > hi * (1 + np.sign(hi) * 2e-7):
            continue
        for e in np.unique(np.linspace(lo, hi, 41).astype(np.float32)):
            t = (B * e).astype(np.float32)
            if np.array_equal((BD * np.float64(m) + t.astype(np.float64)).astype(np.float32), T) and exact_f1(m, e, T):
                found.append((m, e))
    return found


def search_f23(w, span, scales):
    T = (BH * w).astype(np.float32)
    found = []
    for s in scales:
        m1s, m2s = ulps(F(s / 255.0), span), ulps(F(np.float64(w) / s), span)
        c1s = (np.arange(-24, 25) * (s * 2.0 ** -27)).astype(np.float32)
        c2s = (np.arange(-24, 25) * (float(w) * 2.0 ** -27)).astype(np.float32)
        P2 = (BD[None, None, :] * m1s[:, None, None].astype(np.float64) + c1s[None, :, None].astype(np.float64)).astype(np.float32)
        P3 = (BD[None, :] * m1s[:, None].astype(np.float64)).astype(np.float32)
        for m2 in m2s:
            ok = ((P2.astype(np.float64) * np.float64(m2)).astype(np.float32) == T[None, None, :]).all(2)
            found += [("F2", s, m1s[i], c1s[j], m2) for i, j in zip(*np.nonzero(ok))]
            ok = ((P3[:, None, :].astype(np.float64) * np.float64(m2) + c2s[None, :, None].astype(np.float64)).astype(np.float32) == T[None, None, :]).all(2)
            found += [("F3", s, m1s[i], c2s[j], m2) for i, j in zip(*np.nonzero(ok))]
    return found


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    for name, w in (("0.587f (green)", F(0.587)), ("0.229f (red, literal)", F(0.229)), ("0.114f (blue)", F(0.114)), ("0.299f (red, BT.601)", F(0.299))):
        f1 = search_f1(w, 200 if quick else 20000)
        print("%-22s F1 fma(b, m, fl(b*e)): %d exact pairs%s" % (name, len(f1), "" if not f1 else
              "; e.g. m = %s, e = %s" % (float(f1[len(f1) // 2][0]).hex(), float(f1[len(f1) // 2][1]).hex())), flush=True)
        if not f1:
            f23 = search_f23(w, 8 if quick else 40, (1.0, 1.5) if quick else (1.0, 1.5, 1.25, 1.75, 3.0, 5.0, 0.75, 1.1, 1.3, 1.7))
            print("%-22s F2 / F3: %d" % ("", len(f23)), flush=True)


if __name__ == "__main__":
    main()
