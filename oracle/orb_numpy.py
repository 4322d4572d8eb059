"""Second, independently written restatement of the tinyslam ORB front-end, in NumPy.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (same caveat as oracle/orb_oracle.h: the reference
has no tests or golden vectors and cannot run here).  Its only job is to cross-check
oracle/orb_oracle.c so that one misreading of the shaders does not become both the oracle and
the implementation.  Written array-at-a-time from the WGSL text, sharing
no code with the C restatement; binary32 arithmetic is NumPy float32 (one rounding per ufunc).

Reference text followed: src/shaders/grayscale.wgsl:12-38, blit.wgsl:17-36,
gaussian_blur_x.wgsl:14-60, fast.wgsl:25-159, brief.wgsl:20-68 (+ table 70-327), and the stage
order of src/orb.rs:469-557.
"""
import hashlib
import os
import re

import numpy as np

F = np.float32

RING4 = [(3, 0), (-3, 0), (0, 3), (0, -3)]
RING16 = [(-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3), (1, -3), (2, -2), (3, -1),
          (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1)]
BLUR_OFF = [F(-2.2273038885157046), F(-0.4391873198428642), F(1.3243948342247673), F(3.0)]
BLUR_WGT = [F(0.13748623236806098), F(0.5037756553768409), F(0.32748695702046415), F(0.031251155234634016)]


def _pattern():
    """BRIEF pattern parsed from the generated header (data, pinned by SHA-256 in SURVEY.md 8a)."""
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "orb_pattern.h")).read()
    body = text[text.index("{") + 1:text.rindex("}")]
    vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int8)
    assert vals.size == 1024
    assert hashlib.sha256(vals.tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    return vals.reshape(256, 4).astype(np.int32)


PATTERN = _pattern()


def to_f16_bits(a, rtz=0):
    """binary32 -> binary16 bit patterns, round to nearest even (CRD-3); rtz: toward zero (the `f16_round` switch: a store to an
    R16Float target may round either way under Vulkan) -- the nearest-even result stepped back where it overshot."""
    a = np.asarray(a, dtype=np.float32)
    h = a.astype(np.float16)
    if rtz:
        over = np.abs(h.astype(np.float32)) > np.abs(a)  # finite values only on this path
        h = np.where(over, np.nextafter(h, np.float16(0)), h).astype(np.float16)
    return h.view(np.uint16)


def fma32(a, b, c):
    """fl32(a * b + c) with ONE rounding, for binary32 arrays -- NumPy has no fused multiply-add.  a * b is exact in binary64
    (48 significant bits); the sum with c is formed there with its rounding error (TwoSum), rounded TO ODD (of the two
    binary64 neighbours of the exact sum the one with an odd significand), and that is rounded to binary32: with 29 spare bits
    the second rounding cannot see a tie that is not one.  Finite operands whose result is a normal binary32 (all this path has)."""
    p = np.asarray(a, dtype=np.float32).astype(np.float64) * np.asarray(b, dtype=np.float32).astype(np.float64)
    c = np.broadcast_to(np.asarray(c, dtype=np.float32).astype(np.float64), p.shape)
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)  # exact: p + c = s + err
    even = (s.view(np.int64) & 1) == 0
    toward = np.where(err > 0, np.inf, -np.inf)
    s = np.where((err != 0) & even, np.nextafter(s, toward), s)
    return s.astype(np.float32)


def from_f16_bits(a):
    return np.asarray(a, dtype=np.uint16).view(np.float16).astype(np.float32)


def level_sizes(W, H, depth):
    return [(max(1, W >> m), max(1, H >> m)) for m in range(depth)]


# ------------------------------------------------------------------ stages
def luminance(r, g, b, contract=0, dot_order=0):
    """dot(color, vec4(0.229, 0.587, 0.114, 0.0)) (grayscale.wgsl:36) in the four forms a shader compiler can give it: products and
    sums rounded one by one or as one product and a chain of fmas (`contract`), reduced from the first component up or from the
    last one down (`dot_order`; the alpha term is +0 either way)."""
    wr, wg, wb = F(0.229), F(0.587), F(0.114)
    first, last = ((r, wr), (b, wb)) if not dot_order else ((b, wb), (r, wr))
    if contract:
        return fma32(last[0], last[1], fma32(g, wg, first[0] * first[1]))
    return (first[0] * first[1] + g * wg) + last[0] * last[1]


def grayscale(rgba, contract=0, dot_order=0, f16_round=0):
    """grayscale.wgsl: luminance of the vertically mirrored texel, stored as R16Float."""
    img = np.asarray(rgba, dtype=np.uint8)[::-1, :, :]
    chan = img.astype(np.float32) / F(255.0)
    return to_f16_bits(luminance(chan[..., 0], chan[..., 1], chan[..., 2], contract, dot_order), f16_round)


def grayscale_y8(y8, f16_round=0):
    """Y8 input variant (not in the reference's code): the Y sample of the vertically mirrored texel as R16Float."""
    img = np.asarray(y8, dtype=np.uint8)[::-1, :]
    return to_f16_bits(img.astype(np.float32) / F(255.0), f16_round)


def _weight(frac, wbits):
    """A bilinear sampler's weight held in `wbits` fractional bits, halves up (0: the exact binary32 fraction, CRD-5)."""
    frac = np.asarray(frac, dtype=np.float32)
    if not wbits:
        return frac
    scale = F(2 ** wbits)
    return (np.floor(frac * scale + F(0.5)) / scale).astype(np.float32)


def _lerp_axis_coords(n_dst, n_src, wbits=0):
    s = (np.arange(n_dst, dtype=np.float32) + F(0.5)) * (F(n_src) / F(n_dst)) - F(0.5)
    s0 = np.floor(s)
    frac = _weight((s - s0).astype(np.float32), wbits)
    i0 = np.clip(s0.astype(np.int64), 0, n_src - 1)
    i1 = np.clip(s0.astype(np.int64) + 1, 0, n_src - 1)
    return i0, i1, frac


def mip(src_bits, wbits=0, f16_round=0):
    """blit.wgsl: bilinear sample of the previous level at each target texel centre."""
    src = from_f16_bits(src_bits)
    hs, ws = src.shape
    wd, hd = max(1, ws >> 1), max(1, hs >> 1)
    if ws == 2 * wd and hs == 2 * hd:
        top = src[0::2, 0::2] + src[0::2, 1::2]
        bot = src[1::2, 0::2] + src[1::2, 1::2]
        return to_f16_bits((top + bot) * F(0.25), f16_round)
    x0, x1, fx = _lerp_axis_coords(wd, ws, wbits)
    y0, y1, fy = _lerp_axis_coords(hd, hs, wbits)
    a, b = src[np.ix_(y0, x0)], src[np.ix_(y0, x1)]
    c, d = src[np.ix_(y1, x0)], src[np.ix_(y1, x1)]
    top = a + fx[None, :] * (b - a)
    bot = c + fx[None, :] * (d - c)
    return to_f16_bits(top + fy[:, None] * (bot - top), f16_round)


def blur_pass(src_bits, wbits=0, contract=0, f16_round=0):
    """gaussian_blur_x.wgsl used for BOTH passes (orb.rs:399-402); offsets in UV units; flipped v."""
    src = from_f16_bits(src_bits)[::-1, :]
    h, w = src.shape
    fw = F(w)
    u = (np.arange(w, dtype=np.float32) + F(0.5)) / fw
    acc = np.zeros((h, w), dtype=np.float32)
    for off, wgt in zip(BLUR_OFF, BLUR_WGT):
        coord = (u + off) * fw - F(0.5)
        c0 = np.floor(coord)
        frac = _weight((coord - c0).astype(np.float32), wbits)
        i0 = np.clip(c0.astype(np.int64), 0, w - 1)
        i1 = np.clip(c0.astype(np.int64) + 1, 0, w - 1)
        t0, t1 = src[:, i0], src[:, i1]
        sample = t0 + frac[None, :] * (t1 - t0)  # the sampler's filter: not shader arithmetic, never contracted
        acc = fma32(sample, wgt, acc) if contract else acc + sample * wgt  # `result += sample * weight` (gaussian_blur_x.wgsl:58)
    return to_f16_bits(acc, f16_round)


def _streak12(mask):
    """fast.wgsl:51-60 on uint32 arrays."""
    def rot(v, k):
        return (v >> np.uint32(k)) | ((v << np.uint32(16 - k)) & np.uint32(0xFFFF))
    o6 = mask & rot(mask, 6)
    o3 = o6 & rot(o6, 3)
    return o3 & rot(o3, 2) & rot(o3, 1)


def atan2f(y, x):
    """CRD-9 canonical atan2 (same definition as orc_atan2f, re-derived here on arrays)."""
    y = np.asarray(y, dtype=np.float32)
    x = np.asarray(x, dtype=np.float32)
    ax, ay = np.abs(x), np.abs(y)
    mx = np.maximum(ax, ay)
    mn = np.minimum(ax, ay)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(mx > 0, mn / np.where(mx > 0, mx, F(1)), F(0)).astype(np.float32)
    big = a > F(0.41421356)
    t = np.where(big, (a - F(1.0)) / (a + F(1.0)), a).astype(np.float32)
    base = np.where(big, F(0.78539816), F(0.0)).astype(np.float32)
    z = t * t
    p = np.full_like(z, F(8.05374449538e-2))
    p = p * z - F(1.38776856032e-1)
    p = p * z + F(1.99777106478e-1)
    p = p * z - F(3.33329491539e-1)
    r = (p * z) * t + t
    r = base + r
    r = np.where(ay > ax, F(1.57079632679) - r, r)
    r = np.where(x < 0, F(3.14159265) - r, r)
    r = np.where(y < 0, -r, r)
    r = np.where((ax == 0) & (ay == 0), F(0), r)
    return r.astype(np.float32)


def _load(level, xs, ys, oob="zero"):
    """textureLoad.  Outside the level: 0 ("zero", CRD-6), each coordinate clamped into the level ("clamp"), or naga's
    Restrict policy, min(unsigned(coordinate), size - 1) -- a negative coordinate wraps to a huge unsigned one and lands
    on the LAST column / row ("umin")."""
    h, w = level.shape
    xs_b, ys_b = np.broadcast_arrays(np.asarray(xs, dtype=np.int64), np.asarray(ys, dtype=np.int64))
    if oob == "zero":
        ok = (xs_b >= 0) & (ys_b >= 0) & (xs_b < w) & (ys_b < h)
        out = np.zeros(xs_b.shape, dtype=np.float32)
        out[ok] = level[ys_b[ok], xs_b[ok]]
        return out
    if oob == "clamp":
        return level[np.clip(ys_b, 0, h - 1), np.clip(xs_b, 0, w - 1)].astype(np.float32)
    assert oob == "umin"
    ux = np.minimum(xs_b.astype(np.int32).view(np.uint32), np.uint32(w - 1)).astype(np.int64)
    uy = np.minimum(ys_b.astype(np.int32).view(np.uint32), np.uint32(h - 1)).astype(np.int64)
    return level[uy, ux].astype(np.float32)


def angle_code(cy, ang, neg_angle=0):
    """fast.wgsl:153 `u32(angle * 1000.0)`.  A negative angle: 0 (Q7: GPUs saturate; the default), or -- the conversion is undefined in
    SPIR-V -- the low 32 bits of the truncated value ("wrap", 1: x86-64 without AVX-512) or all ones ("ones", 2)."""
    neg_angle = {"zero": 0, "wrap": 1, "ones": 2}.get(neg_angle, neg_angle)
    t = np.trunc(ang.astype(np.float32) * F(1000.0)).astype(np.int64)
    neg = (cy < 0) | (ang < 0)
    if neg_angle == 1:
        return (t & 0xFFFFFFFF).astype(np.uint32)
    if neg_angle == 2:
        return np.where(neg & (t < 0), 0xFFFFFFFF, np.where(neg, 0, t)).astype(np.uint32)
    return np.where(neg, 0, t).astype(np.uint32)


def fast(gray_levels_bits, threshold, oob="zero", neg_angle=0):
    """fast.wgsl compute_fast over all octaves; returns (x, y, angle, octave) rows in raster order."""
    thr = F(threshold)
    H0, W0 = gray_levels_bits[0].shape
    lim_x = (W0 - 16) & 0xFFFFFFFF
    lim_y = (H0 - 16) & 0xFFFFFFFF
    rows = []
    width, height = W0, H0
    for octv, bits in enumerate(gray_levels_bits):
        lvl = from_f16_bits(bits)
        gw, gh = (width + 7) // 8 * 8, (height + 7) // 8 * 8
        width //= 2
        height //= 2
        if gw == 0 or gh == 0:
            continue
        gx, gy = np.meshgrid(np.arange(gw, dtype=np.int64), np.arange(gh, dtype=np.int64))
        guard = (gx > 16) & (gy > 16) & (gx < lim_x) & (gy < lim_y)
        if not guard.any():
            continue
        gx, gy = gx[guard], gy[guard]
        c = _load(lvl, gx, gy, oob)
        over = np.zeros(gx.shape, dtype=np.int32)
        under = np.zeros(gx.shape, dtype=np.int32)
        for dx, dy in RING4:
            diff = _load(lvl, gx + dx, gy + dy, oob) - c
            over += diff > thr
            under += (~(diff > thr)) & (diff < -thr)
        cand = (over >= 3) | (under >= 3)
        gx, gy, c = gx[cand], gy[cand], c[cand]
        m_over = np.zeros(gx.shape, dtype=np.uint32)
        m_under = np.zeros(gx.shape, dtype=np.uint32)
        cx = np.zeros(gx.shape, dtype=np.float32)
        cy = np.zeros(gx.shape, dtype=np.float32)
        for i, (dx, dy) in enumerate(RING16):
            v = _load(lvl, gx + dx, gy + dy, oob)
            diff = v - c
            cx = cx + v * F(dx)
            cy = cy + v * F(dy)
            is_o = diff > thr
            is_u = (~is_o) & (diff < -thr)
            m_over |= (is_o.astype(np.uint32) << np.uint32(i))
            m_under |= (is_u.astype(np.uint32) << np.uint32(i))
        corner = (_streak12(m_over) | _streak12(m_under)) > 0
        gx, gy, cx, cy = gx[corner], gy[corner], cx[corner], cy[corner]
        ang = atan2f(cy, cx)
        code = angle_code(cy, ang, neg_angle)
        for k in range(gx.size):
            rows.append((int(gx[k]), int(gy[k]), int(code[k]), octv))
    return np.array(rows, dtype=np.uint32).reshape(-1, 4)


def rotate(ct, st, x, y, contract=0, dot_order=0):
    """mat2x2f(ct, -st, st, ct) * vec2(x, y) (brief.wgsl:38-54, column-major): x * column 0 + y * column 1.  Unfused, or with one of
    the two products fused into the sum: the second term onto the first product (dot_order 0) or the first onto the second (1)."""
    if not contract:
        return ct * x + st * y, (-st) * x + ct * y
    if not dot_order:
        return fma32(st, y, ct * x), fma32(ct, y, (-st) * x)
    return fma32(ct, x, st * y), fma32(-st, x, ct * y)


def brief(blur_levels_bits, corners, oob="zero", contract=0, dot_order=0):
    """brief.wgsl: rotated BRIEF-256; returns uint32 (n, 8)."""
    corners = np.asarray(corners, dtype=np.uint32).reshape(-1, 4)
    n = corners.shape[0]
    out = np.zeros((n, 8), dtype=np.uint32)
    levels = [from_f16_bits(b) for b in blur_levels_bits]
    theta = corners[:, 2].astype(np.float32) / F(1000.0)
    ct = np.cos(theta.astype(np.float64)).astype(np.float32)
    st = np.sin(theta.astype(np.float64)).astype(np.float32)
    px = corners[:, 0].astype(np.int64)
    py = corners[:, 1].astype(np.int64)
    for j in range(256):
        ax, ay, bx, by = (F(v) for v in PATTERN[j])
        rax, ray = rotate(ct, st, ax, ay, contract, dot_order)
        rbx, rby = rotate(ct, st, bx, by, contract, dot_order)
        tax, tay = np.trunc(rax).astype(np.int64) + px, np.trunc(ray).astype(np.int64) + py
        tbx, tby = np.trunc(rbx).astype(np.int64) + px, np.trunc(rby).astype(np.int64) + py
        va = np.zeros(n, dtype=np.float32)
        vb = np.zeros(n, dtype=np.float32)
        for octv, lvl in enumerate(levels):
            sel = corners[:, 3] == octv
            if sel.any():
                va[sel] = _load(lvl, tax[sel], tay[sel], oob)
                vb[sel] = _load(lvl, tbx[sel], tby[sel], oob)
        out[:, j >> 5] |= ((va > vb).astype(np.uint32) << np.uint32(j & 31))
    return out


CONTRACT_LUM, CONTRACT_BLUR, CONTRACT_ROT = 1, 2, 4


def extract(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, y8=False, oob="zero", weight_bits=0, contract=0, dot_order=0,
            f16_round=0, neg_angle=0):
    """orb.rs:469-557 stage order (y8: the frame is a one-byte-per-pixel Y plane).  oob / weight_bits / contract (a bit per stage) /
    dot_order / f16_round / neg_angle: the implementation-defined switches (out-of-level loads, sampler weight precision, fused
    multiply-adds, reduction order, rounding of the R16Float stores, u32() of a negative angle); defaults = CRD-6 / CRD-5 / CRD-2, -5, -10 / CRD-3."""
    gray = [grayscale_y8(rgba, f16_round) if y8 else grayscale(rgba, contract & CONTRACT_LUM, dot_order, f16_round)]
    for _ in range(1, depth):
        gray.append(mip(gray[-1], weight_bits, f16_round))
    tmp = [blur_pass(g, weight_bits, contract & CONTRACT_BLUR, f16_round) for g in gray]
    blur = [blur_pass(t, weight_bits, contract & CONTRACT_BLUR, f16_round) for t in tmp]
    kps = fast(gray, threshold, oob, neg_angle)
    total = kps.shape[0]
    kps = kps[:max_features]
    desc = brief(blur, kps, oob, contract & CONTRACT_ROT, dot_order)
    return dict(total=total, corners=kps, descriptors=desc, gray=gray, blur=blur)


# ------------------------------------------------------------------ synthetic frames
def _mix32(a):
    a = np.asarray(a, dtype=np.uint32).copy()
    a ^= a >> np.uint32(16)
    a *= np.uint32(0x7FEB352D)
    a ^= a >> np.uint32(15)
    a *= np.uint32(0x846CA68B)
    a ^= a >> np.uint32(16)
    return a


def _rnd(seed, stream, idx):
    with np.errstate(over="ignore"):
        s = _mix32(np.uint32((seed + 0x85EBCA6B) & 0xFFFFFFFF))
        t = _mix32(np.uint32((int(stream) + 0x9E3779B9 + int(s)) & 0xFFFFFFFF))
        return _mix32(np.asarray(idx, dtype=np.uint32) ^ t)


def synth_frame(W, H, seed, flags=15):
    """Same recipe as orc_synth_frame, written with whole-image integer arrays."""
    seed = int(seed) & 0xFFFFFFFF
    x = np.arange(W, dtype=np.int64)[None, :].repeat(H, 0)
    y = np.arange(H, dtype=np.int64)[:, None].repeat(W, 1)
    chans = [np.zeros((H, W), dtype=np.int64) for _ in range(3)]
    if flags & 1:
        chans[0] = 255 * x // (W - 1) if W > 1 else chans[0]
        chans[1] = 255 * y // (H - 1) if H > 1 else chans[1]
        chans[2] = 255 * (x + y) // (W + H - 2) if W + H > 2 else chans[2]
    if flags & 4:
        ncx = (W + 127) // 128
        cx, cy = x // 128, y // 64
        h = _rnd(seed, 2, (cy * ncx + cx).astype(np.uint32)).astype(np.int64)
        rise = 4 + ((h >> 1) & 15) % 13
        run = 2 * rise
        ox = ((h >> 8) & 255) % (128 - run)
        oy = ((h >> 16) & 255) % (64 - rise)
        level = (h >> 24) & 255
        u = x - (cx * 128 + ox)
        v = y - (cy * 64 + oy)
        inside = ((h & 1) == 1) & (u >= 0) & (v >= 0) & (u < run) & (v < rise)
        uu = np.where((h & 32) != 0, run - 1 - u, u)
        vv = np.where((h & 64) != 0, rise - 1 - v, v)
        inside &= vv * run <= uu * rise
        for k in range(3):
            chans[k] = np.where(inside, level, chans[k])
    if flags & 2:
        ncx = (W + 31) // 32
        cx, cy = x // 32, y // 32
        h = _rnd(seed, 1, (cy * ncx + cx).astype(np.uint32)).astype(np.int64)
        s = 1 + ((h >> 1) & 3)
        ox = 1 + ((h >> 4) & 255) % (31 - s)
        oy = 1 + ((h >> 12) & 255) % (31 - s)
        level = 128 + ((h >> 20) & 127)
        bx, by = cx * 32 + ox, cy * 32 + oy
        inside = ((h & 1) == 1) & (x >= bx) & (x < bx + s) & (y >= by) & (y < by + s)
        for k in range(3):
            chans[k] = np.where(inside, level, chans[k])
    if flags & 8:
        n = _rnd(seed, 3, (y * W + x).astype(np.uint32)).astype(np.int64)
        for k in range(3):
            chans[k] = (3 * chans[k] + ((n >> (8 * k)) & 255)) // 4
    out = np.empty((H, W, 4), dtype=np.uint8)
    for k in range(3):
        out[..., k] = chans[k].astype(np.uint8)
    out[..., 3] = 255
    return out


# ------------------------------------------------------------------ opt-in extensions (not in the reference)
def _has_run(mask, arc):
    dbl = mask.astype(np.uint64) | (mask.astype(np.uint64) << np.uint64(16))
    want = np.uint64(0xFFFF if arc >= 16 else (1 << arc) - 1)
    out = np.zeros(mask.shape, dtype=bool)
    for s in range(16):
        out |= ((dbl >> np.uint64(s)) & want) == want
    return out


def fast_ex(gray_levels_bits, threshold, arc=12):
    """Arc length 9..16 (no pre-test: it is only a necessary condition for arc >= 12) + the NMS score
    S = sum over the run's polarity of (|v - c| - thr), binary32, ring order.  Definitions: oracle/orb_oracle.h."""
    thr = F(threshold)
    H0, W0 = gray_levels_bits[0].shape
    lim_x = (W0 - 16) & 0xFFFFFFFF
    lim_y = (H0 - 16) & 0xFFFFFFFF
    rows, scores = [], []
    width, height = W0, H0
    for octv, bits in enumerate(gray_levels_bits):
        lvl = from_f16_bits(bits)
        gw, gh = (width + 7) // 8 * 8, (height + 7) // 8 * 8
        width //= 2
        height //= 2
        if gw == 0 or gh == 0:
            continue
        gx, gy = np.meshgrid(np.arange(gw, dtype=np.int64), np.arange(gh, dtype=np.int64))
        guard = (gx > 16) & (gy > 16) & (gx < lim_x) & (gy < lim_y)
        if not guard.any():
            continue
        gx, gy = gx[guard], gy[guard]
        c = _load(lvl, gx, gy)
        m_over = np.zeros(gx.shape, dtype=np.uint32)
        m_under = np.zeros(gx.shape, dtype=np.uint32)
        cx = np.zeros(gx.shape, dtype=np.float32)
        cy = np.zeros(gx.shape, dtype=np.float32)
        s_over = np.zeros(gx.shape, dtype=np.float32)
        s_under = np.zeros(gx.shape, dtype=np.float32)
        for i, (dx, dy) in enumerate(RING16):
            v = _load(lvl, gx + dx, gy + dy)
            diff = v - c
            cx = cx + v * F(dx)
            cy = cy + v * F(dy)
            is_o = diff > thr
            is_u = (~is_o) & (diff < -thr)
            m_over |= (is_o.astype(np.uint32) << np.uint32(i))
            m_under |= (is_u.astype(np.uint32) << np.uint32(i))
            s_over = np.where(is_o, s_over + (diff - thr), s_over).astype(np.float32)
            s_under = np.where(is_u, s_under + ((-diff) - thr), s_under).astype(np.float32)
        ro, ru = _has_run(m_over, arc), _has_run(m_under, arc)
        corner = ro | ru
        ang = atan2f(cy, cx)
        code = angle_code(cy, ang)
        sc = np.where(ro, s_over, s_under)
        for k in np.nonzero(corner)[0]:
            rows.append((int(gx[k]), int(gy[k]), int(code[k]), octv))
            scores.append(sc[k])
    return np.array(rows, dtype=np.uint32).reshape(-1, 4), np.array(scores, dtype=np.float32)


def nms(corners, scores):
    """3x3 non-maximum suppression per octave; ties keep the earlier raster position."""
    table = {(int(o), int(y), int(x)): (s, i) for i, ((x, y, _, o), s) in enumerate(zip(corners, scores))}
    keep = []
    for i, ((x, y, _, o), s) in enumerate(zip(corners, scores)):
        ok = True
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0:
                    continue
                t = table.get((int(o), int(y) + dy, int(x) + dx))
                if t is None:
                    continue
                later = dy > 0 or (dy == 0 and dx > 0)
                if t[0] > s or (t[0] == s and not later):
                    ok = False
        if ok:
            keep.append(i)
    return corners[keep]


def extract_ex(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, arc=12, use_nms=False):
    gray = [grayscale(rgba)]
    for _ in range(1, depth):
        gray.append(mip(gray[-1]))
    blur = [blur_pass(blur_pass(g)) for g in gray]
    kps, scores = fast_ex(gray, threshold, arc)
    if use_nms:
        kps = nms(kps, scores)
    total = kps.shape[0]
    kps = kps[:max_features]
    return dict(total=total, corners=kps, descriptors=brief(blur, kps))


# ------------------------------------------------------------------ "intended" mode (definitions IM-1..IM-8: oracle/orb_oracle.h)
GAUSS = [F(0.282523781), F(0.221251875), F(0.106235079), F(0.0312511548)]


def grayscale_intended(rgba):
    """IM-1: BT.601 luminance, no mirror."""
    chan = np.asarray(rgba, dtype=np.uint8).astype(np.float32) / F(255.0)
    lum = (F(0.299) * chan[..., 0] + F(0.587) * chan[..., 1]) + F(0.114) * chan[..., 2]
    return to_f16_bits(lum)


def gauss_pass(src_bits, vertical):
    """IM-3: one pass of the 7-tap kernel along x (vertical=False) or y, clamp-to-edge, f16 store."""
    src = from_f16_bits(src_bits)
    if vertical:
        src = src.T
    n = src.shape[1]
    idx = np.arange(n)

    def tap(k):
        return src[:, np.clip(idx + k, 0, n - 1)]

    acc = GAUSS[0] * tap(0)
    for k in (1, 2, 3):
        acc = acc + GAUSS[k] * (tap(-k) + tap(k))
    if vertical:
        acc = acc.T
    return to_f16_bits(acc)


def angle_code_signed(cy, cx):
    """IM-5: milliradians over the full circle, 0..6283."""
    r = atan2f(cy, cx)
    r = np.where(r < 0, r + F(6.28318531), r).astype(np.float32)
    return np.minimum(np.trunc(r * F(1000.0)).astype(np.uint32), np.uint32(6283))


def fast_intended(gray_levels_bits, threshold, arc=9):
    """IM-4/IM-5: segment test with the octave's own guard, score as fast_ex."""
    thr = F(threshold)
    rows, scores = [], []
    for octv, bits in enumerate(gray_levels_bits):
        lvl = from_f16_bits(bits)
        h, w = lvl.shape
        if w <= 33 or h <= 33:
            continue
        gx, gy = np.meshgrid(np.arange(17, w - 16, dtype=np.int64), np.arange(17, h - 16, dtype=np.int64))
        gx, gy = gx.ravel(), gy.ravel()
        c = lvl[gy, gx]
        m_over = np.zeros(gx.shape, dtype=np.uint32)
        m_under = np.zeros(gx.shape, dtype=np.uint32)
        cx = np.zeros(gx.shape, dtype=np.float32)
        cy = np.zeros(gx.shape, dtype=np.float32)
        s_over = np.zeros(gx.shape, dtype=np.float32)
        s_under = np.zeros(gx.shape, dtype=np.float32)
        for i, (dx, dy) in enumerate(RING16):
            v = lvl[gy + dy, gx + dx]  # always inside the level: the guard is 16 px wide
            diff = v - c
            cx = cx + v * F(dx)
            cy = cy + v * F(dy)
            is_o = diff > thr
            is_u = (~is_o) & (diff < -thr)
            m_over |= (is_o.astype(np.uint32) << np.uint32(i))
            m_under |= (is_u.astype(np.uint32) << np.uint32(i))
            s_over = np.where(is_o, s_over + (diff - thr), s_over).astype(np.float32)
            s_under = np.where(is_u, s_under + ((-diff) - thr), s_under).astype(np.float32)
        ro, ru = _has_run(m_over, arc), _has_run(m_under, arc)
        sel = np.nonzero(ro | ru)[0]
        code = angle_code_signed(cy[sel], cx[sel])
        sc = np.where(ro, s_over, s_under)[sel]
        for k, i in enumerate(sel):
            rows.append((int(gx[i]), int(gy[i]), int(code[k]), octv))
            scores.append(sc[k])
    return np.array(rows, dtype=np.uint32).reshape(-1, 4), np.array(scores, dtype=np.float32)


def binned_angle_code(codes, bins):
    """IM-6b: angle codes quantised into `bins` bins of the full circle -- the code of the bin's centre (0: unchanged)."""
    codes = np.minimum(np.asarray(codes, dtype=np.int64), 6283)
    if not bins:
        return codes
    return ((codes * bins // 6284) * 6284 + 3142) // bins


def brief_intended(blur_levels_bits, corners, angle_bins=0):
    """IM-6: pattern rotated by +theta (IM-6b: theta of the keypoint's angle bin)."""
    corners = np.asarray(corners, dtype=np.uint32).reshape(-1, 4)
    n = corners.shape[0]
    out = np.zeros((n, 8), dtype=np.uint32)
    levels = [from_f16_bits(b) for b in blur_levels_bits]
    theta = binned_angle_code(corners[:, 2], angle_bins).astype(np.float32) / F(1000.0)
    ct = np.cos(theta.astype(np.float64)).astype(np.float32)
    st = np.sin(theta.astype(np.float64)).astype(np.float32)
    px = corners[:, 0].astype(np.int64)
    py = corners[:, 1].astype(np.int64)
    for j in range(256):
        ax, ay, bx, by = (F(v) for v in PATTERN[j])
        rax = ct * ax + (-st) * ay
        ray = st * ax + ct * ay
        rbx = ct * bx + (-st) * by
        rby = st * bx + ct * by
        tax, tay = np.trunc(rax).astype(np.int64) + px, np.trunc(ray).astype(np.int64) + py
        tbx, tby = np.trunc(rbx).astype(np.int64) + px, np.trunc(rby).astype(np.int64) + py
        va = np.zeros(n, dtype=np.float32)
        vb = np.zeros(n, dtype=np.float32)
        for octv, lvl in enumerate(levels):
            sel = corners[:, 3] == octv
            if sel.any():
                va[sel] = _load(lvl, tax[sel], tay[sel])
                vb[sel] = _load(lvl, tbx[sel], tby[sel])
        out[:, j >> 5] |= ((va > vb).astype(np.uint32) << np.uint32(j & 31))
    return out


def nms_indices(corners, scores):
    """Indices of the survivors of nms() (same rule)."""
    table = {(int(o), int(y), int(x)): s for (x, y, _, o), s in zip(corners, scores)}
    keep = []
    for i, ((x, y, _, o), s) in enumerate(zip(corners, scores)):
        ok = True
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                t = table.get((int(o), int(y) + dy, int(x) + dx)) if (dx or dy) else None
                if t is not None and (t > s or (t == s and not (dy > 0 or (dy == 0 and dx > 0)))):
                    ok = False
        if ok:
            keep.append(i)
    return np.array(keep, dtype=np.int64)


def topk(corners, scores, k):
    """IM-8: the k best by (score desc, octave, y, x)."""
    if corners.shape[0] <= k:
        return corners
    order = np.lexsort((corners[:, 0], corners[:, 1], corners[:, 3], -scores.astype(np.float64)))
    return corners[order[:k]]


def extract_intended(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, arc=9, use_nms=False, angle_bins=0):
    gray = [grayscale_intended(rgba)]
    for _ in range(1, depth):
        gray.append(mip(gray[-1]))
    blur = [gauss_pass(gauss_pass(g, False), True) for g in gray]
    kps, scores = fast_intended(gray, threshold, arc)
    if use_nms:
        keep = nms_indices(kps, scores)
        kps, scores = kps[keep], scores[keep]
    total = kps.shape[0]
    kps = topk(kps, scores, max_features)
    return dict(total=total, corners=kps, descriptors=brief_intended(blur, kps, angle_bins), gray=gray, blur=blur)


# ------------------------------------------------------------------ descriptor matching (definition: include/tinyorb.h)
def match(desc_a, desc_b):
    """Brute-force Hamming match of every row of desc_a (u32 (n, 8)) against desc_b; ties to the smallest index.
    Returns (index u32, distance u16, second u16)."""
    desc_a = np.asarray(desc_a, dtype=np.uint32).reshape(-1, 8)
    desc_b = np.asarray(desc_b, dtype=np.uint32).reshape(-1, 8)
    na, nb = desc_a.shape[0], desc_b.shape[0]
    index = np.full(na, 0xFFFFFFFF, dtype=np.uint32)
    dist = np.full(na, 0xFFFF, dtype=np.uint16)
    second = np.full(na, 0xFFFF, dtype=np.uint16)
    if nb == 0:
        return index, dist, second
    bits_b = np.unpackbits(desc_b.view(np.uint8), axis=1)
    for i0 in range(0, na, 256):
        bits_a = np.unpackbits(desc_a[i0:i0 + 256].view(np.uint8), axis=1)
        d = (bits_a[:, None, :] != bits_b[None, :, :]).sum(axis=2).astype(np.int64)
        j = d.argmin(axis=1)  # first minimum
        index[i0:i0 + 256] = j
        dist[i0:i0 + 256] = d[np.arange(d.shape[0]), j]
        if nb > 1:
            d[np.arange(d.shape[0]), j] = 1 << 20
            second[i0:i0 + 256] = d.min(axis=1)
    return index, dist, second
