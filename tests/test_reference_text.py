"""The reference's shader TEXT, executed (tests/wgsl_interp.py), against the two restatements that were written by reading it.

What is mechanical here: every expression, constant, loop, guard, bit order and index of the six shaders -- read from the reference checkout at
test time (these tests are skipped where it is absent: the GPU box, a bare clone).  What is not: the fixed-function parts around the shaders,
restated below with the lines of orb.rs they follow (which shader runs over which view, dispatch sizes, the rasteriser's interpolation, the
R16Float store) and the points WGSL leaves to the implementation, bound to the restatement's documented decisions (the
sampler, atan2, cos / sin, u32() of a negative float, loads outside a level).  So this removes TRANSCRIPTION as a source of error between the
shaders and the oracle; it does not pin the oracle to an adapter (parity unpinned)."""
import math
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wgsl_interp as wi  # noqa: E402

SHADERS = "/root/reference/src/shaders"
pytestmark = pytest.mark.skipif(not os.path.isdir(SHADERS), reason="the reference checkout is not on this machine")
F = np.float32


def shader(name, edits=None):
    """The shader's text; edits: {shader: (old, new)} -- a deliberate slip, for the test that shows the comparison notices one."""
    text = open(os.path.join(SHADERS, name + ".wgsl")).read()
    if edits and name in edits:
        old, new = edits[name]
        assert text.count(old) >= 1, "the text to edit is not in %s.wgsl" % name
        text = text.replace(old, new, 1)
    return text


def f16_round(v):
    """The store to an R16Float target (orb.rs:228 and the views of 151, 296, 311): nearest even (CRD-3); kept as the binary32 value."""
    return F(np.float16(v))


class Plane:
    """One texture view: rows x columns of binary32 values (of binary16 precision), or the H x W x 4 byte frame."""

    def __init__(self, a):
        self.a = a
        self.h, self.w = a.shape[:2]


class Pyramid:
    """A texture with mip levels, as fast.wgsl and brief.wgsl see it."""

    def __init__(self, levels):
        self.levels = levels


class Adapter(wi.Hooks):
    """The implementation-defined points as the restatement decided them."""

    def __init__(self, oracle, oob="zero", weight_bits=0, contract=0, dot_order=0, neg_angle="zero"):
        self.oracle, self.oob, self.wq, self.neg_angle = oracle, oob, float(1 << weight_bits) if weight_bits else 0.0, neg_angle
        self.contract_dot, self.contract_muladd, self.contract_matvec = bool(contract & 1), bool(contract & 2), bool(contract & 4)
        self.last_first = bool(dot_order)

    def f32_to_u32(self, v):  # fast.wgsl:153 of a negative angle: 0 (Q7), or what an x86-64 CPU adapter's conversion leaves (orc_impl_t::neg_angle)
        v = float(v)
        if v < 0 and self.neg_angle != "zero":
            return (int(v) & 0xFFFFFFFF) if self.neg_angle == "wrap" else (0xFFFFFFFF if int(v) < 0 else 0)
        return 0 if not v > 0 else min(int(v), 0xFFFFFFFF)

    def atan2(self, y, x):  # CRD-9: the canonical routine shared by oracle and kernels
        return F(self.oracle.lib().orc_atan2f(float(F(y)), float(F(x))))

    def cos(self, a):  # CRD-10: the correctly rounded value (the committed table holds cos of fl32(code / 1000))
        return F(math.cos(float(F(a))))

    def sin(self, a):
        return F(math.sin(float(F(a))))

    def texture_dimensions(self, tex):  # textureDimensions(t) without a level: level 0 (Q8)
        return tex.levels[0].w, tex.levels[0].h

    def texture_load(self, tex, x, y, level):
        p = tex.levels[level]
        if not (0 <= x < p.w and 0 <= y < p.h):
            if self.oob == "zero":  # CRD-6: robust image access -- (0, 0, 0, 0 or 1); only .x is read
                return (F(0), F(0), F(0), F(1))
            if self.oob == "clamp":  # the coordinate clamped into the level
                x, y = min(max(x, 0), p.w - 1), min(max(y, 0), p.h - 1)
            else:  # "umin": naga's Restrict policy, min(u32(coordinate), size - 1) -- a negative coordinate lands on the LAST texel
                x = p.w - 1 if not 0 <= x < p.w else x
                y = p.h - 1 if not 0 <= y < p.h else y
        return (p.a[y, x], F(0), F(0), F(1))

    def texture_sample(self, tex, sampler, u, v):
        w, h = tex.w, tex.h
        # the row: every pass samples rows at texel centres (v = (row + 0.5) / h up to the rounding of the interpolated coordinate): CRD-1
        yf = float(v) * h - 0.5
        if sampler == "box":  # blit.wgsl at an exact half: the sample sits on the corner of four texels -- CRD-4 ((a + b) + (c + d)) * 0.25
            xf = float(u) * w - 0.5
            x0, y0 = math.floor(xf), math.floor(yf)
            assert abs(xf - x0 - 0.5) < 1e-3 and abs(yf - y0 - 0.5) < 1e-3
            a, b, c, d = tex.a[y0, x0], tex.a[y0, x0 + 1], tex.a[y0 + 1, x0], tex.a[y0 + 1, x0 + 1]
            return (((a + b) + (c + d)) * F(0.25), F(0), F(0), F(1))
        row = int(round(yf))
        assert abs(yf - row) < 1e-3 and 0 <= row < h
        if sampler == "texel":  # grayscale.wgsl: texel centres in both directions (CRD-1); RGBA8 unorm -> byte / 255 (IEEE division)
            xf = float(u) * w - 0.5
            col = int(round(xf))
            assert abs(xf - col) < 1e-3 and 0 <= col < w
            return tuple(F(c) / F(255.0) for c in tex.a[row, col])
        assert sampler == "linear_x"  # gaussian_blur_x.wgsl: CRD-5, the coordinate in binary32, clamp to edge, lerp = t0 + f * (t1 - t0)
        coord = F(u) * F(w) - F(0.5)
        i0 = math.floor(float(coord))
        f = coord - F(i0)
        if self.wq:  # a sampler that holds its weights in n fractional bits: nearest multiple of 2^-n, halves up
            f = F(math.floor(float(f * F(self.wq) + F(0.5)))) / F(self.wq)
        t0, t1 = tex.a[row, min(max(i0, 0), w - 1)], tex.a[row, min(max(i0 + 1, 0), w - 1)]
        return (t0 + f * (t1 - t0), F(0), F(0), F(1))


def render(mod, sampler_name, texture_name, src, mode, out_w, out_h):
    """One full-screen draw (orb.rs:478-496, 413-466: `rpass.draw(0..3, 0..1)` into an R16Float view): the vertex shader runs for the three
    vertices, the varyings are the affine function of clip-space position through them (evaluated exactly at the pixel centre, rounded to
    binary32 once), the fragment shader runs per pixel, `.x` of its result is stored as binary16."""
    vs = [mod.run_function("vs_main", wi.V("u32", i)) for i in range(3)]
    vary = [f for f, t, a in mod.structs["VertexOutput"] if "location" in a][0]
    pos = [[Fraction(float(c)) for c in v["position"].e[:2]] for v in vs]
    val = [[Fraction(float(c)) for c in v[vary].e] for v in vs]
    # varying = val0 + s * (val1 - val0) + t * (val2 - val0) with pos = pos0 + s * (pos1 - pos0) + t * (pos2 - pos0)
    (ax, ay), (bx, by) = [(pos[k][0] - pos[0][0], pos[k][1] - pos[0][1]) for k in (1, 2)]
    det = ax * by - ay * bx
    mod.bind(**{sampler_name: mode, texture_name: src})
    out = np.zeros((out_h, out_w), dtype=np.float32)
    for py in range(out_h):
        for px in range(out_w):
            nx, ny = Fraction(2 * px + 1, out_w) - 1, 1 - Fraction(2 * py + 1, out_h)  # pixel centre; framebuffer row 0 is the top (y up in clip space)
            dx, dy = nx - pos[0][0], ny - pos[0][1]
            s, t = (dx * by - dy * bx) / det, (ax * dy - ay * dx) / det
            uv = [F(float(val[0][k] + s * (val[1][k] - val[0][k]) + t * (val[2][k] - val[0][k]))) for k in range(2)]
            frag = {"position": wi.Vec("f32", [F(px + 0.5), F(py + 0.5), F(0), F(1)]), vary: wi.Vec("f32", uv)}
            out[py, px] = f16_round(mod.run_function("fs_main", frag).e[0])
    return Plane(out)


def run_reference_text(oracle, rgba, depth, thr, cap, edits=None, **adapter):
    """extract_corners (orb.rs:469-557) with the shaders interpreted: returns grey and blur levels, the counter, corners and descriptors."""
    hooks = Adapter(oracle, **adapter)
    shader = lambda name: globals()["shader"](name, edits)  # noqa: E731
    H, W = rgba.shape[:2]
    gray = [render(wi.Module(shader("grayscale"), hooks), "texture_sampler", "texture", Plane(rgba), "texel", W, H)]   # orb.rs:478-496
    blit = wi.Module(shader("blit"), hooks)
    for m in range(1, depth):                                                                                          # orb.rs:413-429
        gray.append(render(blit, "r_sampler", "r_color", gray[-1], "box", W >> m, H >> m))
    blur_x = wi.Module(shader("gaussian_blur_x"), hooks)  # BOTH blur pipelines are built from this shader (orb.rs:388-408, Q12)
    tmp = [render(blur_x, "texture_sampler", "texture", g, "linear_x", g.w, g.h) for g in gray]                       # orb.rs:432-448
    blur = [render(blur_x, "texture_sampler", "texture", t, "linear_x", t.w, t.h) for t in tmp]                       # orb.rs:450-466
    fast = wi.Module(shader("fast"), hooks)
    feature = {"x": wi.V("u32", 0), "y": wi.V("u32", 0), "angle": wi.V("u32", 0), "octave": wi.V("u32", 0)}
    corners = [dict(feature) for _ in range(cap)]
    fast.bind(texture=Pyramid(gray), corners=corners, global_counter=wi.V("u32", 0), threshold=wi.V("f32", F(thr)))
    w, h = W, H
    for i in range(depth):                                                                                             # orb.rs:499-520
        fast.bind(octave=wi.V("u32", i))
        fast.dispatch("compute_fast", ((w + 7) // 8, (h + 7) // 8, 1))
        w, h = w // 2, h // 2
    total = fast.globals["global_counter"].v
    brief = wi.Module(shader("brief"), hooks)
    desc = [[wi.V("u32", 0) for _ in range(8)] for _ in range(cap)]
    brief.bind(corners=corners, counter=wi.V("u32", total), descriptors=desc, blur_hierarchy=Pyramid(blur))
    brief.dispatch("brief", (1, (cap + 7) // 8, 1))                                                                    # orb.rs:523-534
    n = min(total, cap)
    c = np.array([[corners[k][f].v for f in ("x", "y", "angle", "octave")] for k in range(n)], dtype=np.uint32).reshape(-1, 4)
    d = np.array([[desc[k][j].v for j in range(8)] for k in range(n)], dtype=np.uint32).reshape(-1, 8)
    return gray, blur, total, c, d


def test_known_answers_from_the_text_itself():
    """SURVEY.md section 8c's known answers, evaluated on the reference's own functions instead of their restatement."""
    fast = wi.Module(shader("fast"))
    streak = lambda x: fast.call("detect_streak_16", [wi.V("u32", x)]).v  # noqa: E731
    assert [streak(x) for x in (0x0FFF, 0x07FF, 0xF0FF, 0xFFFF)] == [0x0001, 0, 0x1000, 0xFFFF]
    for x in range(0, 1 << 16, 97):  # "a circular run of twelve" by brute force
        run = any(all((x >> ((s + k) & 15)) & 1 for k in range(12)) for s in range(16))
        assert (streak(x) != 0) == run, hex(x)
    ring = [tuple(v.e) for v in fast.globals["CORNERS_16"]]
    assert ring[0] == (-3, 0) and ring[4] == (0, -3) and len(set(ring)) == 16 and all(x * x + y * y in (8, 9, 10) for x, y in ring)
    brief = wi.Module(shader("brief"))
    pat = np.array([v.e for v in brief.globals["brief_descriptors"]], dtype=np.int8)
    import hashlib
    assert pat.shape == (256, 4) and hashlib.sha256(pat.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def frame_with_corners(oracle, W, H, seed, flags):
    rgba = oracle.synth_frame(W, H, seed, flags)
    rng = np.random.default_rng(seed)
    for _ in range(10):  # bright and dark squares inside the guard so that corners exist at this size (s = 1..4 fire)
        s = int(rng.integers(1, 5))
        x, y = int(rng.integers(18, W - 22)), int(rng.integers(18, H - 22))
        rgba[y:y + s, x:x + s, :3] = 255 if rng.random() < 0.7 else 0
    return rgba


def differences(oracle, rgba, depth, thr, cap, got, **switches):
    """What differs between an interpreted run and oracle/orb_oracle.c (under the same switches): a list of names (empty: bit for bit the same)."""
    gray, blur, total, c, d = got
    H, W = rgba.shape[:2]
    ref = oracle.extract(rgba, depth=depth, threshold=thr, max_features=cap, planes=True, **switches)
    dims, _ = oracle.level_dims(W, H, depth)
    out = []
    for m, (w, h, off) in enumerate(dims):
        if not np.array_equal(gray[m].a.astype(np.float16).view(np.uint16), ref["gray"][off:off + w * h].reshape(h, w)):
            out.append("grey level %d" % m)
        if not np.array_equal(blur[m].a.astype(np.float16).view(np.uint16), ref["blur"][off:off + w * h].reshape(h, w)):
            out.append("blur level %d" % m)
    if total != ref["total"]:
        out.append("counter")
    order = np.lexsort((c[:, 0], c[:, 1], c[:, 3]))
    rc, rd = oracle.sort_keypoints(ref["corners"], ref["descriptors"])
    want_c = np.stack([rc[k] for k in ("x", "y", "angle", "octave")], 1)
    if c.shape != want_c.shape or not np.array_equal(c[order], want_c):
        out.append("corners")
    elif not np.array_equal(d[order], rd):
        out.append("descriptors")
    return out


@pytest.mark.parametrize("W,H,depth,seed,flags", [(64, 48, 2, 5, 15), (80, 56, 2, 9, 15)])
def test_the_shader_text_executed_equals_both_restatements(oracle, numpy_ref, W, H, depth, seed, flags):
    """grey, mip, both blur passes, FAST with orientation, BRIEF: the interpreted shaders against oracle/orb_oracle.c, plane by plane and
    record by record, bit for bit -- on frames that put keypoints on both octaves, give a third of them a non-zero angle, and leave the
    blur varying in its last eighth of the columns only (the property of the literal blur the fused kernels lean on)."""
    rgba = frame_with_corners(oracle, W, H, seed, flags)
    thr, cap = np.float32(20.0 / 255.0), 256
    got = run_reference_text(oracle, rgba, depth, thr, cap)
    assert differences(oracle, rgba, depth, thr, cap, got) == []
    gray, blur, total, c, d = got
    assert 0 < total <= cap and (c[:, 2] > 0).sum() >= 10, "too few keypoints with a non-zero angle: the rotation was not exercised"
    if (W, H) == (64, 48):
        assert (c[:, 3] == 1).any(), "no keypoint on octave 1"
    varying = int((blur[0].a != blur[0].a[:, :1]).any(0).sum())
    assert 0 < varying <= W // 8 + 1, "the literal blur varies in %d of %d columns" % (varying, W)
    assert numpy_ref.extract(rgba, depth=depth, threshold=thr, max_features=cap)["total"] == total


@pytest.mark.parametrize("edits,expect", [
    ({"grayscale": ("0.229", "0.299")}, "grey level 0"),                          # the red weight (Q1)
    ({"gaussian_blur_x": ("1.3243948342247673", "1.3243948342247")}, None),      # a literal that rounds to the same binary32: no difference
    ({"gaussian_blur_x": ("0.5037756553768409", "0.50378")}, "blur level 0"),     # ... and one that does not
    ({"fast": ("vec2i(-3, -1)", "vec2i(-3, 1)")}, "corners"),                     # one ring offset
    ({"brief": ("global_id.x << 5u | i", "global_id.x << 4u | i")}, "descriptors"),  # the descriptor's bit order
])
def test_a_slip_in_the_text_is_noticed(oracle, edits, expect):
    """The comparison has teeth: one edited token in one shader shows up in the stage it belongs to (and an edit that does not change the
    binary32 constant does not)."""
    W, H, depth, thr, cap = 64, 48, 2, np.float32(20.0 / 255.0), 256
    rgba = frame_with_corners(oracle, W, H, 5, 15)
    diff = differences(oracle, rgba, depth, thr, cap, run_reference_text(oracle, rgba, depth, thr, cap, edits))
    if expect is None:
        assert diff == []
    else:
        assert expect in diff, diff


@pytest.mark.parametrize("switches", [
    dict(oob="clamp"), dict(oob="umin"), dict(weight_bits=8), dict(dot_order=1), dict(contract=7), dict(contract=7, dot_order=1),
    dict(contract=2, weight_bits=8, oob="clamp"), dict(neg_angle="wrap"),
], ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_the_switches_mean_the_same_in_the_interpreter_and_the_oracle(oracle, switches):
    """The implementation-defined points (orc_impl_t) modelled a second time, around the interpreted text: out-of-level loads, sampler weight
    bits, the three contractions with both reduction orders, the conversion of a negative angle -- the oracle under a switch equals the
    interpreter under the same reading; the policies and the weight bits change something on this frame (that equality is not vacuous), the
    arithmetic forms are too rare for 64 x 48 texels and are compared texel by texel in test_oracle.py."""
    W, H, depth, thr, cap = 64, 48, 2, np.float32(20.0 / 255.0), 256
    rgba = frame_with_corners(oracle, W, H, 5, 15)
    got = run_reference_text(oracle, rgba, depth, thr, cap, **switches)
    assert differences(oracle, rgba, depth, thr, cap, got, **switches) == []
    if set(switches) & {"oob", "weight_bits", "neg_angle"}:  # (the arithmetic forms move a dozen texels of a 1280 x 720 frame: none of these 3072)
        assert differences(oracle, rgba, depth, thr, cap, got) != [], "this frame does not tell %r from the defaults" % (switches,)


def test_the_luminance_forms_on_texels_that_tell_them_apart(oracle):
    """grayscale.wgsl's dot() interpreted under the four forms a shader compiler may give it (CRD-2 / CRD-13: fused or not, first or last
    component first), on a frame made of the colours on which the forms disagree -- each form equals the oracle's, and no two forms agree."""
    rng = np.random.default_rng(11)
    cols = rng.integers(0, 256, size=(1, 1 << 21, 4), dtype=np.uint8)  # (one row: the stage mirrors rows)  # the forms disagree on 5 colours in 100 000
    cols[..., 3] = 255
    forms = [(0, 0), (1, 0), (0, 1), (1, 1)]
    grey = {f: oracle.grayscale_fp(cols, f[0], f[1]).ravel() for f in forms}
    pick = np.flatnonzero((grey[0, 0] != grey[1, 0]) | (grey[0, 0] != grey[0, 1]) | (grey[1, 0] != grey[1, 1]))[:64]
    assert len(pick) == 64
    rgba = np.ascontiguousarray(cols[0, pick].reshape(4, 16, 4))
    seen = {}
    for contract, order in forms:
        hooks = Adapter(oracle, contract=contract, dot_order=order)
        got = render(wi.Module(shader("grayscale"), hooks), "texture_sampler", "texture", Plane(rgba), "texel", 16, 4)
        bits = got.a.astype(np.float16).view(np.uint16)
        assert np.array_equal(bits, oracle.grayscale_fp(rgba, contract, order)), (contract, order)
        seen[contract, order] = bits.tobytes()
    assert len(set(seen.values())) == 4


def test_the_rotation_forms_where_they_differ(oracle):
    """brief.wgsl:38-54's matrix * vector under the three forms (unfused; the second term fused onto the first product; the first onto the
    second), at the one angle code (2214) and the three pattern points where their truncations differ -- against oracle.brief_rotate."""
    text = "fn rot(ct: f32, st: f32, p: vec2f) -> vec2f { let m = mat2x2f(ct, -st, st, ct); return m * p; }"  # the expression of brief.wgsl:38-54
    theta = np.float32(2214) / np.float32(1000.0)
    ct, st = F(math.cos(float(theta))), F(math.sin(float(theta)))
    results = {}
    for contract, order in ((0, 0), (4, 0), (4, 1)):
        mod = wi.Module(text, Adapter(oracle, contract=contract, dot_order=order))
        for px, py in ((-3, 4), (6, -8), (8, 6), (5, -7)):
            r = mod.call("rot", [wi.V("f32", ct), wi.V("f32", st), wi.Vec("f32", [F(px), F(py)])])
            want = oracle.brief_rotate(2214, px, py, contract, order)
            assert (r.e[0], r.e[1]) == (want[0], want[1]), (contract, order, px, py)
            results[contract, order, px, py] = tuple(int(v) for v in r.e)
    assert len({results[c, o, -3, 4] for c, o in ((0, 0), (4, 0), (4, 1))}) > 1  # the truncated points do differ between the forms here
