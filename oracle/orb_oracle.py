"""ctypes front-end of the CPU restatement (oracle/orb_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: see oracle/orb_oracle.h.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborb_oracle.so")

MAX_LEVELS = 10
SYN_GRADIENT, SYN_BLOBS, SYN_WEDGES, SYN_NOISE = 1, 2, 4, 8
SYN_ALL = 15

CORNER_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("angle", "<u4"), ("octave", "<u4")])


class _Pyramid(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_uint32),
        ("w", ctypes.c_uint32 * MAX_LEVELS),
        ("h", ctypes.c_uint32 * MAX_LEVELS),
        ("offset", ctypes.c_size_t * MAX_LEVELS),
        ("total", ctypes.c_size_t),
    ]


def build(force=False):
    """Compile the restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, n) for n in ("orb_oracle.c", "orb_oracle.h", "orb_pattern.h", "Makefile")]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(s) <= os.path.getmtime(_LIB_PATH) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, u16p, u32p, vp = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16),
                               ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p)
        u32, f32 = ctypes.c_uint32, ctypes.c_float
        L.orc_pyramid_layout.argtypes = [u32, u32, u32, ctypes.POINTER(_Pyramid)]
        L.orc_f32_to_f16.argtypes = [f32]
        L.orc_f32_to_f16.restype = ctypes.c_uint16
        L.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.orc_f16_to_f32.restype = f32
        L.orc_atan2f.argtypes = [f32, f32]
        L.orc_atan2f.restype = f32
        L.orc_angle_code.argtypes = [f32, f32]
        L.orc_angle_code.restype = u32
        L.orc_angle_code_neg.argtypes = [f32, f32, u32]
        L.orc_angle_code_neg.restype = u32
        L.orc_detect_streak_16.argtypes = [u32]
        L.orc_detect_streak_16.restype = u32
        L.orc_unorm8.argtypes = [ctypes.c_uint8]
        L.orc_unorm8.restype = f32
        L.orc_grayscale.argtypes = [vp, u32, u32, vp]
        L.orc_grayscale_y8.argtypes = [vp, u32, u32, vp]
        L.orc_mip.argtypes = [vp, u32, u32, vp, u32, u32]
        L.orc_blur_pass.argtypes = [vp, u32, u32, vp]
        L.orc_fast.argtypes = [vp, ctypes.POINTER(_Pyramid), f32, vp, u32, u32p]
        L.orc_brief.argtypes = [vp, ctypes.POINTER(_Pyramid), vp, u32, vp]
        L.orc_brief_impl2.argtypes = [vp, ctypes.POINTER(_Pyramid), vp, u32, u32, u32, vp]
        L.orc_brief_impl2.restype = None
        L.orc_brief_fp.argtypes = [vp, ctypes.POINTER(_Pyramid), vp, u32, vp, vp]
        L.orc_brief_fp.restype = None
        L.orc_brief_rotate.argtypes = [u32, ctypes.c_int, ctypes.c_int, u32, u32, ctypes.POINTER(f32), ctypes.POINTER(f32)]
        L.orc_brief_rotate.restype = None
        L.orc_f32_to_f16_mode.argtypes = [f32, u32]
        L.orc_f32_to_f16_mode.restype = ctypes.c_uint16
        L.orc_grayscale_fp.argtypes = [vp, u32, u32, vp, vp]
        L.orc_mip_fp.argtypes = [vp, u32, u32, vp, u32, u32, vp]
        L.orc_blur_pass_fp.argtypes = [vp, u32, u32, vp, vp]
        L.orc_grayscale_impl.argtypes = [vp, u32, u32, vp, u32]
        L.orc_blur_pass_impl2.argtypes = [vp, u32, u32, vp, u32, u32]
        L.orc_extract.argtypes = [vp, u32, u32, u32, f32, u32, vp, vp, u32p, vp, vp]
        L.orc_extract_y8.argtypes = [vp, u32, u32, u32, f32, u32, vp, vp, u32p, vp, vp]
        L.orc_extract.restype = ctypes.c_int
        L.orc_extract_impl.argtypes = [vp, ctypes.c_int, u32, u32, u32, f32, u32, vp, vp, vp, u32p, vp, vp]
        L.orc_extract_impl.restype = ctypes.c_int
        L.orc_mip_impl.argtypes = [vp, u32, u32, vp, u32, u32, u32]
        L.orc_blur_pass_impl.argtypes = [vp, u32, u32, vp, u32]
        L.orc_extract_ex.argtypes = [vp, u32, u32, u32, f32, u32, vp, vp, vp, u32p]
        L.orc_extract_ex.restype = ctypes.c_int
        L.orc_extract_intended.argtypes = [vp, u32, u32, u32, f32, u32, vp, vp, vp, u32p, vp, vp]
        L.orc_extract_intended.restype = ctypes.c_int
        L.orc_gauss_pass.argtypes = [vp, u32, u32, vp, ctypes.c_int]
        L.orc_grayscale_intended.argtypes = [vp, u32, u32, vp]
        L.orc_angle_code_signed.argtypes = [f32, f32]
        L.orc_angle_code_signed.restype = u32
        L.orc_extract_intended_batch.argtypes = [vp, u32, u32, u32, u32, f32, u32, vp, vp, vp, vp, ctypes.c_int]
        L.orc_extract_intended_batch.restype = ctypes.c_int
        L.orc_extract_batch.argtypes = [vp, u32, u32, u32, u32, f32, u32, vp, vp, vp, ctypes.c_int]
        L.orc_extract_batch_y8.argtypes = [vp, u32, u32, u32, u32, f32, u32, vp, vp, vp, ctypes.c_int]
        L.orc_extract_batch.restype = ctypes.c_int
        L.orc_extract_batch_impl.argtypes = [vp, ctypes.c_int, u32, u32, u32, u32, f32, u32, vp, vp, vp, vp, ctypes.c_int]
        L.orc_extract_batch_impl.restype = ctypes.c_int
        L.orc_synth_frame.argtypes = [vp, u32, u32, u32, u32]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pyramid_layout(W, H, depth):
    p = _Pyramid()
    lib().orc_pyramid_layout(W, H, depth, ctypes.byref(p))
    return p


def level_dims(W, H, depth):
    p = pyramid_layout(W, H, depth)
    return [(int(p.w[m]), int(p.h[m]), int(p.offset[m])) for m in range(depth)], int(p.total)


def f32_to_f16(v, rtz=0):
    """CRD-3 (round to nearest even); rtz = 1: toward zero (orc_impl_t::f16_round)."""
    return int(lib().orc_f32_to_f16_mode(float(np.float32(v)), int(rtz)))


def f16_to_f32(h):
    return float(lib().orc_f16_to_f32(int(h)))


def atan2f(y, x):
    return np.float32(lib().orc_atan2f(float(np.float32(y)), float(np.float32(x))))


def angle_code(cy, cx):
    return int(lib().orc_angle_code(float(np.float32(cy)), float(np.float32(cx))))


def detect_streak_16(mask):
    return int(lib().orc_detect_streak_16(int(mask)))


def unorm8(b):
    return np.float32(lib().orc_unorm8(int(b)))


def synth_frame(W, H, seed, flags=SYN_ALL):
    out = np.empty((H, W, 4), dtype=np.uint8)
    lib().orc_synth_frame(_ptr(out), W, H, int(seed) & 0xFFFFFFFF, flags)
    return out


def synth_frame_y8(W, H, seed, flags=SYN_ALL):
    """One-byte-per-pixel test frame for the Y8 input variant: the integer BT.601 luma of the RGBA recipe
    (ORB_SYN_Y8 of include/tinyorb.h generates the same on the device)."""
    px = synth_frame(W, H, seed, flags).astype(np.uint32)
    return ((77 * px[:, :, 0] + 150 * px[:, :, 1] + 29 * px[:, :, 2] + 128) >> 8).astype(np.uint8)


def grayscale(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    H, W = rgba.shape[:2]
    out = np.empty((H, W), dtype=np.uint16)
    lib().orc_grayscale(_ptr(rgba), W, H, _ptr(out))
    return out


def mip(src, wd=None, hd=None, weight_bits=0):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    hs, ws = src.shape
    wd = max(1, ws >> 1) if wd is None else wd
    hd = max(1, hs >> 1) if hd is None else hd
    out = np.empty((hd, wd), dtype=np.uint16)
    lib().orc_mip_impl(_ptr(src), ws, hs, _ptr(out), wd, hd, int(weight_bits))
    return out


def blur_pass(src, weight_bits=0):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    h, w = src.shape
    out = np.empty((h, w), dtype=np.uint16)
    lib().orc_blur_pass_impl(_ptr(src), w, h, _ptr(out), int(weight_bits))
    return out


# The implementation-defined switches of orb_oracle.h (orc_impl_t): what a textureLoad outside the level returns, the precision of a
# bilinear sampler's weights, which stages' products and sums a shader compiler fuses (a bit per stage), the order in which it reduces
# dot() / matrix * vector, and how a store to an R16Float target rounds.  The defaults are CRD-6 / CRD-5 / CRD-2, -5, -10 / CRD-3.
OOB_POLICIES = {"zero": 0, "clamp": 1, "umin": 2}
CONTRACT_LUM, CONTRACT_BLUR, CONTRACT_ROT, CONTRACT_ALL = 1, 2, 4, 7


NEG_ANGLE = {"zero": 0, "wrap": 1, "ones": 2}  # orc_impl_t::neg_angle: u32() of a negative angle (fast.wgsl:153; undefined in SPIR-V)


def _impl(oob, weight_bits, contract=0, dot_order=0, f16_round=0, neg_angle=0):
    return (ctypes.c_uint32 * 6)(OOB_POLICIES[oob] if isinstance(oob, str) else int(oob), int(weight_bits), int(contract), int(dot_order),
                                 int(f16_round), NEG_ANGLE[neg_angle] if isinstance(neg_angle, str) else int(neg_angle))


def grayscale_fp(rgba, contract=0, dot_order=0, f16_round=0):
    """grayscale.wgsl:12-38 under the arithmetic switches (contract & CONTRACT_LUM, dot_order, f16_round)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    H, W = rgba.shape[:2]
    out = np.empty((H, W), dtype=np.uint16)
    impl = _impl("zero", 0, contract, dot_order, f16_round)
    lib().orc_grayscale_fp(_ptr(rgba), W, H, _ptr(out), ctypes.cast(impl, ctypes.c_void_p))
    return out


def blur_pass_fp(src, weight_bits=0, contract=0, f16_round=0):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    h, w = src.shape
    out = np.empty((h, w), dtype=np.uint16)
    impl = _impl("zero", weight_bits, contract, 0, f16_round)
    lib().orc_blur_pass_fp(_ptr(src), w, h, _ptr(out), ctypes.cast(impl, ctypes.c_void_p))
    return out


def mip_fp(src, wd=None, hd=None, weight_bits=0, f16_round=0):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    hs, ws = src.shape
    wd = max(1, ws >> 1) if wd is None else wd
    hd = max(1, hs >> 1) if hd is None else hd
    out = np.empty((hd, wd), dtype=np.uint16)
    impl = _impl("zero", weight_bits, 0, 0, f16_round)
    lib().orc_mip_fp(_ptr(src), ws, hs, _ptr(out), wd, hd, ctypes.cast(impl, ctypes.c_void_p))
    return out


def brief_rotate(code, px, py, contract=0, dot_order=0):
    """One pattern point under brief.wgsl:35-54's rotation at an angle code, before vec2i() truncates it: (rx, ry) as binary32."""
    rx, ry = ctypes.c_float(0), ctypes.c_float(0)
    lib().orc_brief_rotate(int(code), int(px), int(py), 1 if (int(contract) & CONTRACT_ROT) else 0, int(dot_order), ctypes.byref(rx),
                           ctypes.byref(ry))
    return np.float32(rx.value), np.float32(ry.value)


def extract(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, planes=False, oob="zero", weight_bits=0, y8=False, contract=0,
            dot_order=0, f16_round=0, neg_angle=0):
    """Whole frame.  Returns dict(total, corners[structured], descriptors[u32 (n,8)], gray, blur).
    oob / weight_bits / contract (bits CONTRACT_*) / dot_order / f16_round / neg_angle: the implementation-defined switches (orc_impl_t);
    y8: a one-byte-per-pixel frame."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    H, W = rgba.shape[:2]
    corners = np.zeros(max_features, dtype=CORNER_DTYPE)
    desc = np.zeros((max_features, 8), dtype=np.uint32)
    total = ctypes.c_uint32(0)
    _, ntex = level_dims(W, H, depth)
    gray = np.zeros(ntex, dtype=np.uint16) if planes else None
    blur = np.zeros(ntex, dtype=np.uint16) if planes else None
    impl = _impl(oob, weight_bits, contract, dot_order, f16_round, neg_angle)
    rc = lib().orc_extract_impl(_ptr(rgba), 1 if y8 else 0, W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
                                ctypes.cast(impl, ctypes.c_void_p), _ptr(corners), _ptr(desc), ctypes.byref(total),
                                _ptr(gray) if planes else None, _ptr(blur) if planes else None)
    if rc != 0:
        raise ValueError("orc_extract_impl: invalid arguments")
    n = min(total.value, max_features)
    return dict(total=total.value, corners=corners[:n], descriptors=desc[:n], gray=gray, blur=blur)


def brief(blur_pyr, W, H, depth, corners, oob="zero", contract=0, dot_order=0):
    """Descriptors (n, 8) u32 of the given keypoints (CORNER_DTYPE: x, y, angle, octave) over a blur pyramid as `extract(...,
    planes=True)["blur"]` returns it (orc_brief_fp: brief.wgsl:20-68 under an out-of-level policy and a rotation arithmetic).
    tools/pin_oracle.py uses it to ask what the restatement's descriptor is at an angle code somebody else computed."""
    corners = np.ascontiguousarray(corners, dtype=CORNER_DTYPE)
    blur_pyr = np.ascontiguousarray(blur_pyr, dtype=np.uint16)
    lay, ntex = level_dims(W, H, depth)
    assert blur_pyr.size == ntex
    p = _Pyramid()
    lib().orc_pyramid_layout(W, H, depth, ctypes.byref(p))
    out = np.zeros((len(corners), 8), dtype=np.uint32)
    if len(corners):
        impl = _impl(oob, 0, contract, dot_order, 0)
        lib().orc_brief_fp(_ptr(blur_pyr), ctypes.byref(p), _ptr(corners), len(corners), ctypes.cast(impl, ctypes.c_void_p), _ptr(out))
    return out


def grayscale_y8(y8):
    y8 = np.ascontiguousarray(y8, dtype=np.uint8)
    H, W = y8.shape[:2]
    out = np.empty((H, W), dtype=np.uint16)
    lib().orc_grayscale_y8(_ptr(y8), W, H, _ptr(out))
    return out


def extract_y8(y8, depth=2, threshold=20.0 / 255.0, max_features=8192, planes=False, oob="zero", weight_bits=0):
    """Y8 input variant (one byte per pixel; not in the reference's code, see orb_oracle.c)."""
    if oob != "zero" or weight_bits:
        return extract(y8, depth, threshold, max_features, planes, oob, weight_bits, y8=True)
    y8 = np.ascontiguousarray(y8, dtype=np.uint8)
    H, W = y8.shape[:2]
    corners = np.zeros(max_features, dtype=CORNER_DTYPE)
    desc = np.zeros((max_features, 8), dtype=np.uint32)
    total = ctypes.c_uint32(0)
    _, ntex = level_dims(W, H, depth)
    gray = np.zeros(ntex, dtype=np.uint16) if planes else None
    blur = np.zeros(ntex, dtype=np.uint16) if planes else None
    rc = lib().orc_extract_y8(_ptr(y8), W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
                              _ptr(corners), _ptr(desc), ctypes.byref(total),
                              _ptr(gray) if planes else None, _ptr(blur) if planes else None)
    if rc != 0:
        raise ValueError("orc_extract_y8: invalid arguments")
    n = min(total.value, max_features)
    return dict(total=total.value, corners=corners[:n], descriptors=desc[:n], gray=gray, blur=blur)


def extract_ex(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, arc=12, nms=False):
    """Opt-in extensions (arc length 9..16, 3x3 NMS); definitions in orb_oracle.h.  Not in the reference."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    H, W = rgba.shape[:2]
    corners = np.zeros(max_features, dtype=CORNER_DTYPE)
    desc = np.zeros((max_features, 8), dtype=np.uint32)
    total = ctypes.c_uint32(0)
    opt = (ctypes.c_uint32 * 3)(int(arc), 1 if nms else 0, 0)
    rc = lib().orc_extract_ex(_ptr(rgba), W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
                              ctypes.cast(opt, ctypes.c_void_p), _ptr(corners), _ptr(desc), ctypes.byref(total))
    if rc != 0:
        raise ValueError("orc_extract_ex: invalid arguments")
    n = min(total.value, max_features)
    return dict(total=total.value, corners=corners[:n], descriptors=desc[:n])


def binned_angle_code(code, bins):
    """IM-6b: the code a descriptor is rotated by when angles are quantised into `bins` bins (0: the code itself)."""
    lib().orc_binned_angle_code.restype = ctypes.c_uint32
    return int(lib().orc_binned_angle_code(ctypes.c_uint32(int(code)), ctypes.c_uint32(int(bins))))


def extract_intended(rgba, depth=2, threshold=20.0 / 255.0, max_features=8192, arc=9, nms=False, planes=False, angle_bins=0):
    """"intended" mode (IM-1..IM-8 in orb_oracle.h; angle_bins: IM-6b).  Not in the reference."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    H, W = rgba.shape[:2]
    corners = np.zeros(max_features, dtype=CORNER_DTYPE)
    desc = np.zeros((max_features, 8), dtype=np.uint32)
    total = ctypes.c_uint32(0)
    _, ntex = level_dims(W, H, depth)
    gray = np.zeros(ntex, dtype=np.uint16) if planes else None
    blur = np.zeros(ntex, dtype=np.uint16) if planes else None
    opt = (ctypes.c_uint32 * 3)(int(arc), 1 if nms else 0, int(angle_bins))
    rc = lib().orc_extract_intended(_ptr(rgba), W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
                                    ctypes.cast(opt, ctypes.c_void_p), _ptr(corners), _ptr(desc), ctypes.byref(total),
                                    _ptr(gray) if planes else None, _ptr(blur) if planes else None)
    if rc != 0:
        raise ValueError("orc_extract_intended: invalid arguments")
    n = min(total.value, max_features)
    return dict(total=total.value, corners=corners[:n], descriptors=desc[:n], gray=gray, blur=blur)


def gauss_pass(src, vertical):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    h, w = src.shape
    out = np.empty((h, w), dtype=np.uint16)
    lib().orc_gauss_pass(_ptr(src), w, h, _ptr(out), 1 if vertical else 0)
    return out


def angle_code_neg(cy, cx, neg_angle):
    """fast.wgsl:153 under orc_impl_t::neg_angle ("zero" | "wrap" | "ones")."""
    return int(lib().orc_angle_code_neg(float(np.float32(cy)), float(np.float32(cx)), NEG_ANGLE[neg_angle] if isinstance(neg_angle, str) else int(neg_angle)))


def angle_code_signed(cy, cx):
    return int(lib().orc_angle_code_signed(float(np.float32(cy)), float(np.float32(cx))))


def extract_batch(frames, depth=2, threshold=20.0 / 255.0, max_features=8192, n_threads=1, y8=False, oob="zero", weight_bits=0, contract=0,
                  dot_order=0, f16_round=0):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    F, H, W = frames.shape[:3]
    corners = np.zeros((F, max_features), dtype=CORNER_DTYPE)
    desc = np.zeros((F, max_features, 8), dtype=np.uint32)
    totals = np.zeros(F, dtype=np.uint32)
    if (oob, weight_bits, contract, dot_order, f16_round) != ("zero", 0, 0, 0, 0):
        impl = _impl(oob, weight_bits, contract, dot_order, f16_round)
        rc = lib().orc_extract_batch_impl(_ptr(frames), 1 if y8 else 0, F, W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
                                          ctypes.cast(impl, ctypes.c_void_p), _ptr(corners), _ptr(desc), _ptr(totals), int(n_threads))
        if rc != 0:
            raise ValueError("orc_extract_batch_impl failed")
        return totals, corners, desc
    fn = lib().orc_extract_batch_y8 if y8 else lib().orc_extract_batch
    rc = fn(_ptr(frames), F, W, H, depth, ctypes.c_float(np.float32(threshold)), max_features,
            _ptr(corners), _ptr(desc), _ptr(totals), int(n_threads))
    if rc != 0:
        raise ValueError("orc_extract_batch failed")
    return totals, corners, desc


def extract_intended_batch(frames, depth=2, threshold=20.0 / 255.0, max_features=8192, arc=9, nms=False, n_threads=1, angle_bins=0):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    F, H, W = frames.shape[:3]
    corners = np.zeros((F, max_features), dtype=CORNER_DTYPE)
    desc = np.zeros((F, max_features, 8), dtype=np.uint32)
    totals = np.zeros(F, dtype=np.uint32)
    opt = (ctypes.c_uint32 * 3)(int(arc), 1 if nms else 0, int(angle_bins))
    rc = lib().orc_extract_intended_batch(_ptr(frames), F, W, H, depth, ctypes.c_float(np.float32(threshold)),
                                          max_features, ctypes.cast(opt, ctypes.c_void_p), _ptr(corners), _ptr(desc),
                                          _ptr(totals), int(n_threads))
    if rc != 0:
        raise ValueError("orc_extract_intended_batch failed")
    return totals, corners, desc


def sort_keypoints(corners, descriptors=None):
    """Canonical comparison order: by (octave, y, x)."""
    order = np.lexsort((corners["x"], corners["y"], corners["octave"]))
    if descriptors is None:
        return corners[order]
    return corners[order], descriptors[order]
