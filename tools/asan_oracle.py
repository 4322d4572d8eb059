"""The restatement's entry points (round 5's new ones among them) under AddressSanitizer + UBSan, on the CPU (the GPU pool has no sanitizers):
    make -C oracle _build/liborb_oracle_asan.so && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_oracle.py
TEST INFRASTRUCTURE."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orb_oracle as o
o._LIB_PATH = os.path.join(os.path.dirname(o.__file__), '_build', 'liborb_oracle_asan.so')
import numpy as np
f = o.synth_frame(333, 211, 3, 15)
for kw in (dict(), dict(neg_angle="wrap"), dict(neg_angle="ones", contract=4), dict(contract=7), dict(contract=5, dot_order=1, f16_round=1), dict(oob="umin", weight_bits=8, contract=2)):
    r = o.extract(f, depth=3, planes=True, **kw); print(kw, r["total"])
r = o.extract_intended(f, depth=2, nms=True, angle_bins=1024); print("intended bins", r["total"])
r = o.extract_ex(f, depth=2, arc=9, nms=True); print("ex", r["total"])
t, c, d = o.extract_batch(np.stack([f, f]), depth=2, n_threads=2, contract=7, dot_order=1); print("batch", t)
print(o.angle_code_neg(-1.0, 0.5, "wrap"), o.brief_rotate(2214, -3, 4, 4, 1), o.binned_angle_code(6283, 30), o.f32_to_f16(0.3333, 1))
g = o.grayscale_fp(f, 1, 1, 1); b = o.blur_pass_fp(g, 8, 2, 1); m = o.mip_fp(g, weight_bits=8, f16_round=1); print(g.shape, b.shape, m.shape)
print("asan run ok")
