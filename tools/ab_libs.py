"""Library-against-library A/B through the stable subset of the C ABI (round 5).  tools/ab_old_new.sh alternates libraries under the
TREE's bench.py, which needs the tree's ABI; libraries built from commits of earlier rounds lack newer entry points, so this harness
binds only what every ABI since round 2 has -- orb_program_create (OrbConfig + a 32-byte OrbOptions: device, max_batch, flags, fast_arc,
the rest zero), orb_synth_frames_device, orb_extract_batch_device, orb_batch_sync, orb_profile_* -- and times BASELINE configs[3]'s batch
(256 x 1280x720, device-resident, seeds 1000..) the way bench.py does: untimed steps for 300 ms, 3 warm-up steps, 5 repeats of 20 steps
between synchronisations, the median; then one more pass with the library's own HIP events per kernel.

    python tools/ab_libs.py <rounds> name1 name2 ...     (libraries tinyslam_amd/libtinyorb_<name>.so, alternated <rounds> times)
One process per measurement (a fresh HIP context each time), started by this script."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, B, DEPTH, CAP, THR, SEED0 = 1280, 720, 256, 2, 8192, 20.0 / 255.0, 1000


class Cfg(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint32), ("h", ctypes.c_uint32), ("d", ctypes.c_uint32), ("max_features", ctypes.c_uint32),
                ("depth", ctypes.c_uint32), ("thr", ctypes.c_float)]


class Opt(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("max_batch", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("fast_arc", ctypes.c_uint32),
                ("rest", ctypes.c_uint32 * 4)]


def measure(path, steps=20, repeats=5):
    L = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    L.orb_program_create.argtypes = [ctypes.POINTER(Cfg), ctypes.POINTER(Opt), ctypes.POINTER(vp)]
    L.orb_synth_frames_device.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp)]
    L.orb_extract_batch_device.argtypes = [vp, vp, ctypes.c_uint32, vp]
    L.orb_batch_sync.argtypes = [vp]
    L.orb_profile_enable.argtypes = [vp, ctypes.c_int]
    L.orb_profile_reset.argtypes = [vp]
    L.orb_profile_get.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.orb_kernel_name.argtypes = [ctypes.c_int]
    L.orb_kernel_name.restype = ctypes.c_char_p
    L.orb_last_error.argtypes = [vp]
    L.orb_last_error.restype = ctypes.c_char_p
    L.orb_program_destroy.argtypes = [vp]
    cfg, opt, h, dev = Cfg(W, H, 1, CAP, DEPTH, THR), Opt(0, B, 0, 0), vp(), vp()

    def ck(rc):
        if rc not in (0, 3):
            raise RuntimeError("%s: rc %d: %s" % (path, rc, (L.orb_last_error(h) or b"").decode()))
    ck(L.orb_program_create(ctypes.byref(cfg), ctypes.byref(opt), ctypes.byref(h)))
    ck(L.orb_synth_frames_device(h, None, B, SEED0, 15, ctypes.byref(dev)))
    ck(L.orb_batch_sync(h))

    def run(n):
        for _ in range(n):
            ck(L.orb_extract_batch_device(h, dev, B, None))
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        run(1)
        ck(L.orb_batch_sync(h))
    run(3)
    reps = []
    for _ in range(repeats):
        ck(L.orb_batch_sync(h))
        t0 = time.perf_counter()
        run(steps)
        ck(L.orb_batch_sync(h))
        reps.append((time.perf_counter() - t0) / steps * 1e3)
    ck(L.orb_profile_enable(h, 1))
    ck(L.orb_profile_reset(h))
    run(steps)
    ck(L.orb_batch_sync(h))
    prof = {}
    for i in range(32):
        name = L.orb_kernel_name(i)
        if not name:
            break
        ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
        if L.orb_profile_get(h, i, ctypes.byref(ms), ctypes.byref(n)) == 0 and n.value:
            prof[name.decode()] = round(ms.value / steps, 4)
    L.orb_program_destroy(h)
    return {"ms_per_step": sorted(reps)[len(reps) // 2], "min": min(reps), "max": max(reps), "kernels": prof}


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        print(json.dumps(measure(sys.argv[2])))
        sys.exit(0)
    rounds, names = int(sys.argv[1]), sys.argv[2:]
    for r in range(rounds):
        for n in names:
            path = os.path.join(ROOT, "tinyslam_amd", "libtinyorb_%s.so" % n)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", path], capture_output=True, text=True, timeout=300)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if out.returncode or not line:
                print("%s: FAILED %s" % (n, out.stderr[-400:]))
                continue
            d = json.loads(line[-1])
            print("%s round %d: %.4f ms per step (%.4f .. %.4f) %s" % (n, r + 1, d["ms_per_step"], d["min"], d["max"], d["kernels"]), flush=True)
