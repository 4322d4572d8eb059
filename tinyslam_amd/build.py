"""Builds libtinyorb.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m tinyslam_amd.build [--force]

-ffp-contract=off is part of the contract, not an optimisation choice: the kernels must round
every binary32 product and sum on its own to stay bit-identical with the CPU restatement the tests check against.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libtinyorb.so")
# Translation units: (source, extra flags, object name).  k_front's ~160 instances are spread over eight units, one per arithmetic form
# of the adapter's shader compiler (csrc/orb_front_launch.h), compiled in parallel: a kernel is launched from the unit that instantiated
# it, so the objects link as plain host code (no relocatable device code).
UNITS = [("orb_api.hip", [], "orb_api.o"), ("orb_node.hip", [], "orb_node.o")] + \
        [("orb_front_inst.hip", ["-DTINYORB_FRONT_FP=%d" % f], "orb_front_fp%d.o" % f) for f in range(8)]
OBJ_DIR = os.path.join(PKG_DIR, "_obj")
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
    # wave-aggregate LDS/global atomic adds of lane-varying amounts with a DPP scan; the default
    # ("Iterative") serialises over the active lanes with a scalar loop of ~8 instructions per lane
    "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP",
    # no SLP vectorisation: it pairs scalar binary32 products and sums of k_brief_t's literal pattern into v_pk_mul/add_f32
    # (half rate on gfx950) behind hundreds of v_mov that assemble their operands; the kernels that want packed
    # arithmetic ask for it themselves (float2 / half2 types)
    "-fno-slp-vectorize",
    # MFMA results in VGPRs (the register file is unified on gfx950): the one kernel that uses the matrix cores, k_match_mfma, feeds
    # every result to the vector unit at once, and from an AGPR that is one v_accvgpr_read per value on top of the three it costs
    "-mllvm", "-amdgpu-mfma-vgpr-form",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "tinyorb.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def source_hash():
    """SHA-256 over the sources that decide what the kernels do and how they are launched -- the kernel headers (csrc/*.h,
    *.inc), orb_api.hip and the compiler flags -- in name order: stamps measurements that belong to one state of the kernels
    (profiles/traffic*.json; bench.py refuses a stamp that is not the tree's).  The node layer (orb_node.hip) and the public
    header hold no device code and are not part of it."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")) or f in ("orb_api.hip", "orb_front_inst.hip"))
    h.update(" ".join(HIPCC_FLAGS + ["-shared"]).encode())
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_lib(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    flags = list(HIPCC_FLAGS)
    if os.environ.get("TINYORB_BUILD_STAMPS"):  # diagnostic build: in-kernel cycle stamps (tools/stamps.py)
        flags.append("-DTINYORB_STAMPS")
    flags += os.environ.get("TINYORB_BUILD_EXTRA", "").split()  # experiments: extra compiler flags (e.g. -DTINYORB_B1_UNALIGNED)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_unit(unit):
        src, extra, obj = unit
        cmd = [hipcc] + flags + extra + ["-c", "-o", os.path.join(OBJ_DIR, obj), os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return os.path.join(OBJ_DIR, obj)

    jobs = jobs or int(os.environ.get("TINYORB_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
