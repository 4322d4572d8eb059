"""CPU checks of the drop-in boundary: libtinyorb.so loads, exports every symbol that
include/tinyorb.h declares, keeps the reference's record layouts, validates its arguments, and --
without a GPU -- FAILS LOUDLY instead of falling back to anything (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tinyorb.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(tinyorb):
    L = tinyorb.load_library()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libtinyorb.so does not export %s" % n
    assert sorted(tinyorb.EXPORTS) == names
    assert L.orb_abi_version() == 5


def test_record_layouts_match_reference(tinyorb):
    # orb.rs:10-17 CornerData: four u32 = 16 bytes; orb.rs:19-23 CornerDescriptor: [u8; 32]
    assert tinyorb.CORNER_DTYPE.itemsize == 16
    assert [tinyorb.CORNER_DTYPE.fields[k][1] for k in ("x", "y", "angle", "octave")] == [0, 4, 8, 12]
    assert tinyorb.DESCRIPTOR_DTYPE.itemsize == 32
    # orb.rs:40-45 OrbConfig: Extent3d (3 x u32), max_features, hierarchy_depth, initial_threshold
    assert ctypes.sizeof(tinyorb._Config) == 24
    assert tinyorb._Config.max_features.offset == 12 and tinyorb._Config.initial_threshold.offset == 20
    assert ctypes.sizeof(tinyorb._Options) == 32
    # the switches took reserved words one after the other: a zero-initialised OrbOptions has always meant the defaults
    assert [getattr(tinyorb._Options, k).offset for k in ("oob_policy", "sampler_weight_bits", "fp_contract", "angle_bins")] == [16, 20, 24, 28]
    fields = re.search(r"typedef struct OrbOptions \{(.*?)\} OrbOptions;", open(HEADER).read(), re.S).group(1)
    order = re.findall(r"^\s*u?int32_t\s+(\w+)", fields, re.M)
    assert order == ["device", "max_batch", "flags", "fast_arc", "oob_policy", "sampler_weight_bits", "fp_contract", "angle_bins"]


def test_header_constants_match_python_mirror(tinyorb):
    text = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(ORB_[A-Z_0-9]+)\s+(\d+)u?\b", text))
    for name in ("ORB_OK", "ORB_EINVAL", "ORB_EHIP", "ORB_ECAPACITY", "ORB_ESTATE", "ORB_PLANE_GRAY",
                 "ORB_PLANE_BLUR", "ORB_KERNEL_COUNT", "ORB_FLAG_STAGED"):
        assert int(consts[name]) == getattr(tinyorb, name), name
    assert int(consts["ORB_MAX_HIERARCHY_DEPTH"]) == 10


def _create(tinyorb, W=64, H=48, depth=2, cap=128, thr=0.1, dl=1):
    L = tinyorb.load_library()
    cfg = tinyorb._Config(tinyorb._Extent3d(W, H, dl), cap, depth, thr)
    opt = tinyorb._Options(0, 1, 0)
    h = ctypes.c_void_p()
    rc = L.orb_program_create(ctypes.byref(cfg), ctypes.byref(opt), ctypes.byref(h))
    return L, rc, h


@pytest.mark.parametrize("kw", [dict(W=0), dict(H=0), dict(depth=0), dict(depth=11), dict(cap=0), dict(thr=-1.0),
                                dict(dl=2)])
def test_create_rejects_bad_config(tinyorb, kw):
    """The reference panics on these (wgpu validation, label arrays of 10 entries: orb.rs:66-67);
    the C ABI returns ORB_EINVAL with a message and no handle."""
    L, rc, h = _create(tinyorb, **kw)
    assert rc == tinyorb.ORB_EINVAL and not h.value
    assert len(L.orb_last_error(None)) > 0


def test_null_arguments(tinyorb):
    L = tinyorb.load_library()
    assert L.orb_program_create(None, None, None) == tinyorb.ORB_EINVAL
    assert L.orb_set_threshold(None, 0.5) == tinyorb.ORB_EINVAL
    assert L.orb_extract_corners(None, None) == tinyorb.ORB_EINVAL
    L.orb_program_destroy(None)  # no-op
    assert L.orb_kernel_name(0) == b"k_grayscale" and L.orb_kernel_name(99) == b""


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_fails_loudly(tinyorb):
    """No CPU fallback: without a HIP device init() raises ORB_EHIP."""
    prog = tinyorb.OrbProgram(tinyorb.OrbConfig(tinyorb.Extent3d(64, 48)))
    with pytest.raises(tinyorb.OrbError) as e:
        prog.init()
    assert e.value.code == tinyorb.ORB_EHIP
    with pytest.raises(tinyorb.OrbError):
        prog.extract_corners()  # never initialised


def test_missing_library_fails_loudly(tinyorb, tmp_path):
    with pytest.raises(tinyorb.OrbError):
        tinyorb.load_library(str(tmp_path / "libtinyorb.so"))


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/ (it is the checker)."""
    pkg = os.path.join(ROOT, "tinyslam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "oracle/", "orb_oracle", "orb_numpy", "orc_"):
                    assert needle not in text, "%s mentions %r" % (os.path.join(dirpath, f), needle)
    assert "orc_" not in open(HEADER).read() and "oracle/" not in open(HEADER).read()
