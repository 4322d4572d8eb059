"""Host-side mirror of the reference's `tinyslam::orb` module over the C ABI (include/tinyorb.h).

Same names and argument meaning as src/orb.rs: `OrbConfig` (orb.rs:40-45), `OrbProgram` with
`init` (107), `write_input_image` (567), `set_threshold` (585), `extract_corners` (469),
`read_corners` (559), `read_descriptors` (563), and the POD records `CornerData` (10-17) /
`CornerDescriptor` (19-23).  Where the reference panics this raises `OrbError`.

There is no CPU fallback: if libtinyorb.so is missing or no HIP device is present, loading
or `init()` raises.
"""
import ctypes
import importlib.util
import os
from dataclasses import dataclass, field

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TINYORB_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtinyorb.so")  # TINYORB_LIB: A/B of two builds on one box

ORB_OK, ORB_EINVAL, ORB_EHIP, ORB_ECAPACITY, ORB_ESTATE = 0, 1, 2, 3, 4
ORB_PLANE_GRAY, ORB_PLANE_BLUR = 0, 1
ORB_KERNEL_COUNT = 22
ORB_FLAG_STAGED = 1
ORB_FLAG_DOUBLE_OUTPUT = 2
ORB_FLAG_NMS = 4
ORB_FLAG_INTENDED = 8
ORB_FLAG_INPUT_Y8 = 16
ORB_FLAG_SINGLE_BLOCKING_WAIT = 32  # orb_extract_corners: bounded spin, then sleep until the completion interrupt
ORB_OOB_ZERO, ORB_OOB_CLAMP, ORB_OOB_UMIN = 0, 1, 2  # OrbOptions.oob_policy
OOB_POLICIES = {"zero": ORB_OOB_ZERO, "clamp": ORB_OOB_CLAMP, "umin": ORB_OOB_UMIN}
# OrbOptions.fp_contract (CRD-13): which stages' products and sums the adapter's shader compiler fuses, and its reduction order
ORB_FP_CONTRACT_LUMINANCE, ORB_FP_CONTRACT_BLUR, ORB_FP_CONTRACT_ROTATION, ORB_FP_CONTRACT_ALL, ORB_FP_LAST_TERM_FIRST = 1, 2, 4, 7, 8
TRANSPORT_RECORD_WORDS = 10  # ORB_TRANSPORT_RECORD_BYTES / 4
SYN_GRADIENT, SYN_BLOBS, SYN_WEDGES, SYN_NOISE = 1, 2, 4, 8
SYN_ALL = 15

# orb.rs:10-17 / orb.rs:19-23 as numpy record layouts (16 B / 32 B)
CORNER_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("angle", "<u4"), ("octave", "<u4")])
MATCH_DTYPE = np.dtype([("index", "<u4"), ("distance", "<u2"), ("second", "<u2")])
ORB_MATCH_NONE = 0xFFFFFFFF
DESCRIPTOR_DTYPE = np.dtype([("bits", "u1", (32,))])

# Names every build of libtinyorb.so must export (checked by tests against include/tinyorb.h).
EXPORTS = [
    "orb_abi_version", "orb_pipeline", "orb_last_error", "orb_kernel_name", "orb_program_create", "orb_program_destroy",
    "orb_write_input_image", "orb_set_threshold", "orb_extract_corners", "orb_read_corners",
    "orb_read_descriptors", "orb_extract_batch_device", "orb_extract_batch_host", "orb_batch_sync",
    "orb_batch_counts", "orb_batch_read", "orb_batch_select_output", "orb_batch_device_buffers", "orb_level_size",
    "orb_debug_read_plane", "orb_debug_f32_to_f16", "orb_debug_angle_code", "orb_debug_rot_table", "orb_profile_enable",
    "orb_profile_reset", "orb_profile_get", "orb_synth_frames_device", "orb_copy_to_host", "orb_debug_stamps",
    "orb_match_consecutive", "orb_match_read", "orb_corner_level0_xy",
    "orb_batch_read_all", "orb_batch_compact_device", "orb_host_alloc", "orb_host_free", "orb_stream_sync",
    "orb_program_stream", "orb_pipeline_note", "orb_batch_pack", "orb_batch_fetch",
    "orb_batch_pack_transport", "orb_unpack_transport",
    "orb_node_create", "orb_node_destroy", "orb_node_last_error", "orb_node_device_count", "orb_node_program",
    "orb_node_shard", "orb_node_extract_batch", "orb_node_extract_batch_host", "orb_node_collate",
    "orb_node_read_collated", "orb_node_collate_begin", "orb_node_collate_end", "orb_node_pending",
    "orb_extract_batch_pinned", "orb_upload_sync", "orb_node_exchange_backend", "orb_node_rccl_pairs",
    "orb_write_input_image_pinned", "orb_node_set_results", "orb_node_shard_result",
]


def level0_xy(corners):
    """Centres of the keypoints' pixels in level-0 pixel units (orb_corner_level0_xy of include/tinyorb.h)."""
    s = np.left_shift(1, corners["octave"].astype(np.int64)).astype(np.float32)
    return ((corners["x"].astype(np.float32) + np.float32(0.5)) * s - np.float32(0.5),
            (corners["y"].astype(np.float32) + np.float32(0.5)) * s - np.float32(0.5))


class OrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tinyorb error %d: %s" % (code, msg))
        self.code = code


class _Extent3d(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("depth_or_array_layers", ctypes.c_uint32)]


class _Config(ctypes.Structure):
    _fields_ = [("image_size", _Extent3d), ("max_features", ctypes.c_uint32), ("hierarchy_depth", ctypes.c_uint32),
                ("initial_threshold", ctypes.c_float)]


class _Options(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("max_batch", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("fast_arc", ctypes.c_uint32), ("oob_policy", ctypes.c_uint32), ("sampler_weight_bits", ctypes.c_uint32),
                ("fp_contract", ctypes.c_uint32), ("angle_bins", ctypes.c_uint32)]


_lib = None


def _preload_hip_runtime():
    """If PyTorch is installed, bind to ITS bundled libamdhip64 so that a later `import torch`
    (bench.py, torch.distributed) shares one HIP runtime with libtinyorb instead of loading a
    second copy next to the system one."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=None):
    """Loads libtinyorb.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise OrbError(ORB_EHIP, "%s not found: run `python -m tinyslam_amd.build` (hipcc, gfx950)" % path)
    _preload_hip_runtime()
    L = ctypes.CDLL(path)
    vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
    L.orb_abi_version.restype = u32
    L.orb_last_error.restype = ctypes.c_char_p
    L.orb_last_error.argtypes = [vp]
    L.orb_kernel_name.restype = ctypes.c_char_p
    L.orb_pipeline.restype = ctypes.c_char_p
    L.orb_pipeline.argtypes = [vp]
    L.orb_pipeline_note.restype = ctypes.c_char_p
    L.orb_pipeline_note.argtypes = [vp]
    L.orb_kernel_name.argtypes = [ctypes.c_int]
    L.orb_program_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(_Options), ctypes.POINTER(vp)]
    L.orb_program_destroy.argtypes = [vp]
    L.orb_program_destroy.restype = None
    L.orb_write_input_image.argtypes = [vp, vp, sz]
    L.orb_write_input_image_pinned.argtypes = [vp, vp, sz]
    L.orb_set_threshold.argtypes = [vp, ctypes.c_float]
    L.orb_extract_corners.argtypes = [vp, ctypes.POINTER(u32)]
    L.orb_read_corners.argtypes = [vp, vp, sz]
    L.orb_read_descriptors.argtypes = [vp, vp, sz]
    L.orb_extract_batch_device.argtypes = [vp, vp, u32, vp]
    L.orb_extract_batch_host.argtypes = [vp, vp, u32]
    L.orb_batch_sync.argtypes = [vp]
    L.orb_batch_counts.argtypes = [vp, vp, u32]
    L.orb_batch_read.argtypes = [vp, u32, vp, vp, sz]
    L.orb_batch_select_output.argtypes = [vp, u32]
    L.orb_batch_device_buffers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.orb_level_size.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.orb_debug_read_plane.argtypes = [vp, u32, ctypes.c_int, u32, vp, sz]
    L.orb_debug_f32_to_f16.argtypes = [vp, vp, vp, sz]
    L.orb_debug_angle_code.argtypes = [vp, vp, vp, vp, sz]
    L.orb_debug_rot_table.argtypes = [vp, vp, sz, vp, vp]
    L.orb_match_consecutive.argtypes = [vp, u32, vp]
    L.orb_match_read.argtypes = [vp, u32, vp, ctypes.c_size_t]
    L.orb_profile_enable.argtypes = [vp, ctypes.c_int]
    L.orb_profile_reset.argtypes = [vp]
    L.orb_profile_get.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.orb_synth_frames_device.argtypes = [vp, vp, u32, u32, u32, ctypes.POINTER(vp)]
    L.orb_copy_to_host.argtypes = [vp, vp, vp, sz]
    L.orb_debug_stamps.argtypes = [vp, vp, sz]
    L.orb_corner_level0_xy.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.orb_corner_level0_xy.restype = None
    L.orb_batch_read_all.argtypes = [vp, u32, vp, vp, vp, vp, sz, vp]
    L.orb_batch_compact_device.argtypes = [vp, u32, vp, vp, vp, vp, sz, vp]
    L.orb_batch_pack.argtypes = [vp, u32, vp]
    L.orb_batch_fetch.argtypes = [vp, u32, vp, vp, vp, vp, sz, vp]
    L.orb_batch_pack_transport.argtypes = [vp, u32, u32, vp, sz, vp, vp]
    L.orb_unpack_transport.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp]
    L.orb_host_alloc.argtypes = [sz, ctypes.POINTER(vp)]
    L.orb_host_free.argtypes = [vp]
    L.orb_host_free.restype = None
    L.orb_stream_sync.argtypes = [vp, vp]
    L.orb_program_stream.argtypes = [vp]
    L.orb_program_stream.restype = vp
    L.orb_node_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_Config), ctypes.POINTER(_Options),
                                  ctypes.POINTER(vp)]
    L.orb_node_destroy.argtypes = [vp]
    L.orb_node_destroy.restype = None
    L.orb_node_last_error.argtypes = [vp]
    L.orb_node_last_error.restype = ctypes.c_char_p
    L.orb_node_device_count.argtypes = [vp]
    L.orb_node_program.argtypes = [vp, ctypes.c_int]
    L.orb_node_program.restype = vp
    L.orb_node_shard.argtypes = [vp, u32, ctypes.c_int, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.orb_node_extract_batch.argtypes = [vp, ctypes.POINTER(vp), u32]
    L.orb_node_extract_batch_host.argtypes = [vp, vp, u32]
    L.orb_node_collate.argtypes = [vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.orb_node_read_collated.argtypes = [vp, vp, vp, sz]
    L.orb_node_collate_begin.argtypes = [vp]
    L.orb_node_collate_end.argtypes = [vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.orb_node_pending.argtypes = [vp]
    L.orb_node_set_results.argtypes = [vp, ctypes.c_int]
    L.orb_node_shard_result.argtypes = [vp, ctypes.c_int, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.orb_node_exchange_backend.argtypes = [vp]
    L.orb_node_exchange_backend.restype = ctypes.c_char_p
    L.orb_node_rccl_pairs.argtypes = [vp]
    L.orb_node_rccl_pairs.restype = ctypes.c_uint64
    L.orb_extract_batch_pinned.argtypes = [vp, vp, u32]
    L.orb_upload_sync.argtypes = [vp]
    if L.orb_abi_version() != 5 and not os.environ.get("TINYORB_ALLOW_ABI"):  # (tools/ab_old_new.sh alternates libraries of other commits)
        raise OrbError(ORB_EINVAL, "libtinyorb ABI version mismatch")
    if path == LIB_PATH:
        _lib = L
    return L


class PinnedArray:
    """A numpy array over pinned, device-visible host memory from orb_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        L = load_library()
        self._lib = L
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        self.nbytes = max(n * dtype.itemsize, 1)
        ptr = ctypes.c_void_p()
        rc = L.orb_host_alloc(self.nbytes, ctypes.byref(ptr))
        if rc != ORB_OK:
            raise OrbError(rc, (L.orb_last_error(None) or b"").decode())
        self.ptr = ptr
        buf = (ctypes.c_uint8 * self.nbytes).from_address(ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def close(self):
        if self.ptr is not None:
            self.array = None
            self._lib.orb_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatch:
    """Pinned host destination of orb_batch_read_all: counts[n], offsets[n+1], corners[capacity], descriptors[capacity]."""

    def __init__(self, n_frames, capacity):
        self.n_frames, self.capacity = n_frames, capacity
        self._c = PinnedArray((n_frames,), np.uint32)
        self._o = PinnedArray((n_frames + 1,), np.uint64)
        self._k = PinnedArray((capacity,), CORNER_DTYPE)
        self._d = PinnedArray((capacity, 8), np.uint32)
        self.counts, self.offsets, self.corners, self.descriptors = self._c.array, self._o.array, self._k.array, self._d.array

    def frame(self, f):
        lo, hi = int(self.offsets[f]), int(self.offsets[f + 1])
        return self.corners[lo:hi], self.descriptors[lo:hi]

    def close(self):
        self.counts = self.offsets = self.corners = self.descriptors = None
        for a in (self._c, self._o, self._k, self._d):
            a.close()


@dataclass
class Extent3d:
    """wgpu::Extent3d as used by OrbConfig.image_size."""
    width: int
    height: int
    depth_or_array_layers: int = 1


@dataclass
class OrbConfig:
    """orb.rs:40-45."""
    image_size: Extent3d
    max_features: int = 8192
    hierarchy_depth: int = 2
    initial_threshold: float = 20.0 / 255.0
    # build-side options (no counterpart in the reference)
    device: int = 0
    max_batch: int = 1
    flags: int = 0
    fast_arc: int = 0  # 0 -> 12 (reference; 9 with ORB_FLAG_INTENDED); 9..16 opt-in
    # the two implementation-defined points of the reference's WGSL as switches (include/tinyorb.h, OrbOptions)
    oob_policy: int = 0           # ORB_OOB_ZERO / ORB_OOB_CLAMP / ORB_OOB_UMIN: textureLoad outside the level
    sampler_weight_bits: int = 0  # 0: exact bilinear weights; n: weights held in n fractional bits
    angle_bins: int = 0   # ORB_FLAG_INTENDED, IM-6b: 0 = rotate by the milliradian code; 8..6284 = by the centre of the angle bin
    fp_contract: int = 0  # CRD-13: mask of ORB_FP_CONTRACT_LUMINANCE / _BLUR / _ROTATION (that stage's products and sums as fmas) and ORB_FP_LAST_TERM_FIRST


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OrbProgram:
    """orb.rs:47-51 + impl 106-590.  Construct, then `init()` (the reference builds the struct
    by literal and calls `init`, orb.rs:107)."""

    def __init__(self, config: OrbConfig):
        self.config = config
        self._h = None
        self._lib_obj = None

    @property
    def _lib(self):
        if self._lib_obj is None:
            raise OrbError(ORB_ESTATE, "OrbProgram.init() has not been called")
        return self._lib_obj

    # ---- lifetime -------------------------------------------------------------------------
    def init(self):
        L = load_library()
        c = self.config
        cfg = _Config(_Extent3d(c.image_size.width, c.image_size.height, c.image_size.depth_or_array_layers),
                      c.max_features, c.hierarchy_depth, float(np.float32(c.initial_threshold)))
        opt = _Options(c.device, c.max_batch, c.flags, c.fast_arc, c.oob_policy, c.sampler_weight_bits, c.fp_contract, c.angle_bins)
        h = ctypes.c_void_p()
        rc = L.orb_program_create(ctypes.byref(cfg), ctypes.byref(opt), ctypes.byref(h))
        if rc != ORB_OK:
            raise OrbError(rc, (L.orb_last_error(None) or b"").decode())
        self._h, self._lib_obj = h, L
        return self

    def close(self):
        if self._h is not None:
            if not getattr(self, "_borrowed", False):  # programs of an OrbNode belong to the node
                self._lib.orb_program_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self if self._h is not None else self.init()

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != ORB_OK and rc not in allow:
            raise OrbError(rc, (self._lib.orb_last_error(self._h) or b"").decode())
        return rc

    def _handle(self):
        if self._h is None:
            raise OrbError(ORB_ESTATE, "OrbProgram.init() has not been called")
        return self._h

    # ---- the reference's six methods --------------------------------------------------------
    def write_input_image(self, data):
        """orb.rs:567: tightly packed RGBA8 bytes (any buffer / uint8 array of 4*W*H bytes)."""
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview))
                                 else np.asarray(data, dtype=np.uint8))
        self._check(self._lib.orb_write_input_image(self._handle(), _ptr(a), a.size))

    def write_input_image_pinned(self, frame):
        """orb_write_input_image_pinned: `frame` is a numpy view of PINNED memory (PinnedArray.array); asynchronous, one image
        may be written ahead of extract_corners (upload_sync() waits until the array may be reused)."""
        a = np.asarray(frame)
        assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
        self._check(self._lib.orb_write_input_image_pinned(self._handle(), ctypes.c_void_p(a.ctypes.data), a.size))

    def set_threshold(self, threshold):
        """orb.rs:585"""
        self._check(self._lib.orb_set_threshold(self._handle(), float(np.float32(threshold))))

    def extract_corners(self):
        """orb.rs:469: returns the RAW counter (may exceed max_features, like the reference)."""
        n = ctypes.c_uint32(0)
        self._check(self._lib.orb_extract_corners(self._handle(), ctypes.byref(n)), allow=(ORB_ECAPACITY,))
        return n.value

    def read_corners(self, dst):
        """orb.rs:559: fills a CORNER_DTYPE array (the reference's &mut [CornerData])."""
        assert dst.dtype == CORNER_DTYPE and dst.flags.c_contiguous
        self._check(self._lib.orb_read_corners(self._handle(), _ptr(dst), dst.size))
        return dst

    def read_descriptors(self, dst):
        """orb.rs:563: fills a DESCRIPTOR_DTYPE (or uint8 (n,32) / uint32 (n,8)) array."""
        assert dst.flags.c_contiguous and dst.nbytes % 32 == 0
        self._check(self._lib.orb_read_descriptors(self._handle(), _ptr(dst), dst.nbytes // 32))
        return dst

    # ---- conveniences over the six methods -------------------------------------------------
    def extract(self, rgba):
        """One frame in -> (total, corners[stored], descriptors u32 (stored, 8))."""
        self.write_input_image(rgba)
        total = self.extract_corners()
        n = min(total, self.config.max_features)
        corners = self.read_corners(np.zeros(n, dtype=CORNER_DTYPE))
        desc = self.read_descriptors(np.zeros((n, 8), dtype=np.uint32))
        return total, corners, desc

    # ---- batched mode -----------------------------------------------------------------------
    def extract_batch_device(self, frames_dev_ptr, n_frames, stream=None):
        self._check(self._lib.orb_extract_batch_device(self._handle(), ctypes.c_void_p(frames_dev_ptr), n_frames,
                                                       ctypes.c_void_p(stream) if stream else None))

    def extract_batch_host(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        self._check(self._lib.orb_extract_batch_host(self._handle(), _ptr(a), a.shape[0]))

    def extract_batch_pinned(self, frames, n_frames=None):
        """orb_extract_batch_pinned: `frames` is a numpy view of PINNED memory (PinnedArray.array); asynchronous."""
        a = np.asarray(frames)
        assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
        self._check(self._lib.orb_extract_batch_pinned(self._handle(), ctypes.c_void_p(a.ctypes.data),
                                                       a.shape[0] if n_frames is None else n_frames))

    def upload_sync(self):
        self._check(self._lib.orb_upload_sync(self._handle()))

    def batch_sync(self):
        self._check(self._lib.orb_batch_sync(self._handle()))

    def batch_counts(self, n_frames):
        out = np.zeros(n_frames, dtype=np.uint32)
        self._check(self._lib.orb_batch_counts(self._handle(), _ptr(out), n_frames))
        return out

    def batch_read(self, frame, n):
        corners = np.zeros(n, dtype=CORNER_DTYPE)
        desc = np.zeros((n, 8), dtype=np.uint32)
        self._check(self._lib.orb_batch_read(self._handle(), frame, _ptr(corners), _ptr(desc), n))
        return corners, desc

    def batch_read_all(self, n_frames, out=None, stream=None, sync=True):
        """Bulk read-back of the last batch (orb_batch_read_all): returns a `HostBatch` whose arrays live in pinned host
        memory (reused when `out` is passed).  The stored records of all frames are packed back to back in frame
        order; frame f owns records [offsets[f], offsets[f+1])."""
        cap = self.config.max_features
        hb = out if out is not None else HostBatch(n_frames, n_frames * cap)
        assert hb.n_frames >= n_frames
        self._check(self._lib.orb_batch_read_all(self._handle(), n_frames, hb.counts.ctypes.data, hb.offsets.ctypes.data,
                                                 hb.corners.ctypes.data, hb.descriptors.ctypes.data, hb.capacity,
                                                 ctypes.c_void_p(stream) if stream else None))
        if sync:
            if stream:
                self.stream_sync(stream)
            else:
                self.batch_sync()
        return hb

    def batch_pack(self, n_frames, stream=None):
        """Step 1 of the streaming read-back (orb_batch_pack): pack the last batch of the selected output set on the device."""
        self._check(self._lib.orb_batch_pack(self._handle(), n_frames, ctypes.c_void_p(stream) if stream else None))

    def batch_fetch(self, out_set, out, stream=None):
        """Step 2 (orb_batch_fetch): waits for that set's pack, fills out.counts / out.offsets and enqueues the exact-size
        copies of the records into the pinned HostBatch `out` on `stream`; returns the total record count."""
        self._check(self._lib.orb_batch_fetch(self._handle(), out_set, out.counts.ctypes.data, out.offsets.ctypes.data,
                                              out.corners.ctypes.data, out.descriptors.ctypes.data, out.capacity,
                                              ctypes.c_void_p(stream) if stream else None))

    def batch_pack_transport(self, out_set, n_frames, dst_dev_ptr, capacity_records, offsets_dev_ptr=None, stream=None):
        """orb_batch_pack_transport: the stored records of output set `out_set` as 40-byte transport records in device
        memory, enqueued on `stream` (the caller orders it behind the batch)."""
        self._check(self._lib.orb_batch_pack_transport(self._handle(), out_set, n_frames, ctypes.c_void_p(dst_dev_ptr),
                                                       capacity_records, ctypes.c_void_p(offsets_dev_ptr) if offsets_dev_ptr else None,
                                                       ctypes.c_void_p(stream) if stream else None))

    def unpack_transport(self, src_dev_ptr, src_first, count, dst_first, corners_dev_ptr, desc_dev_ptr, stream=None):
        """orb_unpack_transport: runs of transport records -> CornerData / CornerDescriptor arrays on this device."""
        a = [np.ascontiguousarray(v, dtype=np.uint64) for v in (src_first, count, dst_first)]
        assert a[0].shape == a[1].shape == a[2].shape and a[0].ndim == 1
        self._check(self._lib.orb_unpack_transport(self._handle(), ctypes.c_void_p(src_dev_ptr), a[0].shape[0], _ptr(a[0]), _ptr(a[1]),
                                                   _ptr(a[2]), ctypes.c_void_p(corners_dev_ptr), ctypes.c_void_p(desc_dev_ptr),
                                                   ctypes.c_void_p(stream) if stream else None))

    def stream_sync(self, stream=None):
        self._check(self._lib.orb_stream_sync(self._handle(), ctypes.c_void_p(stream) if stream else None))

    def match_consecutive(self, n_frames, stream=None):
        """Hamming-match frame f against f+1 for the first n_frames of the last batch (not in the reference)."""
        self._check(self._lib.orb_match_consecutive(self._handle(), n_frames, ctypes.c_void_p(stream) if stream else None))

    def match_read(self, frame, n):
        out = np.zeros(n, dtype=MATCH_DTYPE)
        self._check(self._lib.orb_match_read(self._handle(), frame, _ptr(out), n))
        return out

    def batch_select_output(self, slot):
        self._check(self._lib.orb_batch_select_output(self._handle(), slot))

    def batch_device_buffers(self):
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.orb_batch_device_buffers(self._handle(), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    # ---- inspection / measurement -------------------------------------------------------------
    def pipeline(self):
        return self._lib.orb_pipeline(self._handle()).decode()

    def pipeline_note(self):
        """Why the staged kernels were chosen although nobody asked for them ('' otherwise)."""
        return self._lib.orb_pipeline_note(self._handle()).decode()

    def level_size(self, level):
        w, h = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._lib.orb_level_size(self._handle(), level, ctypes.byref(w), ctypes.byref(h)))
        return w.value, h.value

    def read_plane(self, kind, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), dtype=np.uint16)
        self._check(self._lib.orb_debug_read_plane(self._handle(), frame, kind, level, _ptr(out), out.size))
        return out

    def device_f32_to_f16(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        out = np.zeros(src.shape, dtype=np.uint16)
        self._check(self._lib.orb_debug_f32_to_f16(self._handle(), _ptr(src), _ptr(out), src.size))
        return out

    def device_angle_code(self, cy, cx):
        cy = np.ascontiguousarray(cy, dtype=np.float32)
        cx = np.ascontiguousarray(cx, dtype=np.float32)
        out = np.zeros(cy.shape, dtype=np.uint32)
        self._check(self._lib.orb_debug_angle_code(self._handle(), _ptr(cy), _ptr(cx), _ptr(out), cy.size))
        return out

    def rot_table(self):
        """The rotated-pattern table as (int16 array [codes, 64, 4, 2] of byte offsets, pitch)."""
        codes, pitch = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self._check(self._lib.orb_debug_rot_table(self._handle(), None, 0, ctypes.byref(codes), ctypes.byref(pitch)))
        out = np.zeros((codes.value, 64, 4, 2), dtype=np.int16)
        self._check(self._lib.orb_debug_rot_table(self._handle(), _ptr(out), out.size, None, None))
        return out, pitch.value

    def profile_enable(self, on=True):
        self._check(self._lib.orb_profile_enable(self._handle(), 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.orb_profile_reset(self._handle()))

    def profile(self):
        """{kernel name: (total_ms, launches)} for kernels launched since the last reset."""
        out = {}
        for i in range(ORB_KERNEL_COUNT):
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            self._check(self._lib.orb_profile_get(self._handle(), i, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[self._lib.orb_kernel_name(i).decode()] = (ms.value, n.value)
        return out

    def synth_frames_device(self, n_frames, seed0, flags=SYN_ALL, frames_dev_ptr=None):
        """Generates synthetic frames on the device; returns the device address."""
        out = ctypes.c_void_p()
        self._check(self._lib.orb_synth_frames_device(self._handle(), ctypes.c_void_p(frames_dev_ptr) if frames_dev_ptr else None,
                                                      n_frames, seed0 & 0xFFFFFFFF, flags, ctypes.byref(out)))
        return out.value

    def debug_stamps(self, n_workgroups):
        out = np.zeros((n_workgroups, 6), dtype=np.uint64)
        self._check(self._lib.orb_debug_stamps(self._handle(), _ptr(out), out.size))
        return out

    def copy_to_host(self, dev_ptr, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        self._check(self._lib.orb_copy_to_host(self._handle(), _ptr(out), ctypes.c_void_p(dev_ptr), nbytes))
        return out


class OrbNode:
    """One process, several GPUs of one node (include/tinyorb.h, orb_node_*): one OrbProgram per device, contiguous
    frame shards, collate on the first device over RCCL.  Not in the reference (single wgpu device, orb.rs:47-51)."""

    def __init__(self, config: OrbConfig, devices):
        self.config = config
        self.devices = list(devices)
        self._h = None
        self._lib = None

    def init(self):
        L = load_library()
        c = self.config
        cfg = _Config(_Extent3d(c.image_size.width, c.image_size.height, c.image_size.depth_or_array_layers),
                      c.max_features, c.hierarchy_depth, float(np.float32(c.initial_threshold)))
        opt = _Options(0, c.max_batch, c.flags, c.fast_arc, c.oob_policy, c.sampler_weight_bits, c.fp_contract, c.angle_bins)
        devs = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        rc = L.orb_node_create(devs, len(self.devices), ctypes.byref(cfg), ctypes.byref(opt), ctypes.byref(h))
        if rc != ORB_OK:
            raise OrbError(rc, (L.orb_node_last_error(None) or b"").decode())
        self._h, self._lib = h, L
        return self

    def close(self):
        if self._h is not None:
            self._lib.orb_node_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self if self._h is not None else self.init()

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != ORB_OK:
            raise OrbError(rc, (self._lib.orb_node_last_error(self._h) or b"").decode())

    def device_count(self):
        return self._lib.orb_node_device_count(self._h)

    def program(self, rank):
        """Borrowed OrbProgram of one rank (do not close it)."""
        h = self._lib.orb_node_program(self._h, rank)
        if not h:
            raise OrbError(ORB_EINVAL, "no such rank")
        p = OrbProgram(self.config)
        p._h, p._lib_obj, p._borrowed = ctypes.c_void_p(h), self._lib, True
        return p

    def shard(self, n_frames, rank):
        lo, hi = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._lib.orb_node_shard(self._h, n_frames, rank, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def extract_batch(self, frames_dev_ptrs, n_frames):
        arr = (ctypes.c_void_p * len(frames_dev_ptrs))(*[ctypes.c_void_p(x) for x in frames_dev_ptrs])
        self._check(self._lib.orb_node_extract_batch(self._h, arr, n_frames))

    def extract_batch_host(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        self._check(self._lib.orb_node_extract_batch_host(self._h, _ptr(a), a.shape[0]))

    def collate(self, n_frames):
        """-> counts (n,), offsets (n+1,), device addresses of the packed corners / descriptors on the first device."""
        counts = np.zeros(n_frames, dtype=np.uint32)
        offsets = np.zeros(n_frames + 1, dtype=np.uint64)
        c, d = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.orb_node_collate(self._h, _ptr(counts), _ptr(offsets), ctypes.byref(c), ctypes.byref(d)))
        return counts, offsets, c.value, d.value

    def collate_begin(self):
        """Stage 2 of the oldest job that has not begun it: enqueue its exchange (waits only for its counters)."""
        self._check(self._lib.orb_node_collate_begin(self._h))

    def collate_end(self, n_frames):
        """Stage 3 of the oldest job: blocks until its collated result is on the first device; returns like collate()."""
        counts = np.zeros(n_frames, dtype=np.uint32)
        offsets = np.zeros(n_frames + 1, dtype=np.uint64)
        c, d = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.orb_node_collate_end(self._h, _ptr(counts), _ptr(offsets), ctypes.byref(c), ctypes.byref(d)))
        return counts, offsets, c.value, d.value

    def pending(self):
        return self._lib.orb_node_pending(self._h)

    def set_results(self, sharded):
        """Results collated on the first device (False, the default) or left packed on the device that computed them (True)."""
        self._check(self._lib.orb_node_set_results(self._h, 1 if sharded else 0))

    def shard_result(self, rank):
        """Sharded results: (frames, records, device addresses of rank's packed corners / descriptors) of the job ended last."""
        nf, nr, c, d = ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.orb_node_shard_result(self._h, rank, ctypes.byref(nf), ctypes.byref(nr), ctypes.byref(c), ctypes.byref(d)))
        return nf.value, nr.value, c.value, d.value

    def exchange_backend(self):
        """'rccl', 'rccl-self' (TINYORB_NODE_LOOPBACK=2), 'copies' (TINYORB_NODE_LOOPBACK=1) or 'none' (one device)."""
        return self._lib.orb_node_exchange_backend(self._h).decode()

    def rccl_pairs(self):
        """ncclSend + ncclRecv pairs this node has enqueued so far."""
        return int(self._lib.orb_node_rccl_pairs(self._h))

    def read_collated(self, total):
        corners = np.zeros(total, dtype=CORNER_DTYPE)
        desc = np.zeros((total, 8), dtype=np.uint32)
        self._check(self._lib.orb_node_read_collated(self._h, _ptr(corners), _ptr(desc), total))
        return corners, desc
