"""ctypes binding of libvjhip.so — the host-side mirror of the reference's clif/clod
interface for the detect path.

The reference's public functions (clod.h:61-81, clif.h:42-73) take OpenCV/OpenCL
types; this module keeps their names, argument order and meaning on top of the C ABI
declared in include/vj.h, with numpy arrays standing in for IplImage / CvMat:

    clodInitEnvironment(device_index)            -> Environment
    clodInitBuffers(env, (width, height))
    clodDetectObjects(image, cascade, env, min_window_size, max_window_size,
                      min_neighbors, flags, use_opencl=True) -> DetectResult
    clifIntegral(image, env)                     -> (sum, square_sum)
    clodReleaseBuffers(env); clodReleaseEnvironment(env)

There is no CPU path here: `use_opencl=False` selects the WINDOW SET of the reference's
CPU variants (their skip after a stage-0 reject), still evaluated on the device, and a
missing/unbuildable libvjhip.so or a missing GPU raises VjError — nothing in this package
falls back to the oracle or to numpy.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from .build import LIB_PATH, build_lib

VJ_MAX_STAGES = 64
VJ_FLAG_COUNTERS = 1 << 0
VJ_FLAG_SIGNED_MEAN = 1 << 1
VJ_FLAG_SKIP_LIST = 1 << 2     # the CLOD_PER_STAGE_ITERATIONS CPU variant's skip over the flattened window list (clod.cpp:729-732)
VJ_FLAG_SKIP_ROW = 1 << 3      # the plain CPU variant: round() positions, skip inside a row (clod.cpp:1409-1432)
VJ_FLAG_GRID_F64 = 1 << 4      # + one of the two above: the same loop of the block variant, whose step is a double (clod.cpp:862)
VJ_FLAG_TILTED_AS_UPRIGHT = 1 << 5   # clod profile: <tilted>1 rectangles count as upright ones, as in the reference (clod.cpp:448-492); else refused

# clod_flags of the reference (clod.h:17-19).  They select among the reference's CPU
# evaluators; the HIP path has one evaluator, so they are accepted and ignored.
CLOD_PRECOMPUTE_FEATURES = 2 << 0
CLOD_BLOCK_IMPLEMENTATION = 2 << 1
CLOD_PER_STAGE_ITERATIONS = 2 << 2

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class VjError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"{what}: {detail}" if detail else what)
        self.code = code


class CascadeInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("win_w", "win_h", "n_stages", "n_trees", "n_nodes", "n_alpha",
                                          "max_trees_per_stage", "max_nodes_per_tree", "n_tilted", "n_three_rect",
                                          "is_stump_based", "is_stage_tree")]


class Params(C.Structure):
    _fields_ = [("min_w", C.c_int32), ("min_h", C.c_int32), ("max_w", C.c_int32), ("max_h", C.c_int32),
                ("scale_factor", C.c_float), ("min_neighbors", C.c_uint32), ("flags", C.c_uint32),
                ("scale_mask", C.c_uint64 * 2)]


class ScaleInfo(C.Structure):
    _fields_ = [("scale_idx", C.c_int32), ("scale", C.c_float), ("step", C.c_float),
                ("win_w", C.c_int32), ("win_h", C.c_int32),
                ("equ_x", C.c_int32), ("equ_y", C.c_int32), ("equ_w", C.c_int32), ("equ_h", C.c_int32),
                ("area", C.c_uint32), ("nx", C.c_int32), ("ny", C.c_int32), ("accepted", C.c_int32)]


class _Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("on_device", C.c_int32), ("channels", C.c_int32)]


class CvParams(C.Structure):
    """vj_cv_params: cvHaarDetectObjects' arguments (OpenCV arithmetic profile)."""
    _fields_ = [("min_w", C.c_int32), ("min_h", C.c_int32), ("scale_factor", C.c_double), ("min_neighbors", C.c_uint32),
                ("flags", C.c_uint32)]


class _Counters(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("stump_evals", C.c_uint64), ("gather_bytes", C.c_uint64),
                ("stage_entered", C.c_uint64 * VJ_MAX_STAGES)]


VJ_MAX_PASSES = 8
VJ_MAX_LAUNCHES = 16
LAUNCH_KINDS = {0: "grid", 1: "queue", 2: "tile", 3: "block"}


class _Launch(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lds_class", C.c_int32), ("stage_begin", C.c_int32), ("stage_end", C.c_int32),
                ("ms", C.c_float), ("lds_bytes", C.c_uint32), ("scale_mask", C.c_uint64 * 2),
                ("stage_entered", C.c_uint64 * VJ_MAX_STAGES)]


class _Timing(C.Structure):
    _fields_ = [("integral_ms", C.c_float), ("cascade_ms", C.c_float), ("total_ms", C.c_float),
                ("n_cascade_launches", C.c_int32), ("pass_ms", C.c_float * VJ_MAX_PASSES),
                ("pass_stage_begin", C.c_int32 * VJ_MAX_PASSES), ("pass_stage_end", C.c_int32 * VJ_MAX_PASSES),
                ("n_launches", C.c_int32), ("launch", _Launch * VJ_MAX_LAUNCHES), ("tile_split", C.c_float),
                ("balance_state", C.c_int32), ("balance_calls", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("rects", C.c_void_p), ("count", C.c_uint32), ("counters", _Counters), ("timing", _Timing)]


RECT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("weight", "<f4"),
                       ("frame", "<i4"), ("scale_idx", "<i4")])
STAGE_DTYPE = np.dtype([("first_tree", "<i4"), ("n_trees", "<i4"), ("threshold", "<f4"), ("parent", "<i4"),
                        ("next", "<i4"), ("child", "<i4")])
TREE_DTYPE = np.dtype([("first_node", "<i4"), ("n_nodes", "<i4"), ("first_alpha", "<i4")])
_RECT_DESC = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("weight", "<f4")])
NODE_DTYPE = np.dtype([("n_rects", "<i4"), ("tilted", "<i4"), ("threshold", "<f4"), ("left", "<i4"),
                       ("right", "<i4"), ("rect", _RECT_DESC, 3)])

# every function include/vj.h declares, with its signature; tests check the library
# exports exactly this set
_SIGNATURES = {
    "vj_strerror": (C.c_char_p, [C.c_int]),
    "vj_last_error": (C.c_char_p, []),
    "vj_cascade_load_xml": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "vj_cascade_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "vj_cascade_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vj_cascade_from_arrays": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vj_cascade_free": (None, [C.c_void_p]),
    "vj_cascade_get_info": (C.c_int, [C.c_void_p, C.POINTER(CascadeInfo)]),
    "vj_cascade_stages": (C.c_void_p, [C.c_void_p]),
    "vj_cascade_trees": (C.c_void_p, [C.c_void_p]),
    "vj_cascade_nodes": (C.c_void_p, [C.c_void_p]),
    "vj_cascade_alpha": (C.c_void_p, [C.c_void_p]),
    "vj_cascade_notice": (C.c_char_p, [C.c_void_p]),
    "vj_params_default": (None, [C.POINTER(Params)]),
    "vj_plan_scales": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(ScaleInfo), C.c_int,
                                 C.POINTER(C.c_int)]),
    "vj_plan_feature_table": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ScaleInfo), C.c_void_p, C.c_void_p]),
    "vj_env_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vj_env_destroy": (None, [C.c_void_p]),
    "vj_env_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vj_env_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "vj_env_configure": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "vj_integral": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vj_integral_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vj_integral_tilted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vj_grayscale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vj_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vj_host_free": (None, [C.c_void_p, C.c_void_p]),
    "vj_detect_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Image), C.c_int, C.POINTER(Params),
                                  C.POINTER(Params), C.POINTER(_Result), C.POINTER(_Result)]),
    "vj_stream_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Params),
                                   C.POINTER(C.c_void_p)]),
    "vj_stream_submit": (C.c_int, [C.c_void_p, C.POINTER(_Image), C.c_int]),
    "vj_stream_collect": (C.c_int, [C.c_void_p, C.POINTER(_Result)]),
    "vj_stream_destroy": (None, [C.c_void_p]),
    "vj_detect_rois": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(_Image), C.c_int, C.c_void_p, C.c_int, C.POINTER(Params),
                                 C.POINTER(_Result)]),
    "vj_detect_opencv": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(_Image), C.c_int, C.POINTER(CvParams),
                                   C.POINTER(_Result)]),
    "vj_cv_params_default": (None, [C.POINTER(CvParams)]),
    "vj_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(_Image), C.c_int, C.POINTER(Params),
                            C.POINTER(_Result)]),
    "vj_result_free": (None, [C.POINTER(_Result)]),
    "vj_group_rectangles": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_double]),
    "vj_count_windows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(C.c_uint64)]),
    "vj_shard_frames": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vj_shard_scales": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
}

_lib = None


def load_library(build: bool = True) -> C.CDLL:
    """dlopen libvjhip.so (building it first when stale).  Raises if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    path = build_lib() if build else LIB_PATH
    if not os.path.exists(path):
        raise VjError(-1, "libvjhip.so is missing", f"expected at {path}; run clfacedetection_amd/build.py")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export what vj.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        lib = load_library()
        raise VjError(rc, f"{what} failed ({lib.vj_strerror(rc).decode()})", lib.vj_last_error().decode())


def default_params(**kw) -> Params:
    p = Params()
    load_library().vj_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "scales":          # iterable of scale indices -> scale_mask; an empty one = no scale (VJ_SCALE_MASK_NONE)
            for i in v:
                if not 0 <= i < 127:
                    raise ValueError("scale indices in a mask must be below 127")
                p.scale_mask[i >> 6] |= 1 << (i & 63)
            if (p.scale_mask[0] | p.scale_mask[1]) == 0:
                p.scale_mask[1] = 1 << 63
        else:
            setattr(p, k, v)
    return p


class Cascade:
    """CvHaarClassifierCascade stand-in (main.cpp:36 `cvLoad`)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        info = CascadeInfo()
        _check(load_library().vj_cascade_get_info(self._h, C.byref(info)), "vj_cascade_get_info")
        self.info = info

    @classmethod
    def load_xml(cls, path: str) -> "Cascade":
        h = C.c_void_p()
        _check(load_library().vj_cascade_load_xml(os.fsencode(path), C.byref(h)), f"vj_cascade_load_xml({path})")
        return cls(h.value)

    @classmethod
    def load(cls, path_or_name: str) -> "Cascade":
        """Load a .vjc file, or a stock cascade by name (e.g. 'frontalface_alt')."""
        path = path_or_name
        if not os.path.exists(path):
            cand = os.path.join(DATA_DIR, f"haarcascade_{path_or_name}.vjc")
            if os.path.exists(cand):
                path = cand
        if path.endswith(".xml"):
            return cls.load_xml(path)
        h = C.c_void_p()
        _check(load_library().vj_cascade_load(os.fsencode(path), C.byref(h)), f"vj_cascade_load({path})")
        return cls(h.value)

    @classmethod
    def from_arrays(cls, win_w: int, win_h: int, stages: np.ndarray, trees: np.ndarray, nodes: np.ndarray,
                    alpha: np.ndarray) -> "Cascade":
        """vj_cascade_from_arrays: a cascade held in memory (STAGE_DTYPE / TREE_DTYPE / NODE_DTYPE arrays + f32 leaf values),
        e.g. converted from a CvHaarClassifierCascade."""
        st = np.ascontiguousarray(stages, STAGE_DTYPE)
        tr = np.ascontiguousarray(trees, TREE_DTYPE)
        nd = np.ascontiguousarray(nodes, NODE_DTYPE)
        al = np.ascontiguousarray(alpha, np.float32)
        h = C.c_void_p()
        _check(load_library().vj_cascade_from_arrays(int(win_w), int(win_h), st.ctypes.data, len(st), tr.ctypes.data, len(tr),
                                                     nd.ctypes.data, len(nd), al.ctypes.data, len(al), C.byref(h)),
               "vj_cascade_from_arrays")
        return cls(h.value)

    def save(self, path: str):
        _check(load_library().vj_cascade_save(self._h, os.fsencode(path)), f"vj_cascade_save({path})")

    def _view(self, fn, dtype, n):
        ptr = fn(self._h)
        buf = (C.c_char * (dtype.itemsize * n)).from_address(ptr)
        return np.frombuffer(buf, dtype, n).copy()

    @property
    def stages(self): return self._view(load_library().vj_cascade_stages, STAGE_DTYPE, self.info.n_stages)
    @property
    def trees(self): return self._view(load_library().vj_cascade_trees, TREE_DTYPE, self.info.n_trees)
    @property
    def nodes(self): return self._view(load_library().vj_cascade_nodes, NODE_DTYPE, self.info.n_nodes)
    @property
    def alpha(self): return self._view(load_library().vj_cascade_alpha, np.dtype("<f4"), self.info.n_alpha)
    @property
    def notice(self) -> str: return load_library().vj_cascade_notice(self._h).decode("latin-1")

    def plan_scales(self, width: int, height: int, params: Params | None = None) -> list[ScaleInfo]:
        p = params or default_params()
        n = C.c_int(0)
        lib = load_library()
        _check(lib.vj_plan_scales(self._h, width, height, C.byref(p), None, 0, C.byref(n)), "vj_plan_scales")
        arr = (ScaleInfo * max(n.value, 1))()
        _check(lib.vj_plan_scales(self._h, width, height, C.byref(p), arr, n.value, C.byref(n)), "vj_plan_scales")
        return list(arr)[:n.value]

    def feature_table(self, width: int, scale: ScaleInfo) -> tuple[np.ndarray, np.ndarray]:
        off = np.zeros((self.info.n_nodes, 3, 4), np.uint32)
        wts = np.zeros((self.info.n_nodes, 3), np.float32)
        _check(load_library().vj_plan_feature_table(self._h, width, C.byref(scale), off.ctypes.data,
                                                    wts.ctypes.data), "vj_plan_feature_table")
        return off, wts

    def shard_scales(self, width: int, height: int, rank: int, world: int, params: Params | None = None) -> list[int]:
        """vj_shard_scales: the scale indices of `rank` when one frame is split over `world` ranks by scale."""
        p = params or default_params()
        m = (C.c_uint64 * 2)()
        _check(load_library().vj_shard_scales(self._h, width, height, C.byref(p), world, rank, m), "vj_shard_scales")
        return [k for k in range(127) if (m[k >> 6] >> (k & 63)) & 1]      # bit 127 = VJ_SCALE_MASK_NONE: an empty share

    def count_windows(self, width: int, height: int, params: Params | None = None) -> int:
        p = params or default_params()
        out = C.c_uint64(0)
        _check(load_library().vj_count_windows(self._h, width, height, C.byref(p), C.byref(out)), "vj_count_windows")
        return out.value

    def close(self):
        if self._h:
            load_library().vj_cascade_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class DetectResult:
    """CLODDetectObjectsResult (clod.h:44-47) + provenance, counters and device timing."""
    rects: np.ndarray            # RECT_DTYPE, sorted by (frame, scale_idx, y, x)
    windows: int
    stump_evals: int
    gather_bytes: int
    stage_entered: list
    integral_ms: float
    cascade_ms: float
    total_ms: float
    n_cascade_launches: int
    passes: list = None          # [(stage_begin, stage_end, ms)] per cascade pass
    launches: list = None        # per kernel launch: dict(kind, lds_class, stage_begin, stage_end, ms, lds_bytes, scales)
    tile_split: float = 0.0      # the chain balance the call's plan was built for
    balance_state: int = 0       # 0 static, 1 the workload's feedback search is running, 2 finished
    balance_calls: int = 0       # calls the search has measured so far

    @property
    def match_count(self) -> int:
        return len(self.rects)


class Environment:
    """CLODEnvironmentData stand-in: one HIP device, its stream and its buffers."""

    def __init__(self, device_index: int = 0):
        h = C.c_void_p()
        _check(load_library().vj_env_create(device_index, C.byref(h)), "vj_env_create")
        self._h = h

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(load_library().vj_env_device_name(self._h, buf, 256), "vj_env_device_name")
        return buf.value.decode()

    def configure(self, key: str, value) -> None:
        """Tunables that never change results: 'pass_split' ("4,9,15"), 'blocks_per_cu'."""
        _check(load_library().vj_env_configure(self._h, key.encode(), str(value).encode()), f"vj_env_configure({key})")

    def reserve(self, width: int, height: int, batch: int = 1):
        _check(load_library().vj_env_reserve(self._h, width, height, batch), "vj_env_reserve")

    def integral(self, gray: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """clifIntegral on a 2-D uint8 image; a 3-D (h, w, 3 | 4) BGR / BGRA image goes through
        clifGrayscaleIntegral's conversion first (vj_integral_image)."""
        if gray.dtype != np.uint8 or gray.ndim not in (2, 3):
            raise ValueError("integral expects a 2-D uint8 image or a (h, w, 3|4) uint8 BGR/BGRA image")
        s = np.empty((gray.shape[0] + 1, gray.shape[1] + 1), np.uint32)
        q = np.empty((gray.shape[0] + 1, gray.shape[1] + 1), np.uint64)
        if gray.ndim == 3:
            g = _pixel_contiguous(gray)
            im = _Image(g.ctypes.data, g.shape[1], g.shape[0], g.strides[0], 0, g.shape[2])
            _check(load_library().vj_integral_image(self._h, C.byref(im), s.ctypes.data, q.ctypes.data), "vj_integral_image")
            return s, q
        g = gray if gray.strides[1] == 1 else np.ascontiguousarray(gray)
        h, w = g.shape
        _check(load_library().vj_integral(self._h, g.ctypes.data, w, h, g.strides[0], s.ctypes.data, q.ctypes.data),
               "vj_integral")
        return s, q

    def _one_image(self, img: np.ndarray):
        if img.dtype != np.uint8 or img.ndim not in (2, 3):
            raise ValueError("expected a 2-D uint8 image or a (h, w, 3|4) uint8 BGR/BGRA image")
        g = _pixel_contiguous(img) if img.ndim == 3 else (img if img.strides[1] == 1 else np.ascontiguousarray(img))
        return _Image(g.ctypes.data, g.shape[1], g.shape[0], g.strides[0], 0, g.shape[2] if g.ndim == 3 else 1), g

    def integral_tilted(self, img: np.ndarray) -> np.ndarray:
        """vj_integral_tilted: cvIntegral's tilted sum, (h+1, w+1) uint32."""
        im, keep = self._one_image(img)
        t = np.empty((img.shape[0] + 1, img.shape[1] + 1), np.uint32)
        _check(load_library().vj_integral_tilted(self._h, C.byref(im), t.ctypes.data), "vj_integral_tilted")
        return t

    def grayscale(self, img: np.ndarray) -> np.ndarray:
        """clifGrayscale: the 8-bit gray image the integral kernels see (BGR / BGRA -> gray on the device)."""
        im, keep = self._one_image(img)
        g = np.empty(img.shape[:2], np.uint8)
        _check(load_library().vj_grayscale(self._h, C.byref(im), g.ctypes.data, g.strides[0]), "vj_grayscale")
        return g

    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """vj_host_alloc: a page-locked numpy array (freed with host_free) for copy-free frame uploads."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        _check(load_library().vj_host_alloc(self._h, n, C.byref(ptr)), "vj_host_alloc")
        buf = (C.c_char * n).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype).reshape(shape)
        self.__dict__.setdefault("_pinned", {})[arr.ctypes.data] = ptr.value
        return arr

    def host_free(self, arr: np.ndarray):
        ptr = self.__dict__.get("_pinned", {}).pop(arr.ctypes.data, None)
        if ptr is not None:
            load_library().vj_host_free(self._h, C.c_void_p(ptr))

    def detect_chain(self, first: Cascade, second: Cascade, frames, params_first: Params | None = None,
                     params_second: Params | None = None, color: bool = False):
        """vj_detect_chain: `second` on every raw candidate of `first`, hand-off on the device.  Returns (result_first,
        result_second); result_second.rects['frame'] indexes result_first.rects, x / y are relative to that region."""
        p1 = params_first or default_params()
        p2 = params_second or default_params()
        imgs, n, keep = self._images(frames, color)
        r1, r2 = _Result(), _Result()
        lib = load_library()
        _check(lib.vj_detect_chain(self._h, first._h, second._h, imgs, n, C.byref(p1), C.byref(p2), C.byref(r1), C.byref(r2)),
               "vj_detect_chain")
        return self._result(lib, r1, first), self._result(lib, r2, second)

    def stream(self, cascade: Cascade, width: int, height: int, max_batch: int, params: Params | None = None,
               channels: int = 1) -> "FrameStream":
        return FrameStream(self, cascade, width, height, max_batch, params or default_params(), channels)

    @staticmethod
    def _images(frames, color: bool):
        """-> (ctypes array of vj_image, n, arrays to keep alive)"""
        keep = []
        if isinstance(frames, DeviceFrames):
            n = frames.n
            imgs = (_Image * max(n, 1))()
            for i in range(n):
                imgs[i] = _Image(frames.ptr + i * frames.stride * frames.height, frames.width, frames.height,
                                 frames.stride, 1, frames.channels)
            return imgs, n, keep
        if isinstance(frames, np.ndarray) and frames.ndim == (3 if color else 2):
            frames = [frames]
        frames = list(frames)
        n = len(frames)
        imgs = (_Image * max(n, 1))()
        for i, f in enumerate(frames):
            if color:
                if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] not in (3, 4):
                    raise ValueError("color frames must be (h, w, 3|4) uint8 (BGR / BGRA)")
                g = _pixel_contiguous(f)
                keep.append(g)
                imgs[i] = _Image(g.ctypes.data, g.shape[1], g.shape[0], g.strides[0], 0, g.shape[2])
                continue
            if f.dtype != np.uint8 or f.ndim != 2:
                raise ValueError("frames must be 2-D uint8 (8-bit single channel)")
            g = f if f.strides[1] == 1 else np.ascontiguousarray(f)
            keep.append(g)
            imgs[i] = _Image(g.ctypes.data, g.shape[1], g.shape[0], g.strides[0], 0, 1)
        return imgs, n, keep

    def detect_rois(self, cascade: Cascade, frames, rois, params: Params | None = None, color: bool = False) -> DetectResult:
        """vj_detect_rois: `rois` = rows of (frame, x, y, w, h), regions of any sizes (frames of one size and a linear
        cascade: one pass for all of them on the frames' integral images; else one batched pass per region size).  In the
        result rects['frame'] is the ROI's row and x / y are relative to the ROI's origin."""
        p = params or default_params()
        imgs, n, keep = self._images(frames, color)
        r = np.ascontiguousarray(np.asarray(rois, np.int32).reshape(-1, 5))
        res = _Result()
        lib = load_library()
        _check(lib.vj_detect_rois(self._h, cascade._h, imgs, n, r.ctypes.data, len(r), C.byref(p), C.byref(res)),
               "vj_detect_rois")
        return self._result(lib, res, cascade)

    def detect_opencv(self, cascade: Cascade, frames, min_size=(0, 0), scale_factor: float = 1.1, min_neighbors: int = 0,
                      flags: int = 0, color: bool = False) -> DetectResult:
        """vj_detect_opencv: cvHaarDetectObjects' scale-cascade path (OpenCV arithmetic profile: f64 sums, threshold
        bias, ystep = max(2, factor), skip after a stage-0 reject, border rule).  result.windows = visited positions."""
        imgs, n, keep = self._images(frames, color)
        p = CvParams(int(min_size[0]), int(min_size[1]), float(scale_factor), int(min_neighbors), int(flags))
        res = _Result()
        lib = load_library()
        _check(lib.vj_detect_opencv(self._h, cascade._h, imgs, n, C.byref(p), C.byref(res)), "vj_detect_opencv")
        return self._result(lib, res, cascade)

    def detect(self, cascade: Cascade, frames, params: Params | None = None, color: bool = False) -> DetectResult:
        """frames: 2-D uint8 array, 3-D (n, h, w) array, list of 2-D arrays, or
        DeviceFrames (frames already resident in HBM).  color=True: every frame is (h, w, 3 | 4) BGR / BGRA
        (one 3-D array, a 4-D batch or a list) and is converted to gray on the device."""
        p = params or default_params()
        imgs, n, keep = self._images(frames, color)
        res = _Result()
        lib = load_library()
        _check(lib.vj_detect(self._h, cascade._h, imgs, n, C.byref(p), C.byref(res)), "vj_detect")
        return self._result(lib, res, cascade)

    @staticmethod
    def _result(lib, res, cascade) -> DetectResult:
        try:
            if res.count:
                buf = (C.c_char * (RECT_DTYPE.itemsize * res.count)).from_address(res.rects)
                rects = np.frombuffer(buf, RECT_DTYPE, res.count).copy()
            else:
                rects = np.zeros(0, RECT_DTYPE)
            k, t = res.counters, res.timing
            return DetectResult(rects, int(k.windows), int(k.stump_evals), int(k.gather_bytes),
                                [int(v) for v in k.stage_entered[:cascade.info.n_stages]],
                                float(t.integral_ms), float(t.cascade_ms), float(t.total_ms),
                                int(t.n_cascade_launches),
                                [(int(t.pass_stage_begin[i]), int(t.pass_stage_end[i]), float(t.pass_ms[i]))
                                 for i in range(min(int(t.n_cascade_launches), VJ_MAX_PASSES))],
                                [dict(kind=LAUNCH_KINDS.get(int(l.kind), "?"), lds_class=int(l.lds_class),
                                      stage_begin=int(l.stage_begin), stage_end=int(l.stage_end), ms=float(l.ms),
                                      lds_bytes=int(l.lds_bytes),
                                      scales=[k for k in range(127) if (l.scale_mask[k >> 6] >> (k & 63)) & 1],
                                      stage_entered=[int(v) for v in l.stage_entered[:cascade.info.n_stages]])
                                 for l in list(t.launch)[:int(t.n_launches)]], float(t.tile_split),
                                int(t.balance_state), int(t.balance_calls))
        finally:
            lib.vj_result_free(C.byref(res))

    def close(self):
        if self._h:
            load_library().vj_env_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameStream:
    """vj_stream: double-buffered upload of host frame batches overlapped with the previous batch's kernels."""

    def __init__(self, env: Environment, cascade: Cascade, width: int, height: int, max_batch: int, params: Params, channels: int = 1):
        self._env, self._cascade = env, cascade
        h = C.c_void_p()
        _check(load_library().vj_stream_create(env._h, cascade._h, width, height, channels, max_batch, C.byref(params), C.byref(h)),
               "vj_stream_create")
        self._h = h
        self._keep = []

    def submit(self, frames, color: bool = False):
        imgs, n, keep = Environment._images(frames, color)
        _check(load_library().vj_stream_submit(self._h, imgs, n), "vj_stream_submit")
        self._keep.append((imgs, keep))

    def collect(self) -> DetectResult:
        res = _Result()
        lib = load_library()
        _check(lib.vj_stream_collect(self._h, C.byref(res)), "vj_stream_collect")
        if self._keep:
            self._keep.pop(0)
        return Environment._result(lib, res, self._cascade)

    def close(self):
        if self._h:
            load_library().vj_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class DeviceFrames:
    """A batch of equal-size 8-bit frames already resident in device memory
    (e.g. a torch.uint8 CUDA tensor's data_ptr()); rows `stride` bytes apart,
    frames `stride * height` bytes apart; channels = 1 (gray), 3 (BGR) or 4 (BGRA), interleaved."""
    ptr: int
    n: int
    height: int
    width: int
    stride: int
    channels: int = 1

    @classmethod
    def from_torch(cls, t) -> "DeviceFrames":
        """(n, h, w) gray or (n, h, w, 3 | 4) BGR / BGRA uint8 CUDA tensor."""
        assert t.is_cuda and t.dtype.itemsize == 1 and t.dim() in (3, 4) and t.is_contiguous()
        if t.dim() == 4:
            n, h, w, ch = t.shape
            assert ch in (3, 4)
            return cls(t.data_ptr(), n, h, w, w * ch, ch)
        n, h, w = t.shape
        return cls(t.data_ptr(), n, h, w, w)


def _pixel_contiguous(img: np.ndarray) -> np.ndarray:
    """(h, w, c) view whose pixels and channels are contiguous (rows may be strided: ROI views stay views)."""
    c = img.shape[2]
    return img if img.strides[2] == 1 and img.strides[1] == c else np.ascontiguousarray(img)


def group_rectangles(rects: np.ndarray, group_threshold: int, eps: float = 0.2) -> np.ndarray:
    """filterResult / cv::groupRectangles on a RECT_DTYPE array sorted by frame (host logic)."""
    buf = np.ascontiguousarray(rects, RECT_DTYPE).copy()
    n = C.c_uint32(len(buf))
    _check(load_library().vj_group_rectangles(buf.ctypes.data if len(buf) else None, C.byref(n), int(group_threshold),
                                              float(eps)), "vj_group_rectangles")
    return buf[:n.value]


# ------------------------------------------------- reference-named entry points
def clodInitEnvironment(device_index: int = 0) -> Environment:
    """clod.h:61-62 / clod.cpp:72-100."""
    return Environment(device_index)


def clodInitBuffers(env: Environment, image_size, batch: int = 1):
    """clod.h:67-69 / clod.cpp:102-163; image_size = (width, height) like CvSize."""
    env.reserve(int(image_size[0]), int(image_size[1]), batch)


def clodReleaseBuffers(env: Environment):
    """clod.h:70-71: buffers are owned by the environment here; nothing to do."""


def clodReleaseEnvironment(env: Environment):
    """clod.h:64-65 / clod.cpp:173-180."""
    env.close()


def clifIntegral(source: np.ndarray, env: Environment, use_opencl: bool = True):
    """clif.h:63-66 / clif.cpp:273-316: returns (sum CV_32S-as-uint32, square_sum as
    uint64), both (h+1, w+1) — cvIntegral's layout."""
    if not use_opencl:
        raise VjError(4, "clifIntegral", "this package has no CPU path; the reference's CPU branch is cvIntegral")
    return env.integral(source)


def clodDetectObjects(image, cascade: Cascade, env: Environment, min_window_size=(0, 0), max_window_size=(0, 0),
                      min_neighbors: int = 0, flags: int = 0, use_opencl: bool = True,
                      vj_flags: int = 0) -> DetectResult:
    """clod.h:72-81 / clod.cpp:1339-1356 with use_opencl=CL_TRUE → clodDetectObjectsOpenCL
    (clod.cpp:1176-1336).  `image` may also be a batch (see Environment.detect).  At the reference's own signature a cascade
    with tilted features behaves as in the reference — precomputeFeatures never reads the flag (clod.cpp:448-492), the
    rectangles count as upright ones — where Environment.detect refuses it unless VJ_FLAG_TILTED_AS_UPRIGHT is given."""
    vj_flags |= VJ_FLAG_TILTED_AS_UPRIGHT
    if not use_opencl:
        # the reference's CPU evaluators (clod.cpp:1358-1499) return the skip-thinned window set; the device computes the
        # same set.  The block variant (clod.cpp:821-1173) keeps `step` in double: two more grids (VJ_FLAG_GRID_F64).
        vj_flags |= VJ_FLAG_SKIP_LIST if flags & CLOD_PER_STAGE_ITERATIONS else VJ_FLAG_SKIP_ROW
        if flags & CLOD_BLOCK_IMPLEMENTATION:
            vj_flags |= VJ_FLAG_GRID_F64
    p = default_params(min_w=int(min_window_size[0]), min_h=int(min_window_size[1]),
                       max_w=int(max_window_size[0]), max_h=int(max_window_size[1]),
                       min_neighbors=int(min_neighbors), flags=int(vj_flags))
    return env.detect(cascade, image, p)


def cvHaarDetectObjects(image, cascade: Cascade, env: Environment, scale_factor: float = 1.1, min_neighbors: int = 3,
                        flags: int = 0, min_size=(0, 0), vj_flags: int = 0) -> DetectResult:
    """The reference demo's OpenCV leg (main.cpp:145: cvHaarDetectObjects(img, cascade, storage, 1.1, ...)) as
    tempcv.cpp:1188-1456 specifies its scale-cascade path, on the device (OpenCV arithmetic profile).  `flags`
    must be 0: canny pruning, find-biggest-object and scale-image are other paths."""
    if flags != 0:
        raise VjError(4, "cvHaarDetectObjects", "only flags = 0 (the scale-cascade path) is implemented")
    return env.detect_opencv(cascade, image, min_size, scale_factor, min_neighbors, vj_flags)
