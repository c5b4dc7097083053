"""Python front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module — never anything under clfacedetection_amd/.  See oracle/vj_oracle.c for the
parity-pinning statement.

Contents
  * parse_xml()  — an independent reader of OpenCV's old-format Haar XML, following
                   icvReadHaarClassifier (tempcv.cpp:1749-2089); used here (where
                   /root/reference exists) to cross-check the product's C++ loader.
  * load_vjc()   — an independent reader of the package's compact .vjc files.
  * Oracle       — ctypes bindings of libvjoracle.so (vj_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import struct
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvjoracle.so")


def build(force: bool = False) -> str:
    """Compile vj_oracle.c (gcc, -ffp-contract=off) if the .so is missing or stale."""
    src = os.path.join(_HERE, "vj_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "libvjoracle.so"], check=True, capture_output=True)
    return _LIB


# --------------------------------------------------------------------- cascade
class CascadeArrays:
    """Flat arrays of one cascade; field names mirror oc_cascade in vj_oracle.c."""

    FIELDS_I32 = ("stage_first_tree", "stage_n_trees", "stage_parent", "stage_next", "stage_child",
                  "tree_first_node", "tree_n_nodes", "tree_first_alpha", "node_rect", "node_left", "node_right")
    FIELDS_F32 = ("stage_threshold", "node_weight", "node_threshold", "alpha")

    def __init__(self):
        self.win_w = self.win_h = 0
        self.name = ""
        self.notice = ""
        for f in self.FIELDS_I32:
            setattr(self, f, np.zeros(0, np.int32))
        for f in self.FIELDS_F32:
            setattr(self, f, np.zeros(0, np.float32))
        self.node_tilted = np.zeros(0, np.int32)

    @property
    def n_stages(self): return len(self.stage_first_tree)
    @property
    def n_trees(self): return len(self.tree_first_node)
    @property
    def n_nodes(self): return len(self.node_threshold)
    @property
    def n_alpha(self): return len(self.alpha)

    def same_as(self, o: "CascadeArrays") -> list[str]:
        """Names of fields that differ (bit-exact comparison)."""
        bad = []
        if (self.win_w, self.win_h) != (o.win_w, o.win_h):
            bad.append("size")
        for f in self.FIELDS_I32 + ("node_tilted",):
            if not np.array_equal(getattr(self, f), getattr(o, f)):
                bad.append(f)
        for f in self.FIELDS_F32:
            a, b = getattr(self, f), getattr(o, f)
            if a.shape != b.shape or not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                bad.append(f)
        return bad


def _f32(text: str) -> np.float32:
    # decimal -> f64 -> f32: OpenCV stores (float)fn->data.f (tempcv.cpp:1932, 1958, 1995, 2054)
    return np.float32(float(text))


def parse_xml(path: str) -> CascadeArrays:
    """Old-format OpenCV Haar XML -> CascadeArrays (tempcv.cpp:1749-2089)."""
    raw = open(path, "r", encoding="latin-1").read()
    m = re.search(r"<!--(.*?)-->", raw, re.S)
    notice = m.group(1) if m else ""
    raw = re.sub(r"<!--.*?-->", "", raw, flags=re.S)   # some stock files carry '--' inside comments
    root = ET.fromstring(raw)
    casc = root[0]
    c = CascadeArrays()
    c.name = casc.tag
    c.notice = notice
    c.win_w, c.win_h = (int(v) for v in casc.find("size").text.split())
    s_first, s_n, s_thr, s_par, s_next = [], [], [], [], []
    t_first, t_n, t_alpha = [], [], []
    n_rect, n_w, n_thr, n_left, n_right, n_tilt = [], [], [], [], [], []
    alpha = []
    for st in casc.find("stages"):
        trees = st.find("trees")
        s_first.append(len(t_first))
        s_n.append(len(trees))
        for tree in trees:
            t_first.append(len(n_thr))
            t_n.append(len(tree))
            t_alpha.append(len(alpha))
            last = 0
            for node in tree:
                feat = node.find("feature")
                rects = [r.text.split() for r in feat.find("rects")]
                rr = [[0, 0, 0, 0]] * 3
                ww = [np.float32(0)] * 3
                for i, r in enumerate(rects):
                    rr[i] = [int(r[0]), int(r[1]), int(r[2]), int(r[3])]
                    ww[i] = _f32(r[4])
                n_rect.append(rr)
                n_w.append(ww)
                n_tilt.append(int(feat.find("tilted").text) != 0)
                n_thr.append(_f32(node.find("threshold").text))
                ln = node.find("left_node")
                if ln is not None:
                    n_left.append(int(ln.text))
                else:
                    n_left.append(-last)
                    alpha.append(_f32(node.find("left_val").text))
                    last += 1
                rn = node.find("right_node")
                if rn is not None:
                    n_right.append(int(rn.text))
                else:
                    n_right.append(-last)
                    alpha.append(_f32(node.find("right_val").text))
                    last += 1
            assert last == len(tree) + 1, "tree structure is broken"
        s_thr.append(_f32(st.find("stage_threshold").text))
        s_par.append(int(st.find("parent").text))
        s_next.append(int(st.find("next").text))
    child = [-1] * len(s_par)
    for i, p in enumerate(s_par):
        if p != -1 and child[p] == -1:
            child[p] = i
    c.stage_first_tree = np.array(s_first, np.int32)
    c.stage_n_trees = np.array(s_n, np.int32)
    c.stage_threshold = np.array(s_thr, np.float32)
    c.stage_parent = np.array(s_par, np.int32)
    c.stage_next = np.array(s_next, np.int32)
    c.stage_child = np.array(child, np.int32)
    c.tree_first_node = np.array(t_first, np.int32)
    c.tree_n_nodes = np.array(t_n, np.int32)
    c.tree_first_alpha = np.array(t_alpha, np.int32)
    c.node_rect = np.array(n_rect, np.int32).reshape(-1)
    c.node_weight = np.array(n_w, np.float32).reshape(-1)
    c.node_threshold = np.array(n_thr, np.float32)
    c.node_left = np.array(n_left, np.int32)
    c.node_right = np.array(n_right, np.int32)
    c.node_tilted = np.array(n_tilt, np.int32)
    c.alpha = np.array(alpha, np.float32)
    return c


def load_vjc(path: str) -> CascadeArrays:
    """Reader of the package's .vjc format (layout documented in vj_cascade.cpp)."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"VJCASC01", "not a VJCASC01 file"
    (nl,) = struct.unpack_from("<I", buf, 8)
    pos = 12
    c = CascadeArrays()
    c.name = os.path.splitext(os.path.basename(path))[0]
    c.notice = buf[pos:pos + nl].decode("latin-1")
    pos += nl
    c.win_w, c.win_h, ns, nt, nn, na = struct.unpack_from("<6i", buf, pos)
    pos += 24
    st = np.frombuffer(buf, np.dtype([("first_tree", "<i4"), ("n_trees", "<i4"), ("thr", "<f4"),
                                      ("parent", "<i4"), ("next", "<i4"), ("child", "<i4")]), ns, pos)
    pos += st.nbytes
    tr = np.frombuffer(buf, np.dtype([("first_node", "<i4"), ("n_nodes", "<i4"), ("first_alpha", "<i4")]), nt, pos)
    pos += tr.nbytes
    rect_dt = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("weight", "<f4")])
    nd = np.frombuffer(buf, np.dtype([("n_rects", "<i4"), ("tilted", "<i4"), ("thr", "<f4"), ("left", "<i4"),
                                      ("right", "<i4"), ("rect", rect_dt, 3)]), nn, pos)
    pos += nd.nbytes
    c.alpha = np.frombuffer(buf, "<f4", na, pos).copy()
    pos += 4 * na
    assert pos == len(buf), "payload size mismatch"
    c.stage_first_tree = st["first_tree"].copy()
    c.stage_n_trees = st["n_trees"].copy()
    c.stage_threshold = st["thr"].copy()
    c.stage_parent = st["parent"].copy()
    c.stage_next = st["next"].copy()
    c.stage_child = st["child"].copy()
    c.tree_first_node = tr["first_node"].copy()
    c.tree_n_nodes = tr["n_nodes"].copy()
    c.tree_first_alpha = tr["first_alpha"].copy()
    r = nd["rect"]
    c.node_rect = np.stack([r["x"], r["y"], r["w"], r["h"]], axis=-1).astype(np.int32).reshape(-1)
    c.node_weight = r["weight"].astype(np.float32).reshape(-1)
    c.node_threshold = nd["thr"].copy()
    c.node_left = nd["left"].copy()
    c.node_right = nd["right"].copy()
    c.node_tilted = nd["tilted"].copy()
    return c


# --------------------------------------------------------------------- ctypes
class _OcCascade(C.Structure):
    _fields_ = [("win_w", C.c_int32), ("win_h", C.c_int32), ("n_stages", C.c_int32), ("n_trees", C.c_int32),
                ("n_nodes", C.c_int32), ("n_alpha", C.c_int32),
                ("stage_first_tree", C.c_void_p), ("stage_n_trees", C.c_void_p), ("stage_threshold", C.c_void_p),
                ("stage_parent", C.c_void_p), ("stage_next", C.c_void_p), ("stage_child", C.c_void_p),
                ("tree_first_node", C.c_void_p), ("tree_n_nodes", C.c_void_p), ("tree_first_alpha", C.c_void_p),
                ("node_rect", C.c_void_p), ("node_weight", C.c_void_p), ("node_threshold", C.c_void_p),
                ("node_left", C.c_void_p), ("node_right", C.c_void_p), ("alpha", C.c_void_p),
                ("node_tilted", C.c_void_p)]


class OcScale(C.Structure):
    _fields_ = [("scale_idx", C.c_int32), ("scale", C.c_float), ("step", C.c_float),
                ("win_w", C.c_int32), ("win_h", C.c_int32),
                ("equ_x", C.c_int32), ("equ_y", C.c_int32), ("equ_w", C.c_int32), ("equ_h", C.c_int32),
                ("area", C.c_uint32), ("nx", C.c_int32), ("ny", C.c_int32), ("accepted", C.c_int32)]


class _OcStats(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("stump_evals", C.c_uint64), ("rect_evals", C.c_uint64),
                ("stage_entered", C.c_uint64 * 64)]


_RECT_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("scale_idx", "<i4")])


class Oracle:
    """ctypes view of libvjoracle.so."""

    def __init__(self, lib_path: str | None = None):
        self.lib = C.CDLL(lib_path or build())
        L = self.lib
        L.oc_integral.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oc_integral.restype = None
        L.oc_scale_count.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        L.oc_scale_count.restype = C.c_int
        L.oc_setup_scale.argtypes = [C.c_float] + [C.c_int] * 8 + [C.POINTER(OcScale)]
        L.oc_setup_scale.restype = C.c_int
        L.oc_feature_table.argtypes = [C.POINTER(_OcCascade), C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.oc_feature_table.restype = None
        L.oc_detect.argtypes = [C.POINTER(_OcCascade), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_OcStats)]
        L.oc_detect.restype = C.c_int
        L.oc_xorshift_noise.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.oc_xorshift_noise.restype = None
        L.oc_xorshift_words.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.oc_xorshift_words.restype = None
        L.oc_u64_to_f32.argtypes = [C.c_uint64]
        L.oc_u64_to_f32.restype = C.c_float
        L.oc_detect_opencvlike.argtypes = [C.POINTER(_OcCascade), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_OcStats)]
        L.oc_detect_opencvlike.restype = C.c_int
        L.oc_detect_opencvlike_all_f64.argtypes = L.oc_detect_opencvlike.argtypes
        L.oc_detect_opencvlike_all_f64.restype = C.c_int
        L.oc_integral_tilted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oc_integral_tilted.restype = None
        L.oc_bgr2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oc_bgr2gray.restype = None
        L.oc_group_rectangles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.oc_group_rectangles.restype = C.c_int

    @staticmethod
    def _cstruct(c: CascadeArrays) -> tuple[_OcCascade, list]:
        keep = []
        s = _OcCascade()
        s.win_w, s.win_h = c.win_w, c.win_h
        s.n_stages, s.n_trees, s.n_nodes, s.n_alpha = c.n_stages, c.n_trees, c.n_nodes, c.n_alpha
        for f in CascadeArrays.FIELDS_I32:
            a = np.ascontiguousarray(getattr(c, f), np.int32)
            keep.append(a)
            setattr(s, f, a.ctypes.data)
        for f in CascadeArrays.FIELDS_F32:
            a = np.ascontiguousarray(getattr(c, f), np.float32)
            keep.append(a)
            setattr(s, f, a.ctypes.data)
        t = np.ascontiguousarray(c.node_tilted, np.int32)
        if len(t) == c.n_nodes and t.any():
            keep.append(t)
            s.node_tilted = t.ctypes.data
        else:
            s.node_tilted = None
        return s, keep

    # a1
    def integral(self, gray: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(sum uint32, sqsum uint64), both (h+1, w+1); sqsum converted from the f64 matrix."""
        assert gray.dtype == np.uint8 and gray.ndim == 2
        h, w = gray.shape
        g = np.ascontiguousarray(gray)
        s = np.zeros((h + 1, w + 1), np.int32)
        q = np.zeros((h + 1, w + 1), np.float64)
        self.lib.oc_integral(g.ctypes.data, w, h, g.strides[0], s.ctypes.data, q.ctypes.data)
        return s.view(np.uint32), q.astype(np.uint64)

    def integral_tilted(self, gray: np.ndarray) -> np.ndarray:
        """cvIntegral's tilted sum, (h+1, w+1) uint32 (32-bit wrap-around)."""
        h, w = gray.shape
        g = np.ascontiguousarray(gray)
        t = np.zeros((h + 1, w + 1), np.int32)
        self.lib.oc_integral_tilted(g.ctypes.data, w, h, g.strides[0], t.ctypes.data)
        return t.view(np.uint32)

    # a2 + a3
    def plan_scales(self, c: CascadeArrays, W: int, H: int, min_size=(0, 0), max_size=(0, 0),
                    scale_factor: float = 1.1) -> list[OcScale]:
        sf = np.float32(scale_factor)
        n = self.lib.oc_scale_count(c.win_w, c.win_h, W, H, sf)
        out = []
        cs = np.float32(1)
        for k in range(n):
            sc = OcScale()
            self.lib.oc_setup_scale(cs, W, H, c.win_w, c.win_h, min_size[0], min_size[1],
                                    max_size[0], max_size[1], C.byref(sc))
            sc.scale_idx = k
            out.append(sc)
            cs = np.float32(cs * sf)
        return out

    # a5
    def feature_table(self, c: CascadeArrays, sc: OcScale, W: int) -> tuple[np.ndarray, np.ndarray]:
        s, keep = self._cstruct(c)
        off = np.zeros((c.n_nodes, 3, 4), np.uint32)
        wts = np.zeros((c.n_nodes, 3), np.float32)
        self.lib.oc_feature_table(C.byref(s), sc.scale, sc.area, W, off.ctypes.data, wts.ctypes.data)
        return off, wts

    # a4 + a6 + a7 (+ a9)
    def detect(self, c: CascadeArrays, gray: np.ndarray, min_size=(0, 0), max_size=(0, 0),
               scale_factor: float = 1.1, signed_mean: bool = False, mode: int | None = None,
               cap: int = 1 << 20):
        """Returns (rects sorted by (scale_idx, y, x), stats dict)."""
        assert gray.dtype == np.uint8 and gray.ndim == 2
        h, w = gray.shape
        g = np.ascontiguousarray(gray)
        s, keep = self._cstruct(c)
        linear = bool(np.all(c.stage_next == -1))
        if mode is None:
            mode = 0 if linear else 1
        assert mode == 1 or linear, "modes 0, 2, 3, 4, 5 (the reference's own loops) are defined for linear cascades only"
        out = np.zeros(cap, _RECT_DT)
        n_total = C.c_int(0)
        st = _OcStats()
        n = self.lib.oc_detect(C.byref(s), g.ctypes.data, w, h, g.strides[0], min_size[0], min_size[1],
                               max_size[0], max_size[1], np.float32(scale_factor), int(signed_mean), mode,
                               out.ctypes.data, cap, C.byref(n_total), C.byref(st))
        assert n_total.value <= cap, "detection buffer too small"
        r = out[:n]
        r = r[np.lexsort((r["x"], r["y"], r["scale_idx"]))]
        stats = {"windows": int(st.windows), "stump_evals": int(st.stump_evals), "rect_evals": int(st.rect_evals),
                 "gather_bytes": 48 * int(st.windows) + 16 * int(st.rect_evals),
                 "stage_entered": [int(v) for v in st.stage_entered[:c.n_stages]]}
        return r, stats

    def bgr2gray(self, img: np.ndarray) -> np.ndarray:
        """(h, w, 3|4) uint8 BGR / BGRA -> (h, w) gray, OpenCV's 8-bit fixed-point formula."""
        g = np.ascontiguousarray(img)
        h, w, ch = g.shape
        out = np.empty((h, w), np.uint8)
        self.lib.oc_bgr2gray(g.ctypes.data, w, h, g.strides[0], ch, out.ctypes.data, out.strides[0])
        return out

    def detect_opencvlike(self, c: CascadeArrays, gray: np.ndarray, min_size=(0, 0), scale_factor: float = 1.1,
                          cap: int = 1 << 20, all_f64: bool = False):
        """cvHaarDetectObjects' scale-cascade path per tempcv.cpp (f64 stage sums over int * float node products,
        f64 products in two_rects stump stages, threshold bias, skip after a reject, stage trees, tilted features):
        the timed CPU baseline and the checker of vj_detect_opencv.  Returns (rects, stats)."""
        h, w = gray.shape
        g = np.ascontiguousarray(gray)
        s, keep = self._cstruct(c)
        out = np.zeros(cap, _RECT_DT)
        n_total = C.c_int(0)
        st = _OcStats()
        fn = self.lib.oc_detect_opencvlike_all_f64 if all_f64 else self.lib.oc_detect_opencvlike   # all_f64: NOT the reference
        n = fn(C.byref(s), g.ctypes.data, w, h, g.strides[0], min_size[0], min_size[1],
               float(scale_factor), out.ctypes.data, cap, C.byref(n_total), C.byref(st))
        return out[:n], {"windows": int(st.windows), "stump_evals": int(st.stump_evals),
                         "stage_entered": [int(v) for v in st.stage_entered[:c.n_stages]]}

    # f1 (next row): grouping
    def group_rectangles(self, xywh: np.ndarray, group_threshold: int, eps: float = 0.2):
        """xywh: (n, 4) int array in detection order -> (grouped (m, 4), weights (m,))."""
        r = np.ascontiguousarray(xywh, np.int32).reshape(-1, 4).copy()
        w = np.zeros(max(len(r), 1), np.int32)
        m = self.lib.oc_group_rectangles(r.ctypes.data, len(r), group_threshold, eps, w.ctypes.data)
        return r[:m], w[:m]

    def xorshift_noise(self, seed: int, h: int, w: int) -> np.ndarray:
        img = np.zeros((h, w), np.uint8)
        self.lib.oc_xorshift_noise(seed, img.ctypes.data, img.size)
        return img

    def xorshift_words(self, seed: int, n: int) -> np.ndarray:
        w = np.zeros(n, np.uint32)
        self.lib.oc_xorshift_words(seed, w.ctypes.data, n)
        return w

    def u64_to_f32(self, v: int) -> np.float32:
        return np.float32(self.lib.oc_u64_to_f32(v))
