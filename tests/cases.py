"""Seeded parity cases shared by the golden generator (tools/make_golden.py), the CPU
tests (oracle vs fixtures) and the GPU tests (HIP path vs oracle and vs fixtures)."""
from __future__ import annotations

import hashlib

import numpy as np

from clfacedetection_amd import synth

# (id, cascade, generator, seed, height, width, min_size, max_size, signed_mean)
DETECT_CASES = [
    ("alt_xs12345_480", "frontalface_alt", "xorshift", 12345, 480, 640, (0, 0), (0, 0), False),
    ("default_xs12345_480", "frontalface_default", "xorshift", 12345, 480, 640, (0, 0), (0, 0), False),
    ("alt_noise_240", "frontalface_alt", "noise", 7, 240, 320, (0, 0), (0, 0), False),
    ("alt_smooth_480", "frontalface_alt", "smooth", 11, 480, 640, (0, 0), (0, 0), False),
    ("alt_blocks_odd", "frontalface_alt", "blocks", 5, 251, 333, (0, 0), (0, 0), False),
    ("default_min40", "frontalface_default", "noise", 21, 480, 640, (40, 40), (0, 0), False),
    ("default_minmax", "frontalface_default", "smooth", 22, 360, 480, (30, 30), (120, 120), False),
    ("eye_noise_300", "eye", "noise", 31, 300, 400, (0, 0), (0, 0), False),
    ("eye_blocks_200", "eye", "blocks", 32, 200, 200, (0, 0), (0, 0), False),
    ("alt_signed_mean", "frontalface_alt", "noise", 41, 240, 320, (0, 0), (0, 0), True),
    ("alt_tiny", "frontalface_alt", "noise", 51, 31, 31, (0, 0), (0, 0), False),
    ("alt_wide", "frontalface_alt", "noise", 52, 40, 700, (0, 0), (0, 0), False),
    ("alt2_noise_240", "frontalface_alt2", "noise", 61, 240, 320, (0, 0), (0, 0), False),
    ("alt2_smooth_300", "frontalface_alt2", "smooth", 62, 300, 400, (0, 0), (0, 0), False),
    ("alt_tree_noise_240", "frontalface_alt_tree", "noise", 71, 240, 320, (0, 0), (0, 0), False),
    ("alt_tree_blocks_300", "frontalface_alt_tree", "blocks", 72, 300, 400, (0, 0), (0, 0), False),
    # cascades the reference ships that the configs do not name: non-square base windows, and tilted features, which the clod
    # path reads as upright rectangles (clod.cpp:448-492; the GPU test calls at the reference's signature, which passes the flag)
    ("eyepair_small_noise", "mcs_eyepair_small", "noise", 91, 200, 320, (0, 0), (0, 0), False),     # 22 x 5
    ("eyepair_big_blocks", "mcs_eyepair_big", "blocks", 92, 240, 400, (0, 0), (0, 0), False),       # 45 x 11
    ("lowerbody_blocks", "lowerbody", "blocks", 93, 240, 320, (0, 0), (0, 0), False),               # 19 x 23
    ("profileface_noise", "profileface", "noise", 94, 240, 320, (0, 0), (0, 0), False),
    ("righteye_2splits_smooth", "righteye_2splits", "smooth", 95, 240, 320, (0, 0), (0, 0), False), # two-node trees with tilted nodes
    ("mcs_mouth_minmax", "mcs_mouth", "blocks", 96, 300, 400, (30, 18), (150, 90), False),          # 25 x 15, size limits
]
# the headline pin: the survey's recorded reference run
HEADLINE_CASE = ("alt_xs12345_1080", "frontalface_alt", "xorshift", 12345, 1080, 1920, (0, 0), (0, 0), False)

# the other evaluation modes (tests/golden/modes.json): P2 skip variants and the OpenCV arithmetic profile
MODE_CASES = [  # (id, cascade, generator, seed, height, width)
    ("m_alt_xs_480", "frontalface_alt", "xorshift", 12345, 480, 640),
    ("m_alt_smooth", "frontalface_alt", "smooth", 11, 300, 420),
    ("m_default_blocks", "frontalface_default", "blocks", 6, 360, 480),
    ("m_alt2_noise", "frontalface_alt2", "noise", 61, 240, 320),
    ("m_eye_noise", "eye", "noise", 31, 200, 260),
    ("m_alt_tree_blocks", "frontalface_alt_tree", "blocks", 72, 300, 400),
    ("m_fullbody_noise", "fullbody", "noise", 81, 240, 320),
    ("m_eyeglasses_smooth", "eye_tree_eyeglasses", "smooth", 84, 240, 320),
    ("m_mcs_nose_noise", "mcs_nose", "noise", 85, 200, 260),                 # 18 x 15, 990 tilted nodes
    ("m_upperbody_blocks", "upperbody", "blocks", 86, 240, 320),             # 22 x 18
    ("m_profileface_smooth", "profileface", "smooth", 87, 240, 320),         # upright stumps: skip modes too
    ("m_lefteye_2splits_noise", "lefteye_2splits", "noise", 88, 240, 320),
    ("m_eyepair_big_blocks", "mcs_eyepair_big", "blocks", 89, 200, 400),     # 45 x 11
]

GROUP_CASES = [  # (id, first cascade, second cascade, seed, height, width, min_neighbors): drawn faces (synth kind "faces")
    ("g_alt2_eye_3", "frontalface_alt2", "eye", 1, 360, 640, 3),
    ("g_alt_eye_1", "frontalface_alt", "eye", 2, 300, 480, 1),
    ("g_default_eye_3", "frontalface_default", "eye", 3, 360, 640, 3),
]

INTEGRAL_CASES = [  # (id, generator, seed, height, width)
    ("i_1x1", "noise", 1, 1, 1), ("i_3x5", "noise", 2, 3, 5), ("i_8x256", "noise", 3, 8, 256),
    ("i_9x257", "noise", 4, 9, 257), ("i_odd", "smooth", 5, 251, 333), ("i_vga", "noise", 6, 480, 640),
    ("i_white", "white", 0, 64, 300), ("i_1080p", "noise", 7, 1080, 1920),
]


# BASELINE.json's configs at their full sizes (tests/golden/fullsize.json, tools/make_fullsize_golden.py): the oracle
# runs once in the build container, the GPU suite compares whole results through rows_sha()
FULLSIZE = {
    "config3": {"cascade": "frontalface_alt", "frames": 64, "height": 1080, "width": 1920, "seed0": 1,
                "kinds": ["noise", "smooth", "blocks"]},
    "config4": {"cascade": "frontalface_alt_tree", "kind": "noise", "seed": 4096, "height": 4096, "width": 4096},
    "config5_raw": {"cascade": "frontalface_alt2", "second": "eye", "frames": 256, "height": 720, "width": 1280,
                    "seed0": 5001, "kinds": ["noise", "smooth", "blocks"]},
    "config5_grouped": {"cascade": "frontalface_alt2", "second": "eye", "frames": 256, "height": 720, "width": 1280,
                        "seed0": 5001, "kinds": ["faces", "noise", "smooth", "blocks"], "min_neighbors": 3},
    # (id, cascade, generator, seed, height, width)
    "opencv": [("cv_alt_pin_1080", "frontalface_alt", "xorshift", 12345, 1080, 1920),
               ("cv_alt_smooth_1080", "frontalface_alt", "smooth", 2, 1080, 1920),
               ("cv_alt_blocks_1080", "frontalface_alt", "blocks", 3, 1080, 1920),
               ("cv_alt_faces_1080", "frontalface_alt", "faces", 4, 1080, 1920),
               ("cv_alt_tree_noise_1080", "frontalface_alt_tree", "noise", 5, 1080, 1920),
               ("cv_alt_tree_faces_1080", "frontalface_alt_tree", "faces", 6, 1080, 1920),
               ("cv_alt2_faces_1080", "frontalface_alt2", "faces", 7, 1080, 1920),
               ("cv_fullbody_smooth_1080", "fullbody", "smooth", 8, 1080, 1920)],
    # (id, cascade, generator, seed, height, width): cascades the configs do not name, at 1080p in BOTH profiles (the clod profile reads
    # tilted rectangles as upright ones, like the reference)
    "shipped": [("s_upperbody_blocks_1080", "upperbody", "blocks", 11, 1080, 1920),
                ("s_mcs_mouth_faces_1080", "mcs_mouth", "faces", 12, 1080, 1920),
                ("s_eyepair_small_noise_1080", "mcs_eyepair_small", "noise", 13, 1080, 1920),
                ("s_lowerbody_smooth_1080", "lowerbody", "smooth", 14, 1080, 1920),
                ("s_righteye_2splits_faces_1080", "righteye_2splits", "faces", 15, 1080, 1920),
                ("s_profileface_faces_1080", "profileface", "faces", 16, 1080, 1920)],
    # (id, cascade, generator, seed, height, width, oracle mode): the CPU variants' window sets at 1080p
    "modes": [(f"m{mode}_{kind}_1080", "frontalface_alt", kind, seed, 1080, 1920, mode)
              for mode in (2, 3, 4, 5) for kind, seed in (("noise", 1), ("smooth", 2), ("faces", 4))],
}


def rows_sha(rects, keys=("scale_idx", "x", "y", "w", "h")) -> str:
    """SHA-256 of the rows of a rectangle list as little-endian int32, in the order given (results are sorted by
    (scale, y, x) on both sides).  `rects`: a structured array (fields `keys`) or a list of int tuples."""
    if isinstance(rects, np.ndarray) and rects.dtype.names:
        a = np.stack([rects[k].astype("<i4") for k in keys], 1) if len(rects) else np.zeros((0, len(keys)), "<i4")
    else:
        a = np.asarray(list(rects), "<i4").reshape(len(rects), -1) if len(rects) else np.zeros((0, 0), "<i4")
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_frame(generator: str, seed: int, h: int, w: int, oracle=None) -> np.ndarray:
    if generator == "xorshift":   # the survey's sequential generator lives in the oracle
        return oracle.xorshift_noise(seed, h, w)
    if generator == "white":
        return np.full((h, w), 255, np.uint8)
    return synth.frame(generator, seed, h, w)


def block_grid_frame(transposed: bool = False) -> np.ndarray:
    """A 900 x 300 (or 300 x 900) frame with a drawn face of 238 pixels at column (row) 650.  Scale 26 of a 20 x 20
    cascade (s = 11.918..., window 238) puts grid index 55 on pixel 656 when the product index * step is the f32 one
    (656.50 after rounding to 24 bits) and on 655 when it is the f64 one (655.4999...): the reference's block variant
    (double step, clod.cpp:862) and its other loops then evaluate different windows over the face, and with
    frontalface_alt they return different rectangles (x = 655 vs 656)."""
    face = np.clip(synth.crude_face(238), 0, 255).astype(np.uint8)
    if transposed:
        img = synth.frame("smooth", 26, 900, 300).copy()
        img[650:888, 20:258] = face
    else:
        img = synth.frame("smooth", 26, 300, 900).copy()
        img[20:258, 650:888] = face
    return img


BLOCK_GRID_LIMITS = {"min_size": (230, 230), "max_size": (270, 270)}     # scales 26 and 27 only


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------- OpenCV-profile arithmetic probes
def crafted_stump_cascade(three_rects: bool, threshold: float):
    """One stage, one stump, 20x20 window; leaves 0 / 1 and stage threshold 0.5: the window is a detection exactly
    when the node sum is >= threshold * vnf.  The rectangles nearly cancel (like real Haar features), so the rounding
    of each product shows in the sum."""
    from oracle.oracle import CascadeArrays
    c = CascadeArrays()
    c.win_w = c.win_h = 20
    c.name = "crafted"
    c.stage_first_tree = np.array([0], np.int32)
    c.stage_n_trees = np.array([1], np.int32)
    c.stage_threshold = np.array([0.5], np.float32)
    c.stage_parent = np.array([-1], np.int32)
    c.stage_next = np.array([-1], np.int32)
    c.stage_child = np.array([-1], np.int32)
    c.tree_first_node = np.array([0], np.int32)
    c.tree_n_nodes = np.array([1], np.int32)
    c.tree_first_alpha = np.array([0], np.int32)
    rects = [[2, 2, 16, 16], [2, 2, 8, 16], [10, 4, 4, 8] if three_rects else [0, 0, 0, 0]]
    c.node_rect = np.array(rects, np.int32).reshape(-1)
    c.node_weight = np.array([-1.0, 2.0, 0.5 if three_rects else 0.0], np.float32)
    c.node_threshold = np.array([threshold], np.float32)
    c.node_left = np.array([0], np.int32)
    c.node_right = np.array([-1], np.int32)
    c.node_tilted = np.array([0], np.int32)
    c.alpha = np.array([0.0, 1.0], np.float32)
    return c


def single_window_frame(seed: int, size: int = 600):
    """(bright noisy frame, factor): at the largest factor of cvHaarDetectObjects' loop for a 20x20 cascade the frame
    holds exactly one window, at (0, 0), whose rectangle sums are far above 2^24."""
    rng = np.random.default_rng(seed)
    img = (180 + rng.integers(0, 76, (size, size))).astype(np.uint8)
    factor, n = 1.0, 0
    while factor * 20 < size - 10:
        last = factor
        factor *= 1.1
    return img, last


def cascade_to_product(c):
    """oracle CascadeArrays -> product Cascade through vj_cascade_from_arrays."""
    from clfacedetection_amd import Cascade
    from clfacedetection_amd.api import NODE_DTYPE, STAGE_DTYPE, TREE_DTYPE
    st = np.zeros(c.n_stages, STAGE_DTYPE)
    st["first_tree"], st["n_trees"], st["threshold"] = c.stage_first_tree, c.stage_n_trees, c.stage_threshold
    st["parent"], st["next"], st["child"] = c.stage_parent, c.stage_next, c.stage_child
    tr = np.zeros(c.n_trees, TREE_DTYPE)
    tr["first_node"], tr["n_nodes"], tr["first_alpha"] = c.tree_first_node, c.tree_n_nodes, c.tree_first_alpha
    nd = np.zeros(c.n_nodes, NODE_DTYPE)
    r = c.node_rect.reshape(-1, 3, 4)
    w = c.node_weight.reshape(-1, 3)
    for k, f in enumerate(("x", "y", "w", "h")):
        nd["rect"][f] = r[:, :, k]
    nd["rect"]["weight"] = w
    nd["n_rects"] = (w != 0).sum(1)
    nd["tilted"] = c.node_tilted if len(c.node_tilted) == c.n_nodes else 0
    nd["threshold"], nd["left"], nd["right"] = c.node_threshold, c.node_left, c.node_right
    return Cascade.from_arrays(c.win_w, c.win_h, st, tr, nd, c.alpha)


def as_stage_tree(c, split_at: int = 4):
    """A copy of a linear cascade re-linked like frontalface_alt_tree: stages 0..split_at form a chain, then two
    chains alternate (split_at+1, +3, +5, ... and split_at+2, +4, ...), the second one taking over when the first rejects
    (parent / next / child as icvReadHaarClassifier would have set them)."""
    import copy
    t = copy.deepcopy(c)
    n = t.n_stages
    parent = np.full(n, -1, np.int32)
    nxt = np.full(n, -1, np.int32)
    for i in range(1, n):
        parent[i] = i - 1 if i <= split_at + 1 else i - 2
    parent[split_at + 2] = split_at
    nxt[split_at + 1] = split_at + 2
    child = np.full(n, -1, np.int32)
    for i in range(n):
        if parent[i] != -1 and child[parent[i]] == -1:
            child[parent[i]] = i
    t.stage_parent, t.stage_next, t.stage_child = parent, nxt, child
    return t
