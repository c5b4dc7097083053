"""Randomized soak of the HIP path against the oracle — more cases than the test suite affords (run by hand on the GPU box:
`python tests/soak_gpu.py [seconds] [first seed] [max width] [max height]`; not collected by pytest).  Every case draws a cascade, a frame kind
and size, size limits, a scale factor, a mode (exhaustive grid, the four CPU variants' skip sets incl. the block variant's f64
grids, the OpenCV profile on tiles and rows, the two-cascade chain with or without grouping, host-supplied regions incl. stage
trees, a batch workload repeated while the chain-balance search runs), a batch size and a few tunables; rectangles and per-stage counts must equal
the oracle's.  Prints one line per failure and a summary; exit code 1 if anything differed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
try:
    import torch  # noqa: F401  (first: see conftest.py)
except Exception:
    pass
from cases import make_frame  # noqa: E402
from clfacedetection_amd import (VJ_FLAG_COUNTERS, VJ_FLAG_GRID_F64, VJ_FLAG_SKIP_LIST, VJ_FLAG_SKIP_ROW, VJ_FLAG_TILTED_AS_UPRIGHT, Cascade, Environment,  # noqa: E402
                                 default_params)
from clfacedetection_amd.api import DATA_DIR  # noqa: E402
from oracle.oracle import Oracle, load_vjc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_w = int(sys.argv[3]) if len(sys.argv) > 3 else 900      # frame sizes are drawn up to max_w x max_h
max_h = int(sys.argv[4]) if len(sys.argv) > 4 else 600
o = Oracle()
env = Environment(0)
NAMES = ["frontalface_alt", "frontalface_default", "frontalface_alt2", "eye", "frontalface_alt_tree", "fullbody", "eye_tree_eyeglasses",
         # the other cascades the reference ships: non-square windows, tilted features (the clod profile reads them as upright rectangles
         # like the reference: VJ_FLAG_TILTED_AS_UPRIGHT), two-node trees with tilted nodes
         "lefteye_2splits", "lowerbody", "mcs_eyepair_big", "mcs_eyepair_small", "mcs_lefteye", "mcs_mouth", "mcs_nose", "mcs_righteye",
         "mcs_upperbody", "profileface", "righteye_2splits", "upperbody"]
CASC = {n: (Cascade.load(n), load_vjc(os.path.join(DATA_DIR, f"haarcascade_{n}.vjc"))) for n in NAMES}
TUNABLES = [("tile_split", ["0", "0.5", "1.3", "0,1.75,2"]), ("blocks_per_cu", ["1", "3", "8"]), ("gather_pairs", ["-1", "0", "2"]),
            ("sp_tail_max", ["0", "16", "48"]), ("thin_pass_spread", ["0", "1"]), ("tree_split_queues", ["0", "1"]), ("concurrent", ["0", "1"]),
            ("tile_classes_kb", ["-2,-1,0", "0,0,0", "24,40,60"]), ("grid_block_w", ["0", "32"]), ("max_subbatch", ["0", "2"]),
            ("group_max", ["2048", "30"]), ("rois_on_device", ["1", "0"]), ("roi_tiles", ["512", "64", "0"]), ("integral_rows", ["2", "0", "1"]), ("wide_tail", ["-1", "0", "1"]), ("min_chunk", ["32", "64", "5"]),
            ("q_slices", ["-1", "1", "5"]), ("gather_waves", ["-1", "3", "4"]), ("cv_tiles", ["1", "0"]), ("cv_tile_ws_max", ["512", "100", "0"]), ("cv_row_blocks", ["-1", "3", "1"]), ("cv_pairs", ["1", "0"]),
            ("cv_tile_min_windows", ["-1", "1536", "256", "64"]), ("cv_tile_min_windows0", ["2048", "512", "64"]), ("auto_balance", ["1", "0"]),
            # round 4: band-major queue pass (switched on for any batch size so that the soak's small batches reach it), chain sweeps of the
            # OpenCV profile's stage trees, the balance keyed on the exact frame count
            ("q_band_px", ["128", "0", "32", "700"]), ("q_group_units", ["4", "1", "16"]), ("q_band_min_frames", ["8", "1", "2"]),
            ("cv_tree_chains", ["1", "0"]), ("cv_tree_chunk", ["64", "256", "100"]), ("cv_tail_max", ["64", "0", "20"]), ("cv_row_band_px", ["128", "0", "37"]), ("cv_tree2", ["1", "0"]), ("cv_tiles_tilted", ["1", "0"]), ("tilted_bands", ["1", "0"]), ("one_pass_max_frames", ["0", "4", "1"]),
            ("cv_tree_chain_blocks", ["2", "1"]), ("cv_tile_min_windows_tree", ["256", "64", "2048"]), ("balance_exact", ["0", "1"])]
DEFAULTS = {"tile_split": "0,1.75,2", "blocks_per_cu": "8", "gather_pairs": "-1", "sp_tail_max": "48", "thin_pass_spread": "1", "tree_split_queues": "1",
            "concurrent": "1", "tile_classes_kb": "-2,-1,0", "grid_block_w": "32", "max_subbatch": "0", "group_max": "2048", "rois_on_device": "1", "roi_tiles": "512", "integral_rows": "2", "wide_tail": "-1", "min_chunk": "32",
            "q_slices": "-1", "gather_waves": "-1", "cv_tiles": "1", "cv_tile_ws_max": "512", "cv_row_blocks": "-1", "cv_tile_min_windows": "-1", "cv_tile_min_windows0": "2048",
            "auto_balance": "1", "q_band_px": "128", "q_group_units": "4", "q_band_min_frames": "8", "cv_tree_chains": "1", "cv_tree_chunk": "64",
            "cv_tail_max": "64", "cv_row_band_px": "128", "cv_tree2": "1", "cv_tiles_tilted": "1", "tilted_bands": "1", "one_pass_max_frames": "0", "cv_pairs": "1", "cv_tree_chain_blocks": "2", "cv_tile_min_windows_tree": "256", "balance_exact": "0"}


def rows(r):
    return [tuple(int(q[k]) for k in ("scale_idx", "x", "y", "w", "h")) for q in r]


t_end = time.time() + budget
n_cases = n_fail = 0
by_mode = {}
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(770000 + seed)
    mode = ["grid", "grid", "grid", "skip_list", "skip_row", "block_row", "block_list", "opencv", "opencv", "chain", "chain_grouped", "rois",
            "rois", "feedback"][int(rng.integers(0, 14))]
    name = NAMES[int(rng.integers(0, len(NAMES)))]
    c, a = CASC[name]
    linear = bool(np.all(a.stage_next == -1))
    tilted = bool(a.node_tilted.any())
    if tilted and mode not in ("opencv", "grid", "skip_list", "skip_row", "block_row", "block_list"):
        mode = "opencv" if rng.random() < 0.5 else "grid"
    tflag = VJ_FLAG_TILTED_AS_UPRIGHT if tilted else 0
    if mode in ("skip_list", "skip_row", "block_row", "block_list") and not linear:
        mode = "grid"
    w = int(rng.integers(c.info.win_w + 11, max_w))
    h = int(rng.integers(c.info.win_h + 11, max_h))
    kind = ["noise", "smooth", "blocks", "faces"][int(rng.integers(0, 4))]
    if kind == "faces" and min(h, w) < 130:
        kind = "blocks"
    img = make_frame(kind, 9000 + seed, h, w)
    nb = int(rng.integers(1, 4))
    tun = {}
    for k, vals in TUNABLES:
        if rng.random() < 0.3:
            tun[k] = vals[int(rng.integers(0, len(vals)))]
    for k, v in tun.items():
        env.configure(k, v)
    desc = (seed, mode, name, kind, h, w, nb, tun)
    ok = True
    try:
        if mode == "feedback":                    # a batch workload repeated: the chain-balance search moves between plans, never the result
            nb = int(rng.integers(8, 12))
            imgs = np.stack([make_frame(kind, 9000 + seed + k, min(h, 420), min(w, 520)) for k in range(nb)])
            p = default_params(scale_factor=[1.1, 1.2][int(rng.integers(0, 2))])
            first = env.detect(c, imgs, p)
            splits = set()
            for _ in range(int(rng.integers(12, 40))):
                r = env.detect(c, imgs, p)
                ok &= np.array_equal(r.rects, first.rects)
                splits.add(r.tile_split)
            f = int(rng.integers(0, nb))
            ro, _ = o.detect(a, imgs[f], scale_factor=[1.1, 1.2][0 if p.scale_factor < 1.15 else 1])
            ok &= rows(first.rects[first.rects["frame"] == f]) == rows(ro)
            desc += (nb, sorted(splits))
        elif mode in ("grid", "skip_list", "skip_row", "block_row", "block_list"):
            mn = (0, 0) if rng.random() < 0.6 else (int(rng.integers(20, 70)),) * 2
            mx = (0, 0) if rng.random() < 0.7 else (int(rng.integers(80, 300)),) * 2
            sf = [1.1, 1.2, 1.05, 1.3, 1.5][int(rng.integers(0, 5))]
            flags = VJ_FLAG_COUNTERS | tflag | {"grid": 0, "skip_list": VJ_FLAG_SKIP_LIST, "skip_row": VJ_FLAG_SKIP_ROW,
                                        "block_row": VJ_FLAG_SKIP_ROW | VJ_FLAG_GRID_F64, "block_list": VJ_FLAG_SKIP_LIST | VJ_FLAG_GRID_F64}[mode]
            p = default_params(flags=flags, min_w=mn[0], min_h=mn[1], max_w=mx[0], max_h=mx[1], scale_factor=sf)
            r = env.detect(c, [img] * nb if nb > 1 else img, p)
            ro, st = o.detect(a, img, min_size=mn, max_size=mx, scale_factor=sf, mode={"grid": None, "skip_list": 2, "skip_row": 3, "block_row": 4, "block_list": 5}[mode])
            for f in range(nb):
                ok &= rows(r.rects[r.rects["frame"] == f]) == rows(ro)
            ok &= r.stage_entered == [v * nb for v in st["stage_entered"]]
            if mode == "grid":      # node evaluations / algorithmic bytes by the oracle's definition (visited nodes), trees included
                ok &= r.stump_evals == st["stump_evals"] * nb and r.gather_bytes == st["gather_bytes"] * nb
            desc += (mn, mx, sf)
        elif mode == "opencv":
            sf = [1.1, 1.2, 1.3][int(rng.integers(0, 3))]
            mn = (0, 0) if rng.random() < 0.7 else (int(rng.integers(24, 60)),) * 2
            r = env.detect_opencv(c, [img] * nb if nb > 1 else img, min_size=mn, scale_factor=sf, flags=VJ_FLAG_COUNTERS)
            ro, st = o.detect_opencvlike(a, img, min_size=mn, scale_factor=sf)
            for f in range(nb):
                ok &= sorted(rows(r.rects[r.rects["frame"] == f])) == sorted(rows(ro))
            ok &= r.stage_entered == [v * nb for v in st["stage_entered"]] and r.windows == st["windows"] * nb
            desc += (mn, sf)
        elif mode == "rois":                      # host-supplied regions of random sizes in a small batch
            if tilted:                            # (stage trees are welcome: the region pass walks them)
                name = "eye"
                c, a = CASC[name]
            imgs = [img, make_frame(kind, 9100 + seed, h, w)][:max(1, min(nb, 2))]
            rois = []
            for _ in range(int(rng.integers(1, 9))):
                rw_, rh_ = int(rng.integers(c.info.win_w + 11, max(c.info.win_w + 12, min(w, 260)))), int(rng.integers(c.info.win_h + 11, max(c.info.win_h + 12, min(h, 260))))
                rw_, rh_ = min(rw_, w), min(rh_, h)
                rois.append((int(rng.integers(0, len(imgs))), int(rng.integers(0, w - rw_ + 1)), int(rng.integers(0, h - rh_ + 1)), rw_, rh_))
            keep = None
            if rng.random() < 0.3:                # a scale mask: the region pass evaluates the selected scales only
                keep = [k for k in range(40) if rng.random() < 0.6] or [0]
            pr = default_params(flags=VJ_FLAG_COUNTERS) if keep is None else default_params(flags=VJ_FLAG_COUNTERS, scales=keep)
            r = env.detect_rois(c, imgs, rois, pr)
            entered = np.zeros(len(r.stage_entered), np.int64)
            for i, (f, x, y, ww, hh) in enumerate(rois):
                ro, st = o.detect(a, np.ascontiguousarray(imgs[f][y:y + hh, x:x + ww]))
                want = rows(ro) if keep is None else [q for q in rows(ro) if q[0] in keep]
                ok &= rows(r.rects[r.rects["frame"] == i]) == want
                entered += np.array(st["stage_entered"], np.int64)
            if keep is None:
                ok &= r.stage_entered == entered.tolist()
            desc += (rois, keep)
        else:
            if not linear or name == "eye":
                name = "frontalface_alt2"
                c, a = CASC[name]
            c2, a2 = CASC["eye"]
            mnb = 0 if mode == "chain" else int(rng.integers(1, 4))
            skip = int(rng.integers(0, 6)) if mode == "chain" else 0      # 1: row skip rule, 2: list skip rule on both cascades (host hand-off)
            flag1, omode = {1: (VJ_FLAG_SKIP_ROW, 3), 2: (VJ_FLAG_SKIP_LIST, 2)}.get(skip, (0, None))
            r1, r2 = env.detect_chain(c, c2, [img] * nb if nb > 1 else img, default_params(min_neighbors=mnb, flags=flag1), default_params(flags=flag1))
            ro, _ = o.detect(a, img, mode=omode)
            if mnb:
                xywh = np.stack([ro[k] for k in ("x", "y", "w", "h")], 1) if len(ro) else np.zeros((0, 4), np.int32)
                g, wt = o.group_rectangles(xywh, mnb)
                want1 = [(int(q[0]), int(q[1]), int(q[2]), int(q[3]), int(n)) for q, n in zip(g, wt)]
                got1 = [(int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"]), int(q["weight"])) for q in r1.rects[r1.rects["frame"] == 0]]
            else:
                want1 = [(int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"]), 0) for q in ro]
                got1 = [(int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"]), 0) for q in r1.rects[r1.rects["frame"] == 0]]
            ok &= got1 == want1
            for i, (x, y, ww, hh, _) in enumerate(want1[:6]):
                r2o, _ = o.detect(a2, np.ascontiguousarray(img[y:y + hh, x:x + ww]), mode=omode)
                ok &= rows(r2.rects[r2.rects["frame"] == i]) == rows(r2o)
            desc += (mnb, len(want1), skip)
    except Exception as e:   # noqa: BLE001
        ok = False
        desc += (repr(e),)
    for k in tun:
        env.configure(k, DEFAULTS[k])
        if k == "tile_split":
            env.configure("auto_balance", "reset")     # (a hand-set split switches the feedback off until it is reset)
    n_cases += 1
    by_mode[mode] = by_mode.get(mode, 0) + 1
    if not ok:
        n_fail += 1
        print("FAIL", desc, flush=True)
    if n_cases % 25 == 0:
        print(f"{n_cases} cases, {n_fail} failures, seed {seed}", flush=True)
    seed += 1
print(f"done: {n_cases} cases {by_mode}, {n_fail} failures, seeds {seed0}..{seed - 1}")
sys.exit(1 if n_fail else 0)
