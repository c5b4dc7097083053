"""Regenerates tests/golden/*.json from the CPU oracle (run in the build container).

The oracle itself is pinned to the reference by tests/test_oracle_pins.py (figures the
survey recorded from runs of the reference's own code); these fixtures then freeze the
oracle's outputs on more inputs so that the GPU box — which has neither the reference
nor its XMLs — can check the HIP path against committed data as well as against the
live oracle.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import DETECT_CASES, HEADLINE_CASE, INTEGRAL_CASES, make_frame, sha  # noqa: E402
from clfacedetection_amd.api import DATA_DIR  # noqa: E402
from oracle.oracle import Oracle, load_vjc  # noqa: E402

o = Oracle()
G = os.path.join(ROOT, "tests", "golden")


def detect_case(case):
    cid, casc, gen, seed, h, w, mn, mx, sm = case
    c = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{casc}.vjc"))
    img = make_frame(gen, seed, h, w, o)
    r, st = o.detect(c, img, min_size=mn, max_size=mx, signed_mean=sm)
    return {"id": cid, "cascade": casc, "generator": gen, "seed": seed, "height": h, "width": w,
            "min_size": list(mn), "max_size": list(mx), "signed_mean": sm, "image_sha256": sha(img),
            "rects": [[int(v) for v in (q["scale_idx"], q["x"], q["y"], q["w"], q["h"])] for q in r],
            "windows": st["windows"], "stump_evals": st["stump_evals"], "rect_evals": st["rect_evals"],
            "gather_bytes": st["gather_bytes"], "stage_entered": st["stage_entered"]}


det = [detect_case(c) for c in DETECT_CASES + [HEADLINE_CASE]]
json.dump(det, open(os.path.join(G, "detect.json"), "w"), indent=1)
for d in det:
    print(d["id"], len(d["rects"]), d["windows"], d["stump_evals"])

integ = []
for cid, gen, seed, h, w in INTEGRAL_CASES:
    img = make_frame(gen, seed, h, w, o)
    s, q = o.integral(img)
    integ.append({"id": cid, "generator": gen, "seed": seed, "height": h, "width": w, "image_sha256": sha(img),
                  "sum_sha256": sha(s), "sqsum_sha256": sha(q), "sum_last": int(s[-1, -1]), "sqsum_last": int(q[-1, -1])})
json.dump(integ, open(os.path.join(G, "integral.json"), "w"), indent=1)

scales = []
for casc, W, H, mn, mx in [("frontalface_default", 640, 480, (0, 0), (0, 0)), ("frontalface_default", 640, 480, (40, 40), (0, 0)),
                           ("frontalface_alt", 640, 480, (0, 0), (0, 0)), ("frontalface_alt", 1920, 1080, (0, 0), (0, 0)),
                           ("frontalface_alt2", 1280, 720, (0, 0), (0, 0)), ("frontalface_alt_tree", 4096, 4096, (0, 0), (0, 0)),
                           ("eye", 100, 80, (0, 0), (60, 60))]:
    c = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{casc}.vjc"))
    sc = o.plan_scales(c, W, H, mn, mx)
    scales.append({"cascade": casc, "width": W, "height": H, "min_size": list(mn), "max_size": list(mx),
                   "windows": sum(s.nx * s.ny for s in sc if s.accepted),
                   "scales": [[s.scale_idx, float(np.float32(s.scale)).hex(), float(np.float32(s.step)).hex(), s.win_w,
                               s.win_h, s.equ_x, s.equ_w, s.equ_h, s.area, s.nx, s.ny, s.accepted] for s in sc]})
json.dump(scales, open(os.path.join(G, "scales.json"), "w"), indent=1)

# the other evaluation modes: the CPU variants' window sets (oracle modes 2 and 3) and the OpenCV-like path
from cases import MODE_CASES  # noqa: E402
modes = []
for cid, casc, gen, seed, h, w in MODE_CASES:
    c = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{casc}.vjc"))
    img = make_frame(gen, seed, h, w, o)
    e = {"id": cid, "cascade": casc, "generator": gen, "seed": seed, "height": h, "width": w, "image_sha256": sha(img)}
    if not c.node_tilted.any() and bool(np.all(c.stage_next == -1)):
        for name, mode in (("skip_list", 2), ("skip_row", 3)):
            r, st = o.detect(c, img, mode=mode)
            e[name] = {"rects": [[int(v) for v in (q["scale_idx"], q["x"], q["y"], q["w"], q["h"])] for q in r],
                       "stage_entered": st["stage_entered"]}
    r, st = o.detect_opencvlike(c, img)
    e["opencv"] = {"rects": sorted([int(v) for v in (q["scale_idx"], q["x"], q["y"], q["w"], q["h"])] for q in r),
                   "windows": st["windows"], "stage_entered": st["stage_entered"]}
    modes.append(e)
    print(cid, {k: len(v["rects"]) for k, v in e.items() if isinstance(v, dict)})
json.dump(modes, open(os.path.join(G, "modes.json"), "w"), indent=1)
print("wrote", os.listdir(G))


# faces -> grouped faces -> eyes inside them (vj_detect_chain with min_neighbors != 0): the oracle's candidates, its
# restatement of cv::groupRectangles, and its candidates of the second cascade on every grouped face's sub-image
from cases import GROUP_CASES  # noqa: E402
groups = []
for cid, c1, c2, seed, h, w, mn in GROUP_CASES:
    a1 = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{c1}.vjc"))
    a2 = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{c2}.vjc"))
    img = make_frame("faces", seed, h, w, o)
    r, _ = o.detect(a1, img)
    xywh = np.stack([r[k] for k in ("x", "y", "w", "h")], 1) if len(r) else np.zeros((0, 4), np.int32)
    g, wt = o.group_rectangles(xywh, max(mn, 1))
    second = []
    for q in g:
        x, y, ww, hh = (int(v) for v in q)
        r2, _ = o.detect(a2, np.ascontiguousarray(img[y:y + hh, x:x + ww]))
        second.append([[int(v) for v in (e["scale_idx"], e["x"], e["y"], e["w"], e["h"])] for e in r2])
    groups.append({"id": cid, "first": c1, "second": c2, "seed": seed, "height": h, "width": w, "min_neighbors": mn,
                   "image_sha256": sha(img), "raw_candidates": len(r),
                   "faces": [[int(v) for v in q] + [int(n)] for q, n in zip(g, wt)], "inside": second})
    print(cid, len(r), len(g), sum(len(x) for x in second))
json.dump(groups, open(os.path.join(G, "groups.json"), "w"), indent=1)
