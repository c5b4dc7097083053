import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from clfacedetection_amd import Cascade, Environment, default_params, synth, VjError, VJ_FLAG_COUNTERS
from clfacedetection_amd.api import DATA_DIR
from oracle.oracle import Oracle, load_vjc
o = Oracle(); env = Environment(0)
c = Cascade.load("frontalface_alt"); a = load_vjc(os.path.join(DATA_DIR, "haarcascade_frontalface_alt.vjc"))
img = synth.frame("faces", 3, 360, 640)
def rows(r): return [tuple(int(q[k]) for k in ("scale_idx", "x", "y", "w", "h")) for q in r]
for sf in (1.01, 1.02, 1.03, 1.5, 3.0, 10.0, 1.0000001):
    try:
        r = env.detect(c, img, default_params(scale_factor=sf, flags=VJ_FLAG_COUNTERS))
        ro, st = o.detect(a, img, scale_factor=sf)
        print("sf", sf, "scales", len(c.plan_scales(640, 360, default_params(scale_factor=sf))) if hasattr(c, "plan_scales") else "?", "match", rows(r.rects) == rows(ro), len(r.rects))
    except Exception as e:
        print("sf", sf, "->", type(e).__name__, str(e)[:150])
for mn, mx in (((700, 700), (0, 0)), ((0, 0), (10, 10)), ((100, 100), (50, 50)), ((-5, -5), (0, 0))):
    try:
        r = env.detect(c, img, default_params(min_w=mn[0], min_h=mn[1], max_w=mx[0], max_h=mx[1]))
        ro, st = o.detect(a, img, min_size=mn, max_size=mx)
        print("min", mn, "max", mx, "match", rows(r.rects) == rows(ro), len(r.rects))
    except Exception as e:
        print("min", mn, "max", mx, "->", type(e).__name__, str(e)[:150])
for sf in (1.01, 1.05, 2.0):
    try:
        r = env.detect_opencv(c, img, scale_factor=sf)
        ro, st = o.detect_opencvlike(a, img, scale_factor=sf)
        print("cv sf", sf, "match", sorted(rows(r.rects)) == sorted(rows(ro)), len(r.rects))
    except Exception as e:
        print("cv sf", sf, "->", type(e).__name__, str(e)[:150])
