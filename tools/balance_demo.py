"""Feedback chain balancing (vj_env auto_balance) against the static tile_split defaults on batch workloads.
Usage on the GPU box:  python tools/balance_demo.py        -> one line per workload: static ms, balanced ms, split found"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
env = Environment(0)
def timed(c, df, n=6):
    ms = []
    for _ in range(n):
        r = env.detect(c, df)
        ms.append(r.cascade_ms)
    return float(np.median(ms)), r
for casc, B, H, W in (("frontalface_alt", 64, 1080, 1920), ("frontalface_default", 64, 1080, 1920), ("frontalface_default", 64, 480, 640),
                      ("frontalface_alt2", 256, 720, 1280), ("eye", 64, 720, 1280), ("frontalface_alt", 16, 1080, 1920),
                      ("frontalface_alt_tree", 1, 4096, 4096), ("frontalface_alt", 1, 4096, 4096)):
    c = Cascade.load(casc)
    t = torch.from_numpy(synth.batch(B, H, W, seed0=1)).cuda(); torch.cuda.synchronize()
    df = DeviceFrames.from_torch(t)
    env.configure("auto_balance", "0")
    for _ in range(3): env.detect(c, df)
    static_ms, rs = timed(c, df)
    env.configure("auto_balance", "1")
    calls = 0
    last = []
    while True:                       # the first calls of a new workload: three per candidate; over when nine calls in a row agree
        r = env.detect(c, df); calls += 1
        last = (last + [(r.tile_split, len(r.launches), tuple(l["lds_class"] for l in r.launches))])[-9:]
        if (len(last) == 9 and len(set(last)) == 1) or calls > 120: break
    bal_ms, rb = timed(c, df)
    same = bool(np.array_equal(rs.rects, rb.rects))
    print(f"{casc} {B}x{W}x{H}: static split {rs.tile_split:.2f} {static_ms:.2f} ms | balanced split {rb.tile_split:.2f} {bal_ms:.2f} ms "
          f"({100 * (static_ms / bal_ms - 1):+.1f} %, {len(rb.launches)} launches against {len(rs.launches)}) after {calls} calls, same rectangles: {same}", flush=True)
    del t
