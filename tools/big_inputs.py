import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, default_params, synth, VJ_FLAG_COUNTERS
env = Environment(0)
c = Cascade.load("frontalface_alt")
for (h, w) in ((8192, 8192), (16000, 12000), (3, 200000)):
    try:
        img = synth.frame("blocks", 7, h, w)
        t0 = time.perf_counter(); r = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS)); t1 = time.perf_counter()
        r2 = env.detect(c, img)
        print(h, w, "windows", r.windows, c.count_windows(w, h), "dets", len(r.rects), len(r2.rects), bool(np.array_equal(r.rects, r2.rects)), f"{(t1-t0)*1e3:.0f} ms, second {r2.total_ms:.1f} ms kernels", flush=True)
    except Exception as e:
        print(h, w, "->", repr(e)[:200], flush=True)
e2 = Cascade.load("eye")
many = synth.batch(2048, 100, 100, seed0=3, kinds=("noise", "faces"))
t0 = time.perf_counter(); r = env.detect(e2, many); t1 = time.perf_counter(); r = env.detect(e2, many); t2 = time.perf_counter()
print("2048 x 100x100 eye:", len(r.rects), f"first {(t1-t0)*1e3:.1f} ms, second {(t2-t1)*1e3:.1f} ms")
