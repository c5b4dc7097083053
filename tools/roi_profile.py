"""What the region pass of BASELINE config 5 (every raw candidate a region) is made of: region sizes, windows and node evaluations of
the second cascade, and the step with the region tiles on / off.  python tools/roi_profile.py [key=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import VJ_FLAG_COUNTERS, Cascade, DeviceFrames, Environment, default_params, synth

env = Environment(0)
for kv in sys.argv[1:]:
    if "=" in kv:
        env.configure(*kv.split("=", 1))
face, eye = Cascade.load("frontalface_alt2"), Cascade.load("eye")
df = DeviceFrames.from_torch(torch.from_numpy(synth.batch(256, 720, 1280, seed0=5001, kinds=("faces", "noise", "smooth", "blocks"))).cuda())
r1, r2 = env.detect_chain(face, eye, df, default_params(flags=VJ_FLAG_COUNTERS), default_params(flags=VJ_FLAG_COUNTERS))
w = r1.rects["w"]
print("regions", len(w), "width percentiles 5/25/50/75/95:", np.percentile(w, [5, 25, 50, 75, 95]).tolist(), "frames with regions", len(np.unique(r1.rects["frame"])))
print("second cascade: windows", r2.windows, "node evaluations", r2.stump_evals, "per window", round(r2.stump_evals / max(1, r2.windows), 2))
print("stage_entered", r2.stage_entered)
print("first cascade: windows", r1.windows, "node evaluations", r1.stump_evals)
for tiles in ("1", "0"):
    env.configure("roi_tiles", tiles)
    p = default_params()
    env.detect_chain(face, eye, df, p)
    lat = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        a, b = env.detect_chain(face, eye, df, p)
        lat.append((time.perf_counter() - t) * 1e3)
    print(f"roi_tiles={tiles}: step {np.median(lat):.2f} ms, first cascade {a.cascade_ms:.2f}, grouping + second {b.cascade_ms:.2f}", flush=True)
