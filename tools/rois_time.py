"""vj_detect_rois on host-supplied regions of many different sizes (the grouped faces of config 5): wall time per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); face = Cascade.load("frontalface_alt2"); eye = Cascade.load("eye")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = synth.batch(B, 720, 1280, seed0=5001, kinds=("faces", "noise", "smooth", "blocks"))
t = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
r1, r2 = env.detect_chain(face, eye, df, default_params(min_neighbors=3))
rois = [(int(r["frame"]), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in r1.rects]
print(len(rois), "regions,", len({(r[3], r[4]) for r in rois}), "distinct sizes; chain found", len(r2.rects), "eyes")
for rep in range(3):
    t0 = time.perf_counter(); h = env.detect_rois(eye, df, rois); dt = (time.perf_counter() - t0) * 1e3
    print(f"vj_detect_rois: {dt:.1f} ms, {len(h.rects)} eyes", flush=True)
