"""OpenCV profile on a stage-tree cascade (frontalface_alt_tree): 16 x 1080p frames and one frame per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
env = Environment(0); c = Cascade.load("frontalface_alt_tree")
t = torch.from_numpy(synth.batch(16, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
env.detect_opencv(c, df)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); r = env.detect_opencv(c, df); ts.append((time.perf_counter() - t0) * 1e3)
print(f"16 x 1080p: {min(ts):.1f} ms, {len(r.rects)} detections")
f = synth.frame("blocks", 3, 1080, 1920)
env.detect_opencv(c, f)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = env.detect_opencv(c, f); ts.append((time.perf_counter() - t0) * 1e3)
print(f"1 x 1080p: {min(ts):.2f} ms, {len(r.rects)} detections")
