import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
env = Environment(0)
t = torch.from_numpy(synth.batch(64, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
for name in ("frontalface_alt",):
    c = Cascade.load(name)
    env.detect_opencv(c, df)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = env.detect_opencv(c, df); ts.append(time.perf_counter() - t0)
    print(name, "ms", round(min(ts) * 1e3, 1), "kernel ms", round(r.cascade_ms, 1), "dets", len(r.rects), flush=True)
