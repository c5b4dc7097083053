"""Small fixed workload for rocprofv3 runs (kernel-trace or --pmc): B 1080p frames, R detect calls."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
B = int(os.environ.get("B", "8")); R = int(os.environ.get("R", "2"))
env = Environment(0)
if os.environ.get("SPLIT") is not None: env.configure("pass_split", os.environ["SPLIT"])
for kv in os.environ.get("CONFIGURE", "").split(";"):   # e.g. CONFIGURE="tile_binned=0;concurrent=0"
    if "=" in kv: env.configure(*kv.split("=", 1))
c = Cascade.load(os.environ.get("CASCADE", "frontalface_alt"))
frames = synth.batch(B, int(os.environ.get("H", "1080")), int(os.environ.get("W", "1920")), seed0=1)
t = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
kw = {}
if os.environ.get("SCALES"): kw["scales"] = [int(x) for x in os.environ["SCALES"].split(",")]
for _ in range(R):
    r = env.detect(c, df, default_params(**kw))
print("cascade_ms", r.cascade_ms, [round(x[2], 3) for x in r.passes], "dets", len(r.rects))
