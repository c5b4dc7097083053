"""OpenCV-profile timing on the bench frames (64 x 1080p by default).  Usage on the GPU box:
     python tools/cv_time.py [cascade[,cascade...]] [frames] [key=value ...]      (keys of vj_env_configure, e.g. cv_tiles=0)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
args = [a for a in sys.argv[1:] if "=" not in a]
names = args[0].split(",") if args else ["frontalface_alt"]
B = int(args[1]) if len(args) > 1 else 64
env = Environment(0)
H, W = int(os.environ.get("H", "1080")), int(os.environ.get("W", "1920"))      # frame size: H=720 W=1280 python tools/cv_time.py ...
t = torch.from_numpy(synth.batch(B, H, W, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
for kv in sys.argv[1:]:
    if "=" in kv:
        env.configure(*kv.split("=", 1))
for name in names:
    c = Cascade.load(name)
    env.detect_opencv(c, df)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = env.detect_opencv(c, df); ts.append(time.perf_counter() - t0)
    print(name, B, "frames", " ".join(a for a in sys.argv[1:] if "=" in a), "ms", round(min(ts) * 1e3, 1), "kernel ms", round(r.cascade_ms, 1), "dets", len(r.rects), flush=True)
