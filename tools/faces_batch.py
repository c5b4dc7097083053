import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
for kinds in (("noise", "smooth", "blocks"), ("faces",)):
    t = torch.from_numpy(synth.batch(64, 1080, 1920, seed0=1, kinds=kinds)).cuda(); torch.cuda.synchronize()
    df = DeviceFrames.from_torch(t)
    for _ in range(2): env.detect(c, df)
    ms = []; lm = None
    for _ in range(5):
        r = env.detect(c, df); ms.append(r.total_ms)
        l = [x["ms"] for x in r.launches]; lm = l if lm is None else [a + b for a, b in zip(lm, l)]
    print(kinds, f"total {np.mean(ms):.2f} ms, {len(r.rects)} rects | " + " ".join(f"{x['kind']}{x['lds_class']}:{y/5:.2f}" for x, y in zip(r.launches, lm)), flush=True)
