import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0)
for name, (H, W) in (("frontalface_alt", (1080, 1920)), ("frontalface_default", (480, 640)), ("frontalface_alt", (240, 320))):
    c = Cascade.load(name)
    f = synth.frame("noise", 3, H, W)
    pin = env.host_alloc((H, W)); pin[...] = f
    d = torch.from_numpy(f[None]).cuda(); df = DeviceFrames.from_torch(d)
    for label, src in (("pageable numpy", f), ("page-locked (vj_host_alloc)", pin), ("device-resident", df)):
        for _ in range(10): env.detect(c, src)
        ws, ks, ig = [], [], []
        for _ in range(100):
            t = time.perf_counter(); r = env.detect(c, src); ws.append((time.perf_counter() - t) * 1e3); ks.append(r.total_ms); ig.append(r.integral_ms)
        print(f"{name} {W}x{H} {label}: wall p50 {np.percentile(ws, 50):.3f} ms, kernels (events) {np.percentile(ks, 50):.3f} ms, integral {np.percentile(ig, 50):.3f} | launches {len(r.launches)}")
    env.host_free(pin)
