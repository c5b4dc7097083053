import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, default_params, synth
env = Environment(0)
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
for name, h, w, mn in (("frontalface_default", 480, 640, 3), ("frontalface_alt", 480, 640, 0), ("frontalface_alt", 240, 320, 0), ("eye", 100, 100, 0)):
    c = Cascade.load(name); img = synth.frame("faces" if h >= 240 else "noise", 2, h, w)
    p = default_params(min_neighbors=mn)
    for _ in range(3): env.detect(c, img, p)
    lat, ker = [], []
    for _ in range(50):
        t0 = time.perf_counter(); r = env.detect(c, img, p); lat.append((time.perf_counter() - t0) * 1e3); ker.append(r.total_ms)
    print(f"{name} {w}x{h} minNeighbors {mn}: wall p50 {np.percentile(lat,50):.3f} ms, kernels p50 {np.percentile(ker,50):.3f} ms, {len(r.rects)} rects | " +
          " ".join(f"{x['kind']}{x['lds_class']}:{x['ms']:.3f}" for x in r.launches) + f" integral {r.integral_ms:.3f}", flush=True)
