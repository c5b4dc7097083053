import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
for kind in ("noise", "faces"):
    frames = [synth.frame(kind, s, 1080, 1920) for s in range(1, 5)]
    for f in frames: env.detect(c, f)
    lat, ker, lm = [], [], None
    for i in range(40):
        t0 = time.perf_counter(); r = env.detect(c, frames[i % 4]); lat.append((time.perf_counter() - t0) * 1e3); ker.append(r.total_ms)
        l = [x["ms"] for x in r.launches]; lm = l if lm is None else [a + b for a, b in zip(lm, l)]
    print(f"{kind}: wall p50 {np.percentile(lat,50):.3f} ms, kernels {np.percentile(ker,50):.3f}, {len(r.rects)} rects | " + " ".join(f"{x['kind']}{x['lds_class']}:{y/40:.3f}" for x, y in zip(r.launches, lm)), flush=True)
