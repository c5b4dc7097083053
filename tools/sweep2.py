"""Single 1080p frame (config 2): queue-pass variants."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
frames = synth.batch(6, 1080, 1920, seed0=1)
for cfg in sys.argv[1:]:
    for kv in cfg.split(";"):
        env.configure(*kv.split("=", 1))
    for f in frames[:2]: env.detect(c, f)
    lat = []; tot = []; lm = None
    for i in range(40):
        t0 = time.perf_counter(); r = env.detect(c, frames[i % 6]); lat.append((time.perf_counter() - t0) * 1e3)
        tot.append(r.total_ms)
        l = [x["ms"] for x in r.launches]
        lm = l if lm is None else [a + b for a, b in zip(lm, l)]
    print(f"{cfg}: p50 {np.percentile(lat,50):.3f} ms kernels {np.percentile(tot,50):.3f} | " +
          " ".join(f"{x['kind']}{x['lds_class']}:{y/40:.3f}" for x, y in zip(r.launches, lm)), flush=True)
