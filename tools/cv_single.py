"""vj_detect_opencv on ONE 1080p frame per call (the reference's per-frame pattern, main.cpp:145): wall time vs kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0)
frames = synth.batch(6, 1080, 1920, seed0=1)
for name in ("frontalface_alt", "frontalface_alt2"):
    c = Cascade.load(name)
    for f in frames[:2]: env.detect_opencv(c, f)
    lat, ker = [], []
    for i in range(30):
        t0 = time.perf_counter(); r = env.detect_opencv(c, frames[i % 6]); lat.append((time.perf_counter() - t0) * 1e3); ker.append(r.total_ms)
    print(f"{name}: wall p50 {np.percentile(lat, 50):.2f} ms, kernels p50 {np.percentile(ker, 50):.2f} ms", flush=True)
