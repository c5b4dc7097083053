import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
for h, w in ((240, 320), (480, 640), (720, 1280)):
    img = synth.frame("faces", 2, h, w)
    for _ in range(3): env.detect_opencv(c, img, min_neighbors=3)
    lat, ker = [], []
    for _ in range(40):
        t0 = time.perf_counter(); r = env.detect_opencv(c, img, min_neighbors=3); lat.append((time.perf_counter() - t0) * 1e3); ker.append(r.total_ms)
    print(f"OpenCV profile {w}x{h}: wall p50 {np.percentile(lat,50):.3f} ms, kernels {np.percentile(ker,50):.3f} ms, {len(r.rects)} faces", flush=True)
