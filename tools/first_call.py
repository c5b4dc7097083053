"""Cost of the first call at a new frame size (plan build: scales, feature tables, tile layouts, uploads) vs a cached one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0)
for name in ("frontalface_alt", "frontalface_alt2", "frontalface_alt_tree"):
    c = Cascade.load(name)
    env.detect(c, synth.frame("noise", 1, 200, 300))                 # everything warm except the plans of the sizes below
    for h, w in ((1080, 1920), (720, 1280), (480, 640), (1081, 1921)):
        img = synth.frame("noise", 2, h, w)
        t0 = time.perf_counter(); env.detect(c, img); t1 = time.perf_counter(); env.detect(c, img); t2 = time.perf_counter()
        print(f"{name} {w}x{h}: first call {(t1 - t0) * 1e3:.1f} ms, second {(t2 - t1) * 1e3:.2f} ms", flush=True)
    t0 = time.perf_counter(); env.detect_opencv(c, img); t1 = time.perf_counter(); env.detect_opencv(c, img); t2 = time.perf_counter()
    print(f"{name} OpenCV profile {w}x{h}: first call {(t1 - t0) * 1e3:.1f} ms, second {(t2 - t1) * 1e3:.2f} ms", flush=True)
