"""Integral kernels alone: event time per batch of 1080p frames (1 and 64 frames) and of one 4096 x 4096 frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clfacedetection_amd import Cascade, Environment, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
p = default_params(); p.min_w = p.min_h = 900          # two or three scales: the call is mostly the integral
for n, h, w in ((1, 1080, 1920), (64, 1080, 1920), (1, 4096, 4096)):
    frames = synth.batch(n, h, w, seed0=3)
    t = []
    for i in range(12):
        r = env.detect(c, frames, p)
        t.append(r.integral_ms)
    print(f"{n} x {w}x{h}: integral {np.median(t[2:]):.4f} ms", flush=True)
