"""Integral kernels' time (HIP events of vj_detect's integral launches): 64 x 1080p batch and one frame per call.
     python tools/integral_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, default_params, synth
env = Environment(0)
for kv in sys.argv[1:]:          # key=value settings of vj_env_configure, e.g. integral_rows=1
    if "=" in kv:
        env.configure(*kv.split("=", 1))
c = Cascade.load("frontalface_alt")
p = default_params(min_w=900, min_h=900)      # (almost no cascade work: the call is the integral)
for B, (H, W) in ((64, (1080, 1920)), (1, (1080, 1920)), (1, (4096, 4096)), (256, (720, 1280)), (8, (1080, 1920)), (1, (480, 640))):
    t = torch.from_numpy(synth.batch(B, H, W, seed0=1)).cuda(); torch.cuda.synchronize()
    df = DeviceFrames.from_torch(t)
    for _ in range(5): env.detect(c, df, p)
    ms = sorted(env.detect(c, df, p).integral_ms for _ in range(21))
    px = B * H * W
    print(" ".join(a for a in sys.argv[1:] if "=" in a), f"{B} x {W}x{H}: integral p50 {ms[10]:.4f} ms min {ms[0]:.4f} ms = {13 * px / (ms[10] * 1e-3) / 1e12:.2f} TB/s algorithmic (13 B per pixel) = {13 * px / (ms[10] * 1e-3) / 8e12:.3f} of the HBM peak", flush=True)
    del df, t
