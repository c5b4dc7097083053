import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0)
for name, (H, W), nf in (("frontalface_alt", (1080, 1920), 1), ("frontalface_alt", (720, 1280), 1), ("frontalface_default", (480, 640), 1), ("frontalface_alt", (240, 320), 1),
                         ("frontalface_default", (1080, 1920), 1), ("eye", (720, 1280), 1), ("frontalface_alt", (1080, 1920), 3), ("frontalface_alt", (4096, 4096), 1)):
    c = Cascade.load(name)
    f = synth.batch(nf, H, W, seed0=3)
    out = []
    for v in ("4", "0", "4", "0"):
        env.configure("one_pass_max_frames", v)
        for _ in range(10): env.detect(c, f)
        ws, ks = [], []
        for _ in range(60):
            t = time.perf_counter(); r = env.detect(c, f); ws.append((time.perf_counter() - t) * 1e3); ks.append(r.total_ms)
        out.append(f"one_pass={v}: wall {np.percentile(ws, 50):.3f} kernels {np.percentile(ks, 50):.3f}")
    print(f"{name} {nf} x {W}x{H} | " + " | ".join(out), flush=True)
