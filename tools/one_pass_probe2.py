import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0)
c = Cascade.load("frontalface_alt")
for kind in ("noise", "smooth", "blocks", "faces"):
    for (H, W) in ((1080, 1920), (720, 1280)):
        f = synth.frame(kind, 5, H, W)
        out = []
        for v in ("4", "0", "4", "0"):
            env.configure("one_pass_max_frames", v)
            for _ in range(10): env.detect(c, f)
            ws, ks = [], []
            for _ in range(60):
                t = time.perf_counter(); r = env.detect(c, f); ws.append((time.perf_counter() - t) * 1e3); ks.append(r.total_ms)
            out.append(f"one_pass={v}: wall {np.percentile(ws, 50):.3f} kernels {np.percentile(ks, 50):.3f}")
        print(f"frontalface_alt {kind} {W}x{H} | " + " | ".join(out[:2]), flush=True)
