"""Per-launch times of BASELINE config 5's first leg (256 x 720p, frontalface_alt2) and of the chain."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); face = Cascade.load("frontalface_alt2"); eye = Cascade.load("eye")
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
B = 256
t = torch.from_numpy(synth.batch(B, 720, 1280, seed0=5001, kinds=("faces", "noise", "smooth", "blocks"))).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
for _ in range(2): env.detect(face, df)
r = env.detect(face, df)
print(f"faces: kernels {r.total_ms:.2f} integral {r.integral_ms:.2f} | " + " ".join(f"{x['kind']}{x['lds_class']}[{x['stage_begin']},{x['stage_end']}):{x['ms']:.2f}" for x in r.launches), flush=True)
env.detect_chain(face, eye, df)
t0 = time.perf_counter(); r1, r2 = env.detect_chain(face, eye, df); dt = (time.perf_counter() - t0) * 1e3
print(f"chain: wall {dt:.2f} ms, first kernels {r1.total_ms:.2f}, second {r2.cascade_ms:.2f} ms on {len(r1.rects)} regions -> {len(r2.rects)}")
p3 = default_params(min_neighbors=3)
env.detect_chain(face, eye, df, p3)
t0 = time.perf_counter(); r1, r2 = env.detect_chain(face, eye, df, p3); dt = (time.perf_counter() - t0) * 1e3
print(f"chain, grouped on the device: wall {dt:.2f} ms, first kernels {r1.total_ms:.2f}, grouping + second {r2.cascade_ms:.2f} ms on {len(r1.rects)} faces -> {len(r2.rects)}")
