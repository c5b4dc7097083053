"""Per-launch times of BASELINE config 4 (4096 x 4096, frontalface_alt_tree) for 1 and 2 frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt_tree")
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
for B in (1, 2):
    t = torch.from_numpy(synth.batch(B, 4096, 4096, seed0=4001, kinds=("blocks",))).cuda(); torch.cuda.synchronize()
    df = DeviceFrames.from_torch(t)
    for _ in range(2): env.detect(c, df)
    lat = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = env.detect(c, df); lat.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B}: p50 {np.percentile(lat,50):.2f} ms kernels {r.total_ms:.2f} integral {r.integral_ms:.2f} | " +
          " ".join(f"{x['kind']}{x['lds_class']}[{x['stage_begin']},{x['stage_end']}):{x['ms']:.2f}" for x in r.launches), flush=True)
