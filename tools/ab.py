"""A/B of one tunable on the bench workload (64 x 1080p, frontalface_alt): per-launch HIP-event times with the chains
overlapped and serialised.  Usage on the GPU box:  python tools/ab.py KEY V1/V2/... [frames] [cascade] [concurrent modes, e.g. 1 or 1/0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
key, vals = sys.argv[1], sys.argv[2].split("/")   # e.g. tile_split 0.5/1/1.5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
casc = sys.argv[4] if len(sys.argv) > 4 else "frontalface_alt"
modes = sys.argv[5].split("/") if len(sys.argv) > 5 else ["1", "0"]
env = Environment(0); c = Cascade.load(casc)
H, W = int(os.environ.get("H", "1080")), int(os.environ.get("W", "1920"))      # frame size: H=480 W=640 python tools/ab.py ...
t = torch.from_numpy(synth.batch(B, H, W, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
ref = None
for conc in modes:
    env.configure("concurrent", conc)
    for rep in range(2):
        for v in vals:
            if key == "set":   # v = "k1=v1;k2=v2"
                for kv in v.split(";"):
                    env.configure(*kv.split("=", 1))
            else:
                env.configure(key, v)
            for _ in range(2): env.detect(c, df)
            ms = []; lm = None
            for _ in range(5):
                r = env.detect(c, df)
                ms.append(r.total_ms)
                l = [x["ms"] for x in r.launches]
                lm = l if lm is None else [a + b for a, b in zip(lm, l)]
            if ref is None: ref = r.rects
            same = bool((r.rects == ref).all()) if len(r.rects) == len(ref) else False
            print(f"concurrent={conc} {key}={v}: total {sum(ms)/len(ms):.2f} ms  launches " +
                  " ".join(f"{x['kind']}{x['lds_class']}:{y/5:.2f}" for x, y in zip(r.launches, lm)) + f" same={same}", flush=True)
