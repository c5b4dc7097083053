"""A/B of one tunable on the bench workload (64 x 1080p, frontalface_alt): per-launch HIP-event times with the chains
overlapped and serialised.  Usage on the GPU box:  python tools/ab.py KEY VALUE_A VALUE_B [B frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
key, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
casc = sys.argv[5] if len(sys.argv) > 5 else "frontalface_alt"
env = Environment(0); c = Cascade.load(casc)
t = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
ref = None
for conc in ("1", "0"):
    env.configure("concurrent", conc)
    for rep in range(2):
        for v in (va, vb):
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
