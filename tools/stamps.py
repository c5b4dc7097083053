"""Diagnostic build only (VJ_STAMPS=1 at build time, VJ_DEBUG_STAMPS=1 at run time): per-phase s_memtime sums of
the tile kernel (thread 0 of each workgroup); the library prints slots 40..59 to stderr after every detect.
  40 wait at loop top | 41 staging + sqsum loads + barrier | 42 variance fill | 43+min(st,8) stage st-1 + re-pack
  52 stump-parallel finish | 53 spill | 54..58 inside the finish: table copy, verdicts, barrier, decision, compaction
Usage on the GPU box:  VJ_STAMPS=1 python -c "from clfacedetection_amd.build import build_lib; build_lib(force=True)"
                       VJ_DEBUG_STAMPS=1 B=64 python tools/stamps.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
B = int(os.environ.get("B", "16"))
t = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
env.configure("tile_repack", ",".join(str(i) for i in range(1, 22)))   # a stamp after every stage
for name, sc in (("class0", list(range(0, 10))), ("class1", list(range(10, 18))), ("scale0", [0]), ("scale9", [9]), ("scale17", [17])):
    p = default_params(scales=sc)
    env.detect(c, df, p)
    print(f"== {name}", file=sys.stderr, flush=True)
    r = env.detect(c, df, p)
    print(f"== {name}: cascade {r.cascade_ms:.2f} ms launches {[(l['kind'], round(l['ms'], 2)) for l in r.launches]}", file=sys.stderr, flush=True)
