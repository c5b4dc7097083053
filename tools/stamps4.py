"""Diagnostic build only (VJ_STAMPS=1 at build time, VJ_DEBUG_STAMPS=1 at run time): config 4's queue passes — time per
chunk (sum / max, s_memtime ticks of 10 ns) and the phases of the stump-parallel tail; the library prints them to stderr."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); c = Cascade.load(os.environ.get("CASCADE", "frontalface_alt_tree"))
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
S = int(os.environ.get("SIZE", "4096"))
t = torch.from_numpy(synth.batch(1, S, S, seed0=4001, kinds=("blocks",))).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
env.detect(c, df)
print("== timed", file=sys.stderr, flush=True)
r = env.detect(c, df)
print(" ".join(f"{x['kind']}{x['lds_class']}[{x['stage_begin']},{x['stage_end']}):{x['ms']:.2f}" for x in r.launches), file=sys.stderr)
