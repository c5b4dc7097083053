"""Config 4 (4096 x 4096, frontalface_alt_tree): windows entering every sweep position, per launch, and stage sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth, VJ_FLAG_COUNTERS
env = Environment(0); c = Cascade.load("frontalface_alt_tree")
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
t = torch.from_numpy(synth.batch(1, 4096, 4096, seed0=4001, kinds=("blocks",))).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
r = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
st = c.stages
nn = [int(s["n_trees"]) if "n_trees" in st.dtype.names else 0 for s in st]
print("stage sizes", nn)
print("parent", [int(s["parent"]) for s in st]); print("next", [int(s["next"]) for s in st]); print("child", [int(s["child"]) for s in st])
print("total entered per stage", r.stage_entered)
for x in r.launches:
    se = x["stage_entered"]
    print(x["kind"], x["lds_class"], (x["stage_begin"], x["stage_end"]), f"{x['ms']:.2f} ms", [int(v) for v in se[:len(nn)]])
