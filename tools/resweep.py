"""One-at-a-time re-sweep of the speed tunables on the bench workload (64 x 1080p, frontalface_alt) around the current defaults:
median kernel time of 7 calls per value, chains overlapped.  python tools/resweep.py [frames]   (results never depend on these keys)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
env = Environment(0)
env.configure("auto_balance", "0")
c = Cascade.load("frontalface_alt")
df = DeviceFrames.from_torch(torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda())
SWEEP = [("pass_cut_nodes", ["35", "10", "50", "75", "150"]), ("blocks_per_cu", ["8", "4", "6", "10", "12"]), ("concurrent_blocks_per_cu", None),
         ("min_chunk", ["32", "16", "48", "64"]), ("q_slices", ["-1", "4", "16", "32"]), ("gather_pairs", ["-1", "0", "2"]),
         ("sp_tail_max", ["48", "32", "64"]), ("wide_tail", ["-1", "0", "1"]), ("thin_pass_spread", ["1", "0"]), ("grid_block_w", ["32", "16", "64"]),
         ("tile_ws_max", ["512", "384", "768"]), ("tile_ws_min", ["48", "32", "64"]), ("tile_sp_begin", ["3", "4", "5"]), ("tile_sp_max", ["192", "128", "256"]),
         ("tile_lds_reserve_kb", ["16", "12", "20", "24"]), ("tile_class_order", ["1", "0"]), ("tile_repack", None)]


def timed():
    for _ in range(2):
        env.detect(c, df)
    return float(np.median([env.detect(c, df).total_ms for _ in range(7)]))


ref = env.detect(c, df).rects
print(f"defaults: {timed():.2f} ms", flush=True)
for key, vals in SWEEP:
    if vals is None:
        continue
    out = []
    for v in vals + [vals[0]]:          # the first value is the default; it is set again at the end
        env.configure(key, v)
        t = timed()
        out.append(f"{v}: {t:.2f}")
    assert np.array_equal(env.detect(c, df).rects, ref)
    print(f"{key:26s} " + "  ".join(out), flush=True)
