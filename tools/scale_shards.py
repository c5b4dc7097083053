"""BASELINE config 3 with the PYRAMID LEVELS sharded instead of the frames: what each rank of a 2 / 4 / 8-GPU job would run, measured
one share after the other on the one GPU (every rank integrates all 64 frames and evaluates its vj_shard_scales share of the scales).
Prints per-rank ms, the job's step (the slowest rank) and the same for whole-frame shards.  python tools/scale_shards.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, default_params, multigpu, synth

what = sys.argv[1] if len(sys.argv) > 1 else "config3"      # config3: 64 x 1080p frontalface_alt; config4: one 4096 x 4096 frame, frontalface_alt_tree
env = Environment(0)
if what == "config4":
    n, H, W = 1, 4096, 4096
    c = Cascade.load("frontalface_alt_tree")
    frames = torch.from_numpy(synth.batch(1, H, W, seed0=4001, kinds=("blocks",))).cuda()
else:
    n, H, W = 64, 1080, 1920
    c = Cascade.load("frontalface_alt")
    frames = torch.from_numpy(synth.batch(n, H, W, seed0=1, kinds=("noise", "smooth", "blocks"))).cuda()
plan = c.plan_scales(W, H)
counts = {int(s.scale_idx): int(s.nx) * int(s.ny) for s in plan if s.accepted}


def timed(df, p, reps=6):
    for _ in range(16 if n >= 8 else 2):     # batches: the chain balance of a new workload is found by feedback over its first calls
        env.detect(c, df, p)
    t = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.detect(c, df, p)
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


whole = DeviceFrames.from_torch(frames)
t1 = timed(whole, default_params())
print(f"1 rank: {t1:.2f} ms per step ({n} frames, all {len(counts)} scales)", flush=True)
for world in (2, 4, 8):
    mine = [c.shard_scales(W, H, r, world) for r in range(world)]          # vj_shard_scales: LPT by window count
    by_scale = [timed(whole, default_params(scales=m)) for m in mine]
    shares = [sum(counts[k] for k in m) / sum(counts.values()) for m in mine]
    by_frame = []
    for r in range(world if n >= world else 0):
        fr = multigpu.shard_frames(n, r, world)
        by_frame.append(timed(DeviceFrames.from_torch(frames[fr.start:fr.stop]), default_params()))
    if not by_frame:
        print(f"{world} ranks, scales sharded: per rank {[round(x, 2) for x in by_scale]} ms (window shares {[round(s, 3) for s in shares]}) -> step {max(by_scale):.2f} ms, "
              f"efficiency {t1 / world / max(by_scale):.2f}", flush=True)
        continue
    print(f"{world} ranks, scales sharded: per rank {[round(x, 2) for x in by_scale]} ms (window shares {[round(s, 3) for s in shares]}) -> step {max(by_scale):.2f} ms, "
          f"efficiency {t1 / world / max(by_scale):.2f} | frames sharded: per rank {[round(x, 2) for x in by_frame]} -> step {max(by_frame):.2f} ms, "
          f"efficiency {t1 / world / max(by_frame):.2f}", flush=True)
