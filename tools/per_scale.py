"""Cascade time of every scale on its own (vj_params.scale_mask with one bit), chain balance off: what a scale costs per window on the
LDS-tile path and on the global-gather path — the basis of vj_shard_scales' cost weights.  64 x 1080p frontalface_alt and one
4096 x 4096 frame with frontalface_alt_tree.  python tools/per_scale.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, default_params, synth
env = Environment(0)
env.configure("tile_split", "0")
for name, n, H, W, kinds, seed in (("frontalface_alt", 64, 1080, 1920, ("noise", "smooth", "blocks"), 1), ("frontalface_alt_tree", 1, 4096, 4096, ("blocks",), 4001)):
    c = Cascade.load(name)
    df = DeviceFrames.from_torch(torch.from_numpy(synth.batch(n, H, W, seed0=seed, kinds=kinds)).cuda())
    plan = [s for s in c.plan_scales(W, H) if s.accepted]
    full = None
    for _ in range(3): full = env.detect(c, df)
    print(name, n, "frames: all scales cascade", round(full.cascade_ms, 3), "total", round(full.total_ms, 3), flush=True)
    for s in plan:
        p = default_params(scales=[s.scale_idx])
        best = 1e9
        for _ in range(4):
            r = env.detect(c, df, p)
            best = min(best, r.cascade_ms)
        nw = s.nx * s.ny
        print(f"  scale {s.scale_idx:2d} s={s.scale:.3f} windows {nw:8d} cascade {best:7.3f} ms  ns/window {best * 1e6 / (nw * n):7.3f}  kinds {[l['kind'] for l in r.launches]}", flush=True)
