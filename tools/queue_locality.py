"""Is the queue pass bound by the cache footprint of its gathers, or by the texture-address unit's lane rate?
(FETCH_SIZE 38-76 GB per launch for a 0.53 GB working set.)

A natural experiment that needs no new kernel: the same cascade on batches whose per-frame sum image is 8.3 MB (1080p: twice an
XCD's L2), 3.7 MB (720p: fits), 1.2 MB (480p) and 0.3 MB (240p) — the queue pass hands chunks out frame-major (q_slices), so an
XCD's waves work on about one frame at a time.  Per size and kernel: launch time (HIP events, chains serialised and
overlapped) over the launch's LANE-GATHERS (from the counted run's per-launch stage counters: 4 dwords per evaluated
rectangle) = ns per giga-gather, and the same in texture-address cycles per lane-gather at the clock the run holds.  Run under
rocprofv3 --pmc for FETCH_SIZE / TCC hit / TA busy per kernel (tools/queue_locality.sh).  If the time per lane-gather does not
move while the bytes fetched per gather fall by an order of magnitude, the pass is not bound by its cache footprint.

    python tools/queue_locality.py [sizes e.g. 1920x1080x64,1280x720x144] [modes: 0/1, 0 or 1] [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from clfacedetection_amd import VJ_FLAG_COUNTERS, Cascade, DeviceFrames, Environment, default_params, synth

args = [a for a in sys.argv[1:] if "=" not in a]
sizes = args[0] if args else "1920x1080x64,1280x720x144,640x480x432,320x240x1728"
modes = (args[1] if len(args) > 1 else "0/1").split("/")      # chains serialised (0) and / or overlapped (1)
env = Environment(0)
for kv in sys.argv[1:]:
    if "=" in kv:
        env.configure(*kv.split("=", 1))
casc = Cascade.load(os.environ.get("CASCADE", "frontalface_alt"))
nodes, trees, stages = casc.nodes, casc.trees, casc.stages
rects_per_stage = []
for st in stages:
    tr = trees[st["first_tree"]:st["first_tree"] + st["n_trees"]]
    rects_per_stage.append(int(sum(int(nodes["n_rects"][t["first_node"]:t["first_node"] + t["n_nodes"]].sum()) for t in tr)))

for spec in sizes.split(","):
    W, H, B = (int(v) for v in spec.split("x"))
    t = torch.from_numpy(synth.batch(B, H, W, seed0=1)).cuda()
    torch.cuda.synchronize()
    df = DeviceFrames.from_torch(t)
    # a fixed balance for every size, so that the queue pass sees the same kind of work; no feedback search
    env.configure("auto_balance", "0")
    counted = env.detect(casc, df, default_params(flags=VJ_FLAG_COUNTERS))
    gathers = {}      # (kind, LDS class, first stage) -> lane-gathers of that launch: the two modes launch in different orders
    for l in counted.launches:
        g = 4 * sum(n * rects_per_stage[s] for s, n in enumerate(l["stage_entered"]))
        if l["kind"] != "queue":
            g += 8 * l["stage_entered"][0]      # the variance's eight corners (four of them 8-byte loads)
        gathers[(l["kind"], l["lds_class"], l["stage_begin"])] = g
    for conc in modes:
        env.configure("concurrent", conc)
        for _ in range(3):
            env.detect(casc, df)
        ms = None
        n_rep = 5
        for _ in range(n_rep):
            r = env.detect(casc, df)
            lm = [x["ms"] for x in r.launches]
            ms = lm if ms is None else [a + b for a, b in zip(ms, lm)]
        out = []
        for i, l in enumerate(r.launches):
            m = ms[i] / n_rep
            g = gathers[(l["kind"], l["lds_class"], l["stage_begin"])]
            out.append(f"{l['kind']}{l['lds_class']}[{l['stage_begin']},{l['stage_end']}) {m:.3f} ms, {g / 1e9:.3f} G lane-gathers, "
                       f"{m * 1e6 / max(g, 1) * 256:.3f} CU-ns per lane-gather")
        print(f"{W}x{H} x {B} frames ({(W + 1) * (H + 3) * 4 / 1e6:.2f} MB sum image per frame), chains "
              f"{'overlapped' if conc == '1' else 'serialised'}: total {r.total_ms:.2f} ms | " + " | ".join(out), flush=True)
    env.configure("concurrent", "1")
    del df, t
