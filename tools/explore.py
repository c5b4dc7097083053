"""Performance exploration on the GPU box (not part of the product): sweeps of the library's tunables on the bench
workload.  Every mode restores the defaults it touches.  The logs under profiles/ were produced by earlier
revisions of this script (modes of the same names); results never depend on a tunable — `same=` re-checks that
against the default configuration on every line.

    B=64 python tools/explore.py MODE [MODE ...]

modes: scales   per-scale cost of the default path
       dense    cost of the dense part of the tile kernel (first pass cut after k stages, no finish)
       finish   wave-split / stump-parallel finish thresholds (tile_sp_begin, tile_ws_max, tile_ws_min, tile_finish)
       classes  LDS classes and tile acceptance (tile_classes_kb, tile_min_windows, tile_accept_windows)
       overlap  the two chains: concurrent, tile_lds_reserve_kb, concurrent_blocks_per_cu, blocks_per_cu
       balance  work split between the chains: tile_split, tile_accept_windows, pass_split, tile_end
       gather   the global-gather chain: grid_block_w, xcd_affinity, global_blocks
       configs  throughput of the other BASELINE configurations (alt2 720p, alt_tree 4096^2, default, eye)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, VJ_FLAG_COUNTERS, default_params, synth

DEFAULTS = {"pass_split": "", "blocks_per_cu": 8, "tile_classes_kb": "-2,-1,0", "tile_min_windows": 768,
            "tile_accept_windows": 768, "tile_end": 64, "tile_sp_begin": 3, "tile_sp_max": 192, "tile_finish": 1,
            "tile_ws_max": 512, "tile_ws_min": 48, "concurrent": 1, "concurrent_blocks_per_cu": 1,
            "tile_lds_reserve_kb": 16, "tile_split": "0,0.5,0.5", "grid_block_w": 32, "xcd_affinity": 1, "global_blocks": 0}

env = Environment(0)
casc = Cascade.load("frontalface_alt")
B = int(os.environ.get("B", "16"))
frames = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda()
torch.cuda.synchronize()
df = DeviceFrames.from_torch(frames)
base = env.detect(casc, df, default_params(flags=VJ_FLAG_COUNTERS))


def configure(**kv):
    for k, v in kv.items():
        env.configure(k, v)


def restore(*keys):
    configure(**{k: DEFAULTS[k] for k in keys})


def show(tag, reps=3):
    """wall time of a step, per-launch HIP-event times, and a parity re-check against the default configuration"""
    env.detect(casc, df, default_params())
    t0 = time.perf_counter()
    for _ in range(reps):
        r = env.detect(casc, df, default_params())
    wall = (time.perf_counter() - t0) / reps * 1e3
    rc = env.detect(casc, df, default_params(flags=VJ_FLAG_COUNTERS))
    same = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
    launches = " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches)
    print(f"{tag}: same={same} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms {launches}", flush=True)


def mode_scales():
    plan = casc.plan_scales(1920, 1080)
    for k in range(0, len(plan), 3):
        rc = env.detect(casc, df, default_params(flags=VJ_FLAG_COUNTERS, scales=[k]))
        env.detect(casc, df, default_params(scales=[k]))
        r = env.detect(casc, df, default_params(scales=[k]))
        print(f"scale {k} s={plan[k].scale:.2f} step={plan[k].step:.2f} windows={rc.windows} "
              f"evals/win={rc.stump_evals / max(rc.windows, 1):.1f} cascade {r.cascade_ms:.3f} ms -> "
              f"{rc.stump_evals / r.cascade_ms / 1e6:.1f} Gevals/s " + " ".join(f"{l['kind'][0]}:{l['ms']:.2f}" for l in r.launches),
              flush=True)


def mode_dense():
    configure(tile_sp_begin=99, concurrent=0)
    for k in (1, 2, 3, 4, 5):
        configure(pass_split=str(k), tile_end=k)
        show(f"first pass [0,{k})", reps=2)
    restore("tile_sp_begin", "concurrent", "pass_split", "tile_end")


def mode_finish():
    for begin in (2, 3, 4):
        for ws_max in (256, 512):
            configure(tile_sp_begin=begin, tile_ws_max=ws_max)
            show(f"tile_sp_begin={begin} tile_ws_max={ws_max}")
    restore("tile_sp_begin", "tile_ws_max")
    for ws_min in (0, 16, 48, 128, 256):
        configure(tile_ws_min=ws_min)
        show(f"tile_ws_min={ws_min}")
    restore("tile_ws_min")
    configure(tile_finish=0)
    show("stump-parallel finish only")
    restore("tile_finish")


def mode_classes():
    for classes in ("-2,-1,0", "-3,-2,-1", "-3,-2,0", "-2,-1,-1"):
        for minw, acc in ((768, 768), (1024, 768), (512, 512), (768, 256)):
            configure(tile_classes_kb=classes, tile_min_windows=minw, tile_accept_windows=acc)
            show(f"classes={classes} min_windows={minw} accept={acc}", reps=2)
    restore("tile_classes_kb", "tile_min_windows", "tile_accept_windows")


def mode_overlap():
    for conc, reserve, bpc in ((0, 0, 1), (0, 18, 1), (1, 0, 1), (1, 8, 1), (1, 18, 1), (1, 18, 2), (1, 26, 1), (1, 36, 2)):
        configure(concurrent=conc, tile_lds_reserve_kb=reserve, concurrent_blocks_per_cu=bpc)
        show(f"concurrent={conc} reserve={reserve} KB gather workgroups/CU={bpc}")
    restore("concurrent", "tile_lds_reserve_kb", "concurrent_blocks_per_cu")
    configure(concurrent=0)
    for bpc in (8, 2, 1):
        configure(blocks_per_cu=bpc)
        show(f"serial blocks_per_cu={bpc}")
    restore("concurrent", "blocks_per_cu")


def mode_balance():
    for split in (0, 0.5, 1.0, 1.5):
        configure(tile_split=split)
        show(f"tile_split={split}")
    restore("tile_split")
    for acc in (256, 512, 768, 1024):
        configure(tile_accept_windows=acc)
        show(f"tile_accept_windows={acc}")
    restore("tile_accept_windows")
    for cuts in ("5,8", "5,8,13", "6,13", "4,8", "5"):
        configure(pass_split=cuts)
        show(f"pass_split={cuts}")
    restore("pass_split")
    for te in (8, 13, 64):
        configure(tile_end=te)
        show(f"tile_end={te}")
    restore("tile_end")


def mode_gather():
    for conc in (0, 1):
        configure(concurrent=conc)
        for bw in (0, 16, 32, 64):
            configure(grid_block_w=bw)
            show(f"concurrent={conc} grid_block_w={bw}")
        restore("grid_block_w")
        for xa in (0, 1):
            configure(xcd_affinity=xa)
            show(f"concurrent={conc} xcd_affinity={xa}")
        restore("xcd_affinity")
        for gb, reserve in ((1, 26),):
            configure(global_blocks=gb, tile_lds_reserve_kb=reserve)
            show(f"concurrent={conc} global_blocks={gb} reserve={reserve} KB")
        restore("global_blocks", "tile_lds_reserve_kb")
    restore("concurrent")


def mode_configs():
    for name, cas, w, h, nb in (("config 5: alt2 720p", "frontalface_alt2", 1280, 720, 64),
                                ("config 4: alt_tree 4096^2", "frontalface_alt_tree", 4096, 4096, 2),
                                ("default 1080p", "frontalface_default", 1920, 1080, 32), ("eye 1080p", "eye", 1920, 1080, 32)):
        c2 = Cascade.load(cas)
        t = torch.from_numpy(synth.batch(nb, h, w, seed0=1)).cuda()
        torch.cuda.synchronize()
        d2 = DeviceFrames.from_torch(t)
        env.detect(c2, d2, default_params())
        t0 = time.perf_counter()
        for _ in range(3):
            r = env.detect(c2, d2, default_params())
        wall = (time.perf_counter() - t0) / 3 * 1e3
        rc = env.detect(c2, d2, default_params(flags=VJ_FLAG_COUNTERS))
        print(f"{name}: {nb} frames wall {wall:.2f} ms -> {rc.windows / wall / 1e6:.2f} Gwin/s, "
              f"{rc.stump_evals / max(rc.windows, 1):.1f} evals/win, dets {len(r.rects)} " +
              " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
        del t, d2


if __name__ == "__main__":
    for m in sys.argv[1:] or ["overlap"]:
        globals()["mode_" + m]()
