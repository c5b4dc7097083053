"""Ad-hoc performance exploration on the GPU box (not part of the product)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth, VJ_FLAG_COUNTERS

env = Environment(0)
c = Cascade.load("frontalface_alt")
B = int(os.environ.get("B", "16"))
frames = synth.batch(B, 1080, 1920, seed0=1)
t = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
scales = c.plan_scales(1920, 1080)

def run(p, reps=3):
    env.detect(c, df, p)
    best = None
    for _ in range(reps):
        r = env.detect(c, df, p)
        if best is None or r.cascade_ms < best.cascade_ms: best = r
    return best

what = sys.argv[1:] or ["occ", "scales", "split"]
if "tile" in what:
    for classes in ("36,64,140",):
        env.configure("tile_classes_kb", classes)
        for sp in ("5,10,16", "5,8,11,14,18", "4,6,8,10,13,17", "3,5,7,9,12,16", "5,7,9,12"):
            for te in (9, 14, 22):
                for ml in (4, 12, 32):
                    env.configure("tile_end", te); env.configure("pass_split", sp); env.configure("tile_min_lanes", ml)
                    r = run(default_params())
                    print(f"split={sp!r} tile_end={te} min_lanes={ml}: cascade {r.cascade_ms:.2f} ms passes {[round(x[2],2) for x in r.passes]}", flush=True)
    env.configure("tile_classes_kb", "-2,-1,0"); env.configure("pass_split", ""); env.configure("tile_end", 8); env.configure("tile_min_lanes", 0)
if "ab" in what:
    env.configure("tile_min_lanes", 4096)   # tile waves always leave at the first boundary
    for k in (0, 4, 8, 10, 12, 14, 16, 20):
        for sp in ("3", "5", "8"):
            env.configure("pass_split", sp)
            out = []
            for classes in ("0,0,0", "36,64,140"):
                env.configure("tile_classes_kb", classes)
                rc = run(default_params(flags=VJ_FLAG_COUNTERS, scales=[k]), 1)
                r = run(default_params(scales=[k]))
                ev = sum(rc.stage_entered[i] * int(c.stages["n_trees"][i]) for i in range(int(sp)))
                out.append(f"{'tile' if classes != '0,0,0' else 'glob'} {r.passes[0][2]:.3f} ms {ev / r.passes[0][2] / 1e6:.1f} Gev/s")
            print(f"scale {k} s={scales[k].scale:.2f} stages[0,{sp}) windows={rc.windows}: " + " | ".join(out), flush=True)
    env.configure("tile_classes_kb", "-2,-1,0"); env.configure("pass_split", ""); env.configure("tile_min_lanes", 0)
if "repack" in what:
    for rp in ("", "3,5", "2,3,4,5,6,7", "3,5,6,7"):
        env.configure("tile_repack", rp)
        for sp, te in (("5,8", 8), ("5,8,11", 11), ("5,8,11,14", 14), ("5,8,10,12", 12), ("3,5,8,11", 11), ("5,10", 10)):
            env.configure("pass_split", sp); env.configure("tile_end", te)
            r = run(default_params())
            print(f"repack={rp!r} split={sp!r} tile_end={te}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_repack", "3,5"); env.configure("pass_split", ""); env.configure("tile_end", 8)
if "conc" in what:
    for conc in (0, 1):
        env.configure("concurrent", conc)
        for sp, te in (("5,8", 8), ("5,8,12,16", 8), ("4,8", 8), ("5,10", 10), ("3,5,8", 8)):
            env.configure("pass_split", sp); env.configure("tile_end", te)
            r = run(default_params())
            print(f"concurrent={conc} split={sp!r} tile_end={te}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("concurrent", 0); env.configure("pass_split", ""); env.configure("tile_end", 8)
if "b64" in what:
    for sp, te in (("5,8", 8), ("5,8,12", 8), ("5,8,11,14,17", 8), ("5,8,10,12,14,17", 8), ("4,6,8,10,12,15,18", 8), ("5,10", 10), ("5,10,13,16", 10), ("5,9,12,15,18", 9), ("5,7,9,11,13,16", 7)):
        env.configure("pass_split", sp); env.configure("tile_end", te)
        r = run(default_params(), 2)
        print(f"split={sp!r} tile_end={te}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("pass_split", ""); env.configure("tile_end", 8)
if "sp" in what:
    rp_all = ",".join(str(i) for i in range(2, 22))
    for spb, spm, sp, te in ((4, 192, "5,8,13", 12), (4, 256, "5,8,13", 12), (3, 256, "5,8,13", 12), (3, 192, "5,8,13", 12), (2, 256, "5,8,13", 12), (5, 256, "5,8,13", 12),
                             (4, 128, "5,8,13", 12), (4, 256, "5,8,13", 22), (4, 256, "5,13", 13)):
        env.configure("tile_sp_begin", spb); env.configure("tile_sp_max", spm); env.configure("tile_repack", rp_all); env.configure("pass_split", sp); env.configure("tile_end", te)
        r = run(default_params(), 2)
        print(f"sp_begin={spb} sp_max={spm} split={sp!r} tile_end={te}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_sp_begin", 3); env.configure("tile_sp_max", 192); env.configure("tile_repack", rp_all); env.configure("pass_split", ""); env.configure("tile_end", 64)
if "large" in what:
    for acc, mdw in ((512, 600), (256, 600), (128, 600), (128, 1200), (64, 1200), (64, 2500), (32, 2500), (16, 5000)):
        env.configure("tile_accept_windows", acc); env.configure("tile_max_dwords_per_window", mdw)
        r = run(default_params(), 2)
        print(f"accept={acc} max_dw/win={mdw}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_accept_windows", 256); env.configure("tile_max_dwords_per_window", 600)
if "minw2" in what:
    for classes in ("-3,-2,-1", "-4,-2,-1", "-3,-2,0", "-4,-3,-2", "-2,-1,0"):
        env.configure("tile_classes_kb", classes)
        for minw, acc in ((1024, 256), (512, 256), (256, 256), (768, 256), (2048, 256), (512, 128)):
            env.configure("tile_min_windows", minw); env.configure("tile_accept_windows", acc)
            r = run(default_params(), 2)
            print(f"classes={classes} minw={minw} accept={acc}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_classes_kb", "-2,-1,0"); env.configure("tile_min_windows", 768); env.configure("tile_accept_windows", 256)
if "accept" in what:
    for classes in ("36,64,140", "36,64,100", "36,72,0", "40,80,0", "52,80,0", "52,80,140"):
        env.configure("tile_classes_kb", classes)
        for minw, acc in ((1024, 128), (1024, 512), (1024, 1024), (512, 512), (2048, 1024), (2048, 512)):
            env.configure("tile_min_windows", minw); env.configure("tile_accept_windows", acc)
            r = run(default_params())
            print(f"classes={classes} minw={minw} accept={acc}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_classes_kb", "-2,-1,0"); env.configure("tile_min_windows", 768); env.configure("tile_accept_windows", 256)
if "minw" in what:
    for minw in (256, 512, 1024, 2048):
        env.configure("tile_min_windows", minw)
        r = run(default_params())
        print(f"minw={minw}: cascade {r.cascade_ms:.2f} ms passes {[round(x[2],2) for x in r.passes]}", flush=True)
    env.configure("tile_min_windows", 768)
if "occ" in what:
    for b in (1, 2, 4, 8, 10):
        env.configure("blocks_per_cu", b)
        r = run(default_params())
        print(f"blocks_per_cu={b}: cascade {r.cascade_ms:.2f} ms passes {[round(x[2],2) for x in r.passes]} integral {r.integral_ms:.2f}", flush=True)
    env.configure("blocks_per_cu", 8)
if "scales" in what:
    for k in (0, 4, 8, 12, 16, 20, 24, 30):
        pc = default_params(flags=VJ_FLAG_COUNTERS, scales=[k])
        rc = run(pc, 1)
        r = run(default_params(scales=[k]))
        s = scales[k]
        print(f"scale {k} s={s.scale:.2f} step={s.step:.2f} windows={rc.windows} evals/win={rc.stump_evals/max(rc.windows,1):.1f} cascade {r.cascade_ms:.3f} ms -> {rc.windows/r.cascade_ms/1e6:.2f} Gwin/s, {rc.stump_evals/r.cascade_ms/1e6:.1f} Gevals/s passes {[round(x[2],3) for x in r.passes]}", flush=True)
if "split" in what:
    for sp in ("22", "3", "5", "4,9,15", "5,10,16", "2,5,10,16", "3,6,9,12,16", "1,2,3,5,8,12,17"):
        env.configure("pass_split", sp)
        r = run(default_params())
        print(f"split {sp}: cascade {r.cascade_ms:.2f} ms passes {[round(x[2],2) for x in r.passes]}", flush=True)
    env.configure("pass_split", "")
if "dense" in what:
    # cost of the dense part of the tile kernel: cut the first pass after k stages, no stump-parallel finish
    env.configure("tile_sp_begin", 99)
    for k in (1, 2, 3, 4, 5):
        env.configure("pass_split", str(k)); env.configure("tile_end", k)
        pc = default_params(flags=VJ_FLAG_COUNTERS)
        rc = run(pc, 1)
        ev = sum(rc.stage_entered[s] * c.stages[s]["n_trees"] for s in range(k))
        r = run(default_params())
        tl = [l for l in r.launches if l["kind"] == "tile"]
        print(f"k={k}: tile launches {[round(l['ms'],2) for l in tl]} all {[ (l['kind'], round(l['ms'],2)) for l in r.launches]} entered {rc.stage_entered[:k+1]} evals[0,k)={ev/1e9:.2f}G", flush=True)
    env.configure("tile_sp_begin", 3); env.configure("pass_split", ""); env.configure("tile_end", 64)
if "ws" in what:
    env.configure("tile_finish", 1)
    for begin in (2, 3, 4):
        for wsmax in (128, 256, 384, 512):
            env.configure("tile_sp_begin", begin); env.configure("tile_ws_max", wsmax)
            r = run(default_params())
            print(f"ws begin={begin} max={wsmax}: cascade {r.cascade_ms:.2f} ms launches {[(l['kind'], round(l['ms'],2)) for l in r.launches]}", flush=True)
    env.configure("tile_finish", 0); env.configure("tile_sp_begin", 3)
    r = run(default_params())
    print(f"sp: cascade {r.cascade_ms:.2f} ms launches {[(l['kind'], round(l['ms'],2)) for l in r.launches]}", flush=True)
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512)
if "wsmin" in what:
    env.configure("tile_finish", 1); env.configure("tile_sp_begin", 3); env.configure("tile_ws_max", 512)
    for wsmin in (0, 8, 16, 24, 32, 48, 64, 96, 128, 256):
        env.configure("tile_ws_min", wsmin)
        r = run(default_params())
        print(f"ws_min={wsmin}: cascade {r.cascade_ms:.2f} ms launches {[(l['kind'], round(l['ms'],2)) for l in r.launches]}", flush=True)
    env.configure("tile_ws_min", 32)
if "large2" in what:
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512); env.configure("tile_ws_min", 48)
    env.configure("tile_repack", ",".join(str(i) for i in range(1, 22)))
    for begin in (1, 2):
        env.configure("tile_sp_begin", begin)
        for acc, mdw in ((256, 600), (128, 1200), (64, 1200), (64, 2500), (64, 5000)):
            env.configure("tile_accept_windows", acc); env.configure("tile_max_dwords_per_window", mdw)
            r = run(default_params(), 2)
            print(f"begin={begin} accept={acc} max_dw/win={mdw}: cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("tile_accept_windows", 256); env.configure("tile_max_dwords_per_window", 600); env.configure("tile_sp_begin", 3)
    env.configure("tile_repack", ",".join(str(i) for i in range(2, 22)))
if "conc2" in what:
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512); env.configure("tile_ws_min", 48)
    def show(tag):
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{tag}: wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.detect(c, df, default_params())
    show("serial")
    for reserve in (0, 20, 36):
        env.configure("tile_lds_reserve_kb", reserve)
        env.configure("concurrent", 0); env.detect(c, df, default_params()); show(f"serial reserve={reserve}")
        for bpc in (1, 2):
            env.configure("concurrent", 1); env.configure("concurrent_blocks_per_cu", bpc)
            env.detect(c, df, default_params())
            show(f"concurrent reserve={reserve} bpc={bpc}")
    env.configure("concurrent", 0); env.configure("tile_lds_reserve_kb", 0)
if "conc3" in what:
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512); env.configure("tile_ws_min", 48)
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{l['ms']:.1f}" for l in r.launches), flush=True)
    for bpc in (8, 4, 2, 1):
        env.configure("blocks_per_cu", bpc); show(f"serial blocks_per_cu={bpc}")
    env.configure("blocks_per_cu", 8)
    for reserve in (0, 8, 20):
        env.configure("tile_lds_reserve_kb", reserve)
        for bpc in (1,):
            env.configure("concurrent", 1); env.configure("concurrent_blocks_per_cu", bpc)
            show(f"concurrent reserve={reserve} bpc={bpc}")
        env.configure("concurrent", 0)
    env.configure("tile_lds_reserve_kb", 0)
if "conc4" in what:
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512); env.configure("tile_ws_min", 48)
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("concurrent", 1); env.configure("concurrent_blocks_per_cu", 1)
    for reserve in (18, 20, 24):
        env.configure("tile_lds_reserve_kb", reserve)
        for mdw in (600, 400, 300, 200, 150):
            env.configure("tile_max_dwords_per_window", mdw)
            show(f"concurrent reserve={reserve} mdw={mdw}")
    env.configure("tile_max_dwords_per_window", 600)
    env.configure("concurrent", 0); env.configure("tile_lds_reserve_kb", 0)
if "conc5" in what:
    env.configure("tile_finish", 1); env.configure("tile_ws_max", 512); env.configure("tile_ws_min", 48)
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("concurrent", 1); env.configure("concurrent_blocks_per_cu", 1)
    env.configure("tile_lds_reserve_kb", 18)
    for acc, minw in ((256, 768), (512, 768), (768, 768), (1024, 1024), (1536, 1536), (2048, 2048)):
        env.configure("tile_accept_windows", acc); env.configure("tile_min_windows", minw)
        show(f"concurrent reserve=18 accept={acc} minw={minw}")
    env.configure("tile_accept_windows", 256); env.configure("tile_min_windows", 768)
    env.configure("tile_lds_reserve_kb", 36); env.configure("concurrent_blocks_per_cu", 2)
    show("concurrent reserve=36 bpc=2")
    env.configure("concurrent", 0); env.configure("tile_lds_reserve_kb", 0)
if "conc6" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    show("default")
    for sp in ("3,6,10", "4,8,13", "6,13", "5,13", "5,8,11,14", "4,6,8,13", "5,8"):
        env.configure("pass_split", sp); show(f"split {sp}")
    env.configure("pass_split", "")
    for acc in (900, 1024, 1280):
        env.configure("tile_accept_windows", acc); show(f"accept {acc}")
    env.configure("tile_accept_windows", 768)
    for te in (8, 13, 22):
        env.configure("tile_end", te); show(f"tile_end {te}")
    env.configure("tile_end", 64)
    for wsmin, begin in ((32, 3), (64, 3), (48, 2), (48, 4)):
        env.configure("tile_ws_min", wsmin); env.configure("tile_sp_begin", begin); show(f"ws_min {wsmin} begin {begin}")
    env.configure("tile_ws_min", 48); env.configure("tile_sp_begin", 3)
if "blocks" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    for conc in (0, 1):
        env.configure("concurrent", conc)
        for gb in (0, 1):
            env.configure("global_blocks", gb); show(f"concurrent={conc} global_blocks={gb}")
    env.configure("global_blocks", 1)
    for bpc in (2, 3):
        env.configure("concurrent_blocks_per_cu", bpc); show(f"concurrent blocks bpc={bpc}")
    env.configure("concurrent_blocks_per_cu", 1)
    for acc in (512, 1024):
        env.configure("tile_accept_windows", acc); show(f"accept {acc}")
    env.configure("tile_accept_windows", 768)
if "grid2d" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("global_blocks", 0); env.configure("tile_lds_reserve_kb", 18)
    for conc in (0, 1):
        env.configure("concurrent", conc)
        for bw in (0, 16, 32, 64, 128):
            env.configure("grid_block_w", bw); show(f"concurrent={conc} grid_block_w={bw}")
    env.configure("grid_block_w", 32); env.configure("global_blocks", 1); env.configure("tile_lds_reserve_kb", 26)
if "split2" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    env.configure("concurrent", 1)
    for gb, reserve in ((0, 18), (1, 26)):
        env.configure("global_blocks", gb); env.configure("tile_lds_reserve_kb", reserve)
        for acc in (768, 256):
            env.configure("tile_accept_windows", acc)
            for sp in (0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0):
                env.configure("tile_split", sp); show(f"blocks={gb} reserve={reserve} accept={acc} tile_split={sp}")
    env.configure("tile_split", 0); env.configure("tile_accept_windows", 768)
if "configs" in what:
    for name, cas, W, H, nb in (("config5 alt2 720p", "frontalface_alt2", 1280, 720, 64), ("config4 alt_tree 4096^2", "frontalface_alt_tree", 4096, 4096, 2),
                                ("default 1080p", "frontalface_default", 1920, 1080, 32), ("eye 1080p", "eye", 1920, 1080, 32)):
        cc = Cascade.load(cas)
        fr = synth.batch(nb, H, W, seed0=1)
        tt = torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
        dd = DeviceFrames.from_torch(tt)
        env.detect(cc, dd, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(cc, dd, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(cc, dd, default_params(flags=VJ_FLAG_COUNTERS))
        print(f"{name}: {nb} frames wall {wall:.2f} ms -> {rc.windows/wall/1e6:.2f} Gwin/s, {rc.stump_evals/max(rc.windows,1):.1f} evals/win, {rc.stump_evals/wall/1e6:.1f} Gevals/s, dets {len(r.rects)} " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
        del tt, dd
if "xcd" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        rc = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
        ok = np.array_equal(rc.rects, base.rects) and rc.stage_entered == base.stage_entered
        print(f"{tag}: same={ok} wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    for conc in (0, 1):
        env.configure("concurrent", conc)
        for xa in (0, 1):
            env.configure("xcd_affinity", xa); show(f"concurrent={conc} xcd_affinity={xa}")
    env.configure("xcd_affinity", 1)
if "split3" in what:
    base = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    def show(tag):
        env.detect(c, df, default_params())
        t0 = time.perf_counter(); n = 3
        for _ in range(n): r = env.detect(c, df, default_params())
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{tag}: wall {wall:.2f} ms cascade {r.cascade_ms:.2f} ms " + " ".join(f"{l['kind'][0]}{l['lds_class']}:{len(l['scales'])}sc:{l['ms']:.1f}" for l in r.launches), flush=True)
    for sp in (0.5, 0.75, 1.0, 1.25, 1.5):
        env.configure("tile_split", sp); show(f"tile_split={sp}")
    env.configure("tile_split", 0.5)
    for bw in (16, 24, 32, 48):
        env.configure("grid_block_w", bw); show(f"grid_block_w={bw}")
    env.configure("grid_block_w", 32)
