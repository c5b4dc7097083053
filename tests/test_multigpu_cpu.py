"""The N>1 path on CPU: world_size-2 gloo processes shard a batch (by frames) and a
single frame (by scales), all-gather the detections and must reproduce the
single-process result.  The per-rank detector here is the oracle (tests may use it);
on the GPU box bench.py runs the same sharding code around the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from clfacedetection_amd import multigpu


def test_shard_frames_covers_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in multigpu.shard_frames(n, r, world)]
            assert got == list(range(n))
            sizes = [len(multigpu.shard_frames(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_scales_lpt_balance(oracle, cascades):
    _, a = cascades("frontalface_alt")
    counts = [s.nx * s.ny if s.accepted else 0 for s in oracle.plan_scales(a, 1920, 1080)]
    for world in (2, 4, 8):
        parts = [multigpu.shard_scales(counts, r, world) for r in range(world)]
        assert sorted(k for p in parts for k in p) == list(range(len(counts)))
        loads = [sum(counts[k] for k in p) for p in parts]
        assert max(loads) <= sum(counts) / world + max(counts)        # LPT bound
    assert multigpu.plan(64, counts, 3, 8) == (list(range(24, 32)), None)
    frames, scales = multigpu.plan(1, counts, 1, 2)
    assert frames == [0] and scales is not None


def test_native_shard_helpers_agree_with_the_python_ones(lib, cascades):
    """vj_shard_frames / vj_shard_scales (what a C++ host uses) give the partitions multigpu.shard_frames / shard_scales
    give: the two N > 1 drivers — bench.py's and examples/multi_gpu — split the work identically."""
    import ctypes as C
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            for r in range(world):
                first, count = C.c_int(), C.c_int()
                assert lib.vj_shard_frames(n, world, r, C.byref(first), C.byref(count)) == 0
                assert list(range(first.value, first.value + count.value)) == list(multigpu.shard_frames(n, r, world))
    assert lib.vj_shard_frames(4, 2, 2, C.byref(first), C.byref(count)) == 1
    for name, (w, h) in (("frontalface_alt", (1920, 1080)), ("frontalface_alt_tree", (4096, 4096)), ("eye", (300, 200))):
        c, _ = cascades(name)
        counts = [s.nx * s.ny if s.accepted else 0 for s in c.plan_scales(w, h)]
        sides = [max(s.win_w, s.win_h) for s in c.plan_scales(w, h)]
        base = max(c.info.win_w, c.info.win_h)
        for world in (1, 2, 4, 8):
            parts = [c.shard_scales(w, h, r, world) for r in range(world)]
            assert parts == [multigpu.shard_scales(counts, r, world, sides, base) for r in range(world)]
            assert sorted(k for p in parts for k in p) == list(range(len(counts)))
            if world > 1 and name != "eye":      # every rank gets LDS-tile scales AND global-gather scales: both of its chains have work
                for p in parts:
                    assert any(sides[k] <= 72 for k in p) and any(sides[k] > 72 for k in p), (name, world, p)


def test_a_rank_without_a_share_gets_the_explicit_empty_mask(lib, cascades):
    """More ranks than scales: an all-zero vj_params.scale_mask means EVERY scale, so a rank that receives nothing must
    not be handed one — vj_shard_scales returns VJ_SCALE_MASK_NONE (bit 127), which vj_detect answers with no rectangles,
    and default_params(scales=[]) builds the same mask."""
    import ctypes as C
    from clfacedetection_amd import default_params
    c, _ = cascades("frontalface_alt")
    n = len(c.plan_scales(64, 48))
    assert 0 < n < 12
    world = 16
    shares = []
    for r in range(world):
        m = (C.c_uint64 * 2)()
        assert lib.vj_shard_scales(c._h, 64, 48, C.byref(default_params()), world, r, m) == 0
        assert (m[0] | m[1]) != 0                                   # never "every scale"
        shares.append(None if (m[0], m[1]) == (0, 1 << 63) else [k for k in range(127) if (m[k >> 6] >> (k & 63)) & 1])
    assert shares.count(None) == world - n and sorted(k for s in shares if s for k in s) == list(range(n))
    assert c.shard_scales(64, 48, world - 1, world) == []
    p = default_params(scales=[])
    assert (p.scale_mask[0], p.scale_mask[1]) == (0, 1 << 63)
    with pytest.raises(ValueError):
        default_params(scales=[127])


def _worker(rank, world, port, mode, q):
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from clfacedetection_amd import synth
    from clfacedetection_amd.api import DATA_DIR, RECT_DTYPE
    from oracle.oracle import Oracle, load_vjc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    a = load_vjc(os.path.join(DATA_DIR, "haarcascade_frontalface_alt.vjc"))
    H, W = 200, 260
    n_frames = 5 if mode == "frames" else 1
    frames = synth.batch(n_frames, H, W, seed0=500, kinds=("noise",))
    counts = [s.nx * s.ny if s.accepted else 0 for s in o.plan_scales(a, W, H)]
    my_frames, my_scales = multigpu.plan(n_frames, counts, rank, world)
    rows = []
    for f in my_frames:
        r, _ = o.detect(a, frames[f])
        for d in r:
            if my_scales is None or int(d["scale_idx"]) in my_scales:
                rows.append((d["x"], d["y"], d["w"], d["h"], float(3 + (int(d["x"]) + int(d["y"])) % 5), f, d["scale_idx"]))   # a neighbour count
    mine = np.array(rows, RECT_DTYPE) if rows else np.zeros(0, RECT_DTYPE)
    # the single-collective form (fixed-capacity [count | rects x cap] blocks): a capacity of 4 rows overflows
    # on the first step — every rank regrows the same way and repeats the collective —, the second step of the same
    # workload then costs exactly one collective
    g = multigpu.RectGather(cap=4)
    allr = g(mine)
    first_step = g.n_collectives
    again = g(mine)
    assert np.array_equal(allr, again) and g.n_collectives == first_step + 1 and g.cap >= len(mine)
    assert np.array_equal(multigpu.allgather_rects(mine), allr)      # the module-level wrapper: same list
    q.put((rank, allr.tolist(), len(mine), first_step))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["frames", "scales"])
def test_two_ranks_reproduce_single_process(oracle, cascades, mode):
    from clfacedetection_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, a = cascades("frontalface_alt")
    n_frames = 5 if mode == "frames" else 1
    frames = synth.batch(n_frames, 200, 260, seed0=500, kinds=("noise",))
    want = []
    for f in range(n_frames):
        r, _ = oracle.detect(a, frames[f])
        want += [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), float(3 + (int(d["x"]) + int(d["y"])) % 5), f, int(d["scale_idx"])) for d in r]
    assert len(want) > 0
    for rank, allr, n_mine, first_step in got:
        assert [(x, y, w, h, wt, fr, sc) for (x, y, w, h, wt, fr, sc) in allr] == want      # weights (neighbour counts) travel too
        assert first_step == (2 if max(g[2] for g in got) > 4 else 1)      # one regrow at most, decided identically on every rank
    assert sum(g[2] for g in got) == len(want)      # shards are disjoint
