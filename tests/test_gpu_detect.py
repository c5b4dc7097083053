"""Parity of the HIP detect path (through the C ABI, vj_detect) with the CPU oracle
and the committed fixtures: raw detections, per-stage survivor counts, stump
evaluations and algorithmic gather bytes — all bit-exact / integer-exact."""
import json
import os

import numpy as np
import pytest

from cases import DETECT_CASES, HEADLINE_CASE, make_frame, sha
from clfacedetection_amd import (VJ_FLAG_COUNTERS, VJ_FLAG_SIGNED_MEAN, DeviceFrames, clodDetectObjects,
                                 default_params, synth)

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DET = {d["id"]: d for d in json.load(open(os.path.join(G, "detect.json")))}
LINEAR = [c for c in DETECT_CASES if c[1] not in ("frontalface_alt_tree",)]


def as_list(rects, with_frame=False):
    if with_frame:   # oracle records carry no frame: 0
        return [(int(r["frame"]) if "frame" in r.dtype.names else 0,) + tuple(int(r[k]) for k in ("scale_idx", "x", "y", "w", "h"))
                for r in rects]
    return [[int(r[k]) for k in ("scale_idx", "x", "y", "w", "h")] for r in rects]


def run(env, c, img, mn=(0, 0), mx=(0, 0), flags=VJ_FLAG_COUNTERS):
    return clodDetectObjects(img, c, env, mn, mx, 0, 0, True, vj_flags=flags)


@pytest.mark.parametrize("case", DETECT_CASES + [HEADLINE_CASE], ids=lambda c: c[0])
def test_detect_matches_fixture_and_oracle(env, oracle, cascades, case):
    cid, casc, gen, seed, h, w, mn, mx, sm = case
    c, a = cascades(casc)
    img = make_frame(gen, seed, h, w, oracle)
    g = DET[cid]
    assert sha(img) == g["image_sha256"]
    r = run(env, c, img, mn, mx, VJ_FLAG_COUNTERS | (VJ_FLAG_SIGNED_MEAN if sm else 0))
    assert as_list(r.rects) == g["rects"]
    assert r.windows == g["windows"] and r.stage_entered == g["stage_entered"]
    if c.info.is_stump_based:
        assert r.stump_evals == g["stump_evals"] and r.gather_bytes == g["gather_bytes"]
    if h <= 480:   # live oracle too (the 1080p case is covered by its fixture = the survey pin)
        ro, st = oracle.detect(a, img, min_size=mn, max_size=mx, signed_mean=sm)
        assert as_list(r.rects) == as_list(ro) and r.stage_entered == st["stage_entered"]
    assert (r.rects["frame"] == 0).all() and (r.rects["weight"] == 0).all()


def test_survey_pin_on_gpu(env, oracle, cascades):
    """The reference-run figures the survey recorded, reproduced by the HIP path."""
    c, _ = cascades("frontalface_alt")
    r = run(env, c, make_frame("xorshift", 12345, 1080, 1920, oracle))
    assert r.stage_entered == [6290352, 4205943, 2030967, 1412523, 643100, 405745, 235217, 206382, 146688, 70924,
                               46128, 22304, 9779, 5930, 3700, 1907, 1036, 561, 319, 175, 90, 50]
    assert r.stump_evals == 267307785 and len(r.rects) == 35 and r.windows == 6290352


@pytest.mark.parametrize("casc,kind,h,w", [("frontalface_alt", "noise", 1080, 1920), ("frontalface_alt", "smooth", 720, 1280),
                                           ("frontalface_default", "blocks", 600, 800), ("eye", "noise", 480, 640),
                                           ("frontalface_alt2", "noise", 720, 1280), ("frontalface_alt2", "blocks", 540, 960)])
def test_detect_matches_live_oracle_large(env, oracle, cascades, casc, kind, h, w):
    c, a = cascades(casc)
    img = synth.frame(kind, 900 + h, h, w)
    r = run(env, c, img)
    ro, st = oracle.detect(a, img)
    assert as_list(r.rects) == as_list(ro)
    assert r.stage_entered == st["stage_entered"]
    # node evaluations and algorithmic bytes as SURVEY.md §8d defines them — the nodes a window's walk VISITS — for stumps and
    # for multi-node trees alike (the counted kernels count the visited nodes below the roots; the roots are every entering window's)
    assert r.stump_evals == st["stump_evals"]
    assert r.gather_bytes == st["gather_bytes"] == 48 * st["windows"] + 16 * st["rect_evals"]
    if casc == "frontalface_alt2":
        all_nodes = sum(n * t["n_nodes"] for n, sg in zip(r.stage_entered, c.stages)
                        for t in c.trees[sg["first_tree"]:sg["first_tree"] + sg["n_trees"]])
        assert sum(r.stage_entered) < r.stump_evals < all_nodes          # more than the roots, fewer than every node


def test_batch_equals_single_frames(env, oracle, cascades):
    """Frames of a batch are independent: batch result == per-frame results, frame index kept — for batch sizes
    below, at and above the number of queue parts (frames are grouped into 8 parts by index)."""
    c, a = cascades("frontalface_alt")
    frames = synth.batch(19, 270, 360, seed0=300)
    per_frame = [oracle.detect(a, f) for f in frames]
    for n in (1, 7, 8, 9, 19):
        rb = env.detect(c, frames[:n], default_params(flags=VJ_FLAG_COUNTERS))
        total = [0] * c.info.n_stages
        for f in range(n):
            ro, st = per_frame[f]
            assert as_list(rb.rects[rb.rects["frame"] == f]) == as_list(ro), (n, f)
            total = [x + y for x, y in zip(total, st["stage_entered"])]
        assert rb.stage_entered == total, n
    assert np.all(np.diff(rb.rects["frame"]) >= 0)


def test_device_resident_frames(env, cascades):
    """Frames already in HBM (a torch CUDA tensor) give the same result as host frames."""
    import torch
    c, _ = cascades("frontalface_alt")
    frames = synth.batch(3, 300, 400, seed0=40)
    t = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    rd = env.detect(c, DeviceFrames.from_torch(t), default_params(flags=VJ_FLAG_COUNTERS))
    rh = env.detect(c, frames, default_params(flags=VJ_FLAG_COUNTERS))
    assert np.array_equal(rd.rects, rh.rects) and rd.stage_entered == rh.stage_entered


def test_band_major_queue_pass_equals_the_chunked_one(env, oracle, cascades):
    """The band-major queue pass (first-pass units ordered by image band, the queue pass drawing groups of units through the run
    table: vj_kernels.hip, CascadeArgs::run_table) against the chunk-by-chunk one and the oracle: rectangles, per-stage
    counts, and per-launch counters; band heights / group sizes that leave one unit per group or one group per scale; frame
    counts that do not divide by the eight queue parts; a batch below the switch-over keeps the chunked pass."""
    c, a = cascades("frontalface_alt")
    p = default_params(flags=VJ_FLAG_COUNTERS)
    try:
        for n_frames, (h, w) in ((11, (300, 420)), (8, (480, 640)), (3, (270, 500))):
            frames = synth.batch(n_frames, h, w, seed0=333, kinds=("noise", "faces", "blocks"))
            env.configure("q_band_px", 0)
            base = env.detect(c, frames, p)
            want = []
            for f in range(n_frames):
                ro, _ = oracle.detect(a, frames[f])
                want += [(f,) + t[1:] for t in as_list(ro, True)]
            assert as_list(base.rects, True) == want
            for band, group, min_frames in ((128, 8, 1), (32, 1, 1), (4096, 64, 1), (64, 3, 1), (128, 8, 8)):
                env.configure("q_band_px", band)
                env.configure("q_group_units", group)
                env.configure("q_band_min_frames", min_frames)
                for tile_split in ("0", "1.5", "99"):      # nothing / the default share / every tile scale on the gather chain
                    env.configure("tile_split", tile_split)
                    r = env.detect(c, frames, p)
                    assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, (n_frames, band, group, tile_split)
                    assert [sum(l["stage_entered"][s] for l in r.launches) for s in range(len(r.stage_entered))] == r.stage_entered
                    r2 = env.detect(c, frames)          # the timed (uncounted) kernel variants
                    assert np.array_equal(r2.rects, base.rects)
    finally:
        env.configure("q_band_px", 128)
        env.configure("q_group_units", 8)
        env.configure("q_band_min_frames", 8)
        env.configure("tile_split", "0,1.75,2")


def test_pass_split_and_occupancy_do_not_change_results(env, cascades):
    c, _ = cascades("frontalface_alt")
    frames = synth.batch(2, 480, 640, seed0=70)
    p = default_params(flags=VJ_FLAG_COUNTERS)
    base = env.detect(c, frames, p)
    try:
        for split in ("1", "2,3,5,8,13", "21", "4,9,15", "0"):
            env.configure("pass_split", split)
            r = env.detect(c, frames, p)
            assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, split
        env.configure("pass_split", "")
        for b in (1, 3, 16):
            env.configure("blocks_per_cu", b)
            r = env.detect(c, frames, p)
            assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, b
        env.configure("blocks_per_cu", 8)
        env.configure("thin_pass_spread", 0)       # every workgroup of a queue pass draws tickets / only the first ones do
        r = env.detect(c, frames, p)
        assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
        env.configure("thin_pass_spread", 1)
        for de, x4 in ((0, 1), (0, 0), (1, 0)):
            env.configure("tile_deinterleave", de)
            env.configure("tile_stage_x4", x4)
            r = env.detect(c, frames, p)
            assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, (de, x4)
        env.configure("tile_deinterleave", 1)
        env.configure("tile_stage_x4", 1)
        for conc, reserve, blocks, split, gbw in ((0, 26, 1, 0, 32), (0, 0, 0, 0.4, 0), (1, 0, 1, 1.3, 32), (1, 40, 0, 2.5, 16),
                                                  (1, 26, 0, 0.7, 64), (1, 26, 0, 99, 32)):
            env.configure("concurrent", conc)
            env.configure("tile_lds_reserve_kb", reserve)
            env.configure("global_blocks", blocks)
            env.configure("tile_split", split)
            env.configure("grid_block_w", gbw)
            r = env.detect(c, frames, p)
            assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, (conc, reserve, blocks, split)
            assert [sum(l["stage_entered"][s] for l in r.launches) for s in range(len(r.stage_entered))] == r.stage_entered
        env.configure("concurrent", 1)
        env.configure("tile_lds_reserve_kb", 16)
        env.configure("global_blocks", 0)
        env.configure("tile_split", "0,1.75,2")
        env.configure("grid_block_w", 32)
        # LDS-tile path off / small / large tiles / shallow / deep: the tile and the
        # global-gather paths agree bit for bit
        for classes, tile_end, minw in (("0,0,0", 10, 1024), ("24,40,60", 3, 1024), ("36,64,140", 10, 1024),
                                        ("36,64,140", 22, 64), ("140,140,140", 14, 2048), ("20,0,0", 7, 128)):
            env.configure("tile_classes_kb", classes)
            env.configure("tile_end", tile_end)
            env.configure("tile_min_windows", minw)
            env.configure("tile_min_lanes", {3: 0, 10: 12, 22: 64, 14: 1, 7: 200}[tile_end])
            env.configure("tile_repack", {3: "", 10: "3,5", 22: "1,2,3,4,5,6,7,9,11,13,17", 14: "2", 7: "6"}[tile_end])
            env.configure("tile_sp_begin", {3: 64, 10: 8, 22: 4, 14: 1, 7: 6}[tile_end])
            env.configure("tile_sp_max", {3: 96, 10: 48, 22: 256, 14: 200, 7: 45}[tile_end])
            env.configure("tile_finish", {3: 1, 10: 1, 22: 0, 14: 1, 7: 0}[tile_end])
            env.configure("tile_ws_max", {3: 512, 10: 100, 22: 512, 14: 512, 7: 64}[tile_end])
            for split in ("", "22", "7", "2,4,6,9,12,15,18"):
                env.configure("pass_split", split)
                r = env.detect(c, frames, p)
                assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, \
                    (classes, tile_end, minw, split)
    finally:
        env.configure("concurrent", 1)
        env.configure("tile_lds_reserve_kb", 16)
        env.configure("global_blocks", 0)
        env.configure("tile_split", "0,1.75,2")
        env.configure("grid_block_w", 32)
        env.configure("pass_split", "")
        env.configure("blocks_per_cu", 8)
        env.configure("tile_classes_kb", "-2,-1,0")
        env.configure("tile_end", 64)
        env.configure("tile_min_windows", 768)
        env.configure("tile_min_lanes", 0)
        env.configure("tile_repack", ",".join(str(i) for i in range(2, 22)))
        env.configure("tile_sp_begin", 3)
        env.configure("tile_sp_max", 192)
        env.configure("tile_finish", 1)
        env.configure("tile_ws_max", 512)


@pytest.mark.parametrize("casc", ["frontalface_alt", "frontalface_alt2"])
def test_finish_variants_agree(env, cascades, casc):
    """The stump-parallel and the wave-split finish of the tile kernel (and neither) give the same rectangles and
    per-stage counts as the oracle-checked default, on frames that keep many windows alive (noise) and few (smooth).
    frontalface_alt2 (two-node trees) takes the wave-split finish only."""
    c, _ = cascades(casc)
    frames = np.stack([make_frame("noise", 31, 540, 960), make_frame("smooth", 32, 540, 960),
                       make_frame("blocks", 33, 540, 960)])
    p = default_params(flags=VJ_FLAG_COUNTERS)
    try:
        env.configure("tile_sp_begin", 99)
        base = env.detect(c, frames, p)
        env.configure("tile_sp_begin", 3)
        env.configure("tile_classes_kb", "0,0,0")   # no LDS tiles: every scale runs as grid pass + queue passes
        r = env.detect(c, frames, p)
        assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
        env.configure("global_blocks", 1)           # ... and as unstaged 2-D blocks in the tile kernel
        r = env.detect(c, frames, p)
        assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
        assert [l["kind"] for l in r.launches] == (["block"] if casc == "frontalface_alt" else ["grid", "queue"])
        env.configure("global_blocks", 0)
        env.configure("tile_classes_kb", "-2,-1,0")
        for finish, begin, ws_max, sp_max, ws_min in ((1, 3, 512, 192, 32), (1, 1, 512, 192, 0), (1, 2, 200, 192, 100),
                                                      (1, 5, 64, 192, 8), (1, 3, 512, 192, 256), (0, 3, 512, 192, 32),
                                                      (0, 4, 512, 256, 32)):
            env.configure("tile_finish", finish)
            env.configure("tile_ws_min", ws_min)
            env.configure("tile_sp_begin", begin)
            env.configure("tile_ws_max", ws_max)
            env.configure("tile_sp_max", sp_max)
            r = env.detect(c, frames, p)
            assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered, (finish, begin, ws_max)
    finally:
        env.configure("global_blocks", 0)
        env.configure("tile_classes_kb", "-2,-1,0")
        env.configure("tile_finish", 1)
        env.configure("tile_sp_begin", 3)
        env.configure("tile_ws_max", 512)
        env.configure("tile_ws_min", 48)
        env.configure("tile_sp_max", 192)


def test_scale_mask_partitions_the_result(env, cascades):
    """Scales are independent: the union over a partition of the scales is the full result."""
    c, _ = cascades("frontalface_alt")
    img = make_frame("noise", 5, 480, 640)
    full = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS))
    n = len(c.plan_scales(640, 480))
    parts = [env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS, scales=range(k, n, 3))) for k in range(3)]
    merged = np.concatenate([p.rects for p in parts])
    merged = merged[np.lexsort((merged["x"], merged["y"], merged["scale_idx"], merged["frame"]))]
    assert np.array_equal(merged, full.rects)
    assert sum(p.windows for p in parts) == full.windows
    assert [sum(v) for v in zip(*[p.stage_entered for p in parts])] == full.stage_entered


def test_repeatable_and_counter_invariants_at_full_size(env, cascades):
    """BASELINE config 3 shape (batch of 1080p frames, reduced to 8 here for time): idempotence and
    integer invariants that hold at any size."""
    c, _ = cascades("frontalface_alt")
    frames = synth.batch(8, 1080, 1920, seed0=1)
    p = default_params(flags=VJ_FLAG_COUNTERS)
    r1 = env.detect(c, frames, p)
    r2 = env.detect(c, frames, p)
    assert np.array_equal(r1.rects, r2.rects) and r1.stage_entered == r2.stage_entered
    assert r1.windows == 8 * 6290352 == r1.stage_entered[0]
    assert all(a >= b for a, b in zip(r1.stage_entered, r1.stage_entered[1:]))
    r0 = env.detect(c, frames, default_params())          # counters off: same detections
    assert np.array_equal(r0.rects, r1.rects)
    for rect in r1.rects:                                   # every window lies inside its frame
        assert 0 <= rect["x"] and rect["x"] + rect["w"] <= 1920 and rect["y"] + rect["h"] <= 1080


def test_flat_and_extreme_images(env, oracle, cascades):
    """variance == 0 (flat), the 'variance = 1' branch, all-white, all-black."""
    c, a = cascades("frontalface_alt")
    for img in (np.zeros((120, 160), np.uint8), np.full((120, 160), 255, np.uint8), np.full((120, 160), 7, np.uint8),
                np.tile(np.array([[0, 255], [255, 0]], np.uint8), (60, 80))):
        r = run(env, c, img)
        ro, st = oracle.detect(a, img)
        assert as_list(r.rects) == as_list(ro) and r.stage_entered == st["stage_entered"]


def test_empty_and_too_small_inputs(env, cascades):
    c, _ = cascades("frontalface_alt")
    assert env.detect(c, [], default_params()).match_count == 0
    r = env.detect(c, np.zeros((25, 25), np.uint8), default_params(flags=VJ_FLAG_COUNTERS))   # no scale fits
    assert r.match_count == 0 and r.windows == 0
    from clfacedetection_amd import VjError
    with pytest.raises(VjError):
        env.detect(c, [np.zeros((40, 40), np.uint8), np.zeros((41, 40), np.uint8)], default_params())


@pytest.mark.parametrize("min_neighbors", [1, 2, 3])
def test_min_neighbors_grouping(env, oracle, cascades, min_neighbors):
    """BASELINE config 1 shape: 640x480, frontalface_default, minNeighbors = 3 — raw candidates from the
    HIP path, grouped on the host as the reference does (clod.cpp:1325-1326)."""
    c, a = cascades("frontalface_default")
    # a frame with a planted cluster of detections: tile one detected window of a noise frame
    frames = synth.batch(3, 480, 640, seed0=120, kinds=("noise",))
    raw = env.detect(c, frames, default_params())
    got = clodDetectObjects(frames, c, env, (0, 0), (0, 0), min_neighbors, 0, True)
    want = []
    for f in range(len(frames)):
        ro, _ = oracle.detect(a, frames[f])
        xywh = np.stack([ro[k] for k in ("x", "y", "w", "h")], 1) if len(ro) else np.zeros((0, 4), np.int32)
        g, w = oracle.group_rectangles(xywh, max(min_neighbors, 1))
        want += [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(n), f) for r, n in zip(g, w)]
    assert [(int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"]), int(r["weight"]), int(r["frame"])) for r in got.rects] == want
    assert len(got.rects) <= len(raw.rects)


def test_grouping_of_dense_detections(env, oracle, cascades):
    """Force many overlapping raw candidates (an all-stages-pass image does not exist, so re-use raw
    detections of several noise frames as one frame's list) and group them through the C ABI."""
    from clfacedetection_amd import group_rectangles
    c, a = cascades("frontalface_alt")
    frames = synth.batch(6, 360, 480, seed0=900, kinds=("noise",))
    raw = env.detect(c, frames, default_params()).rects.copy()
    raw["frame"] = 0
    raw = np.concatenate([raw] * 3)          # every rect three times: classes of >= 3 members
    got = group_rectangles(raw, 2)
    g, w = oracle.group_rectangles(np.stack([raw[k] for k in ("x", "y", "w", "h")], 1), 2)
    assert np.array_equal(np.stack([got[k] for k in ("x", "y", "w", "h")], 1), g)
    assert got["weight"].astype(int).tolist() == w.tolist() and len(got) > 0


def test_config4_4096_alt_tree(env, oracle, cascades):
    """BASELINE config 4: one 4096x4096 frame, frontalface_alt_tree (stage tree, 8468 stumps, 56 scales,
    53,305,712 windows).  Full-size checks by size-independent properties + exact parity on the largest
    scales (selected through min_window_size, which both sides implement)."""
    c, a = cascades("frontalface_alt_tree")
    img = synth.frame("noise", 4096, 4096, 4096)
    p = default_params(flags=VJ_FLAG_COUNTERS)
    r1 = env.detect(c, img, p)
    assert r1.windows == 53305712 == r1.stage_entered[0]
    r2 = env.detect(c, img, default_params())
    assert np.array_equal(r1.rects, r2.rects)                                   # repeatable, counters on/off
    n = len(c.plan_scales(4096, 4096))
    assert n == 56
    parts = [env.detect(c, img, default_params(scales=range(k, n, 2))).rects for k in range(2)]
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["x"], merged["y"], merged["scale_idx"], merged["frame"]))]
    assert np.array_equal(merged, r1.rects)                                     # scales are independent
    # exact parity on the scales with windows >= 900 px (few windows: the oracle is quick there)
    big = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS, min_w=900, min_h=900))
    ro, st = oracle.detect(a, img, min_size=(900, 900))
    assert as_list(big.rects) == as_list(ro) and big.stage_entered == st["stage_entered"]
    assert as_list(r1.rects[r1.rects["w"] >= 900]) == as_list(ro)
    # the linear prefix of the stage tree (stages 0..4) on the linear kernels + general pass from its queue == the
    # one-pass general kernel
    try:
        env.configure("general_prefix", 0)
        r0 = env.detect(c, img, p)
        assert [l["kind"] for l in r0.launches] == ["grid"] and len(r1.launches) > 3     # prefix + the two chains' passes
        assert np.array_equal(r0.rects, r1.rects) and r0.stage_entered == r1.stage_entered
        env.configure("general_prefix", 1)
        env.configure("tile_segments", 0)        # tiles hand over after the prefix instead of running the chains
        r2 = env.detect(c, img, p)
        assert np.array_equal(r2.rects, r1.rects) and r2.stage_entered == r1.stage_entered
    finally:
        env.configure("general_prefix", 1)
        env.configure("tile_segments", 1)


def test_config5_two_cascades_on_rois(env, oracle, cascades):
    """BASELINE config 5 shape: frontalface_alt2 on 1280x720 frames, then haarcascade_eye on every face
    ROI (ROI = sub-image view: pointer + stride).  Both legs against the oracle."""
    from clfacedetection_amd import group_rectangles
    c2, a2 = cascades("frontalface_alt2")
    ce, ae = cascades("eye")
    frames = synth.batch(4, 720, 1280, seed0=777, kinds=("noise", "blocks"))
    faces = env.detect(c2, frames, default_params(flags=VJ_FLAG_COUNTERS))
    n_rois = 0
    for f in range(len(frames)):
        ro, st = oracle.detect(a2, frames[f])
        mine = faces.rects[faces.rects["frame"] == f]
        assert as_list(mine) == as_list(ro)
        for face in mine[:6]:
            x, y, w, h = (int(face[k]) for k in ("x", "y", "w", "h"))
            roi = frames[f][y:y + h, x:x + w]                  # strided view, not a copy
            if roi.shape[0] < 30 or roi.shape[1] < 30:
                continue
            eyes = env.detect(ce, roi, default_params(flags=VJ_FLAG_COUNTERS))
            eo, est = oracle.detect(ae, np.ascontiguousarray(roi))
            assert as_list(eyes.rects) == as_list(eo) and eyes.stage_entered == est["stage_entered"]
            n_rois += 1
    assert len(faces.rects) > 0 and n_rois > 0
    # the same second leg through vj_detect_rois: every face of every frame in ONE call, ROIs of equal size
    # batched; host frames, then the device-resident batch
    import torch
    rois = np.array([(int(r["frame"]), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in faces.rects
                     if r["w"] >= 30], np.int32)
    assert len(rois) > 10 and len({(w, h) for _, _, _, w, h in rois.tolist()}) < len(rois)   # sizes repeat
    pe = default_params(flags=VJ_FLAG_COUNTERS)
    got = env.detect_rois(ce, frames, rois, pe)
    want, entered = [], np.zeros(len(got.stage_entered), np.int64)
    for i, (f, x, y, w, h) in enumerate(rois.tolist()):
        eo, est = oracle.detect(ae, np.ascontiguousarray(frames[f][y:y + h, x:x + w]))
        want += [(i,) + t[1:] for t in as_list(eo, with_frame=True)]
        entered += np.array(est["stage_entered"], np.int64)
    assert as_list(got.rects, with_frame=True) == want and got.stage_entered == entered.tolist()
    dev = DeviceFrames.from_torch(torch.from_numpy(frames).cuda())
    got_d = env.detect_rois(ce, dev, rois, pe)
    assert np.array_equal(got_d.rects, got.rects)
    assert env.detect_rois(ce, frames, np.zeros((0, 5), np.int32)).rects.size == 0
    with pytest.raises(Exception):
        env.detect_rois(ce, frames, [(0, 1270, 0, 40, 40)])          # outside the frame
    # ROIs of BGR frames: the view starts at the ROI's first pixel, 3 bytes per pixel
    col = np.repeat(frames[..., None], 3, 3)
    got_c = env.detect_rois(ce, col, rois[:12], pe, color=True)
    got_g = env.detect_rois(ce, frames, rois[:12], pe)
    assert np.array_equal(got_c.rects, got_g.rects) and got_c.stage_entered == got_g.stage_entered


def test_subbatching_and_detection_buffer_growth(env, cascades):
    """Batches are cut into sub-batches when offsets or queues would overflow, and the detection buffer
    grows (the cascade passes are re-run) when it overflows: force both and compare with the plain run."""
    c, _ = cascades("frontalface_alt")
    frames = synth.batch(7, 300, 400, seed0=2000, kinds=("noise",))
    p = default_params(flags=VJ_FLAG_COUNTERS)
    base = env.detect(c, frames, p)
    assert len(base.rects) >= 3
    try:
        env.configure("max_subbatch", 3)
        env.configure("det_cap", 1)
        r = env.detect(c, frames, p)
        assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
        r = env.detect(c, frames, default_params())
        assert np.array_equal(r.rects, base.rects)
    finally:
        env.configure("max_subbatch", 0)
        env.configure("det_cap", 65536)


def test_native_library_is_the_one_running(env):
    """The GPU tests must run hand-written HIP: libvjhip.so is mapped into this process."""
    maps = open("/proc/self/maps").read()
    assert "libvjhip.so" in maps
    assert "gfx950" in env.device_name


def test_color_frames_detect_like_their_gray(env, oracle, cascades):
    """Image ingest: BGR / BGRA frames converted inside the integral kernels give exactly the
    detections of the oracle-converted gray frames — host arrays, strided ROI views and device-resident batches."""
    import torch
    c, a = cascades("frontalface_alt")
    rng = np.random.default_rng(77)
    gray = [make_frame(k, 40 + i, 300, 420) for i, k in enumerate(("noise", "blocks"))]
    for ch in (3, 4):
        # color frames whose gray value is the synthetic frame +- channel noise
        col = [np.clip(g[..., None].astype(np.int16) + rng.integers(-20, 21, g.shape + (ch,)), 0, 255).astype(np.uint8) for g in gray]
        want = [oracle.bgr2gray(f) for f in col]
        base = env.detect(c, want, default_params(flags=VJ_FLAG_COUNTERS))
        r = env.detect(c, col, default_params(flags=VJ_FLAG_COUNTERS), color=True)
        assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
        for f in range(2):
            ro, _ = oracle.detect(a, want[f])
            assert as_list(r.rects[r.rects["frame"] == f]) == as_list(ro)
        t = torch.from_numpy(np.stack(col)).cuda()
        rd = env.detect(c, DeviceFrames.from_torch(t), default_params(flags=VJ_FLAG_COUNTERS))
        assert np.array_equal(rd.rects, base.rects)
        roi = col[0][5:250, 7:333]
        rr = env.detect(c, roi, default_params(), color=True)
        rg = env.detect(c, np.ascontiguousarray(want[0][5:250, 7:333]), default_params())
        assert np.array_equal(rr.rects, rg.rects)


@pytest.mark.parametrize("seed", range(40))
def test_randomized_parity(env, oracle, cascades, seed):
    """Random frame sizes (odd widths, thin frames, frames barely larger than the window), cascades, frame kinds,
    size limits and scale factors against the oracle — rectangles, per-stage counts and window counts."""
    rng = np.random.default_rng(1000 + seed)
    casc = ["frontalface_alt", "frontalface_default", "frontalface_alt2", "eye", "frontalface_alt_tree"][seed % 5]
    c, a = cascades(casc)
    w = int(rng.integers(c.info.win_w + 11, 1100))
    h = int(rng.integers(c.info.win_h + 11, 700))
    if seed % 7 == 3:
        h = c.info.win_h + 12          # a strip: few window rows
    kind = ["noise", "smooth", "blocks"][int(rng.integers(0, 3))]
    img = make_frame(kind, 5000 + seed, h, w)
    mn = (0, 0) if seed % 3 else (int(rng.integers(20, 60)),) * 2
    mx = (0, 0) if seed % 4 else (int(rng.integers(80, 300)),) * 2
    sf = [1.1, 1.2, 1.05, 1.3][seed % 4]
    p = default_params(flags=VJ_FLAG_COUNTERS, min_w=mn[0], min_h=mn[1], max_w=mx[0], max_h=mx[1], scale_factor=sf)
    r = env.detect(c, img, p)
    ro, st = oracle.detect(a, img, min_size=mn, max_size=mx, scale_factor=sf)
    assert as_list(r.rects) == as_list(ro), (casc, w, h, kind, mn, mx, sf)
    assert r.stage_entered == st["stage_entered"] and r.windows == st["windows"]
    # and in a batch with its mirror image (independent frames, queue parts by frame)
    rb = env.detect(c, [img, img[:, ::-1], img], p)
    assert as_list(rb.rects[rb.rects["frame"] == 0]) == as_list(ro) == as_list(rb.rects[rb.rects["frame"] == 2])


def test_reserve_then_detect_other_sizes(oracle, cascades):
    """clodInitBuffers (vj_env_reserve) pre-sizes the device buffers; detection on smaller and on larger frames
    afterwards — the zeroed slack rows belong to a layout, not to a buffer — still matches the oracle."""
    from clfacedetection_amd import clodInitBuffers, clodInitEnvironment, clodReleaseEnvironment
    c, a = cascades("frontalface_alt")
    env2 = clodInitEnvironment(0)
    try:
        clodInitBuffers(env2, (640, 480), 2)
        for h, w in ((240, 320), (480, 640), (300, 900), (240, 320)):
            img = make_frame("noise", h + w, h, w)
            r = env2.detect(c, [img, img[::-1].copy()], default_params(flags=VJ_FLAG_COUNTERS))
            ro, st = oracle.detect(a, img)
            assert as_list(r.rects[r.rects["frame"] == 0]) == as_list(ro)
    finally:
        clodReleaseEnvironment(env2)


def test_balance_is_keyed_by_batch_size_class_and_travels(env, cascades, tmp_path):
    """The feedback's table names a workload by the cascade's CONTENT, the frame size, the parameters and the batch-size CLASS
    (8-15, 16-31, ...): another size of the same class runs the found split without a search of its own, another class searches
    for itself, and an exported table lets a second environment start on the found split (results never depend on any of it)."""
    from clfacedetection_amd import Environment, VjError
    c, _ = cascades("frontalface_default")
    frames = synth.batch(20, 360, 640, seed0=300)
    env.configure("auto_balance", "reset")
    env.configure("auto_balance", "1")
    try:
        want12 = env.detect(c, frames[:12])
        r = want12
        for _ in range(60):
            if r.balance_state == 2:
                break
            r = env.detect(c, frames[:12])
            assert np.array_equal(r.rects, want12.rects)
        assert r.balance_state == 2
        s12 = env.detect(c, frames[:12]).tile_split       # (the call that ended the search still ran its last candidate)
        r9 = env.detect(c, frames[:9])                    # 9 frames: the class of 12
        assert r9.balance_state == 2 and r9.tile_split == s12
        r20 = env.detect(c, frames)                       # 20 frames: the next class starts its own search
        assert r20.balance_state == 1 and r20.balance_calls <= 1
        path = str(tmp_path / "balance.txt")
        env.configure("balance_export", path)
        lines = [l.split() for l in open(path) if l.startswith("vjbal1")]
        assert len(lines) >= 2 and any(int(l[12]) == 8 and float(l[13]) == s12 and int(l[15]) == 3 for l in lines)
        e2 = Environment(0)
        try:
            e2.configure("balance_import", path)
            r2 = e2.detect(c, frames[:12])                # (a second load of the same cascade would hash the same too)
            assert r2.balance_state == 2 and r2.tile_split == s12 and np.array_equal(r2.rects, want12.rects)
            with pytest.raises(VjError):
                e2.configure("balance_import", str(tmp_path / "missing.txt"))
        finally:
            e2.close()
    finally:
        env.configure("auto_balance", "reset")


def test_chain_balance_feedback_does_not_change_results(env, oracle, cascades):
    """vj_detect finds the chain balance of a batch workload (Plan::tile_split) by a short hill climb on the measured
    cascade time of its first calls, then freezes it: every call of the search returns the same rectangles (the split only
    moves work between the two chains), the split stops moving, and configuring tile_split by hand switches the feedback off."""
    c, a = cascades("frontalface_default")
    frames = synth.batch(12, 360, 640, seed0=300)
    env.configure("auto_balance", "reset")
    env.configure("auto_balance", "1")
    first = env.detect(c, frames)
    splits = []
    for _ in range(100):
        r = env.detect(c, frames)
        assert np.array_equal(r.rects, first.rects)
        splits.append(r.tile_split)
    assert len(set(splits[-6:])) == 1 and all(0.0 <= s <= 3.0 for s in splits)
    assert r.balance_state == 2 and 0 < r.balance_calls <= 45      # the search reports itself: finished, within its budget of measured calls
    ro, _ = oracle.detect(a, frames[5])
    assert as_list(first.rects[first.rects["frame"] == 5]) == as_list(ro)
    env.configure("tile_split", "0,1.75,2")             # static values: the feedback is off
    try:
        assert {env.detect(c, frames).tile_split for _ in range(7)} == {1.75}      # (12 frames: the value for 8 .. 31)
        assert np.array_equal(env.detect(c, frames).rects, first.rects)
    finally:
        env.configure("auto_balance", "reset")
    # a pyramid with ONE tile scale (frontalface_alt, min size 66: the 69-px scale runs on tiles, everything above on gathers) and
    # a search that starts with that scale moved to the gather chain: a one-chain plan.  The search measures it like any other
    # candidate and goes on (up, then back below the start) — it used to end on such a plan, whatever it cost.
    alt, _ = cascades("frontalface_alt")
    env.configure("tile_split", "1.0")
    env.configure("auto_balance", "reset")               # feedback on again, the start values stay
    try:
        p = default_params(min_w=66, min_h=66)
        ref = env.detect(alt, frames, p)
        assert ref.tile_split == 1.0 and not any(l["kind"] == "tile" for l in ref.launches)
        seen = []
        for _ in range(100):
            r = env.detect(alt, frames, p)
            assert np.array_equal(r.rects, ref.rects)
            seen.append(r.tile_split)
        assert len(set(seen[-6:])) == 1 and len(set(seen)) >= 3, seen
    finally:
        env.configure("tile_split", "0,1.75,2")
        env.configure("auto_balance", "reset")
    # vj_detect_chain searches the balance of its first cascade the same way: same two results in every call of the search
    eye, _ = cascades("eye")
    c1, c2 = env.detect_chain(c, eye, frames)
    seen = []
    for _ in range(100):
        r1, r2 = env.detect_chain(c, eye, frames)
        assert np.array_equal(r1.rects, c1.rects) and np.array_equal(r2.rects, c2.rects)
        seen.append(r1.tile_split)
    assert len(set(seen[-6:])) == 1 and len(set(seen)) >= 2, seen
    # one 16-megapixel frame per call is a workload of its own: the split search runs, then ONE more candidate — the lower
    # tile thresholds (more scales on the tile chain) — and the search ends; same rectangles in every call
    tree, _ = cascades("frontalface_alt_tree")
    big = synth.batch(1, 4096, 4096, seed0=4001, kinds=("blocks",))
    want = env.detect(tree, big)
    seen = []
    for _ in range(80):
        r = env.detect(tree, big)
        assert np.array_equal(r.rects, want.rects)
        seen.append((r.tile_split, len(r.launches)))
    assert len(set(seen[-8:])) == 1 and len(set(s for s, _ in seen)) >= 2, seen
    del big
    # a share of the scales (a mask) starts without a move: the default is a fraction of the last tile scale of a WHOLE pyramid
    assert env.detect(c, frames, default_params(scales=[0, 1, 20, 21])).tile_split == 0.0
    env.configure("auto_balance", "reset")
