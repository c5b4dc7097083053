"""OpenCV arithmetic profile (vj_detect_opencv) vs the oracle's restatement of the same lines of
tempcv.cpp (oc_detect_opencvlike).  OpenCV itself cannot be run here: parity unpinned at that boundary."""
import numpy as np
import pytest

from cases import make_frame
from clfacedetection_amd import VJ_FLAG_COUNTERS, cvHaarDetectObjects, synth

pytestmark = pytest.mark.gpu


def rows(rects):
    return [tuple(int(r[k]) for k in ("scale_idx", "x", "y", "w", "h")) for r in rects]


@pytest.mark.parametrize("casc,kind,seed,h,w", [
    ("frontalface_alt", "xorshift", 12345, 480, 640),       # the pinned frame of the clod profile: both find the same 2 faces
    ("frontalface_alt", "noise", 3, 300, 420),
    ("frontalface_alt", "smooth", 4, 360, 500),
    ("frontalface_alt", "blocks", 5, 720, 1280),
    ("frontalface_default", "blocks", 6, 480, 640),
    ("eye", "noise", 7, 240, 320),
    ("frontalface_alt2", "noise", 8, 300, 400),              # two-node trees
    ("frontalface_alt2", "blocks", 9, 540, 960),
])
def test_matches_oracle_restatement(env, oracle, cascades, casc, kind, seed, h, w):
    c, a = cascades(casc)
    img = make_frame(kind, seed, h, w, oracle)
    r = env.detect_opencv(c, img, flags=VJ_FLAG_COUNTERS)
    ro, st = oracle.detect_opencvlike(a, img)
    assert sorted(rows(r.rects)) == sorted(rows(ro))
    assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"]


def test_min_size_scale_factor_batch_and_grouping(env, oracle, cascades):
    c, a = cascades("frontalface_alt")
    frames = synth.batch(3, 400, 600, seed0=21)
    r = env.detect_opencv(c, frames, min_size=(40, 40), scale_factor=1.2, flags=VJ_FLAG_COUNTERS)
    got = {f: rows(r.rects[r.rects["frame"] == f]) for f in range(3)}
    entered = np.zeros(len(r.stage_entered), np.int64)
    for f in range(3):
        ro, st = oracle.detect_opencvlike(a, frames[f], min_size=(40, 40), scale_factor=1.2)
        assert sorted(got[f]) == sorted(rows(ro)) and all(t[3] >= 40 for t in got[f])
        entered += np.array(st["stage_entered"], np.int64)
    assert r.stage_entered == entered.tolist()
    # minNeighbors: cv::groupRectangles on the raw candidates, as the clod profile applies it
    from clfacedetection_amd import group_rectangles
    raw = env.detect_opencv(c, frames[0])
    grouped = cvHaarDetectObjects(frames[0], c, env, 1.1, 2)
    assert np.array_equal(grouped.rects, group_rectangles(raw.rects, 2))


def test_flat_extreme_and_skip_rule_images(env, oracle, cascades):
    """Flat frames (variance 0 -> vnf 0 or 1), saturated frames and a frame whose left half rejects at stage 0
    (long reject runs: the skip rule's parity carries across 64-position chunks) against the oracle."""
    c, a = cascades("frontalface_alt")
    noise = make_frame("noise", 77, 240, 700, oracle)
    half = noise.copy()
    half[:, :350] = (half[:, :350] // 64) + 100          # nearly flat left half
    for img in (np.full((240, 320), 128, np.uint8), np.zeros((200, 260), np.uint8), np.full((200, 260), 255, np.uint8), half):
        r = env.detect_opencv(c, img, flags=VJ_FLAG_COUNTERS)
        ro, st = oracle.detect_opencvlike(a, img)
        assert sorted(rows(r.rects)) == sorted(rows(ro))
        assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"]
    assert r.windows < sum((700 - 10) // 2 * ((240 - 10) // 2) for _ in range(1)) * 40   # sanity: a bounded count


def test_refuses_stage_trees(env, cascades):
    c, _ = cascades("frontalface_alt_tree")
    with pytest.raises(Exception):
        env.detect_opencv(c, np.zeros((200, 200), np.uint8))


def test_color_frames_subbatches_and_many_detections(env, oracle, cascades):
    """BGR input goes through the same fused conversion as the clod profile; sub-batching (max_subbatch) and the
    growth of the detection buffer (eye cascade on noise: tens of thousands of raw candidates) keep results equal."""
    c, a = cascades("eye")
    rng = np.random.default_rng(3)
    gray = list(synth.batch(5, 300, 400, seed0=90, kinds=("blocks", "smooth")))
    col = [np.repeat(g[..., None], 3, 2) for g in gray]                 # B = G = R: converts back to g exactly
    base = env.detect_opencv(c, gray, flags=VJ_FLAG_COUNTERS)
    assert len(base.rects) > 100
    r = env.detect_opencv(c, col, flags=VJ_FLAG_COUNTERS, color=True)
    assert np.array_equal(r.rects, base.rects) and r.stage_entered == base.stage_entered
    try:
        env.configure("max_subbatch", 2)
        r2 = env.detect_opencv(c, gray, flags=VJ_FLAG_COUNTERS)
        assert np.array_equal(r2.rects, base.rects) and r2.stage_entered == base.stage_entered and r2.windows == base.windows
    finally:
        env.configure("max_subbatch", 0)
    ro, st = oracle.detect_opencvlike(a, gray[0])
    assert sorted(rows(base.rects[base.rects["frame"] == 0])) == sorted(rows(ro))


@pytest.mark.parametrize("seed", range(12))
def test_randomized_parity(env, oracle, cascades, seed):
    """Random sizes, cascades, min sizes and scale factors: rectangles, visited windows and per-stage counts equal the
    oracle's restatement."""
    rng = np.random.default_rng(2000 + seed)
    casc = ["frontalface_alt", "frontalface_default", "frontalface_alt2", "eye"][seed % 4]
    c, a = cascades(casc)
    w = int(rng.integers(c.info.win_w + 12, 800))
    h = int(rng.integers(c.info.win_h + 12, 560))
    img = make_frame(["noise", "smooth", "blocks"][seed % 3], 6000 + seed, h, w, oracle)
    mn = (0, 0) if seed % 2 else (int(rng.integers(20, 70)),) * 2
    sf = [1.1, 1.25, 1.07][seed % 3]
    r = env.detect_opencv(c, img, min_size=mn, scale_factor=sf, flags=VJ_FLAG_COUNTERS)
    ro, st = oracle.detect_opencvlike(a, img, min_size=mn, scale_factor=sf)
    assert sorted(rows(r.rects)) == sorted(rows(ro)), (casc, w, h, mn, sf)
    assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"]
