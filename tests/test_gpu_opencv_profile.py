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


@pytest.mark.parametrize("kind,seed,h,w", [("noise", 71, 240, 320), ("blocks", 72, 300, 400), ("smooth", 73, 360, 500)])
def test_stage_tree_cascade(env, oracle, cascades, kind, seed, h, w):
    """frontalface_alt_tree: cvRunHaarClassifierCascadeSum's is_tree walk (tempcv.cpp:834-861) returns 0 on ANY reject, so
    the invoker skips the next window after every rejected one (:1163), not only after a stage-0 reject."""
    c, a = cascades("frontalface_alt_tree")
    img = make_frame(kind, seed, h, w, oracle)
    r = env.detect_opencv(c, img, flags=VJ_FLAG_COUNTERS)
    ro, st = oracle.detect_opencvlike(a, img)
    assert sorted(rows(r.rects)) == sorted(rows(ro))
    assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"]


@pytest.mark.parametrize("casc,kind,seed,h,w", [
    ("fullbody", "noise", 81, 240, 320),                 # 14x28 stumps, 201 tilted features
    ("fullbody", "blocks", 82, 400, 300),
    ("eye_tree_eyeglasses", "noise", 83, 240, 320),      # trees of up to 3 nodes, 577 tilted features
    ("eye_tree_eyeglasses", "smooth", 84, 300, 420),
])
def test_tilted_features(env, oracle, cascades, casc, kind, seed, h, w):
    """Tilted rectangles read four corners of the tilted integral image (tempcv.cpp:743-750), weight correction 0.5 (:731)."""
    c, a = cascades(casc)
    assert c.info.n_tilted > 0
    img = make_frame(kind, seed, h, w, oracle)
    r = env.detect_opencv(c, img, flags=VJ_FLAG_COUNTERS)
    ro, st = oracle.detect_opencvlike(a, img)
    assert sorted(rows(r.rects)) == sorted(rows(ro))
    assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"]


@pytest.mark.parametrize("bands", [1, 0], ids=["banded_prefix_sums", "row_recurrence"])
def test_tilted_integral_and_gray_image(env, oracle, bands):
    """vj_integral_tilted (cvIntegral's tilted output) and vj_grayscale (clifGrayscale) against the oracle, odd sizes,
    BGR input and the 32-bit wrap-around included — with both kernels: the three banded prefix sums (default) and the row-by-row
    recurrence (`tilted_bands` = 0)."""
    rng = np.random.default_rng(5)
    env.configure("tilted_bands", bands)
    try:
        for (h, w) in [(1, 1), (3, 5), (2, 9), (9, 2), (8, 8), (17, 1), (1, 40), (64, 300), (251, 333), (480, 640), (1080, 1920)]:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            assert np.array_equal(env.integral_tilted(img), oracle.integral_tilted(img)), (h, w)
            assert np.array_equal(env.grayscale(img), img)
        white = np.full((3000, 3000), 255, np.uint8)              # tilted sums pass 2^32
        assert np.array_equal(env.integral_tilted(white), oracle.integral_tilted(white))
        bgr = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
        g = oracle.bgr2gray(bgr)
        assert np.array_equal(env.grayscale(bgr), g)
        assert np.array_equal(env.integral_tilted(bgr), oracle.integral_tilted(g))
    finally:
        env.configure("tilted_bands", 1)
    bgra = rng.integers(0, 256, (77, 131, 4), dtype=np.uint8)
    assert np.array_equal(env.grayscale(bgra[20:60, 10:100]), oracle.bgr2gray(np.ascontiguousarray(bgra[20:60, 10:100])))


@pytest.mark.parametrize("three_rects", [True, False])
def test_node_products_follow_the_scalar_branch(env, oracle, three_rects):
    """The window of tests/test_oracle_cv.py whose verdict depends on HOW the rectangle sums are multiplied: binary32
    products (int * float, tempcv.cpp:907-911) when the stage has a three-rectangle node, f64 products (:872-888) in a
    two_rects stage.  The device must land on the reference's side of the threshold in both."""
    from cases import cascade_to_product, crafted_stump_cascade, single_window_frame
    from test_oracle_cv import numpy_single_window
    img, factor = single_window_frame(seed=5)
    c0 = crafted_stump_cascade(three_rects, threshold=0.0)
    s32 = numpy_single_window(c0, img, factor, False)[0]
    s64 = numpy_single_window(c0, img, factor, True)[0]
    vnf = numpy_single_window(crafted_stump_cascade(three_rects, threshold=1.0), img, factor, False)[1]
    a = crafted_stump_cascade(three_rects, threshold=float(np.float32((s32 + s64) / 2 / vnf)))
    win = int(np.rint(a.win_w * factor))
    ro, _ = oracle.detect_opencvlike(a, img, min_size=(win, win))
    r = env.detect_opencv(cascade_to_product(a), img, min_size=(win, win), flags=VJ_FLAG_COUNTERS)
    assert r.windows == 1 and len(r.rects) == len(ro)
    literal = numpy_single_window(a, img, factor, f64_products=not three_rects)[2]
    other = numpy_single_window(a, img, factor, f64_products=three_rects)[2]
    assert literal != other and (len(r.rects) == 1) == literal


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


@pytest.mark.parametrize("seed", range(14))
def test_randomized_parity(env, oracle, cascades, seed):
    """Random sizes, cascades, min sizes and scale factors: rectangles, visited windows and per-stage counts equal the
    oracle's restatement."""
    rng = np.random.default_rng(2000 + seed)
    casc = ["frontalface_alt", "frontalface_default", "frontalface_alt2", "eye", "frontalface_alt_tree", "fullbody",
            "eye_tree_eyeglasses"][seed % 7]
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


def test_plan_cache_of_the_profile(env, oracle, cascades):
    """vj_detect_opencv keeps what it derives from (cascade, frame size, parameters) per environment: repeated calls give
    the same rectangles, another size / scale factor / min size / cascade gets its own plan, the cache is bounded, and an
    evicted plan is rebuilt."""
    c, a = cascades("frontalface_alt")
    c2, a2 = cascades("frontalface_alt2")
    img = synth.frame("blocks", 5, 300, 400)
    want, _ = oracle.detect_opencvlike(a, img)
    for _ in range(3):
        assert sorted(rows(env.detect_opencv(c, img).rects)) == sorted(rows(want))
    w2, _ = oracle.detect_opencvlike(a, img, scale_factor=1.2)
    assert sorted(rows(env.detect_opencv(c, img, scale_factor=1.2).rects)) == sorted(rows(w2))
    w3, _ = oracle.detect_opencvlike(a, img, min_size=(40, 40))
    assert sorted(rows(env.detect_opencv(c, img, min_size=(40, 40)).rects)) == sorted(rows(w3))
    w4, _ = oracle.detect_opencvlike(a2, img)
    assert sorted(rows(env.detect_opencv(c2, img).rects)) == sorted(rows(w4))
    assert sorted(rows(env.detect_opencv(c, img).rects)) == sorted(rows(want))
    env.configure("plan_cache_max", 4)
    try:
        for k in range(12):                                   # more sizes than the cache holds
            sub = np.ascontiguousarray(img[:200 + 5 * k, :300 + 7 * k])
            ws, _ = oracle.detect_opencvlike(a, sub)
            assert sorted(rows(env.detect_opencv(c, sub).rects)) == sorted(rows(ws)), k
        assert sorted(rows(env.detect_opencv(c, img).rects)) == sorted(rows(want))
    finally:
        env.configure("plan_cache_max", 48)



@pytest.mark.parametrize("casc,kind,seed,h,w,batch", [
    ("frontalface_alt", "noise", 31, 300, 420, 1),
    ("frontalface_alt", "faces", 32, 540, 960, 3),            # tiles of both LDS classes, detections inside the finish
    ("frontalface_alt", "smooth", 33, 720, 1280, 2),          # crowded tiles: many windows pass the early stages
    ("frontalface_default", "blocks", 34, 480, 640, 2),       # 24 x 24 window, 25 stages
    ("eye", "noise", 35, 200, 333, 4),                        # window rows that are not a multiple of the tile width
    ("frontalface_alt", "white", 0, 200, 500, 1),             # flat: every window rejects at stage 0 (variance 0), long skip runs
    ("frontalface_alt2", "faces", 39, 540, 960, 2),            # two-node trees: both nodes' gathers in flight, 2-bit leaf codes in the finish
    ("frontalface_alt2", "noise", 40, 300, 420, 3),
    ("frontalface_alt_tree", "faces", 36, 540, 960, 2),       # stage tree: prefix on tiles, cv_tree_walk, skip_resolve, cv_tree_emit
    ("frontalface_alt_tree", "noise", 37, 300, 420, 3),       # (the counted call of a stage tree walks the rows: the plain one is the tile path)
    ("frontalface_alt_tree", "smooth", 38, 400, 700, 1),
    ("fullbody", "blocks", 42, 540, 960, 2),                  # tilted features: the tilted integral's tile staged behind the sum's
    ("mcs_mouth", "noise", 43, 480, 640, 3),                  # 25 x 15 window, 223 tilted nodes
    ("upperbody", "smooth", 45, 720, 1280, 1),
    ("lefteye_2splits", "faces", 44, 540, 960, 2),            # two-node trees with tilted nodes
])
def test_lds_tile_path_equals_the_row_kernel_and_the_oracle(env, oracle, cascades, casc, kind, seed, h, w, batch):
    """The profile's small scales run on LDS tiles (vj_cv_tile.hip: reject bits of stage 0, skip_resolve, the cascade on
    the visited windows), the large ones on cv_profile_pass; cv_tiles = 0 sends every scale through cv_profile_pass.  Both
    give the same rectangles, visited-window counts and per-stage counts, and they are the oracle's."""
    c, a = cascades(casc)
    frames = np.stack([make_frame(kind, seed + i, h, w, oracle) for i in range(batch)])
    tiled = env.detect_opencv(c, frames, flags=VJ_FLAG_COUNTERS)
    plain_tiled = env.detect_opencv(c, frames)
    try:
        env.configure("cv_tiles", 0)
        rows_only = env.detect_opencv(c, frames, flags=VJ_FLAG_COUNTERS)
    finally:
        env.configure("cv_tiles", 1)
    assert np.array_equal(tiled.rects, rows_only.rects) and np.array_equal(plain_tiled.rects, tiled.rects)
    assert tiled.windows == rows_only.windows and tiled.stage_entered == rows_only.stage_entered
    entered = np.zeros(len(tiled.stage_entered), np.int64)
    vis = 0
    for f in range(batch):
        ro, st = oracle.detect_opencvlike(a, frames[f])
        assert sorted(rows(tiled.rects[tiled.rects["frame"] == f])) == sorted(rows(ro)), f
        entered += np.array(st["stage_entered"], np.int64)
        vis += st["windows"]
    assert tiled.stage_entered == entered.tolist() and tiled.windows == vis
    if casc == "frontalface_alt_tree":       # more prefix survivors than the tree queue holds: the call falls back to the rows
        try:
            env.configure("cv_tree_queue_cap", 16)
            assert np.array_equal(env.detect_opencv(c, frames).rects, tiled.rects)
        finally:
            env.configure("cv_tree_queue_cap", 0)
    defaults = {"cv_tile_ws_max": 512, "cv_tile_min_windows": -1, "cv_tile_min_windows0": 2048, "cv_row_blocks": -1, "concurrent": 1, "cv_row_band_px": 128, "cv_tree2": 1, "cv_tiles_tilted": 1}
    for setting in ({"cv_tile_ws_max": 64}, {"cv_tile_ws_max": 0}, {"cv_tile_min_windows": 64, "cv_tile_min_windows0": 64}, {"cv_row_band_px": 0}, {"cv_row_band_px": 40}, {"cv_tree2": 0}, {"cv_tiles_tilted": 0},
                    {"cv_row_blocks": 1, "cv_tile_min_windows0": 512, "cv_tile_min_windows": 512}, {"concurrent": 0}):
        try:                         # finish thresholds, small tiles of both LDS classes, other occupancies, one stream: same result
            for key, val in setting.items():
                env.configure(key, val)
            assert np.array_equal(env.detect_opencv(c, frames).rects, tiled.rects), setting
        finally:
            for key in setting:
                env.configure(key, defaults[key])
