"""The oracle against the committed golden fixtures (tests/golden/*.json) — guards the
fixtures the GPU box relies on against drift in the oracle or the synthetic generator."""
import json
import os

import numpy as np
import pytest

from cases import make_frame, sha

G = os.path.join(os.path.dirname(__file__), "golden")
DET = json.load(open(os.path.join(G, "detect.json")))
INT = json.load(open(os.path.join(G, "integral.json")))


@pytest.mark.parametrize("g", [d for d in DET if d["height"] <= 480], ids=lambda d: d["id"])
def test_detect_fixture(oracle, cascades, g):
    _, a = cascades(g["cascade"])
    img = make_frame(g["generator"], g["seed"], g["height"], g["width"], oracle)
    assert sha(img) == g["image_sha256"]
    r, st = oracle.detect(a, img, min_size=tuple(g["min_size"]), max_size=tuple(g["max_size"]),
                          signed_mean=g["signed_mean"])
    assert [[int(v) for v in (q["scale_idx"], q["x"], q["y"], q["w"], q["h"])] for q in r] == g["rects"]
    for k in ("windows", "stump_evals", "rect_evals", "gather_bytes", "stage_entered"):
        assert st[k] == g[k], k
    assert st["stage_entered"][0] == st["windows"]


@pytest.mark.parametrize("g", INT, ids=lambda d: d["id"])
def test_integral_fixture(oracle, g):
    img = make_frame(g["generator"], g["seed"], g["height"], g["width"], oracle)
    assert sha(img) == g["image_sha256"]
    s, q = oracle.integral(img)
    assert sha(s) == g["sum_sha256"] and sha(q) == g["sqsum_sha256"]
    assert int(s[-1, -1]) == g["sum_last"] == int(img.astype(np.uint64).sum() & 0xFFFFFFFF)
    assert int(q[-1, -1]) == g["sqsum_last"] == int((img.astype(np.uint64) ** 2).sum())
    assert not s[0].any() and not s[:, 0].any() and not q[0].any() and not q[:, 0].any()


MODES = json.load(open(os.path.join(G, "modes.json")))


@pytest.mark.parametrize("g", MODES, ids=lambda d: d["id"])
def test_mode_fixture(oracle, cascades, g):
    """The CPU variants' window sets (oracle modes 2 / 3) and the OpenCV-like path, frozen for the GPU box."""
    _, a = cascades(g["cascade"])
    img = make_frame(g["generator"], g["seed"], g["height"], g["width"], oracle)
    assert sha(img) == g["image_sha256"]
    rows = lambda r: [[int(v) for v in (q["scale_idx"], q["x"], q["y"], q["w"], q["h"])] for q in r]
    for name, mode in (("skip_list", 2), ("skip_row", 3)):
        if name in g:
            r, st = oracle.detect(a, img, mode=mode)
            assert rows(r) == g[name]["rects"] and st["stage_entered"] == g[name]["stage_entered"]
    r, st = oracle.detect_opencvlike(a, img)
    assert sorted(rows(r)) == g["opencv"]["rects"]
    assert st["windows"] == g["opencv"]["windows"] and st["stage_entered"] == g["opencv"]["stage_entered"]


def test_integral_against_numpy_cumsum(oracle):
    img = make_frame("smooth", 9, 67, 129)
    s, q = oracle.integral(img)
    ref = np.cumsum(np.cumsum(img.astype(np.uint64), 0), 1)
    refq = np.cumsum(np.cumsum(img.astype(np.uint64) ** 2, 0), 1)
    assert np.array_equal(s[1:, 1:], ref.astype(np.uint32)) and np.array_equal(q[1:, 1:], refq)


def test_integral_wraps_like_cv32s(oracle):
    # 4200 x 4200 of 255 exceeds 2^32: the sum image wraps (CV_32S), the squared sum does not
    img = np.full((4200, 4200), 255, np.uint8)
    s, q = oracle.integral(img)
    assert int(s[-1, -1]) == (255 * 4200 * 4200) % (1 << 32)
    assert int(q[-1, -1]) == 255 * 255 * 4200 * 4200


@pytest.mark.parametrize("g", json.load(open(os.path.join(G, "groups.json"))), ids=lambda d: d["id"])
def test_group_fixture(oracle, cascades, g):
    """Drawn faces -> candidates -> cv::groupRectangles -> second cascade inside every face: guards the `faces` generator,
    the oracle's grouping and the fixture the GPU box checks vj_detect_chain against."""
    _, a1 = cascades(g["first"])
    _, a2 = cascades(g["second"])
    img = make_frame("faces", g["seed"], g["height"], g["width"], oracle)
    assert sha(img) == g["image_sha256"]
    r, _ = oracle.detect(a1, img)
    assert len(r) == g["raw_candidates"]
    xywh = np.stack([r[k] for k in ("x", "y", "w", "h")], 1)
    faces, wt = oracle.group_rectangles(xywh, max(g["min_neighbors"], 1))
    assert [[int(v) for v in q] + [int(n)] for q, n in zip(faces, wt)] == g["faces"]
    assert max(f[4] for f in g["faces"]) >= 10          # real clusters, not isolated candidates
    x, y, w, h, _ = g["faces"][0]
    r2, _ = oracle.detect(a2, np.ascontiguousarray(img[y:y + h, x:x + w]))
    assert [[int(v) for v in (e["scale_idx"], e["x"], e["y"], e["w"], e["h"])] for e in r2] == g["inside"][0]


FULL = json.load(open(os.path.join(G, "fullsize.json")))


def test_fullsize_fixture_sample(oracle, cascades):
    """tests/golden/fullsize.json (tools/make_fullsize_golden.py) on a sample that fits the CPU suite: the fixture names
    the inputs cases.FULLSIZE describes, and the oracle reproduces a config-3 frame, a block-variant 1080p frame, an
    OpenCV-profile 1080p frame and a config-5 frame with its regions."""
    from cases import FULLSIZE, rows_sha
    from clfacedetection_amd import synth
    for name in ("config3", "config4", "config5_raw", "config5_grouped"):
        for k, v in FULLSIZE[name].items():
            assert FULL[name][k] == v, (name, k)
    assert [[e[k] for k in ("id", "cascade", "generator", "seed", "height", "width")] for e in FULL["opencv"]] == \
           [list(t) for t in FULLSIZE["opencv"]]
    assert [[e[k] for k in ("id", "cascade", "generator", "seed", "height", "width", "mode")] for e in FULL["modes"]] == \
           [list(t) for t in FULLSIZE["modes"]]
    g = FULL["config3"]
    _, a = cascades(g["cascade"])
    f = 5
    img = synth.frame(g["kinds"][f % 3], g["seed0"] + f, g["height"], g["width"])
    r, st = oracle.detect(a, img)
    assert (len(r), rows_sha(r)) == (g["n"][f], g["sha"][f]) and st["stage_entered"] == g["stage_entered_per_frame"][f]
    assert sum(g["n"]) > 1000 and g["stage_entered"][0] == 64 * 6290352
    assert [sum(col) for col in zip(*g["stage_entered_per_frame"])] == g["stage_entered"]
    e = next(e for e in FULL["modes"] if e["id"] == "m4_faces_1080")
    r, st = oracle.detect(a, make_frame(e["generator"], e["seed"], e["height"], e["width"], oracle), mode=4)
    assert (len(r), rows_sha(r)) == (e["n"], e["sha"]) and st["stage_entered"] == e["stage_entered"]
    e3 = next(e for e in FULL["modes"] if e["id"] == "m3_faces_1080")
    assert e3["stage_entered"] != e["stage_entered"]          # the f64 grid is another grid at 1080p
    e = next(e for e in FULL["opencv"] if e["id"] == "cv_alt_faces_1080")
    r, st = oracle.detect_opencvlike(a, make_frame(e["generator"], e["seed"], e["height"], e["width"], oracle))
    r = r[np.lexsort((r["x"], r["y"], r["scale_idx"]))]
    assert (len(r), rows_sha(r)) == (e["n"], e["sha"]) and st["windows"] == e["windows"]
    assert [[e[k] for k in ("id", "cascade", "generator", "seed", "height", "width")] for e in FULL["shipped"]] == [list(t) for t in FULLSIZE["shipped"]]
    e = next(e for e in FULL["shipped"] if e["id"] == "s_lowerbody_smooth_1080")      # 19 x 23 window, tilted nodes read as upright in the clod path
    _, al = cascades(e["cascade"])
    img = make_frame(e["generator"], e["seed"], e["height"], e["width"], oracle)
    r, st = oracle.detect(al, img)
    assert (len(r), rows_sha(r)) == (e["clod"]["n"], e["clod"]["sha"]) and st["stage_entered"] == e["clod"]["stage_entered"]
    r, st = oracle.detect_opencvlike(al, img)
    r = r[np.lexsort((r["x"], r["y"], r["scale_idx"]))]
    assert (len(r), rows_sha(r)) == (e["opencv"]["n"], e["opencv"]["sha"]) and st["windows"] == e["opencv"]["windows"]
    g = FULL["config5_grouped"]
    _, a1 = cascades(g["cascade"])
    _, a2 = cascades(g["second"])
    f = 4          # a frame with drawn faces
    img = synth.frame(g["kinds"][f % 4], g["seed0"] + f, g["height"], g["width"])
    r, _ = oracle.detect(a1, img)
    faces, wt = oracle.group_rectangles(np.stack([r[k] for k in ("x", "y", "w", "h")], 1), g["min_neighbors"])
    rws = [(int(q[0]), int(q[1]), int(q[2]), int(q[3]), int(n)) for q, n in zip(faces, wt)]
    assert (len(r), len(rws), rows_sha(rws)) == (g["n_raw"][f], g["n_faces"][f], g["sha_faces"][f]) and len(rws) > 0
    inside = []
    for i, (x, y, w, h, _) in enumerate(rws):
        r2, _ = oracle.detect(a2, np.ascontiguousarray(img[y:y + h, x:x + w]))
        inside += [(i, int(q["scale_idx"]), int(q["x"]), int(q["y"]), int(q["w"]), int(q["h"])) for q in r2]
    assert (len(inside), rows_sha(inside)) == (g["n_second"][f], g["sha_second"][f])
    g4 = FULL["config4"]
    assert g4["windows"] == 53305712 == g4["stage_entered"][0] and sum(g4["n_per_scale"]) == g4["n"] and g4["n_scales"] == 56
