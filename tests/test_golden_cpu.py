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
