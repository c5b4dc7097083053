"""Grouping / NMS: the product's host implementation
(vj_group_rectangles, csrc/vj_group.cpp) against the oracle's restatement of
cv::groupRectangles + cv::partition (tempcv.cpp:130-243).  Parity with the reference is
UNPINNED for this row: the reference's own port (clod.cpp:182-357) is buggy and never
executed by its demo."""
import numpy as np
import pytest

from clfacedetection_amd import group_rectangles
from clfacedetection_amd.api import RECT_DTYPE


def make(rows, frame=0):
    a = np.zeros(len(rows), RECT_DTYPE)
    for i, (x, y, w, h) in enumerate(rows):
        a[i] = (x, y, w, h, 0.0, frame, 0)
    return a


def clusters(rng, n_clusters, per, jitter, size=(20, 200)):
    rows = []
    for _ in range(n_clusters):
        s = int(rng.integers(*size))
        x, y = int(rng.integers(0, 1000)), int(rng.integers(0, 600))
        for _ in range(int(rng.integers(1, per + 1))):
            j = rng.integers(-jitter, jitter + 1, 4)
            rows.append((x + int(j[0]), y + int(j[1]), max(1, s + int(j[2])), max(1, s + int(j[3]))))
    order = rng.permutation(len(rows))
    return [rows[i] for i in order]


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("thr", [1, 2, 3])
def test_matches_oracle_on_random_clusters(oracle, seed, thr):
    rng = np.random.default_rng(seed)
    rows = clusters(rng, int(rng.integers(1, 12)), 9, int(rng.integers(0, 6)))
    got = group_rectangles(make(rows), thr)
    want, w = oracle.group_rectangles(np.array(rows, np.int32), thr)
    assert np.array_equal(np.stack([got[k] for k in "xywh"], 1), want)
    assert got["weight"].astype(np.int32).tolist() == w.tolist()
    assert (got["scale_idx"] == -1).all()


def test_known_small_cases(oracle):
    # three near-identical rects + one outlier, threshold 2: one group of weight 3 (truncated average)
    rows = [(10, 10, 50, 50), (12, 11, 50, 50), (11, 12, 51, 49), (300, 300, 40, 40)]
    got = group_rectangles(make(rows), 2)
    assert [(int(g["x"]), int(g["y"]), int(g["w"]), int(g["h"]), int(g["weight"])) for g in got] == [(11, 11, 50, 49, 3)]
    # threshold 0 keeps everything with weight 1 (tempcv.cpp:147-157)
    got = group_rectangles(make(rows), 0)
    assert len(got) == 4 and (got["weight"] == 1).all()
    # small rectangle inside a larger, better supported one is suppressed (tempcv.cpp:214-231)
    big = [(100, 100, 100, 100)] * 5
    small = [(130, 130, 30, 30)] * 2
    got = group_rectangles(make(big + small), 1)
    assert [(int(g["x"]), int(g["w"]), int(g["weight"])) for g in got] == [(100, 100, 5)]
    assert len(group_rectangles(make([]), 3)) == 0


def test_grouping_is_per_frame(oracle):
    rows = [(10, 10, 50, 50), (11, 10, 50, 50), (10, 11, 50, 50)]
    a = np.concatenate([make(rows, 0), make(rows, 1), make(rows[:1], 2)])
    got = group_rectangles(a, 2)
    assert got["frame"].tolist() == [0, 1] and got["weight"].tolist() == [3.0, 3.0]


def test_chained_similarity_merges_transitively(oracle):
    # a chain where only neighbours are similar must still form one class (union-find semantics)
    rows = [(100 + 4 * i, 100, 60, 60) for i in range(10)]
    got = group_rectangles(make(rows), 1)
    want, w = oracle.group_rectangles(np.array(rows, np.int32), 1)
    assert len(got) == len(want) == 1 and int(got["weight"][0]) == 10
