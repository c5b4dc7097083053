"""Host planning through the C ABI (vj_plan_scales, vj_plan_feature_table,
vj_count_windows) against the oracle and the committed scale tables."""
import json
import os

import numpy as np
import pytest

from clfacedetection_amd import VjError, default_params

G = os.path.join(os.path.dirname(__file__), "golden")
FIELDS = ["scale_idx", "scale", "step", "win_w", "win_h", "equ_x", "equ_y", "equ_w", "equ_h", "area", "nx", "ny",
          "accepted"]
SIZES = [(640, 480), (1920, 1080), (1280, 720), (317, 211), (31, 31), (40, 700), (2048, 64)]


@pytest.mark.parametrize("name", ["frontalface_default", "frontalface_alt", "frontalface_alt2", "eye"])
@pytest.mark.parametrize("W,H", SIZES)
def test_scales_match_oracle(oracle, cascades, name, W, H):
    c, a = cascades(name)
    ps, os_ = c.plan_scales(W, H), oracle.plan_scales(a, W, H)
    assert len(ps) == len(os_)
    for p, q in zip(ps, os_):
        for f in FIELDS:
            assert getattr(p, f) == getattr(q, f), (f, p.scale_idx)
    assert c.count_windows(W, H) == sum(s.nx * s.ny for s in os_ if s.accepted)


@pytest.mark.parametrize("mn,mx", [((40, 40), (0, 0)), ((0, 0), (100, 100)), ((30, 50), (200, 90))])
def test_min_max_window(oracle, cascades, mn, mx):
    c, a = cascades("frontalface_default")
    p = default_params(min_w=mn[0], min_h=mn[1], max_w=mx[0], max_h=mx[1])
    ps, os_ = c.plan_scales(640, 480, p), oracle.plan_scales(a, 640, 480, mn, mx)
    assert [s.accepted for s in ps] == [s.accepted for s in os_]
    assert [(s.nx, s.ny) for s in ps] == [(s.nx, s.ny) for s in os_]


def test_golden_scale_tables(cascades):
    for g in json.load(open(os.path.join(G, "scales.json"))):
        c, _ = cascades(g["cascade"])
        p = default_params(min_w=g["min_size"][0], min_h=g["min_size"][1], max_w=g["max_size"][0],
                           max_h=g["max_size"][1])
        ps = c.plan_scales(g["width"], g["height"], p)
        got = [[s.scale_idx, float(np.float32(s.scale)).hex(), float(np.float32(s.step)).hex(), s.win_w, s.win_h,
                s.equ_x, s.equ_w, s.equ_h, s.area, s.nx, s.ny, s.accepted] for s in ps]
        assert got == g["scales"]
        assert c.count_windows(g["width"], g["height"], p) == g["windows"]


@pytest.mark.parametrize("name,W,H", [("frontalface_alt", 1920, 1080), ("frontalface_default", 640, 480),
                                      ("frontalface_alt2", 1280, 720), ("frontalface_alt_tree", 4096, 4096),
                                      ("eye", 317, 211)])
def test_feature_tables_match_oracle(oracle, cascades, name, W, H):
    c, a = cascades(name)
    for p, q in zip(c.plan_scales(W, H), oracle.plan_scales(a, W, H)):
        if not p.accepted:
            continue
        off, w = c.feature_table(W, p)
        off2, w2 = oracle.feature_table(a, q, W)
        assert np.array_equal(off, off2), p.scale_idx
        assert np.array_equal(w.view(np.uint32), w2.view(np.uint32)), p.scale_idx


def test_scale_chain_is_f32(cascades):
    c, _ = cascades("frontalface_alt")
    s = np.float32(1)
    for sc in c.plan_scales(1920, 1080):
        assert np.float32(sc.scale) == s
        s = np.float32(s * np.float32(1.1))


def test_degenerate_inputs(cascades):
    c, _ = cascades("frontalface_alt")
    assert c.plan_scales(20, 20) == [] and c.count_windows(30, 30) == 0
    with pytest.raises(VjError):
        c.plan_scales(640, 480, default_params(scale_factor=1.0))
    with pytest.raises(VjError):
        c.plan_scales(0, 480)
