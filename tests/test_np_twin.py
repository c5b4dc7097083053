"""The C oracle against its independent numpy twin (oracle/np_oracle.py) — for the tree
cascades (frontalface_alt2: 2-node trees; frontalface_alt_tree: stage tree) this mutual
check is all the pinning there is."""
import numpy as np
import pytest

from cases import make_frame
from oracle import np_oracle


@pytest.mark.parametrize("casc,gen,seed,h,w,mn,mx,sm", [
    ("frontalface_alt", "noise", 7, 120, 160, (0, 0), (0, 0), False),
    ("frontalface_alt", "blocks", 8, 97, 131, (0, 0), (0, 0), True),
    ("frontalface_default", "smooth", 9, 110, 150, (30, 30), (80, 80), False),
    ("eye", "noise", 10, 90, 120, (0, 0), (0, 0), False),
    ("frontalface_alt2", "noise", 11, 120, 160, (0, 0), (0, 0), False),
    ("frontalface_alt2", "blocks", 12, 100, 140, (0, 0), (0, 0), False),
    ("frontalface_alt_tree", "noise", 13, 120, 160, (0, 0), (0, 0), False),
    ("frontalface_alt_tree", "smooth", 14, 100, 130, (0, 0), (0, 0), False),
])
def test_c_oracle_equals_numpy_twin(oracle, cascades, casc, gen, seed, h, w, mn, mx, sm):
    _, a = cascades(casc)
    img = make_frame(gen, seed, h, w, oracle)
    r, st = oracle.detect(a, img, min_size=mn, max_size=mx, signed_mean=sm)
    dets, entered = np_oracle.detect(a, img, mn, mx, 1.1, sm)
    assert [tuple(int(q[k]) for k in ("scale_idx", "x", "y", "w", "h")) for q in r] == dets
    assert st["stage_entered"] == entered


def test_integral_twin(oracle):
    img = make_frame("noise", 3, 77, 91)
    s, q = oracle.integral(img)
    s2, q2 = np_oracle.integral(img)
    assert np.array_equal(s, s2) and np.array_equal(q, q2)


def test_scales_twin(oracle, cascades):
    _, a = cascades("frontalface_default")
    for (W, H, mn, mx) in [(640, 480, (0, 0), (0, 0)), (317, 211, (30, 30), (100, 100)), (1920, 1080, (0, 0), (0, 0))]:
        cs = oracle.plan_scales(a, W, H, mn, mx)
        ns = np_oracle.scales(a, W, H, mn, mx)
        assert len(cs) == len(ns)
        for p, q in zip(cs, ns):
            assert bool(p.accepted) == q["accepted"] and np.float32(p.scale) == q["scale"]
            if p.accepted:
                assert (p.win_w, p.win_h, p.equ_x, p.equ_w, p.equ_h, p.area, p.nx, p.ny) == \
                    (q["win_w"], q["win_h"], q["equ_x"], q["equ_w"], q["equ_h"], q["area"], q["nx"], q["ny"])
