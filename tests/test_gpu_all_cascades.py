"""Every cascade the reference ships (19 XML files next to its sources; here as converted .vjc data) through both arithmetic
profiles against the oracle: non-square base windows (45 x 11, 22 x 5, 14 x 28 ...), trees of two and three nodes, tilted
features.  The clod profile takes tilted rectangles as UPRIGHT ones exactly as the reference does (precomputeFeatures never reads
the flag, clod.cpp:448-492) — behind VJ_FLAG_TILTED_AS_UPRIGHT, refused without it; the OpenCV profile evaluates them on the
tilted integral (tempcv.cpp:743-750)."""
import numpy as np
import pytest

from cases import make_frame
from clfacedetection_amd import (VJ_FLAG_COUNTERS, VJ_FLAG_TILTED_AS_UPRIGHT, VjError, clodDetectObjects, default_params)

pytestmark = pytest.mark.gpu

ALL = ["eye", "eye_tree_eyeglasses", "frontalface_alt", "frontalface_alt2", "frontalface_alt_tree", "frontalface_default", "fullbody",
       "lefteye_2splits", "lowerbody", "mcs_eyepair_big", "mcs_eyepair_small", "mcs_lefteye", "mcs_mouth", "mcs_nose", "mcs_righteye",
       "mcs_upperbody", "profileface", "righteye_2splits", "upperbody"]


def as_list(rects):
    return [tuple(int(r[k]) for k in ("scale_idx", "x", "y", "w", "h")) for r in rects]


@pytest.mark.parametrize("name", ALL)
def test_clod_profile_on_every_shipped_cascade(env, oracle, cascades, name):
    c, a = cascades(name)
    k = ALL.index(name)
    h, w = 200 + 13 * k, 260 + 17 * k
    p = default_params(flags=VJ_FLAG_COUNTERS | VJ_FLAG_TILTED_AS_UPRIGHT)
    if c.info.n_tilted:
        with pytest.raises(VjError):                       # without the flag: refused, not silently evaluated
            env.detect(c, make_frame("noise", 1, h, w), default_params())
    for kind in ("noise", "blocks", "smooth"):
        img = make_frame(kind, 7100 + k, h, w)
        r = env.detect(c, img, p)
        ro, st = oracle.detect(a, img)
        assert as_list(r.rects) == as_list(ro), (name, kind)
        assert r.stage_entered == st["stage_entered"] and r.windows == st["windows"], (name, kind)
    # a batch through the chains' balance classes, and the reference's own signature (which passes the flag itself)
    frames = [make_frame("blocks", 7300 + k + i, h, w) for i in range(9)]
    rb = env.detect(c, frames, p)
    for i in (0, 8):
        ro, _ = oracle.detect(a, frames[i])
        assert as_list(rb.rects[rb.rects["frame"] == i]) == as_list(ro), (name, i)
    assert as_list(clodDetectObjects(frames[0], c, env).rects) == as_list(rb.rects[rb.rects["frame"] == 0])


@pytest.mark.parametrize("name", ALL)
def test_opencv_profile_on_every_shipped_cascade(env, oracle, cascades, name):
    c, a = cascades(name)
    k = ALL.index(name)
    h, w = 180 + 11 * k, 250 + 19 * k
    for kind in ("noise", "blocks"):
        img = make_frame(kind, 7500 + k, h, w)
        r = env.detect_opencv(c, img, flags=VJ_FLAG_COUNTERS)
        ro, st = oracle.detect_opencvlike(a, img)
        assert sorted(as_list(r.rects)) == sorted(as_list(ro)), (name, kind)
        assert r.windows == st["windows"] and r.stage_entered == st["stage_entered"], (name, kind)
    frames = np.stack([make_frame("blocks", 7700 + k + i, h, w) for i in range(6)])      # batch: tiles and row kernel side by side
    rb = env.detect_opencv(c, frames)
    for i in (0, 5):
        ro, _ = oracle.detect_opencvlike(a, frames[i])
        assert sorted(as_list(rb.rects[rb.rects["frame"] == i])) == sorted(as_list(ro)), (name, i)
