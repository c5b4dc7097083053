"""CPU checks of the oracle's OpenCV-like path (oracle/vj_oracle.c: oc_detect_opencvlike, oc_integral_tilted) against
independent numpy restatements of the same reference lines (tempcv.cpp:549-972): the tilted integral by its definition,
and the node-sum arithmetic — binary32 products outside two_rects stump stages, f64 products inside them — on a
hand-built cascade whose single stump sits exactly between the two roundings."""
import numpy as np
import pytest

from cases import crafted_stump_cascade, single_window_frame
from oracle.oracle import CascadeArrays


def test_tilted_integral_matches_its_definition(oracle):
    rng = np.random.default_rng(0)
    for (h, w) in [(1, 1), (2, 3), (5, 4), (9, 13), (17, 6), (23, 31)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        t = oracle.integral_tilted(img)
        ref = np.zeros((h + 1, w + 1), np.uint64)
        for Y in range(h + 1):
            for X in range(w + 1):
                ref[Y, X] = sum(int(img[y, x]) for y in range(Y) for x in range(w) if abs(x - X + 1) <= Y - y - 1)
        assert np.array_equal(t, ref.astype(np.uint32)), (h, w)
    # 32-bit wrap-around like CV_32S: a white 4096-wide strip overflows 2^32 in its lower rows
    img = np.full((3000, 3000), 255, np.uint8)
    t = oracle.integral_tilted(img)
    Y, X = 3000, 1500      # full triangle of height 3000 clipped by the image: count its pixels exactly
    n = sum(min(3000, X - 1 + (Y - y - 1) + 1) - max(0, X - 1 - (Y - y - 1)) for y in range(Y))
    assert int(t[Y, X]) == (255 * n) % (1 << 32)


def numpy_single_window(c: CascadeArrays, img: np.ndarray, factor: float, f64_products: bool):
    """cvSetImagesForHaarClassifierCascade + one window at (0, 0) of a one-stage one-stump cascade, in numpy."""
    cvr = lambda v: int(np.rint(v))                       # cvRound: half to even
    ii = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int64)
    ii[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    qq = np.zeros_like(ii)
    qq[1:, 1:] = (img.astype(np.int64) ** 2).cumsum(0).cumsum(1)
    ex, ew, eh = cvr(factor), cvr((c.win_w - 2) * factor), cvr((c.win_h - 2) * factor)
    ws = 1.0 / (ew * eh)
    box = lambda a, x, y, w, h: int(a[y, x] - a[y, x + w] - a[y + h, x] + a[y + h, x + w])
    mean = box(ii, ex, ex, ew, eh) * ws
    vnf = box(qq, ex, ex, ew, eh) * ws - mean * mean
    vnf = np.sqrt(vnf) if vnf >= 0 else 1.0
    rects = c.node_rect.reshape(-1, 3, 4)[0]
    wts = c.node_weight.reshape(-1, 3)[0]
    nr = 3 if wts[2] != 0 else 2
    tr, w32 = [], []
    sum0 = 0.0
    for k in range(nr):
        x, y, w, h = (cvr(v * factor) for v in rects[k])
        tr.append((x, y, w, h))
        w32.append(np.float32(float(wts[k]) * ws))
        if k == 0:
            area0 = w * h
        else:
            sum0 += float(np.float32(np.float32(w32[k] * np.float32(w)) * np.float32(h)))   # float * int * int in binary32
    w32[0] = np.float32(-sum0 / area0)
    s = 0.0
    for k in range(nr):
        r = box(ii, *tr[k])
        s += float(r) * float(w32[k]) if f64_products else float(np.float32(np.float32(r) * w32[k]))
    return s, float(c.node_threshold[0]) * vnf, s >= float(c.node_threshold[0]) * vnf


@pytest.mark.parametrize("three_rects", [True, False])
def test_node_sum_arithmetic_is_the_scalar_branch(oracle, three_rects):
    """A window whose rectangle sums exceed 2^24: (float)int rounds, the binary32 product rounds again, and the
    cancelling rectangles amplify the difference.  With three rectangles the stage is not two_rects: the reference
    multiplies in binary32 (tempcv.cpp:907-911); with two it multiplies in f64 (:872-888).  The node threshold is put
    between the two candidate sums, so the two formulas disagree about this window."""
    img, factor = single_window_frame(seed=5)
    c0 = crafted_stump_cascade(three_rects, threshold=0.0)
    s32, _, _ = numpy_single_window(c0, img, factor, f64_products=False)
    s64, t0, _ = numpy_single_window(c0, img, factor, f64_products=True)
    assert s32 != s64, "the crafted window does not separate the two formulas"
    _, tv, _ = numpy_single_window(crafted_stump_cascade(three_rects, threshold=1.0), img, factor, False)   # = vnf
    thr = np.float32((s32 + s64) / 2 / tv)
    c = crafted_stump_cascade(three_rects, threshold=float(thr))
    v32 = numpy_single_window(c, img, factor, False)[2]
    v64 = numpy_single_window(c, img, factor, True)[2]
    assert v32 != v64, "threshold resolution too coarse for this window"
    win = int(np.rint(c.win_w * factor))
    ro, st = oracle.detect_opencvlike(c, img, min_size=(win, win))
    assert st["windows"] == 1
    literal = v32 if three_rects else v64          # what the reference's scalar branch computes
    assert (len(ro) == 1) == literal
    ra, _ = oracle.detect_opencvlike(c, img, min_size=(win, win), all_f64=True)   # round 1's formula, for contrast
    assert (len(ra) == 1) == v64
