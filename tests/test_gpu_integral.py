"""HIP integral / squared-integral kernels through the C ABI (vj_integral) vs the oracle
and the committed fixtures.  Exact integers: every element must match."""
import json
import os

import numpy as np
import pytest

from cases import INTEGRAL_CASES, make_frame, sha

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ROWS_DEFAULT = 2


@pytest.fixture(params=[0, 1, 2], autouse=True, ids=lambda m: f"rows{m}")
def rows_mode(env, request):
    """Every test of this file with each form of the row kernel: one wave walking a band's chunks (0), the chunks of a band side
    by side in one workgroup (1), and the default's choice between them by call size (2)."""
    env.configure("integral_rows", request.param)
    yield request.param
    env.configure("integral_rows", ROWS_DEFAULT)


@pytest.mark.parametrize("g", json.load(open(os.path.join(G, "integral.json"))), ids=lambda d: d["id"])
def test_integral_fixture(env, oracle, g):
    img = make_frame(g["generator"], g["seed"], g["height"], g["width"], oracle)
    s, q = env.integral(img)
    assert sha(s) == g["sum_sha256"] and sha(q) == g["sqsum_sha256"]
    so, qo = oracle.integral(img)
    assert np.array_equal(s, so) and np.array_equal(q, qo)


@pytest.mark.parametrize("h,w", [(1, 1), (1, 300), (300, 1), (2, 2), (7, 5), (8, 255), (9, 256), (16, 257), (17, 1023),
                                 (33, 1025), (100, 2049), (719, 1279), (720, 1280)])
def test_integral_shapes(env, oracle, h, w):
    img = make_frame("noise", h * 131 + w, h, w)
    s, q = env.integral(img)
    so, qo = oracle.integral(img)
    assert np.array_equal(s, so) and np.array_equal(q, qo)


def test_integral_very_wide_rows(env, oracle):
    """Rows of more than 66051 pixels: a row's prefix of squares no longer fits 32 bits (band_rows<uint64_t>); bright
    pixels so that it really overflows, and a height that is not a multiple of the band."""
    img = np.full((11, 70001), 255, np.uint8)
    img[::3, ::7] = make_frame("noise", 5, 11, 70001)[::3, ::7]
    s, q = env.integral(img)
    so, qo = oracle.integral(img)
    assert int(qo[2, -1]) - int(qo[1, -1]) > 2**32 and np.array_equal(s, so) and np.array_equal(q, qo)


def test_integral_strided_rows(env, oracle):
    big = make_frame("noise", 77, 200, 400)
    view = big[10:150, 37:300]           # row stride 400, unaligned start
    s, q = env.integral(view)
    so, qo = oracle.integral(np.ascontiguousarray(view))
    assert np.array_equal(s, so) and np.array_equal(q, qo)


def test_integral_4096_wraps_and_totals(env):
    """BASELINE config 4 size: size-independent properties instead of a full compare."""
    img = make_frame("noise", 404, 4096, 4096)
    s, q = env.integral(img)
    a = img.astype(np.uint64)
    assert int(s[-1, -1]) == int(a.sum() & 0xFFFFFFFF) and int(q[-1, -1]) == int((a * a).sum())
    # row / column marginals
    assert np.array_equal(s[-1, 1:], (np.cumsum(a.sum(0)) & 0xFFFFFFFF).astype(np.uint32))
    assert np.array_equal(q[1:, -1], np.cumsum((a * a).sum(1)))
    # 2-D second difference recovers the image (checks every element)
    d = s[1:, 1:].astype(np.int64) - s[:-1, 1:] - s[1:, :-1] + s[:-1, :-1]
    assert np.array_equal(d & 0xFFFFFFFF, a)
    white = np.full((4200, 4200), 255, np.uint8)   # > 2^32: the sum wraps like CV_32S
    s, q = env.integral(white)
    assert int(s[-1, -1]) == (255 * 4200 * 4200) % (1 << 32) and int(q[-1, -1]) == 255 * 255 * 4200 * 4200


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,ch", [(1, 1, 3), (7, 5, 4), (33, 130, 3), (64, 257, 4), (480, 641, 3), (720, 1280, 4)])
def test_color_ingest_integral(env, oracle, h, w, ch):
    """clifGrayscaleIntegral: BGR / BGRA converted inside the integral kernels == oracle gray -> oracle integral."""
    rng = np.random.default_rng(h * 1000 + w + ch)
    img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    s, q = env.integral(img)
    so, qo = oracle.integral(oracle.bgr2gray(img))
    assert np.array_equal(s, so) and np.array_equal(q, qo)
    # a strided ROI view (rows not 4-byte aligned) takes the byte-load path
    if h > 8 and w > 8:
        roi = img[3:h - 2, 1:w - 3]
        s, q = env.integral(roi)
        so, qo = oracle.integral(oracle.bgr2gray(roi))
        assert np.array_equal(s, so) and np.array_equal(q, qo)
