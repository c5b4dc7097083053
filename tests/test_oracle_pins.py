"""Pins the CPU oracle to the REFERENCE.

The reference ships no tests or fixtures and cannot be built in this image, so the only
reference-run outputs available are the figures the survey recorded while driving the
reference's own clod.cpp / clod.cl:
scale counts, candidate-window counts, per-stage survivor counts, stump-evaluation
totals and raw detection counts on xorshift32(13,17,5) noise.  The generator seed is
not written down in the survey; seed 12345 reproduces every recorded figure exactly
(22 per-stage counts, 3 totals, 3 detection counts), which no other seed tried does.
"""
import os

import numpy as np
import pytest

from cases import make_frame

SURVEY_STAGE_ENTERED_1080P_ALT = [
    6290352, 4205943, 2030967, 1412523, 643100, 405745, 235217, 206382, 146688, 70924, 46128,
    22304, 9779, 5930, 3700, 1907, 1036, 561, 319, 175, 90, 50]


@pytest.mark.parametrize("casc,W,H,n_scales,windows", [
    ("frontalface_default", 640, 480, 32, 810001),
    ("frontalface_alt", 640, 480, None, 839321),
    ("frontalface_alt2", 1280, 720, 38, 2700015),
    ("frontalface_alt", 1920, 1080, 42, 6290352),
    ("frontalface_alt_tree", 4096, 4096, 56, 53305712),
])
def test_scale_and_window_counts(oracle, cascades, casc, W, H, n_scales, windows):
    _, a = cascades(casc)
    sc = oracle.plan_scales(a, W, H)
    if n_scales is not None:
        assert len(sc) == n_scales
    assert sum(s.nx * s.ny for s in sc if s.accepted) == windows


def test_demo_min_window(oracle, cascades):
    # 640x480 / default with the demo's 40x40 minimum: 399,792 windows
    _, a = cascades("frontalface_default")
    sc = oracle.plan_scales(a, 640, 480, min_size=(40, 40))
    assert sum(s.nx * s.ny for s in sc if s.accepted) == 399792


def test_per_scale_extremes_1080p(oracle, cascades):
    # "Per-scale at 1080p ranges 503,500 (s=1) -> 38 (s=49.8)"; s up to 49.785
    _, a = cascades("frontalface_alt")
    sc = oracle.plan_scales(a, 1920, 1080)
    assert sc[0].nx * sc[0].ny == 503500 and sc[-1].nx * sc[-1].ny == 38
    assert abs(sc[-1].scale - 49.785) < 1e-3
    _, t = cascades("frontalface_alt_tree")
    assert abs(oracle.plan_scales(t, 4096, 4096)[-1].scale - 189.06) < 1e-2


def test_vga_default_noise(oracle, cascades):
    # 640x480 noise, frontalface_default: 810,001 windows,
    # 23,490,298 stump evaluations (29.0/window), 3 raw detections
    _, a = cascades("frontalface_default")
    r, st = oracle.detect(a, make_frame("xorshift", 12345, 480, 640, oracle))
    assert st["windows"] == 810001 and st["stump_evals"] == 23490298 and len(r) == 3


def test_vga_alt_noise(oracle, cascades):
    # the reference's five CPU variants and its kernel route all
    # returned 2 rects on the 640x480 noise image
    _, a = cascades("frontalface_alt")
    r, st = oracle.detect(a, make_frame("xorshift", 12345, 480, 640, oracle))
    assert st["windows"] == 839321 and len(r) == 2


@pytest.mark.slow
def test_1080p_alt_noise_per_stage_survivors(oracle, cascades):
    # SURVEY §8a-6 / BASELINE.md §2: run of the reference's own runStage kernel
    _, a = cascades("frontalface_alt")
    r, st = oracle.detect(a, make_frame("xorshift", 12345, 1080, 1920, oracle))
    assert st["stage_entered"] == SURVEY_STAGE_ENTERED_1080P_ALT
    assert st["stump_evals"] == 267307785
    assert len(r) == 35
    assert abs(st["stump_evals"] / st["windows"] - 42.49) < 0.01


@pytest.mark.slow
def test_1080p_alt_smooth_is_consistent_with_the_survey_to_12_evaluations(oracle, cascades):
    """BASELINE.md §2's second reference run: 1080p `smooth`, frontalface_alt, 294,264,545 stump evaluations, 2 raw detections.
    The survey does not say how a generator word becomes the noise, in which precision the sines are taken, or how the sum
    is rounded; of 1,717 readings tried (tools/pin_smooth_search.py) the closest — seed 12345, noise = the
    signed word modulo 17 (in [0, 17)) minus 8, the field in double stored to float, (int)(field + noise + 0.5f) — gives the
    2 detections and 294,264,533 evaluations: 12 short (4e-8), one or two pixels of libm's last bit.  NOT a pin: the test
    records how close the oracle comes and that it stays there."""
    _, a = cascades("frontalface_alt")
    H, W = 1080, 1920
    n = (oracle.xorshift_words(12345, H * W).astype(np.int32).astype(np.int64) % 17 - 8).reshape(H, W)
    y, x = np.mgrid[0:H, 0:W]
    field = (128 + 60 * np.sin(.05 * x) * np.cos(.07 * y) + 40 * np.sin(.013 * (x + y))).astype(np.float32)
    img = np.clip((field + n.astype(np.float32) + np.float32(0.5)).astype(np.int64), 0, 255).astype(np.uint8)
    r, st = oracle.detect(a, img)
    assert st["windows"] == 6290352 and len(r) == 2
    assert abs(st["stump_evals"] - 294264545) <= 12
    assert abs(st["stump_evals"] / st["windows"] - 46.78) < 0.005


def test_walk_mode_equals_list_mode(oracle, cascades):
    # per-window stage walk (tempcv.cpp:834-861) == per-stage list compaction
    # (clod.cpp:1271-1302) on a linear cascade
    _, a = cascades("frontalface_alt")
    img = make_frame("noise", 3, 200, 260)
    r0, s0 = oracle.detect(a, img, mode=0)
    r1, s1 = oracle.detect(a, img, mode=1)
    assert (r0 == r1).all() and s0 == s1


def test_u64_to_f32_known_answers(oracle):
    # ties-to-even of the u64 -> f32 conversion used for the squared sum (clod.cpp:432)
    import numpy as np
    assert oracle.u64_to_f32((1 << 24) + 1) == np.float32(16777216.0)        # tie -> even (down)
    assert oracle.u64_to_f32((1 << 24) + 3) == np.float32(16777220.0)        # tie -> even (up)
    assert oracle.u64_to_f32((1 << 40) + (1 << 16)) == np.float32(2.0 ** 40)  # tie -> even
    assert oracle.u64_to_f32((1 << 40) + (1 << 16) + 1) == np.float32(2.0 ** 40 + 2.0 ** 17)


@pytest.mark.skipif(not os.path.isdir("/root/reference/CLFaceDetection"), reason="reference tree not present")
def test_reference_is_not_buildable_here():
    # DESIGN.md claims the reference cannot be compiled in this image: its headers are absent
    import shutil
    import subprocess
    r = subprocess.run([shutil.which("g++") or "g++", "-fsyntax-only", "-x", "c++",
                        "/root/reference/CLFaceDetection/clod.cpp"], capture_output=True, text=True)
    assert r.returncode != 0 and ("opencv2" in r.stderr or "CLEnvironment.h" in r.stderr)


def test_opencvlike_baseline_sanity(oracle, cascades):
    """The OpenCV-style restatement is a timing baseline only (unpinned); sanity: it walks fewer windows than
    the clod grid (ystep >= 2, stage-0 skip), every hit is a legal window, and on the pinned 640x480 frame the
    two arithmetic profiles agree on the two alt detections."""
    from cases import make_frame
    a = cascades("frontalface_alt")[1]
    img = make_frame("xorshift", 12345, 480, 640, oracle)
    r, st = oracle.detect_opencvlike(a, img)
    r2, st2 = oracle.detect(a, img)
    assert 0 < st["windows"] < st2["windows"]
    assert all(0 <= x and x + w < 641 and 0 <= y and y + h < 481 for x, y, w, h in zip(r["x"], r["y"], r["w"], r["h"]))
    assert sorted(zip(r["x"], r["y"], r["w"])) == sorted(zip(r2["x"], r2["y"], r2["w"]))
    # min_size skips whole scales
    r3, st3 = oracle.detect_opencvlike(a, img, min_size=(60, 60))
    assert st3["windows"] < st["windows"] and all(w >= 60 for w in r3["w"])


def test_bgr2gray_known_answers(oracle):
    import numpy as np
    """OpenCV's 8-bit BGR2GRAY (third-party formula, unpinned by the reference): primaries, white, the rounding
    term, alpha ignored; cross-checked against numpy integer arithmetic on random pixels."""
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [1, 1, 1], [4, 0, 0], [5, 0, 0]]], np.uint8)
    assert oracle.bgr2gray(px).tolist() == [[255, 29, 150, 76, 0, 1, 0, 1]]     # 4*1868+8192 = 15664 < 16384 <= 5*1868+8192
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    want = ((img[..., 0].astype(np.uint32) * 1868 + img[..., 1].astype(np.uint32) * 9617 + img[..., 2].astype(np.uint32) * 4899 + 8192) >> 14)
    assert np.array_equal(oracle.bgr2gray(img), want.astype(np.uint8))
    assert np.array_equal(oracle.bgr2gray(img[..., :3]), want.astype(np.uint8))
    g = rng.integers(0, 256, (9, 11), dtype=np.uint8)                          # B = G = R returns the gray value exactly
    assert np.array_equal(oracle.bgr2gray(np.repeat(g[..., None], 3, 2)), g)


def test_block_variant_modes_are_their_own_grids(oracle):
    """Oracle modes 4 / 5 restate clodDetectObjectsBlock (clod.cpp:821-1173), whose `step` is a double (:862).  Where no
    product index * step comes within an f32 rounding of a half they visit the windows of modes 3 / 2; on the crafted
    frame they do not (cases.block_grid_frame), and the rectangles differ by one column."""
    import os
    from cases import BLOCK_GRID_LIMITS, block_grid_frame
    from clfacedetection_amd import synth
    from clfacedetection_amd.api import DATA_DIR
    from oracle.oracle import load_vjc
    a = load_vjc(os.path.join(DATA_DIR, "haarcascade_frontalface_alt.vjc"))
    img = synth.frame("smooth", 3, 240, 320)
    for f32_mode, f64_mode in ((3, 4), (2, 5)):
        r32, s32 = oracle.detect(a, img, mode=f32_mode)
        r64, s64 = oracle.detect(a, img, mode=f64_mode)
        assert np.array_equal(r32, r64) and s32 == s64
    for transposed in (False, True):
        img = block_grid_frame(transposed)
        k = "y" if transposed else "x"
        for f32_mode, f64_mode in ((3, 4), (2, 5)):
            r32, _ = oracle.detect(a, img, mode=f32_mode, **BLOCK_GRID_LIMITS)
            r64, _ = oracle.detect(a, img, mode=f64_mode, **BLOCK_GRID_LIMITS)
            c32, c64 = r32[r32["scale_idx"] == 26][k], r64[r64["scale_idx"] == 26][k]
            assert 656 in c32 and 655 not in c32 and 655 in c64 and 656 not in c64
