"""Seeded parity cases shared by the golden generator (tools/make_golden.py), the CPU
tests (oracle vs fixtures) and the GPU tests (HIP path vs oracle and vs fixtures)."""
from __future__ import annotations

import hashlib

import numpy as np

from clfacedetection_amd import synth

# (id, cascade, generator, seed, height, width, min_size, max_size, signed_mean)
DETECT_CASES = [
    ("alt_xs12345_480", "frontalface_alt", "xorshift", 12345, 480, 640, (0, 0), (0, 0), False),
    ("default_xs12345_480", "frontalface_default", "xorshift", 12345, 480, 640, (0, 0), (0, 0), False),
    ("alt_noise_240", "frontalface_alt", "noise", 7, 240, 320, (0, 0), (0, 0), False),
    ("alt_smooth_480", "frontalface_alt", "smooth", 11, 480, 640, (0, 0), (0, 0), False),
    ("alt_blocks_odd", "frontalface_alt", "blocks", 5, 251, 333, (0, 0), (0, 0), False),
    ("default_min40", "frontalface_default", "noise", 21, 480, 640, (40, 40), (0, 0), False),
    ("default_minmax", "frontalface_default", "smooth", 22, 360, 480, (30, 30), (120, 120), False),
    ("eye_noise_300", "eye", "noise", 31, 300, 400, (0, 0), (0, 0), False),
    ("eye_blocks_200", "eye", "blocks", 32, 200, 200, (0, 0), (0, 0), False),
    ("alt_signed_mean", "frontalface_alt", "noise", 41, 240, 320, (0, 0), (0, 0), True),
    ("alt_tiny", "frontalface_alt", "noise", 51, 31, 31, (0, 0), (0, 0), False),
    ("alt_wide", "frontalface_alt", "noise", 52, 40, 700, (0, 0), (0, 0), False),
    ("alt2_noise_240", "frontalface_alt2", "noise", 61, 240, 320, (0, 0), (0, 0), False),
    ("alt2_smooth_300", "frontalface_alt2", "smooth", 62, 300, 400, (0, 0), (0, 0), False),
    ("alt_tree_noise_240", "frontalface_alt_tree", "noise", 71, 240, 320, (0, 0), (0, 0), False),
    ("alt_tree_blocks_300", "frontalface_alt_tree", "blocks", 72, 300, 400, (0, 0), (0, 0), False),
]
# the headline pin: the survey's recorded reference run
HEADLINE_CASE = ("alt_xs12345_1080", "frontalface_alt", "xorshift", 12345, 1080, 1920, (0, 0), (0, 0), False)

INTEGRAL_CASES = [  # (id, generator, seed, height, width)
    ("i_1x1", "noise", 1, 1, 1), ("i_3x5", "noise", 2, 3, 5), ("i_8x256", "noise", 3, 8, 256),
    ("i_9x257", "noise", 4, 9, 257), ("i_odd", "smooth", 5, 251, 333), ("i_vga", "noise", 6, 480, 640),
    ("i_white", "white", 0, 64, 300), ("i_1080p", "noise", 7, 1080, 1920),
]


def make_frame(generator: str, seed: int, h: int, w: int, oracle=None) -> np.ndarray:
    if generator == "xorshift":   # the survey's sequential generator lives in the oracle
        return oracle.xorshift_noise(seed, h, w)
    if generator == "white":
        return np.full((h, w), 255, np.uint8)
    return synth.frame(generator, seed, h, w)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
