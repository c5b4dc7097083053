"""Synthetic 8-bit frames for tests and bench (numpy only, deterministic everywhere).

Pixels come from a counter-based generator — a xorshift-multiply hash of
(seed, pixel index) — so any frame can be produced vectorised, on any machine, from
its (kind, seed, size) alone; nothing is stored.  Kinds:
  noise   uniform bytes
  smooth  128 + 60 sin(.05x) cos(.07y) + 40 sin(.013(x+y)) + noise in [-8, 8], clamped
  blocks  piecewise-constant 16x16 blocks of uniform bytes + noise in [-8, 8]
           (large flat regions: variance ~0, exercises the sqrt / "variance = 1" branch)
  faces   the smooth background with a few crude frontal faces drawn on it (dark hair and eye band, darker eyes,
           bright nose bridge and cheeks, dark mouth) at hashed positions and sizes: the frontal-face cascades answer
           each with a CLUSTER of overlapping candidates — what grouping (min_neighbors) and the face -> eye chain need
"""
from __future__ import annotations

import numpy as np


def _hash_u32(seed: int, n: int) -> np.ndarray:
    x = np.arange(n, dtype=np.uint64)
    x = (x + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def frame(kind: str, seed: int, height: int, width: int) -> np.ndarray:
    n = height * width
    r = _hash_u32(seed, n)
    if kind == "noise":
        return (r & np.uint32(0xFF)).astype(np.uint8).reshape(height, width)
    jitter = ((r >> np.uint32(8)) % np.uint32(17)).astype(np.int32).reshape(height, width) - 8
    if kind == "smooth":
        y, x = np.mgrid[0:height, 0:width].astype(np.float64)
        base = 128.0 + 60.0 * np.sin(0.05 * x) * np.cos(0.07 * y) + 40.0 * np.sin(0.013 * (x + y))
        return np.clip(np.rint(base).astype(np.int32) + jitter, 0, 255).astype(np.uint8)
    if kind == "blocks":
        by, bx = (height + 15) // 16, (width + 15) // 16
        b = (_hash_u32(seed ^ 0x5BD1E995, by * bx) & np.uint32(0xFF)).astype(np.int32).reshape(by, bx)
        base = np.kron(b, np.ones((16, 16), np.int32))[:height, :width]
        return np.clip(base + jitter, 0, 255).astype(np.uint8)
    if kind == "faces":
        img = frame("smooth", seed, height, width).astype(np.int32)
        h = _hash_u32(seed ^ 0x2545F491, 64)
        lo, hi = 40, max(41, min(height, width) // 3)
        for k in range(3 + int(h[0] % 6)):
            s = lo + int(h[1 + 3 * k] % (hi - lo))
            if s + 2 > height or s + 2 > width:
                continue
            x0, y0 = int(h[2 + 3 * k] % (width - s)), int(h[3 + 3 * k] % (height - s))
            img[y0:y0 + s, x0:x0 + s] = crude_face(s) + jitter[y0:y0 + s, x0:x0 + s] // 2
        return np.clip(img, 0, 255).astype(np.uint8)
    raise ValueError(f"unknown kind {kind!r}")


def crude_face(s: int) -> np.ndarray:
    """s x s gray levels of a schematic frontal face (int32); comparisons of exact quotients only."""
    f = np.full((s, s), 200, np.int32)
    yy, xx = np.mgrid[0:s, 0:s] / float(s)
    f[yy < 0.18] = 90                                                              # hair
    f[(yy > 0.28) & (yy < 0.42)] = 120                                             # eye band
    for cx in (0.3, 0.7):
        f[((xx - cx) / 0.11) ** 2 + ((yy - 0.35) / 0.06) ** 2 < 1] = 40            # eyes
    f[(yy > 0.28) & (yy < 0.62) & (abs(xx - 0.5) < 0.07)] = 225                    # nose bridge
    f[(yy > 0.42) & (yy < 0.62) & (abs(xx - 0.5) > 0.12) & (abs(xx - 0.5) < 0.38)] = 215   # cheeks
    f[((xx - 0.5) / 0.2) ** 2 + ((yy - 0.76) / 0.05) ** 2 < 1] = 80                # mouth
    f[(xx < 0.08) | (xx > 0.92)] = 100
    f[yy > 0.92] = 110
    return f


def batch(n: int, height: int, width: int, seed0: int = 1, kinds=("noise", "smooth", "blocks")) -> np.ndarray:
    """n frames, seeds seed0.. and kinds cycling — the mix BASELINE.json's config 3 names."""
    out = np.empty((n, height, width), np.uint8)
    for i in range(n):
        out[i] = frame(kinds[i % len(kinds)], seed0 + i, height, width)
    return out
