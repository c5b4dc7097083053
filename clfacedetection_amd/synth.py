"""Synthetic 8-bit frames for tests and bench (numpy only, deterministic everywhere).

Pixels come from a counter-based generator — a xorshift-multiply hash of
(seed, pixel index) — so any frame can be produced vectorised, on any machine, from
its (kind, seed, size) alone; nothing is stored.  Kinds:
  noise   uniform bytes
  smooth  128 + 60 sin(.05x) cos(.07y) + 40 sin(.013(x+y)) + noise in [-8, 8], clamped
  blocks  piecewise-constant 16x16 blocks of uniform bytes + noise in [-8, 8]
           (large flat regions: variance ~0, exercises the sqrt / "variance = 1" branch)
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
    raise ValueError(f"unknown kind {kind!r}")


def batch(n: int, height: int, width: int, seed0: int = 1, kinds=("noise", "smooth", "blocks")) -> np.ndarray:
    """n frames, seeds seed0.. and kinds cycling — the mix BASELINE.json's config 3 names."""
    out = np.empty((n, height, width), np.uint8)
    for i in range(n):
        out[i] = frame(kinds[i % len(kinds)], seed0 + i, height, width)
    return out
