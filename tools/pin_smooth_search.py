"""Search for the generator details of the survey's second 1080p reference run.

BASELINE.md §2 records the reference's own kernel on a 1920 x 1080 `smooth` frame with frontalface_alt:
294,264,545 stump evaluations (46.78 per window), 2 raw detections.  SURVEY.md §8d defines the kind as
`128 + 60 sin(.05x) cos(.07y) + 40 sin(.013(x+y))` + noise in +-8, clamped, noise from the 32-bit xorshift (13,17,5)
generator; it does not say how a generator word becomes a value in [-8, 8], whether the sines are evaluated in float or
double, how the sum is rounded, nor the seed.  For the noise frame seed 12345 reproduced every recorded figure
(tests/test_oracle_pins.py); this script tries the same seed (fresh, and continued behind earlier frames of a probe run)
with every plausible reading of the rest and prints the variants whose oracle run gives the recorded count.

    python tools/pin_smooth_search.py [--procs 8] [--quick]

CPU only (the oracle, ~1.2 s per variant and core).  Output: one line per variant, hits marked with ***."""
import argparse
import itertools
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TARGET_EVALS, TARGET_DETS = 294264545, 2
H, W = 1080, 1920


def xorshift_words(seed, n, skip=0):
    """n generator words after `skip` earlier draws (vectorised in blocks through the sequential recurrence)."""
    s = np.uint32(seed if seed else 1)
    out = np.empty(n, np.uint32)
    # the recurrence is sequential; run it in C-speed chunks with Python ints on a bytearray would be slow: use numpy scalar loop
    # through a small compiled helper instead
    import ctypes
    lib = _helper()
    lib.xs_fill(ctypes.c_uint32(int(s)), ctypes.c_size_t(skip), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n))
    return out


_HELPER = None


def _helper():
    global _HELPER
    if _HELPER is None:
        import ctypes
        import subprocess
        import tempfile
        d = tempfile.mkdtemp(prefix="xs_")
        src = os.path.join(d, "xs.c")
        open(src, "w").write("""
#include <stdint.h>
#include <stddef.h>
void xs_fill(uint32_t s, size_t skip, uint32_t* dst, size_t n) {
    if (!s) s = 1;
    for (size_t i = 0; i < skip; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; }
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; dst[i] = s; }
}
""")
        so = os.path.join(d, "xs.so")
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so], check=True)
        _HELPER = ctypes.CDLL(so)
    return _HELPER


NOISE_MAPS = {
    "mod17": lambda r: (r % 17).astype(np.int64) - 8,
    "mod17_lowbyte": lambda r: ((r & 0xFF) % 17).astype(np.int64) - 8,
    "mod17_hi": lambda r: ((r >> 8) % 17).astype(np.int64) - 8,
    "mod17_hi16": lambda r: ((r >> 16) % 17).astype(np.int64) - 8,
    "mod16": lambda r: (r % 16).astype(np.int64) - 8,
    "and15": lambda r: (r & 15).astype(np.int64) - 8,
    "mod17_signed": lambda r: (r.astype(np.int32).astype(np.int64) % 17) - 8,          # Python-style modulo of a signed word
    "mod17_ctrunc": lambda r: np.fmod(r.astype(np.int32).astype(np.int64), 17),         # C's % on a signed word: [-16, 16] -> as is
    "mod17_ctrunc8": lambda r: np.clip(np.fmod(r.astype(np.int32).astype(np.int64), 17), -8, 8),
    "byte_scaled": lambda r: ((r & 0xFF).astype(np.int64) * 17 >> 8) - 8,
    "byte_div16": lambda r: ((r & 0xFF).astype(np.int64) >> 4) - 8,
    "float_unit": lambda r: np.rint((r.astype(np.float64) / 4294967296.0) * 16.0 - 8.0).astype(np.int64),
    "float_unit_trunc": lambda r: np.trunc((r.astype(np.float64) / 4294967296.0) * 16.0 - 8.0).astype(np.int64),
    "float_unit17": lambda r: np.floor((r.astype(np.float64) / 4294967296.0) * 17.0).astype(np.int64) - 8,
}


def base_field(prec, order):
    """The smooth field before noise; prec: f64 / f32 (float sines: sinf of float arguments); order: how x, y index the formula."""
    y, x = np.mgrid[0:H, 0:W]
    if order == "yx":      # formula written with x = row, y = column
        x, y = y, x
    if prec == "f64":
        x = x.astype(np.float64); y = y.astype(np.float64)
        return 128.0 + 60.0 * np.sin(0.05 * x) * np.cos(0.07 * y) + 40.0 * np.sin(0.013 * (x + y))
    x = x.astype(np.float32); y = y.astype(np.float32)
    f = np.float32
    return (f(128.0) + f(60.0) * np.sin(f(0.05) * x) * np.cos(f(0.07) * y) + f(40.0) * np.sin(f(0.013) * (x + y))).astype(np.float32)


ROUNDINGS = {
    "trunc_sum": lambda b, nz: np.trunc(b + nz),                 # (int)(base + noise)
    "rint_sum": lambda b, nz: np.rint(b + nz),                   # lrint / nearbyint
    "round_sum": lambda b, nz: np.floor(b + nz + 0.5),           # (int)(v + 0.5)
    "trunc_base": lambda b, nz: np.trunc(b) + nz,                # (int)base + noise
    "rint_base": lambda b, nz: np.rint(b) + nz,
}

_ORACLE = None


def run_variant(v):
    global _ORACLE
    seed, skip, nmap, prec, order, rnd = v
    from oracle.oracle import Oracle, load_vjc
    from clfacedetection_amd.api import DATA_DIR
    if _ORACLE is None:
        _ORACLE = (Oracle(), load_vjc(os.path.join(DATA_DIR, "haarcascade_frontalface_alt.vjc")))
    o, a = _ORACLE
    r = xorshift_words(seed, H * W, skip)
    nz = NOISE_MAPS[nmap](r).reshape(H, W)
    b = base_field(prec, order)
    img = np.clip(ROUNDINGS[rnd](b.astype(np.float64), nz), 0, 255).astype(np.uint8)
    rects, st = o.detect(a, img)
    return v, st["stump_evals"], len(rects)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--quick", action="store_true", help="only the most likely reading of each dimension")
    ap.add_argument("--seeds", default="12345")
    args = ap.parse_args()
    seeds = [int(s) for s in args.seeds.split(",")]
    # generator position: a fresh generator, or one that already produced the probe's earlier frames
    skips = [0, H * W, 640 * 480, 640 * 480 * 2, 640 * 480 + H * W, 640 * 480 * 2 + H * W]
    if args.quick:
        variants = list(itertools.product(seeds, [0], ["mod17", "mod17_lowbyte", "and15"], ["f64"], ["xy"], ["trunc_sum", "rint_sum"]))
    else:
        variants = list(itertools.product(seeds, skips, NOISE_MAPS, ["f64", "f32"], ["xy", "yx"], ROUNDINGS))
    print(f"{len(variants)} variants", flush=True)
    hits = []
    with Pool(args.procs) as pool:
        for v, evals, dets in pool.imap_unordered(run_variant, variants, chunksize=1):
            hit = evals == TARGET_EVALS
            mark = "***" if hit else ("  *" if dets == TARGET_DETS else "   ")
            print(f"{mark} seed {v[0]} skip {v[1]} noise {v[2]} {v[3]} {v[4]} {v[5]}: {evals} evals ({evals - TARGET_EVALS:+d}), {dets} detections", flush=True)
            if hit:
                hits.append((v, evals, dets))
    print("hits:", hits, flush=True)


if __name__ == "__main__":
    main()
