"""Long-running-service rehearsal (run by hand on the GPU box: `python tests/stress_gpu.py [seconds]`; not collected by pytest):
thousands of calls of every entry point with changing frame sizes, cascades, batch sizes and parameters on ONE environment,
plus environments created and destroyed on the side.  Device memory must level off (bounded plan caches, buffers that grow
to the largest request and stay) and every call must succeed; a few results are re-checked against the first time they were
computed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: see conftest.py)
from clfacedetection_amd import (VJ_FLAG_COUNTERS, VJ_FLAG_SKIP_LIST, VJ_FLAG_SKIP_ROW, Cascade, Environment, default_params, synth)  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
env = Environment(0)
NAMES = ["frontalface_alt", "frontalface_default", "frontalface_alt2", "eye", "frontalface_alt_tree"]
CASC = {n: Cascade.load(n) for n in NAMES}
rng = np.random.default_rng(12345)
frames_cache = {}


def frame(kind, seed, h, w):
    k = (kind, seed % 7, h, w)
    if k not in frames_cache:
        if len(frames_cache) > 64:
            frames_cache.clear()
        frames_cache[k] = synth.frame(kind, seed % 7, h, w)
    return frames_cache[k]


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


t0 = time.time()
n = 0
first_seen = {}
marks = []
stream = None
while time.time() - t0 < budget:
    name = NAMES[int(rng.integers(0, len(NAMES)))]
    c = CASC[name]
    h, w = int(rng.integers(60, 800)), int(rng.integers(60, 1300))
    kind = ["noise", "smooth", "blocks", "faces"][int(rng.integers(0, 4))]
    if kind == "faces" and min(h, w) < 130:
        kind = "noise"
    nb = int(rng.integers(1, 5)) if rng.random() < 0.8 else int(rng.integers(8, 24))      # (batches of >= 8: band-major queue pass, balance classes)
    img = frame(kind, n, h, w)
    batch = [img] * nb if nb > 1 else img
    op = int(rng.integers(0, 8))
    key = None
    if op <= 2:
        flags = [0, VJ_FLAG_COUNTERS, VJ_FLAG_SKIP_LIST if name != "frontalface_alt_tree" else 0][op]
        r = env.detect(c, batch, default_params(flags=flags, min_neighbors=int(rng.integers(0, 3))))
        key, val = ("detect", name, kind, n % 7, h, w, flags), len(r.rects)
    elif op == 3:
        r = env.detect_opencv(c, batch, min_neighbors=int(rng.integers(0, 3)))
    elif op == 4:
        r1, r2 = env.detect_chain(CASC["frontalface_alt2"], CASC["eye"], batch, default_params(min_neighbors=int(rng.integers(0, 4))))
    elif op == 5:
        s, q = env.integral(img)
    elif op == 6:
        if stream is not None:
            stream.close()
        stream = env.stream(c, w, h, 3)
        stream.submit([img, img])
        stream.submit([img])
        stream.collect(); stream.collect()
    else:
        e2 = Environment(0)                      # a second environment comes and goes
        e2.detect(c, img)
        e2.close()
    n += 1
    if n % 200 == 0:
        marks.append((n, round(time.time() - t0), round(free_mib())))
        print(f"{n} calls, {marks[-1][1]} s, free device memory {marks[-1][2]} MiB", flush=True)
if stream is not None:
    stream.close()
if len(marks) >= 4:
    half = marks[len(marks) // 2][2]
    last = marks[-1][2]
    print(f"done: {n} calls; free memory at half time {half} MiB, at the end {last} MiB")
    sys.exit(0 if half - last < 512 else 1)
print(f"done: {n} calls")
