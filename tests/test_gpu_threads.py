"""One environment per thread (include/vj.h: an environment is not thread-safe, different environments are independent):
four threads with their own environments on the same device run detections, chains and the OpenCV profile concurrently;
every result equals the oracle's."""
import os
import threading

import numpy as np
import pytest

from clfacedetection_amd import Cascade, Environment, default_params, synth

pytestmark = pytest.mark.gpu
NAMES = ["frontalface_alt", "frontalface_alt2", "eye", "frontalface_alt_tree"]


def rows(rects):
    return [tuple(int(r[k]) for k in ("scale_idx", "x", "y", "w", "h")) for r in rects]


def test_environments_on_four_threads(oracle, cascades):
    imgs, want, want_cv = {}, {}, {}
    for i, n in enumerate(NAMES):
        _, a = cascades(n)
        imgs[n] = synth.frame(["noise", "blocks", "faces", "smooth"][i], 40 + i, 300 + 17 * i, 400 + 31 * i)
        want[n] = rows(oracle.detect(a, imgs[n])[0])
        want_cv[n] = sorted(rows(oracle.detect_opencvlike(a, imgs[n])[0]))
    errors = []

    def worker(tid):
        try:
            env = Environment(0)
            cs = {n: Cascade.load(n) for n in NAMES}
            for it in range(60):
                n = NAMES[(it + tid) % len(NAMES)]
                r = env.detect(cs[n], imgs[n] if it % 3 else [imgs[n], imgs[n]])
                if rows(r.rects[r.rects["frame"] == 0]) != want[n]:
                    errors.append((tid, it, n))
                if it % 10 == tid:
                    env.detect_chain(cs["frontalface_alt2"], cs["eye"], imgs["eye"], default_params(min_neighbors=it % 3))
                    if sorted(rows(env.detect_opencv(cs[n], imgs[n]).rects)) != want_cv[n]:
                        errors.append((tid, it, n, "opencv"))
            env.close()
        except Exception as e:   # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
