"""BASELINE configs 3 and 5 at their FULL sizes (64 x 1080p; 256 x 720p, two cascades) through size-independent
properties, plus the oracle on a bounded sample — the oracle needs ~2 s per 1080p frame, so it cannot check 64 of them
inside the suite's budget."""
import numpy as np
import pytest

from clfacedetection_amd import VJ_FLAG_COUNTERS, DeviceFrames, default_params, synth

pytestmark = pytest.mark.gpu


def rows(rects, frame=None):
    r = rects if frame is None else rects[rects["frame"] == frame]
    return [tuple(int(x[k]) for k in ("scale_idx", "x", "y", "w", "h")) for x in r]


def test_config3_batch_of_64_1080p(env, oracle, cascades):
    import torch
    c, a = cascades("frontalface_alt")
    B, H, W = 64, 1080, 1920
    frames = synth.batch(B, H, W, seed0=1)
    dev = torch.from_numpy(frames).cuda()
    df = DeviceFrames.from_torch(dev)
    full = env.detect(c, df, default_params(flags=VJ_FLAG_COUNTERS))
    # counters: the candidate-window count of the metric, monotone per-stage populations, consistent totals
    assert full.windows == B * 6290352 == full.stage_entered[0]
    assert all(x >= y for x, y in zip(full.stage_entered, full.stage_entered[1:]))
    n_nodes = [int(t["n_trees"]) for t in c.stages]
    assert full.stump_evals == sum(n * k for n, k in zip(full.stage_entered, n_nodes))
    assert list(full.rects["frame"]) == sorted(full.rects["frame"]) and len(full.rects) > 0
    # the timed (uncounted) kernel variants return the same rectangles
    plain = env.detect(c, df)
    assert np.array_equal(plain.rects, full.rects)
    # batch-size independence: eight batches of eight frames, host frames, give the same rectangles
    parts = []
    for k in range(0, B, 8):
        r = env.detect(c, frames[k:k + 8]).rects.copy()
        r["frame"] += k
        parts.append(r)
    assert np.array_equal(np.concatenate(parts), full.rects)
    # scales partition the result (the multi-GPU split of one frame): even + odd scale indices = everything
    ev = env.detect(c, df, default_params(scales=range(0, 42, 2))).rects
    od = env.detect(c, df, default_params(scales=range(1, 42, 2))).rects
    both = np.concatenate([ev, od])
    both = both[np.lexsort((both["x"], both["y"], both["scale_idx"], both["frame"]))]
    assert np.array_equal(both, full.rects)
    # the oracle on a sample of the batch
    for f in (0, 37):
        ro, st = oracle.detect(a, frames[f])
        assert rows(full.rects, f) == rows(ro)


def test_config5_256_frames_two_cascades(env, oracle, cascades):
    import torch
    face, face_a = cascades("frontalface_alt2")
    eye, eye_a = cascades("eye")
    B, H, W = 256, 720, 1280
    frames = synth.batch(B, H, W, seed0=5001)
    dev = torch.from_numpy(frames).cuda()
    df = DeviceFrames.from_torch(dev)
    r1, r2 = env.detect_chain(face, eye, df, default_params(flags=VJ_FLAG_COUNTERS), default_params(flags=VJ_FLAG_COUNTERS))
    assert r1.windows == B * 2700015 and len(r1.rects) > 100
    # first leg = vj_detect; both legs = the same chain on four sub-batches of 64 frames
    assert np.array_equal(env.detect(face, df).rects, r1.rects)
    n1 = n2 = 0
    for k in range(0, B, 64):
        a1, a2 = env.detect_chain(face, eye, frames[k:k + 64])
        sel = r1.rects[(r1.rects["frame"] >= k) & (r1.rects["frame"] < k + 64)]
        b1 = a1.rects.copy()
        b1["frame"] += k
        assert np.array_equal(b1, sel)
        want2 = r2.rects[(r2.rects["frame"] >= n1) & (r2.rects["frame"] < n1 + len(a1.rects))].copy()
        want2["frame"] -= n1
        assert np.array_equal(a2.rects, want2)
        n1 += len(a1.rects)
        n2 += len(a2.rects)
    assert n1 == len(r1.rects) and n2 == len(r2.rects)
    # second leg against the host hand-off (regions through the host, integral images of the sub-images) for the regions
    # of the first 16 frames, and against the oracle on a few sub-images
    sub = r1.rects[r1.rects["frame"] < 16]
    rois = [(int(r["frame"]), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in sub]
    host = env.detect_rois(eye, frames[:16], rois)
    mine = r2.rects[r2.rects["frame"] < len(sub)]
    key = lambda rr: sorted(tuple(int(r[k]) for k in ("frame", "scale_idx", "y", "x", "w", "h")) for r in rr)
    assert key(mine) == key(host.rects)
    for i in range(0, len(rois), max(1, len(rois) // 6)):
        f, x, y, w, h = rois[i]
        ro, _ = oracle.detect(eye_a, np.ascontiguousarray(frames[f][y:y + h, x:x + w]))
        assert rows(r2.rects[r2.rects["frame"] == i]) == rows(ro)
    ro, _ = oracle.detect(face_a, frames[3])
    assert rows(r1.rects, 3) == rows(ro)


def test_config5_256_frames_eyes_inside_grouped_faces(env, oracle, cascades):
    """Config 5 as the reference's caller would run it: faces with minNeighbors 3, eyes inside every FACE.  Grouping on the
    device equals grouping on the host at the full batch size; the chain on sub-batches gives the same; the second leg
    equals the host hand-off and the oracle on sub-images."""
    import torch
    face, face_a = cascades("frontalface_alt2")
    eye, eye_a = cascades("eye")
    B, H, W = 256, 720, 1280
    frames = synth.batch(B, H, W, seed0=5001, kinds=("faces", "noise", "smooth", "blocks"))
    df = DeviceFrames.from_torch(torch.from_numpy(frames).cuda())
    p1 = default_params(min_neighbors=3)
    r1, r2 = env.detect_chain(face, eye, df, p1)
    assert np.array_equal(env.detect(face, df, p1).rects, r1.rects)
    assert len(r1.rects) >= 64 * 3 and len(r2.rects) > 0
    n1 = 0
    for k in range(0, B, 128):
        a1, a2 = env.detect_chain(face, eye, frames[k:k + 128], p1)
        b1 = a1.rects.copy()
        b1["frame"] += k
        assert np.array_equal(b1, r1.rects[(r1.rects["frame"] >= k) & (r1.rects["frame"] < k + 128)])
        want2 = r2.rects[(r2.rects["frame"] >= n1) & (r2.rects["frame"] < n1 + len(a1.rects))].copy()
        want2["frame"] -= n1
        assert np.array_equal(a2.rects, want2)
        n1 += len(a1.rects)
    assert n1 == len(r1.rects)
    sub = r1.rects[r1.rects["frame"] < 24]
    rois = [(int(r["frame"]), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in sub]
    host = env.detect_rois(eye, frames[:24], rois)
    key = lambda rr: [tuple(int(r[k]) for k in ("frame", "scale_idx", "y", "x", "w", "h")) for r in rr]
    assert key(r2.rects[r2.rects["frame"] < len(sub)]) == key(host.rects)
    for i in range(0, len(rois), max(1, len(rois) // 6)):
        f, x, y, w, h = rois[i]
        ro, _ = oracle.detect(eye_a, np.ascontiguousarray(frames[f][y:y + h, x:x + w]))
        assert rows(r2.rects[r2.rects["frame"] == i]) == rows(ro)
    ro, _ = oracle.detect(face_a, frames[4])
    xywh = np.stack([ro[k] for k in ("x", "y", "w", "h")], 1)
    g, w = oracle.group_rectangles(xywh, 3)
    mine = r1.rects[r1.rects["frame"] == 4]
    assert [(int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"]), int(r["weight"])) for r in mine] == \
           [(int(q[0]), int(q[1]), int(q[2]), int(q[3]), int(n)) for q, n in zip(g, w)]
