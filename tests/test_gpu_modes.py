"""P2 skip modes, the device-side two-cascade chain (§8f-4), frame streams (§8f-3) and the smaller
boundary additions, each against the oracle or against the plain vj_detect path."""
import numpy as np
import pytest

from cases import make_frame
from clfacedetection_amd import (CLOD_PER_STAGE_ITERATIONS, CLOD_PRECOMPUTE_FEATURES, VJ_FLAG_COUNTERS, VJ_FLAG_SKIP_LIST,
                                 VJ_FLAG_SKIP_ROW, VjError, clodDetectObjects, default_params, synth)

pytestmark = pytest.mark.gpu


def rows(rects):
    return [tuple(int(r[k]) for k in ("scale_idx", "x", "y", "w", "h")) for r in rects]


# ----------------------------------------------------------------------------------------------- P2: skip modes
@pytest.mark.parametrize("casc,kind,seed,h,w", [
    ("frontalface_alt", "xorshift", 12345, 480, 640),
    ("frontalface_alt", "smooth", 11, 480, 640),          # long stage-0 reject runs
    ("frontalface_alt", "blocks", 5, 251, 333),
    ("frontalface_default", "noise", 21, 360, 500),
    ("eye", "blocks", 32, 200, 260),
    ("frontalface_alt2", "noise", 61, 240, 320),          # two-node trees, linear stages
    ("frontalface_alt", "noise", 52, 40, 700),            # rows of > 64 windows, few rows
])
@pytest.mark.parametrize("flag,mode", [(VJ_FLAG_SKIP_LIST, 2), (VJ_FLAG_SKIP_ROW, 3)])
def test_skip_modes_match_the_cpu_variants(env, oracle, cascades, casc, kind, seed, h, w, flag, mode):
    """VJ_FLAG_SKIP_LIST = the per-stage-list CPU variant (clod.cpp:1434-1482, skip over the flattened list :729-732);
    VJ_FLAG_SKIP_ROW = the plain CPU loop (clod.cpp:1409-1432: round() positions, skip inside a row :1430).  Rectangles
    and the per-stage counts of the windows actually evaluated equal the oracle's restatement of those loops."""
    c, a = cascades(casc)
    img = make_frame(kind, seed, h, w, oracle)
    r = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS | flag))
    ro, st = oracle.detect(a, img, mode=mode)
    assert rows(r.rects) == rows(ro)
    assert r.stage_entered == st["stage_entered"] and r.windows == st["windows"]
    p1 = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS))
    assert r.stage_entered[0] <= p1.stage_entered[0]
    if flag == VJ_FLAG_SKIP_LIST:          # same grid positions as the kernel contract: P2 is a subset of P1
        assert set(rows(r.rects)) <= set(rows(p1.rects))


@pytest.mark.parametrize("seed", range(10))
def test_skip_modes_randomized(env, oracle, cascades, seed):
    rng = np.random.default_rng(3000 + seed)
    casc = ["frontalface_alt", "frontalface_default", "eye", "frontalface_alt2"][seed % 4]
    c, a = cascades(casc)
    w = int(rng.integers(c.info.win_w + 12, 900))
    h = int(rng.integers(c.info.win_h + 12, 600))
    img = make_frame(["noise", "smooth", "blocks"][seed % 3], 7000 + seed, h, w, oracle)
    mn = (0, 0) if seed % 2 else (int(rng.integers(20, 60)),) * 2
    for flag, mode in ((VJ_FLAG_SKIP_LIST, 2), (VJ_FLAG_SKIP_ROW, 3)):
        r = env.detect(c, img, default_params(flags=VJ_FLAG_COUNTERS | flag, min_w=mn[0], min_h=mn[1]))
        ro, st = oracle.detect(a, img, min_size=mn, mode=mode)
        assert rows(r.rects) == rows(ro), (casc, w, h, mn, mode)
        assert r.stage_entered == st["stage_entered"]


def test_skip_modes_batches_tunables_and_refusals(env, oracle, cascades):
    c, a = cascades("frontalface_alt")
    frames = synth.batch(5, 300, 420, seed0=40)
    for flag, mode in ((VJ_FLAG_SKIP_LIST, 2), (VJ_FLAG_SKIP_ROW, 3)):
        r = env.detect(c, frames, default_params(flags=flag))
        for f in range(len(frames)):
            ro, _ = oracle.detect(a, frames[f], mode=mode)
            assert rows(r.rects[r.rects["frame"] == f]) == rows(ro)
        # every window goes through the global-gather chain, or the tiles take all they can: same result
        for key, val, back in (("tile_classes_kb", "0,0,0", "-2,-1,0"), ("tile_split", "0", "0.5"), ("concurrent", "0", "1")):
            env.configure(key, val)
            try:
                assert np.array_equal(env.detect(c, frames, default_params(flags=flag)).rects, r.rects), (key, mode)
            finally:
                env.configure(key, back)
    # the reference-named entry point: use_opencl = False selects the CPU variants' window sets
    img = frames[0]
    plain = clodDetectObjects(img, c, env, flags=CLOD_PRECOMPUTE_FEATURES, use_opencl=False)
    assert rows(plain.rects) == rows(oracle.detect(a, img, mode=3)[0])
    per_stage = clodDetectObjects(img, c, env, flags=CLOD_PRECOMPUTE_FEATURES | CLOD_PER_STAGE_ITERATIONS, use_opencl=False)
    assert rows(per_stage.rects) == rows(oracle.detect(a, img, mode=2)[0])
    with pytest.raises(VjError):
        env.detect(c, img, default_params(flags=VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW))
    tree, _ = cascades("frontalface_alt_tree")
    with pytest.raises(VjError):
        env.detect(tree, img, default_params(flags=VJ_FLAG_SKIP_ROW))


# ----------------------------------------------------------------------------------------------- two cascades, hand-off on the device
def test_chain_equals_the_host_hand_off_and_the_oracle(env, oracle, cascades):
    """vj_detect_chain: faces (frontalface_alt2) -> eyes (haarcascade_eye) inside every raw face candidate, regions built and
    consumed on the device.  Equal to vj_detect followed by vj_detect_rois (regions through the host, integrals of the
    sub-images), and to the oracle run on each sub-image."""
    face, _ = cascades("frontalface_alt2")
    eye, eye_a = cascades("eye")
    frames = synth.batch(4, 720, 1280, seed0=777, kinds=("noise", "blocks"))
    p2 = default_params(flags=VJ_FLAG_COUNTERS)
    r1, r2 = env.detect_chain(face, eye, frames, default_params(), p2)
    base = env.detect(face, frames)
    assert np.array_equal(r1.rects, base.rects)
    assert len(r1.rects) > 0, "the synthetic frames must produce face candidates"
    rois = [(int(r["frame"]), int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in r1.rects]
    host = env.detect_rois(eye, frames, rois, p2)
    key = lambda rr: [tuple(int(r[k]) for k in ("frame", "scale_idx", "y", "x", "w", "h")) for r in rr]
    assert sorted(key(r2.rects)) == sorted(key(host.rects))
    assert key(r2.rects) == sorted(key(r2.rects))
    assert r2.stage_entered == host.stage_entered and r2.windows == host.windows
    for i in list(range(0, len(rois), max(1, len(rois) // 12)))[:12]:      # the oracle on the sub-image itself
        f, x, y, w, h = rois[i]
        ro, _ = oracle.detect(eye_a, np.ascontiguousarray(frames[f][y:y + h, x:x + w]))
        mine = r2.rects[r2.rects["frame"] == i]
        assert rows(mine) == rows(ro), rois[i]


def test_chain_overflow_paths_and_refusals(env, oracle, cascades):
    """Small detection / unit / region-detection buffers make the chain grow them and run again; results do not change."""
    face, _ = cascades("frontalface_alt2")
    eye, _ = cascades("eye")
    frames = synth.batch(2, 720, 1280, seed0=777, kinds=("noise", "blocks"))
    ref1, ref2 = env.detect_chain(face, eye, frames)
    env.configure("det_cap", 4)
    try:
        a1, a2 = env.detect_chain(face, eye, frames)
    finally:
        env.configure("det_cap", 65536)
    assert np.array_equal(a1.rects, ref1.rects) and np.array_equal(a2.rects, ref2.rects)
    tree, _ = cascades("frontalface_alt_tree")
    with pytest.raises(VjError):
        env.detect_chain(face, tree, frames)                        # second cascade must be linear
    with pytest.raises(VjError):
        env.detect_chain(face, eye, frames, default_params(min_neighbors=2))


# ----------------------------------------------------------------------------------------------- frame streams
def test_stream_results_equal_detect(env, cascades):
    """vj_stream: batches uploaded on a copy stream into one of two device buffers while the previous batch's kernels
    run.  Pageable and page-locked host frames, partial batches, BGR frames; results equal vj_detect's."""
    c, _ = cascades("frontalface_alt")
    H, W, B = 270, 480, 4
    batches = [synth.batch(B, H, W, seed0=700 + 10 * k) for k in range(5)]
    want = [env.detect(c, b, default_params(flags=VJ_FLAG_COUNTERS)) for b in batches]
    st = env.stream(c, W, H, B, default_params(flags=VJ_FLAG_COUNTERS))
    pinned = env.host_alloc((B, H, W))
    try:
        got = []
        st.submit(batches[0])                       # pageable: goes through the stream's pinned staging buffer
        for k in range(1, 5):
            if k % 2:
                pinned[...] = batches[k]
                st.submit(pinned)                   # page-locked: DMA straight from the caller's buffer
            else:
                st.submit(batches[k][:3])           # a partial batch
            got.append(st.collect())
        got.append(st.collect())
        with pytest.raises(VjError):
            st.collect()                            # nothing pending
        for k in range(5):
            w = want[k] if (k % 2 or k == 0) else env.detect(c, batches[k][:3], default_params(flags=VJ_FLAG_COUNTERS))
            assert np.array_equal(got[k].rects, w.rects), k
            assert got[k].stage_entered == w.stage_entered
        st.submit(batches[0])
        st.submit(batches[1])
        with pytest.raises(VjError):
            st.submit(batches[2])                   # two batches pending
        assert np.array_equal(st.collect().rects, want[0].rects) and np.array_equal(st.collect().rects, want[1].rects)
    finally:
        st.close()
        env.host_free(pinned)
    col = np.repeat(batches[0][..., None], 3, 3)
    st = env.stream(c, W, H, B, default_params(), channels=3)
    try:
        st.submit(col, color=True)
        assert np.array_equal(st.collect().rects, want[0].rects)
    finally:
        st.close()


# ----------------------------------------------------------------------------------------------- environment housekeeping
def test_plan_cache_is_bounded(env, cascades):
    """A stream of distinct region sizes (eyes inside faces of any size) must not grow device memory without bound:
    the plan cache evicts its least recently used plans."""
    import torch
    c, _ = cascades("eye")
    rng = np.random.default_rng(1)
    img = synth.frame("noise", 1, 400, 400)
    env.configure("plan_cache_max", 8)
    try:
        env.detect(c, img[:100, :100])
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        first = None
        for k in range(60):
            w, h = 60 + 3 * k, 50 + 2 * k
            r = env.detect(c, np.ascontiguousarray(img[:h, :w]))
            if k == 0:
                first = r.rects.copy()
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 < 256 << 20, (free0, free1)
        assert np.array_equal(env.detect(c, np.ascontiguousarray(img[:50, :60])).rects, first)   # an evicted plan is rebuilt
    finally:
        env.configure("plan_cache_max", 48)


def test_two_environments_and_second_device(cascades, oracle):
    """The LDS attribute of the tile kernel is set per environment (per device): a second environment — on device 1 when
    the box has one — runs tile launches with more than 64 KiB of dynamic LDS as well."""
    import torch
    from clfacedetection_amd import Environment
    c, a = cascades("frontalface_alt")
    img = synth.frame("blocks", 3, 480, 640)
    ro, _ = oracle.detect(a, img)
    envs = [Environment(0), Environment(1 if torch.cuda.device_count() > 1 else 0)]
    try:
        for e in envs:
            r = e.detect(c, img)
            assert rows(r.rects) == rows(ro)
            assert any(l["kind"] == "tile" and l["lds_bytes"] > 65536 for l in r.launches)
    finally:
        for e in envs:
            e.close()
