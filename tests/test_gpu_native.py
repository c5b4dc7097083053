"""The C ABI from real C / C++ translation units on the GPU box (tests/capi_smoke.c, examples/clod_shim, examples/multi_gpu),
and the N > 1 path around the HIP kernels: two gloo ranks sharing the one GPU."""
import os
import socket
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "clfacedetection_amd")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, **kw)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


def test_c_translation_unit_uses_the_header(lib, tmp_path):
    exe = str(tmp_path / "capi_smoke")
    _run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Iinclude", "tests/capi_smoke.c", f"-L{LIBDIR}", "-lvjhip",
          f"-Wl,-rpath,{LIBDIR}", "-o", exe])
    out = _run([exe])
    assert "capi_smoke: OK" in out and "2 raw detections" in out


def test_clod_shim_keeps_the_reference_call_sites(lib, tmp_path):
    """clodInitEnvironment / clifInitBuffers / clodInitBuffers / clifIntegral / clifGrayscale / clifGrayscaleIntegral /
    clodDetectObjects(IplImage*, CvHaarClassifierCascade*, ...) / free() over the library (examples/clod_shim): both integral
    images element by element, then the OpenCL route and the four CPU-variant window sets on the survey's pin frame."""
    exe = str(tmp_path / "clod_demo")
    _run(["g++", "-std=c++17", "-Wall", "-Iinclude", "examples/clod_shim/clod_hip.cpp", "examples/clod_shim/clif_hip.cpp", "examples/clod_shim/demo_main.cpp",
          f"-L{LIBDIR}",
          "-lvjhip", f"-Wl,-rpath,{LIBDIR}", "-o", exe])
    out = _run([exe])
    assert "clod shim demo: OK" in out and out.count("2 matches") == 5 and out.count("all elements equal") == 2
    assert "clifGrayscaleIntegral on the B=G=R frame: equal" in out


def test_native_multi_gpu_host_with_rccl(lib, tmp_path):
    """A C++ host: one thread + one vj_env per visible device, vj_shard_frames, ncclAllGather of the rectangles
    (include/vj_rccl.h); every rank's gathered list equals a single-device run of the whole batch."""
    exe = str(tmp_path / "multi_gpu_detect")
    _run(["hipcc", "-O2", "-std=c++17", "-Iinclude", "examples/multi_gpu/multi_gpu_detect.cpp", f"-L{LIBDIR}", "-lvjhip", "-lrccl",
          f"-Wl,-rpath,{LIBDIR}", "-o", exe])
    out = _run([exe], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert "OK" in out and "MISMATCH" not in out
    import re
    m = re.search(r"(\d+) device\(s\), rccl_ranks (\d+)", out)
    assert m and m.group(1) == m.group(2)          # the all-gather spanned as many ranks as the box has devices (1 here)
    m = re.search(r"collectives per step (\d+) (\d+) (\d+)", out)      # 4 slots at first: one regrow, then ONE collective per step
    assert m and int(m.group(1)) in (1, 2) and m.group(2) == m.group(3) == "1", out
    print(out.strip().splitlines()[-1])


def _rank(rank, world, port, mode, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from clfacedetection_amd import Cascade, Environment, default_params, multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = Environment(0)                       # both ranks on the box's one GPU
    c = Cascade.load("frontalface_alt")
    H, W = 270, 360
    n_frames = 5 if mode == "frames" else 1
    frames = synth.batch(n_frames, H, W, seed0=900, kinds=("noise", "blocks"))
    counts = [s.nx * s.ny if s.accepted else 0 for s in c.plan_scales(W, H)]
    sides = [max(s.win_w, s.win_h) for s in c.plan_scales(W, H)]
    my_frames, my_scales = multigpu.plan(n_frames, counts, rank, world, sides, max(c.info.win_w, c.info.win_h))
    if my_scales is not None:
        assert my_scales == c.shard_scales(W, H, rank, world)          # the native helper picks the same scales
    p = default_params(scales=my_scales) if my_scales is not None else default_params()
    r = env.detect(c, frames[my_frames], p) if my_frames else None
    mine = r.rects.copy() if r is not None else np.zeros(0, multigpu_dtype())
    if len(mine):
        mine["frame"] = np.asarray(my_frames)[mine["frame"]]
    allr = multigpu.allgather_rects(mine)
    q.put((rank, allr.tolist()))
    dist.barrier()
    dist.destroy_process_group()
    env.close()


def multigpu_dtype():
    from clfacedetection_amd.api import RECT_DTYPE
    return RECT_DTYPE


@pytest.mark.parametrize("mode", ["frames", "scales"])
def test_two_ranks_around_the_hip_path(env, cascades, mode):
    """world_size 2 over gloo, both ranks computing on the one GPU with the HIP kernels (frames mode: whole frames per
    rank; scales mode: one frame, scales split by vj_shard_scales' LPT rule): the gathered result equals one rank's."""
    import torch.multiprocessing as mp
    from clfacedetection_amd import default_params, synth
    c, _ = cascades("frontalface_alt")
    n_frames = 5 if mode == "frames" else 1
    frames = synth.batch(n_frames, 270, 360, seed0=900, kinds=("noise", "blocks"))
    want = env.detect(c, frames, default_params()).rects
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert got[r] == want.tolist(), (mode, r)
