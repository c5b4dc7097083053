"""The C-ABI library loads, exports every symbol include/vj.h declares (and nothing
else of ours), and refuses to compute without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from clfacedetection_amd import VjError, load_library
from clfacedetection_amd.api import _SIGNATURES
from clfacedetection_amd.build import LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in_header():
    text = open(os.path.join(ROOT, "include", "vj.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(vj_[a-z0-9_]+)\s*\(", text))


def test_header_and_binding_agree():
    assert declared_in_header() == set(_SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    ours = {s for s in exported if s.startswith("vj_")}
    assert ours == declared_in_header()
    for name in declared_in_header():
        assert getattr(lib, name) is not None


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "vj.h")).read()
    assert "torch" not in text.lower().replace("no torch types", "") and "at::" not in text and "std::" not in text


def test_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "clfacedetection_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "vj_oracle" not in src and "libvjoracle" not in src, f


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_env_create_fails_loudly_without_gpu(lib):
    h = C.c_void_p()
    rc = lib.vj_env_create(0, C.byref(h))
    assert rc == 5 and not h.value                      # VJ_ERR_NO_DEVICE, no CPU fallback
    assert b"no CPU fallback" in lib.vj_last_error()
    from clfacedetection_amd import Environment
    with pytest.raises(VjError):
        Environment(0)


def test_argument_errors(lib):
    assert lib.vj_env_create(0, None) == 1
    assert lib.vj_detect(None, None, None, 0, None, None) == 1
    assert lib.vj_integral(None, None, 0, 0, 0, None, None) == 1
    assert lib.vj_strerror(0) == b"ok" and lib.vj_strerror(99) == b"unknown error"


def test_cpu_variants_never_compute_on_the_cpu():
    """use_opencl=False picks the WINDOW SET of the reference's CPU loops (skip modes), still on the device; the block
    variant selects its own two grids the same way; clifIntegral's CPU branch (cvIntegral) is refused.  Nothing computes
    without an environment."""
    import numpy as np
    from clfacedetection_amd import CLOD_BLOCK_IMPLEMENTATION, clodDetectObjects, clifIntegral
    with pytest.raises(AttributeError):
        clodDetectObjects(np.zeros((40, 40), np.uint8), None, None, flags=CLOD_BLOCK_IMPLEMENTATION, use_opencl=False)
    with pytest.raises(AttributeError):   # no environment -> nothing to run on: there is no host evaluator to fall back to
        clodDetectObjects(np.zeros((40, 40), np.uint8), None, None, use_opencl=False)
    with pytest.raises(VjError):
        clifIntegral(np.zeros((4, 4), np.uint8), None, use_opencl=False)


def test_cascade_from_arrays_round_trip_and_validation(lib):
    """vj_cascade_from_arrays: what a cvLoad-ed CvHaarClassifierCascade converts to; equals the file-loaded cascade,
    derives `child`, and rejects broken links."""
    import numpy as np
    from clfacedetection_amd import Cascade
    for name in ("frontalface_alt", "frontalface_alt2", "frontalface_alt_tree"):
        c = Cascade.load(name)
        st = c.stages.copy()
        st["child"] = -1                                   # as a caller that only has parent / next would pass them
        c2 = Cascade.from_arrays(c.info.win_w, c.info.win_h, st, c.trees, c.nodes, c.alpha)
        assert np.array_equal(c2.stages, c.stages) and np.array_equal(c2.trees, c.trees)
        assert c2.nodes.tobytes() == c.nodes.tobytes() and c2.alpha.tobytes() == c.alpha.tobytes()
        assert c2.count_windows(1920, 1080) == c.count_windows(1920, 1080)
    c = Cascade.load("frontalface_alt")
    bad = c.trees.copy()
    bad["first_node"][5] = c.info.n_nodes                  # out of range
    with pytest.raises(VjError):
        Cascade.from_arrays(20, 20, c.stages, bad, c.nodes, c.alpha)
    badn = c.nodes.copy()
    badn["left"][0] = 3                                    # a stump pointing at a node that does not exist
    with pytest.raises(VjError):
        Cascade.from_arrays(20, 20, c.stages, c.trees, badn, c.alpha)
    bads = c.stages.copy()
    bads["parent"][3] = 7                                  # a parent that follows its child
    with pytest.raises(VjError):
        Cascade.from_arrays(20, 20, bads, c.trees, c.nodes, c.alpha)
    assert lib.vj_cascade_from_arrays(20, 20, None, 0, None, 0, None, 0, None, 0, None) == 1


def test_reference_shaped_shims_compile_and_fail_loudly_without_a_device(lib, tmp_path):
    """examples/clod_shim: clod.h's five and clif.h's seven functions with the reference's argument lists, and a plain C
    translation unit on include/vj.h, compile and link against libvjhip.so here; without a GPU they stop with the
    library's "no CPU fallback" error instead of computing anywhere else (the GPU box runs them: tests/test_gpu_native.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "clfacedetection_amd")
    exe = str(tmp_path / "clod_demo")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-Iinclude", "examples/clod_shim/clod_hip.cpp", "examples/clod_shim/clif_hip.cpp",
                    "examples/clod_shim/demo_main.cpp", f"-L{libdir}", "-lvjhip", f"-Wl,-rpath,{libdir}", "-o", exe], cwd=root, check=True)
    r = subprocess.run([exe], cwd=root, capture_output=True, text=True)
    assert (r.returncode == 0 and "clod shim demo: OK" in r.stdout) or "no CPU fallback" in r.stderr
    hdr = open(os.path.join(root, "examples", "clod_shim", "clif_hip.h")).read()
    for name in ("clifInitEnvironment", "clifReleaseEnvironment", "clifInitBuffers", "clifReleaseBuffers", "clifGrayscale",
                 "clifIntegral", "clifGrayscaleIntegral"):      # clif.h:42-73
        assert name + "(" in hdr
