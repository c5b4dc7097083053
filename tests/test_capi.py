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


def test_cpu_variants_are_refused():
    import numpy as np
    from clfacedetection_amd import clodDetectObjects, clifIntegral
    with pytest.raises(VjError):
        clodDetectObjects(np.zeros((40, 40), np.uint8), None, None, use_opencl=False)
    with pytest.raises(VjError):
        clifIntegral(np.zeros((4, 4), np.uint8), None, use_opencl=False)
