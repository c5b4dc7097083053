"""Build recipe of libvjhip.so (the C-ABI library declared in include/vj.h).

Everything is compiled in-tree with hipcc for gfx950 only; the resulting .so is
git-ignored but travels to the GPU box with the snapshot.  -ffp-contract=off is part
of the arithmetic contract: hipcc's default would fuse the reference's
separate multiply/add into FMAs and break bit-exact parity.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvjhip.so")
SOURCES = ["vj_cascade.cpp", "vj_plan.cpp", "vj_group.cpp", "vj_env.cpp", "vj_cv.cpp", "vj_kernels.hip",
           "vj_cv_profile.hip", "vj_cv_tile.hip", "vj_group_dev.hip"]
HEADERS = ["vj_internal.hpp", "vj_device.hpp", "vj_env_internal.hpp", "vj_devutil.hpp", os.path.join("..", "..", "include", "vj.h")]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
         "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-DVJ_BUILDING"] + \
    (["-DVJ_STAMPS=1"] if os.environ.get("VJ_STAMPS") else []) + \
    [f"-D{k}={v}" for k, v in (kv.split("=", 1) for kv in os.environ.get("VJ_DEFINES", "").split(",") if "=" in kv)]   # experiment switches


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libvjhip.so cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile libvjhip.so if missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    objs = []
    build_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    hipcc = _hipcc()
    for src in SOURCES:
        obj = os.path.join(build_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        objs.append(obj)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
