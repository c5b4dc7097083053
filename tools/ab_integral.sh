#!/bin/bash
# GPU box: compile-time variants of the integral's row kernels (VJ_DEFINES=NAME=VALUE, one build per value) timed with tools/integral_time.py
#   bash tools/ab_integral.sh NAME=V1 NAME=V2 ...      (the last build stays)
cd "$GRAFT_REPO_ROOT" || exit 1
for d in "$@"; do
    VJ_DEFINES=$d python -c "from clfacedetection_amd.build import build_lib; build_lib(force=True)" || exit 1
    echo "== $d"; python tools/integral_time.py 2>&1 | grep -v amdgpu.ids
done
