#!/bin/bash
# A/B of a COMPILE-TIME switch on the GPU box: builds the library once per value of VJ_DEFINES and runs tools/ab.py on each.
#   tools/ab_build.sh "VJ_WS_QUAD=0 VJ_WS_QUAD=1" [ab.py arguments...]      (results: gpurun_out/ab_build_<defines>.log)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
variants=$1; shift
for d in $variants; do
    VJ_DEFINES=$d python -c "from clfacedetection_amd.build import build_lib; build_lib(force=True)" || exit 1
    echo "== $d"
    python tools/ab.py "$@" 2>&1 | grep -v amdgpu.ids | tee "gpurun_out/ab_build_${d//[^A-Za-z0-9_=]/_}.log"
done
