#!/bin/bash
# GPU box: compile-time variants of band_rows' register budget (VJ_ROWS_WPE) timed with tools/integral_time.py
cd "$GRAFT_REPO_ROOT" || exit 1
for d in VJ_ROWS_WPE=3 VJ_ROWS_WPE=4 VJ_ROWS_WPE=0; do
    VJ_DEFINES=$d python -c "from clfacedetection_amd.build import build_lib; build_lib(force=True)" || exit 1
    echo "== $d"; python tools/integral_time.py 2>&1 | grep -v amdgpu.ids
done
