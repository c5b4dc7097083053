#!/bin/bash
# GPU box: OpenCV profile, linear cascades, after the row kernel got the pair / stump-parallel forms: the two balance knobs
# (row-kernel workgroups per CU x smallest tile) on several workloads
cd "$GRAFT_REPO_ROOT"
for cfg in "frontalface_alt 64 1080 1920" "frontalface_default 64 1080 1920" "frontalface_alt2 64 1080 1920" "frontalface_alt 256 720 1280" "eye 64 720 1280" "frontalface_alt 16 1080 1920"; do
  set -- $cfg
  for knobs in "1 1536" "2 1536" "2 2048" "3 1536"; do
    set -- $cfg $knobs
    H=$3 W=$4 timeout -k 10 150 python tools/cv_time.py $1 $2 cv_row_blocks=$5 cv_tile_min_windows=$6 2>/dev/null | grep frames | sed "s/^/${4}x${3} /"
  done
done
