#!/bin/bash
# GPU box: stump-parallel threshold of the chain sweeps x chunk (frontalface_alt_tree, OpenCV profile)
cd "$GRAFT_REPO_ROOT"
for tm in 0 16 32 48 64; do for ch in 64 128; do
  timeout -k 10 200 python tools/cv_time.py frontalface_alt_tree 64 cv_tail_max=$tm cv_tree_chunk=$ch 2>/dev/null | grep frames
done; done
for tm in 32 64; do timeout -k 10 100 python tools/cv_time.py frontalface_alt_tree 1 cv_tail_max=$tm 2>/dev/null | grep frames; done
for c in frontalface_alt frontalface_alt2 frontalface_default; do timeout -k 10 100 python tools/cv_time.py $c 64 2>/dev/null | grep frames; done
