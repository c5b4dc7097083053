#!/bin/bash
# GPU box: chunk size / workgroups per CU of cv_tree_chain_pass and the tile threshold (frontalface_alt_tree, 64 x 1080p and one frame).
cd "$GRAFT_REPO_ROOT"
for ch in 64 128 256; do for wb in 1 2 3 4; do
  timeout -k 10 200 python tools/cv_time.py frontalface_alt_tree 64 cv_tree_chunk=$ch cv_tree_chain_blocks=$wb 2>/dev/null | grep frames
done; done
for ch in 64 128 256; do timeout -k 10 100 python tools/cv_time.py frontalface_alt_tree 1 cv_tree_chunk=$ch 2>/dev/null | grep frames; done
