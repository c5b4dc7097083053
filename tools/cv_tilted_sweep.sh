#!/bin/bash
# OpenCV profile, cascades with tilted features: LDS tiles (two images per tile) against the row kernel alone, and the balance
# (row-kernel workgroups per CU x smallest tile) — 16 x 1080p.      bash tools/cv_tilted_sweep.sh > gpurun_out/cv_tilted.log
for c in ${CASCADES:-fullbody mcs_righteye upperbody}; do
  python tools/cv_time.py $c 16 cv_tiles_tilted=0 2>/dev/null | grep frames
  python tools/cv_time.py $c 16 2>/dev/null | grep frames
  for rb in ${BLOCKS:-1 2 3}; do for mw in ${WINDOWS:-256 384 512 1024 1536 2048}; do
    python tools/cv_time.py $c 16 cv_row_blocks=$rb cv_tile_min_windows=$mw cv_tile_min_windows0=$mw 2>/dev/null | grep frames
  done; done
done
