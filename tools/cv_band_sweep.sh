#!/bin/bash
# OpenCV profile: row order of the row kernel (cv_row_band_px; 0 = scale after scale), 64 x 1080p.
#   bash tools/cv_band_sweep.sh > gpurun_out/cv_band.log     (on the GPU box)
for b in 0 32 64 128 256 512; do
  python tools/cv_time.py frontalface_alt,frontalface_alt2,frontalface_default,frontalface_alt_tree 64 cv_row_band_px=$b 2>/dev/null | grep frames
done
echo "--- every scale on the row kernel (cv_tiles=0)"
for b in 0 64 128 256; do
  python tools/cv_time.py frontalface_alt,frontalface_alt_tree 64 cv_tiles=0 cv_row_band_px=$b 2>/dev/null | grep frames
done
