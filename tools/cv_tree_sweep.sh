#!/bin/bash
# GPU box: the OpenCV profile on a stage tree (frontalface_alt_tree, 64 x 1080p): kernel trace of the default settings, then the
# two balance knobs (row-kernel workgroups per CU x smallest tile) with the chain sweeps on.  Output: gpurun_out/cvtree/.
set -o pipefail
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cvtree; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/cv_time.py frontalface_alt_tree 64 > $O/kt.log 2>&1
cat $O/kt/*/*kernel_stats.csv | cut -d, -f1-6 | head -12
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ser -- python tools/cv_time.py frontalface_alt_tree 64 concurrent=0 > $O/kt_ser.log 2>&1
cat $O/kt_ser/*/*kernel_stats.csv | cut -d, -f1-6 | head -12
for rb in 1 2 3; do for mw in 128 256 512 1024; do
  timeout -k 10 200 python tools/cv_time.py frontalface_alt_tree 64 cv_row_blocks_tree=$rb cv_tile_min_windows_tree=$mw 2>/dev/null | grep frames
done; done | tee $O/sweep.log
