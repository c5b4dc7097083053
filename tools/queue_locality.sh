#!/bin/bash
# GPU box: the queue-pass locality experiment (tools/queue_locality.py) plain and under rocprofv3 --pmc, one counter group per
# pass and one process per frame size and chain mode (separate passes, program after "--": MI355X_MICROARCH.md).  Every
# profiled run is bounded by `timeout` (a counter set the hardware refuses aborts rocprofv3 and can leave it hanging).
# Output: gpurun_out/qloc/.
set -o pipefail
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/qloc; rm -rf $O; mkdir -p $O
python tools/queue_locality.py > $O/plain.log 2>&1 || exit 1
cat $O/plain.log
for spec in 1920x1080x64 1280x720x144 640x480x432; do
 for mode in 1 0; do
  [ $mode = 0 ] && [ $spec != 1920x1080x64 ] && continue
  T=${spec}_m$mode
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$T -- python tools/queue_locality.py $spec $mode > $O/fetch_$T.log 2>&1 || echo "fetch pass failed for $T"
  timeout -k 10 240 rocprofv3 --pmc TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/cache_$T -- python tools/queue_locality.py $spec $mode > $O/cache_$T.log 2>&1 || echo "cache pass failed for $T"
  for g in fetch cache; do echo "== $g $T"; python tools/pmc_agg.py $O/${g}_$T cascade_pass; done > $O/agg_$T.txt 2>&1
  cat $O/agg_$T.txt
 done
done
timeout -k 10 240 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta_1080 -- python tools/queue_locality.py 1920x1080x64 1 > $O/ta_1080.log 2>&1 || echo "ta pass failed"
echo "== ta 1080p overlapped"; python tools/pmc_agg.py $O/ta_1080 cascade_pass | tee $O/agg_ta_1080.txt
find $O -name "*.csv" -size +2M -delete
