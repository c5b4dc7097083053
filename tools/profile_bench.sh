#!/bin/bash
# Runs on the GPU box: the default bench command under rocprofv3, separate passes as MI355X_MICROARCH.md prescribes
# (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE; one SQ / LDS pass).  The program itself follows "--" (no env /
# bash wrappers).  Outputs land in gpurun_out/prof/; tools/pmc_traffic.py turns them into profiles/rNN_*.
set -o pipefail
STEPS=${STEPS:-5}; WARMUP=${WARMUP:-2}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
B="python bench.py --cpu-frames 0 --extras= --no-pipeline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- $B --steps $STEPS --warmup $WARMUP > gpurun_out/prof/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -- $B --steps 2 --warmup 1 > gpurun_out/prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -- $B --steps 2 --warmup 1 > gpurun_out/prof/write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/prof/sq -- $B --steps 2 --warmup 1 > gpurun_out/prof/sq.log 2>&1
# the other configs: kernel trace + stats only (tools/pmc_traffic.py copies the summaries to profiles/rNN_<config>_kernel_stats.csv)
for cfg in "config2 40" "config4 20" "config5 4" "cv 4" "cvtree 3"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt_$1 -- python tools/configs.py $1 $2 > gpurun_out/prof/kt_$1.log 2>&1
    grep -h "wall ms" gpurun_out/prof/kt_$1.log
done
grep -h '"metric"' gpurun_out/prof/kt.log | tail -1
