#!/bin/bash
# Runs on the GPU box: the default bench command under rocprofv3, three separate passes
# (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# The program itself follows "--" (no env/bash wrappers).  Outputs land in gpurun_out/prof/.
set -o pipefail
STEPS=${STEPS:-5}; WARMUP=${WARMUP:-2}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py --steps $STEPS --warmup $WARMUP --cpu-frames 0 > gpurun_out/prof/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > gpurun_out/prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > gpurun_out/prof/write.log 2>&1
grep -h '"metric"' gpurun_out/prof/kt.log | tail -1
