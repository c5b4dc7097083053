#!/bin/bash
# usage: tools/pmc_ab.sh TAG "CONFIGURE string" — SQ / LDS counters of the cascade kernels for one configuration (16 frames x 2 calls,
# chains serialised so that the counters of one kernel are not mixed with its neighbour's).  Program itself after "--".
tag=$1; export CONFIGURE="$2;concurrent=0"; export B=16 R=2
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_$tag -- python tools/prof_run.py > gpurun_out/pmc_$tag.log 2>&1
python tools/pmc_sum.py gpurun_out/pmc_$tag > gpurun_out/pmc_${tag}_sum.txt
cat gpurun_out/pmc_${tag}_sum.txt
