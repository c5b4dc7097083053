#!/bin/bash
# usage: tools/pmc.sh OUTDIR "COUNTERS..." -- env for prof_run via exported vars
# Runs one rocprofv3 --pmc pass of tools/prof_run.py (program itself after --, no wrappers).
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $@ --output-format csv -d gpurun_out/$out -- python tools/prof_run.py > gpurun_out/$out.log 2>&1
