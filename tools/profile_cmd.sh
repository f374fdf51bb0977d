#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of an arbitrary command, then separate PMC
# passes for HBM read / write traffic (never combined with tracing domains other than --kernel-trace).
#   tools/profile_cmd.sh <tag> "<command>"
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); summaries to gpurun_out/prof_<tag>/summary (copy to profiles/).
set -u
TAG=$1
CMD=$2
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace_stdout.log 2>&1
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch_stdout.log 2>&1
echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write_stdout.log 2>&1
echo "pmc write rc=$?"
cd $REPO
python tools/summarize_profile.py $OUT $TAG > $OUT/summary_stdout.log 2>&1
tail -2 $OUT/trace_stdout.log
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
